// Diagnostic: the "f32s" (float16 pieces) GEMM main loop in isolation -- LDS-DMA of the next stages from a buffer
// far larger than the caches, fragment ds_read_b128, three v_mfma_f32_32x32x16_f16 per product block -- in the
// product's configuration and in candidates with less LDS traffic per product:
//   P0  128 x 128 tile, 4 waves of 64 x 64, 2 LDS stages, 2 workgroups per CU, hipcc's schedule of the reads (product)
//   P1  P0 with the fragment reads software-pipelined across the stage barrier (k-step t+1 requested before the
//       MFMAs of k-step t; the first k-step of the next stage right behind the barrier)
//   P2  128 x 256 tile, 4 waves of 64 x 128 (a quarter fewer LDS bytes per product), 3 LDS stages (two of lead),
//       1 workgroup per CU, pipelined like P1
//   P3  256 x 128 tile, 4 waves of 128 x 64, otherwise P2
//   P4  P2 with 2 LDS stages
// Prints f32-equivalent TFLOP/s (2 M N K per product block, whatever the number of MFMAs behind it; roof 833).
// Measured (MI355X, 72 stages, 16 workgroups per source window, windows within 24 MiB / 192 MiB / 3 GiB):
//   P0 430 / 494 / 470    P1 437 / 486 / 456    P2 413 / 428 / 421    P3 438 / 433 / 422    P4 451 / 453 / 416
// i.e. the product's configuration is as good as any of them and explicit pipelining of the reads buys nothing here
// (two workgroups per CU already cover each other's LDS round trips): the loop itself runs at 52-59 % of the pieces
// roof, 1.25 us per stage like the product's workgroups; a whole 3x3 layer (310 in the product) then loses its
// partly filled last round of tiles and the tile prologues / epilogues.
// P5 / P6 (the pieces loop in the 256 x 128 / 64-byte-stage geometry, two workgroups per CU): 550-560, +12..29 % -- the
// pieces loop in the product's configuration moves the same 32 KB per stage as the float16 one and sits on the same
// ~15 TB/s L2 -> LDS stream.  Built as a kernel like conv_gemm_wide16.hip and measured: 3x3 class +4.5 %, end to end
// +1.1 % in one A/B and -0.8 % in a threshold sweep at 720p batch 16, nothing at batch 64 -- the f32s layers that are
// not HBM-bound are too small a share.  Removed again.
// float16 mode loop (H*): 128 x 128 tiles 495-505 whatever the wave layout (the L2 -> LDS stream, ~15 TB/s, bounds it);
// 128 x 256 tiles, 3 stages, 1 workgroup per CU 570-610 -- but built into conv_gemm_kernel that configuration lost
// 4.5-6 % end to end (nobody multiplies during a tile's prologue / epilogue with one workgroup per CU).
// H8-H10: 256 x 128 tiles with 64-byte (32-k) stages, 48 KB of LDS, TWO workgroups per CU: 620-665 -- the geometry of
// conv_gemm_wide16.hip (+2.7 % end to end at 720p batch 16, +5.7 % at 4K batch 32).
// H11-H15 (round 3): 256 x 256 tiles, 8 waves of 64 x 128 or 128 x 64, ONE workgroup per CU, 3-4 stages of 32 KB or 2 of
// 64 KB -- a third fewer L2 -> LDS bytes per product than 256 x 128: 680-695 (64-byte stages), 686-732 (128-byte stages):
// +4 % over H8.  At 660+ (1.3 PFLOP/s executed) the loop is no longer bound by the L2 -> LDS stream but by the matrix
// pipe at the clock the chip holds under this load; the product's distance to it (441 for the 3x3 class, 535 for the
// 1x1 class) is not tile geometry.  Not built.
// H16 / H17: H8 fed like the product -- every row 64 bytes of a pixel's channel run, i.e. HALF a 128-byte line, the other
// half wanted in the next stage (1x1 layers) or nine stages later (3x3 layers): 477-575 / 334-346 instead of 632-660.
// H18-H22: whole-line rows (128-byte stages): 192 x 128 at two workgroups per CU 453-469 on pitched rows, 256 x 128 at one
// workgroup per CU 547-569.
// H23-H26: the 256 x 128 tile as 4 FAT waves of 128 x 64 (0.75 KB of LDS fragment reads per MFMA instead of 1, two waves per
// SIMD): 480 / 349 where H8's eight waves run 660 / 635 -- two waves per SIMD do not cover each other's round trips.
//   hipcc -O3 --offload-arch=gfx950 tools/pieces_loop_bench.hip -o build/pieces_loop_bench && build/pieces_loop_bench
#include <hip/hip_runtime.h>

#include <cstdio>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;


template <int BM, int BN, int WM, int WN, int NS, bool PIPE, bool F16 = false, int ROWB = 128, int HALF = 0, bool BCONT = false>
__global__ __launch_bounds__(64 * WM * WN) void loop_kernel(float *out, int stages, const char *src, size_t src_bytes) {
  constexpr int NW = WM * WN, MI = BM / WM / 32, NI = BN / WN / 32;
  constexpr int STAGE = (BM + BN) * ROWB;
  constexpr int PER = ((BM + BN) * ROWB / 1024 + NW - 1) / NW;   // LDS-DMA instructions (1 KiB each) per wave and stage
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN, r = lane & 31, h = lane >> 5;
  // 128-byte rows: 8 chunks, chunk ^ (row >> 1) & 7; 64-byte rows (F16 only): 4 chunks, chunk ^ (row >> 2) & 3
  const int sw = ROWB == 128 ? (r >> 1) & 7 : (r >> 2) & 3;
  floatx16 acc[MI][NI];
  for (int mi = 0; mi < MI; ++mi)
    for (int ni = 0; ni < NI; ++ni)
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;
  // stage s, row group g of 8 rows = 1 KiB.  The product's operands mostly come out of L2 (weights; the taps and the
  // n-tiles re-reading an activation tile), so 16 workgroups share a source window and the windows together (src_bytes)
  // can be sized to sit in L2, in the Infinity Cache or in neither
  const size_t wg_span = (size_t)stages * STAGE;
  const char *my = src + ((size_t)(blockIdx.x / 16) * wg_span) % (src_bytes - wg_span);
  auto issue = [&](int s) __attribute__((always_inline)) {
    char *dst = lds + (s % NS) * STAGE;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
      const int g = wave + NW * i;
      if (g >= STAGE / 1024) break;   // (wave-uniform; only geometries whose stage is not a multiple of NW KiB)
      if (HALF == 0 || (BCONT && g >= BM * ROWB / 1024)) {   // (BCONT: the weight rows of a stage pre-packed contiguously)
        __builtin_amdgcn_global_load_lds((gptr_t)(my + (size_t)s * STAGE + g * 1024 + lane * 16), (lptr_t)(dst + g * 1024), 16, 0, 0);
      } else if constexpr (HALF < 0) {
        // whole 128-byte lines, one per row, rows at a 512-byte pitch (a pixel's channel run in a 256-channel tensor)
        __builtin_amdgcn_global_load_lds((gptr_t)(my + (size_t)(s / 4) * 4 * STAGE + (s % 4) * 128 + (size_t)(g * 8 + (lane >> 3)) * 512 + (lane & 7) * 16),
                                         (lptr_t)(dst + g * 1024), 16, 0, 0);
      } else {
        // the product's activation rows: 64 bytes of a pixel's channel run per stage, i.e. HALF a 128-byte line per row;
        // the line's other half is read HALF stages later (1: the next 32-channel chunk of a 1x1 layer; 9: of a 3x3 layer)
        const int grp = s / (2 * HALF), pos = s % (2 * HALF), slot = pos % HALF, half = pos / HALF;
        const char *base = my + ((size_t)grp * HALF + slot) * 2 * STAGE;
        __builtin_amdgcn_global_load_lds((gptr_t)(base + g * 2048 + (lane >> 2) * 128 + half * 64 + (lane & 3) * 16),
                                         (lptr_t)(dst + g * 1024), 16, 0, 0);
      }
    }
  };
  // F16 (the float16 mode: f16 activations, [hi | lo] weight rows stacked along N): four 16-k steps per 128-byte
  // stage, one fragment per operand block and one MFMA per block pair; else the pieces loop (two steps, hi / lo
  // fragments, three MFMAs per block pair)
  constexpr int TSTEPS = F16 ? ROWB / 32 : ROWB / 64;   // pieces: a 64-byte row is one 16-k step (2 hi chunks, 2 lo chunks)
  constexpr int LO = ROWB / 32;                          // pieces: chunk distance between a value's hi and lo piece
  struct Frags {
    halfx8 ahi[MI], alo[F16 ? 1 : MI], bhi[NI], blo[F16 ? 1 : NI];
  };
  auto read = [&](Frags &f, int s, int t) __attribute__((always_inline)) {
    const char *a = lds + (s % NS) * STAGE + (wm * (BM / WM) + r) * ROWB;
    const char *b = lds + (s % NS) * STAGE + BM * ROWB + (wn * (BN / WN) + r) * ROWB;
    const int g = 2 * t + h;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
      f.ahi[mi] = *reinterpret_cast<const halfx8 *>(a + mi * 32 * ROWB + 16 * (g ^ sw));
      if constexpr (!F16) f.alo[mi] = *reinterpret_cast<const halfx8 *>(a + mi * 32 * ROWB + 16 * ((LO + g) ^ sw));
    }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      f.bhi[ni] = *reinterpret_cast<const halfx8 *>(b + ni * 32 * ROWB + 16 * (g ^ sw));
      if constexpr (!F16) f.blo[ni] = *reinterpret_cast<const halfx8 *>(b + ni * 32 * ROWB + 16 * ((LO + g) ^ sw));
    }
  };
  auto mma = [&](const Frags &f) __attribute__((always_inline)) {
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi[mi], f.bhi[ni], acc[mi][ni], 0, 0, 0);
        if constexpr (!F16) {
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.alo[mi], f.bhi[ni], acc[mi][ni], 0, 0, 0);
          acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi[mi], f.blo[ni], acc[mi][ni], 0, 0, 0);
        }
      }
  };
  if constexpr (!PIPE) {
    // the product's order: barrier, DMA of the stage NS - 1 ahead, then the stage's reads and MFMAs
#pragma unroll
    for (int s = 0; s < NS - 1; ++s) issue(s);
    for (int s = 0; s < stages; ++s) {
      // stage s has landed once at most the NS - 2 younger stages are outstanding
      if (NS == 2 || s + NS - 2 >= stages) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * PER) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + NS - 1 < stages) issue(s + NS - 1);
      __builtin_amdgcn_sched_barrier(0);
      Frags f;
#pragma unroll
      for (int t = 0; t < TSTEPS; ++t) {
        read(f, s, t);
        mma(f);
      }
      __builtin_amdgcn_sched_barrier(0);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  } else {
    // all NS buffers in flight; the barrier sits in the MIDDLE of a stage, behind the request of its last fragments:
    // past it every read of stage s has returned (its buffer takes stage s + NS) and stage s + 1 is visible, whose
    // first fragments are requested before the second half of stage s's MFMAs is issued
#pragma unroll
    for (int s = 0; s < NS; ++s) issue(s);
    Frags fr[2];
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 1) * PER) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read(fr[0], 0, 0);
    for (int s = 0; s < stages; ++s) {
#pragma unroll
      for (int t = 0; t < TSTEPS - 1; ++t) {
        read(fr[(t + 1) & 1], s, t + 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(fr[t & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (NS == 2 || s + NS > stages) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((NS - 2) * PER) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (s + NS < stages) issue(s + NS);
      if (s + 1 < stages) read(fr[0], s + 1, 0);   // TSTEPS is even: the last step of a stage sits in fr[1]
      __builtin_amdgcn_sched_barrier(0);
      mma(fr[1]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float sum = 0.f;
  for (int mi = 0; mi < MI; ++mi)
    for (int ni = 0; ni < NI; ++ni)
      for (int q = 0; q < 16; ++q) sum += acc[mi][ni][q];
  if (sum == 12345.678f) out[tid] = sum;  // keep the accumulators live
}

template <int BM, int BN, int WM, int WN, int NS, bool PIPE, bool F16 = false, int ROWB = 128, int HALF = 0, bool BCONT = false>
void run(const char *name, int wgs_per_cu, const char *src, size_t src_bytes, float *out) {
  const int stages = 72, grid = 256 * wgs_per_cu * 6;
  const size_t lds = (size_t)NS * (BM + BN) * ROWB;
  auto k = loop_kernel<BM, BN, WM, WN, NS, PIPE, F16, ROWB, HALF, BCONT>;
  (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  float best = 1e30f;
  for (int rep = 0; rep < 12; ++rep) {   // back to back: the clock governor settles
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k, dim3(grid), dim3(64 * WM * WN), lds, 0, out, stages, src, src_bytes);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 4 && ms < best) best = ms;
  }
  // F16: a 128-byte stage is 64 k, and half of the tile's N rows are the lo halves of the other half's channels
  const double flops = F16 ? 2.0 * BM * (BN / 2) * (ROWB / 2.0) * stages * grid : 2.0 * BM * BN * (ROWB / 4.0) * stages * grid;
  std::printf("%-64s %8.3f ms  %7.1f f32-equivalent TFLOP/s  (%s)\n", name, best, flops / best / 1e9,
              hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t alloc = (size_t)3 << 30;
  char *src;
  float *out;
  if (hipMalloc(&src, alloc) != hipSuccess || hipMalloc(&out, 4096) != hipSuccess) return 1;
  (void)hipMemset(src, 0x3c, alloc);   // finite float16 values (0x3c3c = 1.06)
  for (size_t src_bytes : {(size_t)24 << 20, (size_t)192 << 20, alloc}) {
  std::printf("== source windows within %zu MiB\n", src_bytes >> 20);
  run<128, 128, 2, 2, 2, false>("P0 128x128, 4 x (64x64), 2 stages, 2 WG/CU (product)", 2, src, src_bytes, out);
  run<128, 128, 2, 2, 2, true>("P1 P0 + reads pipelined across the barrier", 2, src, src_bytes, out);
  run<128, 256, 2, 2, 3, true>("P2 128x256, 4 x (64x128), 3 stages, 1 WG/CU, pipelined", 1, src, src_bytes, out);
  run<256, 128, 2, 2, 3, true>("P3 256x128, 4 x (128x64), 3 stages, 1 WG/CU, pipelined", 1, src, src_bytes, out);
  run<128, 256, 2, 2, 2, true>("P4 128x256, 4 x (64x128), 2 stages, 1 WG/CU, pipelined", 1, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, false, 64>("P5 pieces 256x128, 8 x (64x64), 64-byte stages, 2 stages, 2 WG/CU", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 3, false, false, 64>("P6 P5 with 3 stages (72 KiB), 2 WG/CU", 2, src, src_bytes, out);
  run<128, 128, 2, 4, 2, false, true>("H0 f16 128x128, 8 x (64x32), 2 stages, 2 WG/CU (product)", 2, src, src_bytes, out);
  run<128, 128, 2, 4, 2, true, true>("H1 H0 + reads pipelined across the barrier", 2, src, src_bytes, out);
  run<128, 128, 2, 2, 2, false, true>("H2 f16 128x128, 4 x (64x64), 2 stages, 2 WG/CU", 2, src, src_bytes, out);
  run<128, 128, 2, 2, 2, true, true>("H3 H2 + pipelined", 2, src, src_bytes, out);
  run<128, 256, 2, 2, 2, true, true>("H4 f16 128x256, 4 x (64x128), 2 stages, 1 WG/CU, pipelined", 1, src, src_bytes, out);
  run<128, 256, 2, 4, 3, true, true>("H5 f16 128x256, 8 x (64x64), 3 stages, 1 WG/CU, pipelined", 1, src, src_bytes, out);
  run<128, 256, 2, 4, 2, false, true>("H6 f16 128x256, 8 x (64x64), 2 stages, 1 WG/CU, product's loop order", 1, src, src_bytes, out);
  run<128, 256, 2, 4, 2, true, true>("H7 H6 pipelined", 1, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 64>("H8 f16 256x128, 8 x (64x64), 64-byte stages, 2 stages, 2 WG/CU", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 2, true, true, 64>("H9 H8 pipelined", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 3, false, true, 64>("H10 H8 with 3 stages (72 KiB), 2 WG/CU", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 64, 1>("H16 H8, rows = half lines, other half in the next stage (1x1 layers)", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 64, 9>("H17 H8, rows = half lines, other half 9 stages later (3x3 layers)", 2, src, src_bytes, out);
  run<192, 128, 3, 2, 2, false, true, 128>("H18 f16 192x128, 6 x (64x64), 128-byte stages, 2 stages (80 KiB), 2 WG/CU", 2, src, src_bytes, out);
  run<192, 128, 3, 2, 2, false, true, 128, -1>("H19 H18, rows = whole lines at a 512-byte pitch", 2, src, src_bytes, out);
  run<128, 128, 2, 4, 2, false, true, 128, -1>("H20 H0 (128x128, 128-byte stages), rows = whole lines at a 512-byte pitch", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 3, false, true, 128, -1>("H21 f16 256x128, 8 x (64x64), 128-byte stages, 3 stages (144 KiB), 1 WG/CU, whole lines", 1, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 128, -1>("H22 H21 with 2 stages (96 KiB)", 1, src, src_bytes, out);
  run<256, 128, 2, 2, 2, false, true, 64>("H23 f16 256x128, 4 FAT waves of 128x64 (0.75 KB of LDS reads per MFMA), 64-byte stages, 2 WG/CU", 2, src, src_bytes, out);
  run<256, 128, 2, 2, 2, true, true, 64>("H24 H23 pipelined", 2, src, src_bytes, out);
  run<256, 128, 2, 2, 2, false, true, 64, 1>("H25 H23, half-line rows, other half next stage (1x1)", 2, src, src_bytes, out);
  run<256, 128, 2, 2, 2, false, true, 64, 9>("H26 H23, half-line rows, other half 9 stages later (3x3)", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 64, 1, true>("H27 H16 (1x1 pattern) with the weight rows of a stage contiguous", 2, src, src_bytes, out);
  run<256, 128, 4, 2, 2, false, true, 64, 9, true>("H28 H17 (3x3 pattern) with the weight rows of a stage contiguous", 2, src, src_bytes, out);
  run<256, 256, 4, 2, 3, false, true, 64>("H11 f16 256x256, 8 x (64x128), 64-byte stages, 3 stages (96 KiB), 1 WG/CU", 1, src, src_bytes, out);
  run<256, 256, 4, 2, 3, true, true, 64>("H12 H11 pipelined", 1, src, src_bytes, out);
  run<256, 256, 4, 2, 4, false, true, 64>("H13 H11 with 4 stages (128 KiB)", 1, src, src_bytes, out);
  run<256, 256, 2, 4, 3, false, true, 64>("H14 f16 256x256, 8 x (128x64), 3 stages, 1 WG/CU", 1, src, src_bytes, out);
  run<256, 256, 4, 2, 2, false, true, 128>("H15 f16 256x256, 8 x (64x128), 128-byte stages, 2 stages (128 KiB), 1 WG/CU", 1, src, src_bytes, out);
  }
  return 0;
}
