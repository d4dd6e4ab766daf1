// The float16 mode's implicit-GEMM convolution (conv_gemm.hip: float16 activations against [hi | lo] float16 weight
// rows stacked along N) in a second tile geometry, for launches that are several rounds of tiles:
//
//     256 pixels x 128 weight rows ([64 hi | 64 lo] of one 64-channel group), K stages of 64 BYTES (32 k).
//
// Why: that loop is bound by the L2 -> LDS stream, not by the matrix pipe, the LDS or HBM -- in isolation every
// 128 x 128 configuration of it (4 or 8 waves, fragment reads pipelined or not) runs at the same ~15 TB/s of LDS-DMA,
// 500 algorithmic TFLOP/s (tools/pieces_loop_bench.hip, H0-H3).  A 256 x 128 tile moves a quarter fewer bytes per
// product; with 128-byte stages it needs 96 KB of LDS and one workgroup per CU, and that version (as 128 x 256) lost
// 4.5-6 % in the product because nothing multiplies during a tile's prologue, first-stage wait and transpose.
// Halving the stage depth instead keeps the stage at 24 KB, the ring of two at 48 KB and TWO workgroups per CU:
// 660 TFLOP/s in the isolated loop (H8), +32 %.
//
// Layout of a stage: rows of 64 bytes = 4 chunks of 16 bytes; an LDS-DMA wave-instruction writes 64 x 16 B = 16 rows
// linearly, lane -> row 16 g + lane / 4, chunk position lane % 4, which receives global chunk pos ^ ((row >> 2) & 3).
// A fragment read (ds_read_b128, 16 lanes per pass = rows r .. r + 15 of one MFMA operand) then hits bank groups
// 4 (row & 3) + (chunk ^ (row >> 2) & 3): all 16 distinct.  Wave (wm, wn), 4 x 2: pixels [64 wm, 64 wm + 64) x channels
// [32 wn, 32 wn + 32) of the group, hi and lo rows in separate accumulators (two 32-pixel blocks each: 64 registers),
// 8 MFMAs (v_mfma_f32_32x32x16_f16) per wave and stage.  Epilogue: hi + 2^-11 lo through a 128-row float32 transpose in the
// (idle) stage buffers, two rounds, then bias (+ residual) (+ ReLU) and 8-byte float16 stores.
// Plain tiles only (one workgroup per tile, no split-K / stream-K): launch_conv_gemm picks this kernel for the big
// launches and keeps conv_gemm_kernel for the rest.
#include "cnn_device.h"
#include "cnn_kernels.h"
#ifdef DVSG_STAMPS
#include <cstdio>
#endif

namespace dvsg {
namespace {

constexpr int WBM = 256;     // pixels per tile
constexpr int WBN = 128;     // stacked weight rows per tile (64 output channels)
constexpr int WROWB = 64;    // bytes of k per row and stage (32 float16)
constexpr int WBKE = 32;     // k elements per stage

__device__ const floatx4 g_zero16w = {0.f, 0.f, 0.f, 0.f};

struct ConvWide16Dev {
  const _Float16 *x, *wt, *res;
  const _Float16 *wtp;   // the weights again, packed stage by stage (pack_wide16_kernel), or nullptr
  int arows;             // 128-byte activation rows (conv_wide16a_kernel; wtp packed in order 1)
  const float *bias;
  _Float16 *y;
  int H, W, Cin, Ho, Wo, Cout;
  int stride, pad;
  int res_H, res_W, res_stride;
  int M, K, mtiles, ntiles;
#ifdef DVSG_STAMPS
  int stamp;   // this launch writes its per-workgroup phase times to g_w16_stamps
#endif
};

#ifdef DVSG_STAMPS  // diagnostic build (tools/stamp_probe_wide16.py): per-workgroup phase times, wave 0 of each workgroup
__device__ unsigned long long g_w16_stamps[8 * 65536];
#define W16_STAMP() __builtin_amdgcn_s_memtime()
#endif

// SPLIT = false: the 128 weight rows of a tile are 128 output channels of a PLAIN float16 weight matrix [Cout][K] (layers
// whose weights do not need the lo piece -- locnet.hip's pair policy): the "lo" accumulators are simply the tile's second
// 64 channels, same loop, half the bytes and MFMAs per output channel.
template <int KS, bool RELU, int RES, bool SPLIT = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv_wide16_kernel(ConvWide16Dev p) {
  constexpr int NW = 8, MI = 2;
  constexpr int AG = WBM / 16 / NW;   // 2 LDS-DMA instructions (16 rows each) per wave and stage for the pixels
  constexpr int PER = AG + 1;         // + 1 for the 128 weight rows
  constexpr float kLoScale = 1.0f / 2048.0f;
  __shared__ __attribute__((aligned(16))) char lds[2 * (WBM + WBN) * WROWB];   // 48 KiB
  char *As = lds;
  char *Bs = lds + 2 * WBM * WROWB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = lane >> 2, lpos = lane & 3;
  const int sw = (r >> 2) & 3;   // rows of a fragment are 32-aligned + r
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  // nt fastest: the n-tiles of a pixel tile run back to back on one XCD and re-read its activations out of L2
#ifdef DVSG_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = W16_STAMP();
#endif
  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * WBM;

  long a_off[AG];
  unsigned a_mask[AG];
  const bool dense = KS == 1 && p.stride == 1;   // a 1x1 / stride 1 layer is a row-major GEMM (conv_gemm.hip)
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = 16 * (wave + NW * i) + lrow;
    const int chunk = lpos ^ ((row >> 2) & 3);
    const int m = m0 + row;
    const int mm = m < p.M ? m : 0;
    if (dense) {
      a_off[i] = (long)mm * p.Cin + 8 * chunk;
      a_mask[i] = m < p.M ? 0x11u : 0u;
      continue;
    }
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + 8 * chunk;
    unsigned mk = 0;
    if (m < p.M) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        if (hi0 + q >= 0 && hi0 + q < p.H) mk |= 1u << q;
        if (wi0 + q >= 0 && wi0 + q < p.W) mk |= 16u << q;
      }
    }
    a_mask[i] = mk;
  }
  const _Float16 *wsrc;
  {
    const int row = 16 * wave + lrow;   // stacked weight row of the tile (0..127)
    wsrc = p.wt + ((size_t)nt * WBN + row) * p.K + 8 * (lpos ^ ((row >> 2) & 3));
  }
  // K order as in conv_gemm.hip: channel chunk outer, the KS x KS taps inner
  // The weight rows of a stage: 64 bytes out of each of 128 rows that lie K elements apart -- 128 half lines, the other
  // half of each wanted one stage (1x1) or nine stages (3x3) later.  The packed copy holds every stage of a tile as 8 KB
  // in a row, already in the LDS image's chunk order: the same bytes as 64 whole lines, lane l of wave w fetching
  // 16 bytes at 1024 w + 16 l (isolated loop, tools/pieces_loop_bench.hip H27 / H28: +15 % / +30 % with such stages).
  const _Float16 *wpk = p.wtp ? p.wtp + (size_t)nt * p.K * WBN + wave * 512 + lane * 8 : nullptr;
  int s_kh = 0, s_kw = 0, s_c0 = 0;
  auto issue_stage = [&](int buf) __attribute__((always_inline)) {
    const _Float16 *xa = p.x + ((long)s_kh * p.W + s_kw) * p.Cin + s_c0;
    const int wk = KS > 1 ? (s_kh * KS + s_kw) * p.Cin + s_c0 : s_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16w);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * WBM + 16 * (wave + NW * i)) * WROWB), 16, 0, 0);
    }
    const _Float16 *wfrom = wsrc + wk;
    if (wpk) {   // (uniform)
      wfrom = wpk;
      wpk += WBN * WBKE;
    }
    __builtin_amdgcn_global_load_lds((gptr_t)wfrom, (lptr_t)(Bs + (buf * WBN + 16 * wave) * WROWB), 16, 0, 0);
    if (KS > 1) {
      if (++s_kw == KS) {
        s_kw = 0;
        if (++s_kh == KS) {
          s_kh = 0;
          s_c0 += WBKE;
        }
      }
    } else {
      s_c0 += WBKE;
    }
  };
  floatx16 acc_hi[MI], acc_lo[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_hi[mi][q] = acc_lo[mi][q] = 0.f;
  auto compute_stage = [&](int buf) __attribute__((always_inline)) {
    const char *a_base = As + (buf * WBM + wm * 64 + r) * WROWB;
    const char *b_base = Bs + (buf * WBN + wn * 32 + r) * WROWB;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int co = 16 * ((2 * t + h) ^ sw);
      const halfx8 bh = *reinterpret_cast<const halfx8 *>(b_base + co);
      const halfx8 bl = *reinterpret_cast<const halfx8 *>(b_base + 64 * WROWB + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const halfx8 a = *reinterpret_cast<const halfx8 *>(a_base + mi * 32 * WROWB + co);
        acc_hi[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc_hi[mi], 0, 0, 0);
        acc_lo[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc_lo[mi], 0, 0, 0);
      }
    }
  };

  const int KT = p.K / WBKE;   // >= 2 (Cin % 64 == 0)
  issue_stage(0);
  issue_stage(1);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
#ifdef DVSG_STAMPS
  const unsigned long long t_first = W16_STAMP();
  st[0] = t_first - t_begin;   // index setup + the first stage's round trip
#endif
  __builtin_amdgcn_sched_barrier(0);
  compute_stage(0);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 1; kt < KT - 1; ++kt) {
#ifdef DVSG_STAMPS
    const unsigned long long b0 = W16_STAMP();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long b1 = W16_STAMP();
    __syncthreads();
    st[1] += b1 - b0;             // this wave's part of the next stage has not landed
    st[2] += W16_STAMP() - b1;    // at the barrier
#else
    __syncthreads();   // vmcnt(0): stage kt has landed; everyone has read stage kt - 1
#endif
    issue_stage((kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  compute_stage((KT - 1) & 1);
#ifdef DVSG_STAMPS
  asm volatile("s_nop 0" ::: "memory");
  const unsigned long long t_loop = W16_STAMP();
  st[3] = t_loop - t_first;   // the K loop, waits included
#endif

  // ---- epilogue: 64 output channels at a time, rounds of 128 pixels through a [128][64] float32 transpose
  // (SPLIT: one channel half, hi + 2^-11 lo; plain: two channel halves, the accumulators as they are)
  float *Cs = reinterpret_cast<float *>(lds);
  const int col4 = tid & 15, row0 = tid >> 4;   // 16 float4 per row, 32 rows per pass
  constexpr int NHALF = SPLIT ? 1 : 2;
#pragma unroll
  for (int ch = 0; ch < NHALF; ++ch) {
  const int n = nt * (SPLIT ? 64 : 128) + 64 * ch + 4 * col4;
  const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
  for (int rho = 0; rho < 2; ++rho) {
    float4 rv[4];
    if (RES != 0) {   // residual of this round's rows: in flight under the two barriers and the transpose
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mr = m0 + 128 * rho + row0 + 32 * i;
        const int m = mr < p.M ? mr : p.M - 1;
        size_t roff;
        if (RES == 1) {
          roff = (size_t)m * p.Cout + n;
        } else {  // slim `subsample`: shortcut = x[:, ::s, ::s, :]
          const int wo = m % p.Wo;
          const int t = m / p.Wo;
          const int ho = t % p.Ho;
          const int b = t / p.Ho;
          roff = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.Cout + n;
        }
        rv[i] = load4(p.res + roff);
      }
    }
#ifdef DVSG_STAMPS
    const unsigned long long e0 = W16_STAMP();
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the stage buffers (first round) / the previous round's rows have been read
    asm volatile("" ::: "memory");
#ifdef DVSG_STAMPS
    const unsigned long long e1 = W16_STAMP();
    st[4] += e1 - e0;   // barrier before the transpose (the other waves' MFMAs / reads)
#endif
    if ((wm >> 1) == rho) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          Cs[((wm & 1) * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 64 + wn * 32 + r] =
              SPLIT ? acc_hi[mi][q] + acc_lo[mi][q] * kLoScale : (ch == 0 ? acc_hi[mi][q] : acc_lo[mi][q]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef DVSG_STAMPS
    const unsigned long long e2 = W16_STAMP();
    st[5] += e2 - e1;   // transpose stores + barrier
    if (RES != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long e3 = W16_STAMP();
    st[6] += e3 - e2;   // residual rows not there yet (and, second round, the first round's stores still in flight)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 32 * i;
      const int m = m0 + 128 * rho + row;
      if (m < p.M) {
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * 64 + 4 * col4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (RES != 0) {
          v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
        }
        if (RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        store4(p.y + (size_t)m * p.Cout + n, v);
      }
    }
  }
  }
#ifdef DVSG_STAMPS
  const unsigned long long t_issued = W16_STAMP();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t_end = W16_STAMP();
  if (p.stamp && tid == 0 && blockIdx.x < 65536) {
    unsigned long long *o = g_w16_stamps + (size_t)blockIdx.x * 8;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = st[4]; o[5] = st[5]; o[6] = st[6];
    o[7] = ((t_end - t_begin) << 24) | ((t_end - t_issued) & 0xffffffull);   // lifetime | drain of the last stores
  }
#endif
}

// ----------------------------------------------------------------------------------------
// The same tile with the ACTIVATION rows staged 128 bytes (64 k) at a time: a pixel's 64-byte piece is half a 128-byte
// line, and the kernel above asks for a line's halves in different stages -- one stage apart in a 1x1 layer, nine in a
// 3x3 layer (isolated loop, tools/pieces_loop_bench.hip: 660 with contiguous stages, 490-575 / 335 with such rows).  Here
// an activation stage is a SUPER-stage of 64 k (whole lines: 256 rows x 128 bytes, two buffers = 64 KB) consumed as two
// half-stages against two of the 8 KB weight stages (packed: launch_pack_wide16, order 1), still one barrier per 32 k
// and 80 KB per workgroup, two per CU.  K order of a 3x3 layer: 64-channel chunk outer, taps inner, the chunk's two
// halves innermost.  Packed weights only.
// SPLIT = false: the 128 weight rows of a tile are 128 output channels of a PLAIN float16 weight matrix [Cout][K] (layers
// whose weights do not need the lo piece -- locnet.hip's pair policy): the "lo" accumulators are simply the tile's second
// 64 channels, same loop, half the bytes and MFMAs per output channel.
template <int KS, bool RELU, int RES, bool SPLIT = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv_wide16a_kernel(ConvWide16Dev p) {
  constexpr int NW = 8, MI = 2;
  constexpr int AG = WBM / 8 / NW;    // 4 LDS-DMA instructions (8 rows of 128 bytes each) per wave and super-stage
  constexpr int AROWB = 2 * WROWB;    // 128
  constexpr float kLoScale = 1.0f / 2048.0f;
  __shared__ __attribute__((aligned(16))) char lds[2 * WBM * AROWB + 2 * WBN * WROWB];   // 64 + 16 KiB
  char *As = lds;
  char *Bs = lds + 2 * WBM * AROWB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = lane >> 3, lpos = lane & 7;   // activation DMA: 8 rows x 8 chunks per instruction
  const int sw = (r >> 2) & 3;   // weight rows (64 bytes): chunk ^ (row >> 2) & 3
  const int swa = (r >> 1) & 7;  // activation rows (128 bytes): chunk ^ (row >> 1) & 7
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  // nt fastest: the n-tiles of a pixel tile run back to back on one XCD and re-read its activations out of L2
  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * WBM;

  long a_off[AG];
  unsigned a_mask[AG];
  const bool dense = KS == 1 && p.stride == 1;   // a 1x1 / stride 1 layer is a row-major GEMM (conv_gemm.hip)
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    const int chunk = lpos ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const int mm = m < p.M ? m : 0;
    if (dense) {
      a_off[i] = (long)mm * p.Cin + 8 * chunk;
      a_mask[i] = m < p.M ? 0x11u : 0u;
      continue;
    }
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + 8 * chunk;
    unsigned mk = 0;
    if (m < p.M) {
#pragma unroll
      for (int q = 0; q < KS; ++q) {
        if (hi0 + q >= 0 && hi0 + q < p.H) mk |= 1u << q;
        if (wi0 + q >= 0 && wi0 + q < p.W) mk |= 16u << q;
      }
    }
    a_mask[i] = mk;
  }
  // weight half-stages: 8 KB each, contiguous in the packed copy, lane l of wave w fetching 16 bytes at 1024 w + 16 l
  const _Float16 *wpk = p.wtp + (size_t)nt * p.K * WBN + wave * 512 + lane * 8;
  auto issue_b = [&](int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gptr_t)wpk, (lptr_t)(Bs + (buf * WBN + 16 * wave) * WROWB), 16, 0, 0);
    wpk += WBN * WBKE;
  };
  // activation super-stages: 64 channels of one tap
  int s_kh = 0, s_kw = 0, s_c0 = 0;
  auto issue_a = [&](int buf) __attribute__((always_inline)) {
    const _Float16 *xa = p.x + ((long)s_kh * p.W + s_kw) * p.Cin + s_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16w);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * WBM + 8 * (wave + NW * i)) * AROWB), 16, 0, 0);
    }
    if (KS > 1) {
      if (++s_kw == KS) {
        s_kw = 0;
        if (++s_kh == KS) {
          s_kh = 0;
          s_c0 += 2 * WBKE;
        }
      }
    } else {
      s_c0 += 2 * WBKE;
    }
  };
  floatx16 acc_hi[MI], acc_lo[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_hi[mi][q] = acc_lo[mi][q] = 0.f;
  auto compute_half = [&](int hs) __attribute__((always_inline)) {   // half-stage hs: super-stage hs / 2, weights hs
    const char *a_base = As + ((((hs >> 1) & 1) * WBM + wm * 64 + r)) * AROWB;
    const char *b_base = Bs + ((hs & 1) * WBN + wn * 32 + r) * WROWB;
    const int half = hs & 1;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int co = 16 * ((2 * t + h) ^ sw);
      const int ca = 16 * ((4 * half + 2 * t + h) ^ swa);
      const halfx8 bh = *reinterpret_cast<const halfx8 *>(b_base + co);
      const halfx8 bl = *reinterpret_cast<const halfx8 *>(b_base + 64 * WROWB + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        const halfx8 a = *reinterpret_cast<const halfx8 *>(a_base + mi * 32 * AROWB + ca);
        acc_hi[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc_hi[mi], 0, 0, 0);
        acc_lo[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc_lo[mi], 0, 0, 0);
      }
    }
  };

  const int KT = p.K / WBKE;   // half-stages: even, >= 2 (Cin % 64 == 0)
  issue_a(0);
  issue_b(0);
  issue_b(1);
  for (int hs = 0; hs < KT; ++hs) {
    __syncthreads();   // vmcnt(0): what half-stage hs needs has landed; everyone has read half-stage hs - 1
    if (hs >= 1 && hs + 1 < KT) issue_b((hs + 1) & 1);              // into the buffer of half-stage hs - 1
    if ((hs & 1) == 0 && hs + 2 < KT) issue_a(((hs >> 1) + 1) & 1);   // into the buffer of super-stage hs / 2 - 1
    __builtin_amdgcn_sched_barrier(0);
    compute_half(hs);
    __builtin_amdgcn_sched_barrier(0);
  }

  // ---- epilogue: 64 output channels at a time, rounds of 128 pixels through a [128][64] float32 transpose
  // (SPLIT: one channel half, hi + 2^-11 lo; plain: two channel halves, the accumulators as they are)
  float *Cs = reinterpret_cast<float *>(lds);
  const int col4 = tid & 15, row0 = tid >> 4;   // 16 float4 per row, 32 rows per pass
  constexpr int NHALF = SPLIT ? 1 : 2;
#pragma unroll
  for (int ch = 0; ch < NHALF; ++ch) {
  const int n = nt * (SPLIT ? 64 : 128) + 64 * ch + 4 * col4;
  const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
  for (int rho = 0; rho < 2; ++rho) {
    float4 rv[4];
    if (RES != 0) {   // residual of this round's rows: in flight under the two barriers and the transpose
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mr = m0 + 128 * rho + row0 + 32 * i;
        const int m = mr < p.M ? mr : p.M - 1;
        size_t roff;
        if (RES == 1) {
          roff = (size_t)m * p.Cout + n;
        } else {  // slim `subsample`: shortcut = x[:, ::s, ::s, :]
          const int wo = m % p.Wo;
          const int t = m / p.Wo;
          const int ho = t % p.Ho;
          const int b = t / p.Ho;
          roff = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.Cout + n;
        }
        rv[i] = load4(p.res + roff);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the stage buffers (first round) / the previous round's rows have been read
    asm volatile("" ::: "memory");
    if ((wm >> 1) == rho) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          Cs[((wm & 1) * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 64 + wn * 32 + r] =
              SPLIT ? acc_hi[mi][q] + acc_lo[mi][q] * kLoScale : (ch == 0 ? acc_hi[mi][q] : acc_lo[mi][q]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 32 * i;
      const int m = m0 + 128 * rho + row;
      if (m < p.M) {
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * 64 + 4 * col4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (RES != 0) {
          v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
        }
        if (RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        store4(p.y + (size_t)m * p.Cout + n, v);
      }
    }
  }
  }
}

// ----------------------------------------------------------------------------------------
// 3x3, stride 1: the three taps of a kernel ROW from ONE staged run of pixels.  conv_wide16a_kernel stages 256 pixels x
// 64 channels nine times per 64-channel chunk -- once per tap -- although the taps (kh, 0), (kh, 1), (kh, 2) read the same
// input row shifted by one pixel.  Here a super-stage is (64-channel chunk, kernel row kh): LDS rows 0..255 hold the
// pixels m0 .. m0 + 255 (linear NHWC index) of input row ho + kh - 1 at their OWN column, and output pixel p multiplies
// rows p - 1, p, p + 1 for kw = 0, 1, 2 -- a third of the activation traffic and DMA instructions, six weight half-stages
// (kw x two 32-k halves) per activation stage.  Consequences:
//   * a tile computes the 254 pixels m0 + 1 .. m0 + 254 (tiles overlap by two; rows 0 and 255 are only ever neighbours):
//     the two buffers stay 2 x 32 KB and the workgroup 80 KB, two per CU, at 0.8 % more MFMAs;
//   * the linear neighbour of a pixel in the first / last column of the image is not its spatial neighbour: those lanes'
//     fragments are zeroed for kw = 0 / kw = 2 (what the padding would have supplied);
//   * K order: 64-channel chunk, kh, kw, half (launch_pack_wide16 order 2).
// ----------------------------------------------------------------------------------------
constexpr int WHM = WBM - 2;   // output pixels per tile

template <bool RELU, int RES, bool SPLIT = true>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv_wide16h_kernel(ConvWide16Dev p) {
  constexpr int NW = 8, MI = 2;
  constexpr int AG = WBM / 8 / NW;    // 4 LDS-DMA instructions (8 rows of 128 bytes each) per wave and super-stage
  constexpr int AROWB = 2 * WROWB;    // 128
  constexpr float kLoScale = 1.0f / 2048.0f;
  __shared__ __attribute__((aligned(16))) char lds[2 * WBM * AROWB + 2 * WBN * WROWB];   // 64 + 16 KiB
  char *As = lds;
  char *Bs = lds + 2 * WBM * AROWB;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  const int lrow = lane >> 3, lpos = lane & 7;
  const int sw = (r >> 2) & 3;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * WHM - 1;   // LDS row i <-> pixel m0 + i; outputs for i = 1 .. 254

  // staging: LDS row `row` of a super-stage (kh) is pixel m0 + row of the image row above / at / below its own
  int a_pix[AG];         // (b, ho - 1, wo) as one index into [B][H][W] (may be negative: the row above the image)
  unsigned a_mask[AG];   // bit kh: that image row exists
  int a_chunk[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = 8 * (wave + NW * i) + lrow;
    a_chunk[i] = 8 * (lpos ^ ((row >> 1) & 7));
    const int m = m0 + row;
    const bool in = m >= 0 && m < p.M;
    const int mm = in ? m : 0;
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    a_pix[i] = (b * p.H + ho - 1) * p.W + wo;
    unsigned mk = 0;
    if (in) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (ho - 1 + q >= 0 && ho - 1 + q < p.H) mk |= 1u << q;
    }
    a_mask[i] = mk;
  }
  const _Float16 *wpk = p.wtp + (size_t)nt * p.K * WBN + wave * 512 + lane * 8;
  auto issue_b = [&](int buf) __attribute__((always_inline)) {
    __builtin_amdgcn_global_load_lds((gptr_t)wpk, (lptr_t)(Bs + (buf * WBN + 16 * wave) * WROWB), 16, 0, 0);
    wpk += WBN * WBKE;
  };
  int s_kh = 0, s_c0 = 0;
  auto issue_a = [&](int buf) __attribute__((always_inline)) {
    const _Float16 *xa = p.x + (long)s_kh * p.W * p.Cin + s_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + ((long)a_pix[i] * p.Cin + a_chunk[i])) : static_cast<const void *>(&g_zero16w);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * WBM + 8 * (wave + NW * i)) * AROWB), 16, 0, 0);
    }
    if (++s_kh == 3) {
      s_kh = 0;
      s_c0 += 2 * WBKE;
    }
  };
  // fragment rows of this lane: output pixel 64 wm + 32 mi + r reads LDS rows (that) + kw - 1, kept inside the buffer for
  // the two pixels nobody stores; its byte offset and chunk swizzle per (mi, kw), and whether that neighbour is real
  const int pr0 = wm * 64 + r;   // (+ 32 mi)
  unsigned a_zero = 0;   // bit 3 mi + kw: the tap falls outside the image row
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int m = m0 + pr0 + 32 * mi;
    const int mm = m >= 0 && m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    if (wo == 0) a_zero |= 1u << (3 * mi);
    if (wo == p.W - 1) a_zero |= 4u << (3 * mi);
  }
  floatx16 acc_hi[MI], acc_lo[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_hi[mi][q] = acc_lo[mi][q] = 0.f;
  // half-stage: activation buffer abuf, tap kw, half of the 64 channels, weight buffer bbuf
  auto compute_half = [&](int abuf, int kw, int half, int bbuf) __attribute__((always_inline)) {
    const char *a_base = As + abuf * WBM * AROWB;
    const char *b_base = Bs + (bbuf * WBN + wn * 32 + r) * WROWB;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int co = 16 * ((2 * t + h) ^ sw);
      const halfx8 bh = *reinterpret_cast<const halfx8 *>(b_base + co);
      const halfx8 bl = *reinterpret_cast<const halfx8 *>(b_base + 64 * WROWB + co);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) {
        int rr = pr0 + 32 * mi + kw - 1;
        asm volatile("" : "+v"(rr));   // (recomputed every time: hoisted out of the K loop the 24 fragment addresses spill)
        rr = rr < 0 ? 0 : (rr > WBM - 1 ? WBM - 1 : rr);
        halfx8 a = *reinterpret_cast<const halfx8 *>(a_base + rr * AROWB + 16 * ((4 * half + 2 * t + h) ^ ((rr >> 1) & 7)));
        if ((a_zero >> (3 * mi + kw)) & 1u) a = halfx8{0, 0, 0, 0, 0, 0, 0, 0};
        acc_hi[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc_hi[mi], 0, 0, 0);
        acc_lo[mi] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc_lo[mi], 0, 0, 0);
      }
    }
  };

  const int NS = 3 * (p.Cin / 64);   // super-stages (64-channel chunk, kh), six half-stages each
  issue_a(0);
  issue_b(0);
  issue_b(1);
  int hs = 0;
  const int KT = 6 * NS;
  for (int S = 0; S < NS; ++S) {
#pragma unroll
    for (int sub = 0; sub < 6; ++sub, ++hs) {
      __syncthreads();   // vmcnt(0): what this half-stage needs has landed; everyone has read the previous one
      if (hs >= 1 && hs + 1 < KT) issue_b((hs + 1) & 1);
      if (sub == 0 && S + 1 < NS) issue_a((S + 1) & 1);   // into the buffer of super-stage S - 1
      __builtin_amdgcn_sched_barrier(0);
      compute_half(S & 1, sub >> 1, sub & 1, hs & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: as in conv_wide16_kernel, for the tile's rows 1 .. 254
  float *Cs = reinterpret_cast<float *>(lds);
  const int col4 = tid & 15, row0 = tid >> 4;
  constexpr int NHALF = SPLIT ? 1 : 2;
#pragma unroll
  for (int ch = 0; ch < NHALF; ++ch) {
  const int n = nt * (SPLIT ? 64 : 128) + 64 * ch + 4 * col4;
  const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll
  for (int rho = 0; rho < 2; ++rho) {
    float4 rv[4];
    if (RES != 0) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mr = m0 + 128 * rho + row0 + 32 * i;
        const int m = mr < 0 ? 0 : (mr < p.M ? mr : p.M - 1);
        size_t roff;
        if (RES == 1) {
          roff = (size_t)m * p.Cout + n;
        } else {
          const int wo = m % p.Wo;
          const int t = m / p.Wo;
          const int ho = t % p.Ho;
          const int b = t / p.Ho;
          roff = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.Cout + n;
        }
        rv[i] = load4(p.res + roff);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    if ((wm >> 1) == rho) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          Cs[((wm & 1) * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 64 + wn * 32 + r] =
              SPLIT ? acc_hi[mi][q] + acc_lo[mi][q] * kLoScale : (ch == 0 ? acc_hi[mi][q] : acc_lo[mi][q]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int row = row0 + 32 * i;
      const int pr = 128 * rho + row;
      const int m = m0 + pr;
      if (pr >= 1 && pr <= WHM && m < p.M) {
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * 64 + 4 * col4);
        v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
        if (RES != 0) {
          v.x += rv[i].x; v.y += rv[i].y; v.z += rv[i].z; v.w += rv[i].w;
        }
        if (RELU) {
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
        }
        store4(p.y + (size_t)m * p.Cout + n, v);
      }
    }
  }
  }
}

template <bool SPLIT>
int launch_h(const ConvWide16Dev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid(d.mtiles * d.ntiles), block(512);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_wide16h_kernel<R, Q, SPLIT>), grid, block, 0, s, d)
  if (relu) {
    if (res == 0) DVSG_LAUNCH(true, 0);
    else if (res == 1) DVSG_LAUNCH(true, 1);
    else DVSG_LAUNCH(true, 2);
  } else {
    if (res == 0) DVSG_LAUNCH(false, 0);
    else if (res == 1) DVSG_LAUNCH(false, 1);
    else DVSG_LAUNCH(false, 2);
  }
#undef DVSG_LAUNCH
  return check_launch("conv_wide16h_kernel");
}

template <int KS, bool SPLIT>
int launch_ks(const ConvWide16Dev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid(d.mtiles * d.ntiles), block(512);
#define DVSG_LAUNCH(R, Q)                                                                           \
  do {                                                                                              \
    if (d.arows) hipLaunchKernelGGL((conv_wide16a_kernel<KS, R, Q, SPLIT>), grid, block, 0, s, d);   \
    else hipLaunchKernelGGL((conv_wide16_kernel<KS, R, Q, SPLIT>), grid, block, 0, s, d);            \
  } while (0)
  if (relu) {
    if (res == 0) DVSG_LAUNCH(true, 0);
    else if (res == 1) DVSG_LAUNCH(true, 1);
    else DVSG_LAUNCH(true, 2);
  } else {
    if (res == 0) DVSG_LAUNCH(false, 0);
    else if (res == 1) DVSG_LAUNCH(false, 1);
    else DVSG_LAUNCH(false, 2);
  }
#undef DVSG_LAUNCH
  return check_launch("conv_wide16_kernel");
}

// Packed weights for the kernel above: [tile of 128 rows][stage][row][chunk position][8 halves]; stage s of a tile is the
// 32 k the kernel visits s-th (channel chunk outer, taps inner), position c' of row r holds chunk c' ^ ((r >> 2) & 3).
// order 1 (conv_wide16a_kernel): stage s = half (s & 1) of the 64-channel chunk of super-stage s / 2 (64-channel chunk outer,
// taps inner).
__global__ void pack_wide16_kernel(const _Float16 *__restrict__ wt, _Float16 *__restrict__ out, int K, int Cin, int taps,
                                   int order, size_t nchunks) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const int KT = K / WBKE;
  const int pos = (int)(i & 3), row = (int)((i >> 2) & 127);
  const size_t ts = i >> 9;   // tile * KT + stage
  const int st = (int)(ts % KT);
  const size_t tile = ts / KT;
  const int chunk = pos ^ ((row >> 2) & 3);
  int k0;
  if (taps == 1) k0 = st * WBKE;
  else if (order == 0) k0 = (st % taps) * Cin + (st / taps) * WBKE;
  else if (order == 1) k0 = ((st >> 1) % taps) * Cin + ((st >> 1) / taps) * 2 * WBKE + (st & 1) * WBKE;
  else {   // order 2 (conv_wide16h_kernel): 64-channel chunk, kh, kw, half
    const int S = st / 6, sub = st % 6;
    k0 = ((S % 3) * 3 + (sub >> 1)) * Cin + (S / 3) * 2 * WBKE + (sub & 1) * WBKE;
  }
  const int k = k0 + 8 * chunk;
  *reinterpret_cast<floatx4 *>(out + i * 8) = *reinterpret_cast<const floatx4 *>(wt + (tile * 128 + row) * K + k);
}

}  // namespace

int g_wide16_packed = 1;   // dvsg_debug_set_option("wide16_packed", 0): weight stages fetched from the [rows][K] layout
void set_wide16_packed(int v) { g_wide16_packed = v; }
int g_wide16_hreuse = 1;   // dvsg_debug_set_option("wide16_hreuse", 0): 3x3 stride-1 layers through conv_wide16a_kernel
void set_wide16_hreuse(int v) { g_wide16_hreuse = v; }
int g_wide16_arows = 1;    // dvsg_debug_set_option("wide16_arows", v): 0 = 64-byte activation rows everywhere (conv_wide16_kernel),
                           // 2 = 128-byte rows for K = 128 too (tests)
void set_wide16_arows(int v) { g_wide16_arows = v; }
int wide16_pack_order() { return g_wide16_arows ? 1 : 0; }

size_t wide16_packed_bytes(int rows, int Cin, int ksize) { return (size_t)rows * ksize * ksize * Cin * sizeof(_Float16); }

// wt: `rows` weight rows of K = ksize^2 Cin float16 (stacked hi / lo rows, or plain ones), rows % 128 == 0, Cin % 64 == 0
int launch_pack_wide16(const void *wt, void *out, int rows, int Cin, int ksize, int order, hipStream_t s) {
  DVSG_REQUIRE(wt && out && rows % 128 == 0 && Cin % 64 == 0 && (ksize == 1 || ksize == 3), "pack_wide16: bad arguments");
  const int K = ksize * ksize * Cin;
  const size_t nchunks = (size_t)rows * K / 8;
  hipLaunchKernelGGL(pack_wide16_kernel, dim3((unsigned)((nchunks + 255) / 256)), dim3(256), 0, s, static_cast<const _Float16 *>(wt),
                     static_cast<_Float16 *>(out), K, Cin, ksize * ksize, order, nchunks);
  return check_launch("pack_wide16_kernel");
}

namespace {
}  // namespace

#ifdef DVSG_STAMPS
int g_w16_stamp_sel = 0;   // KS * 100000000 + Cin * 10000 + Cout of the launches that record
extern "C" int dvsg_debug_wide16_stamp_select(int sel) {   // and forget the stamps recorded so far
  g_w16_stamp_sel = sel;
  void *sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_w16_stamps)) != hipSuccess) return -3;
  return hipMemset(sym, 0, sizeof(unsigned long long) * 8 * 65536) == hipSuccess && hipDeviceSynchronize() == hipSuccess ? 0 : -3;
}
extern "C" int dvsg_debug_read_wide16_stamps(void *host, size_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_w16_stamps), bytes) == hipSuccess ? 0 : -3;
}
#endif

// p: a float16 layer -- stacked [hi | lo] weights (p.wsplit, Cout % 64 == 0) or plain ones (Cout % 128 == 0) -- with
// Cin % 64 == 0; the caller has opened the ProfScope
int launch_conv_wide16(const ConvGemm &p, hipStream_t s) {
  const long M = (long)p.B * p.Ho * p.Wo;
  ConvWide16Dev d;
  d.x = static_cast<const _Float16 *>(p.x); d.wt = static_cast<const _Float16 *>(p.wt);
  d.res = static_cast<const _Float16 *>(p.res); d.bias = p.bias; d.y = static_cast<_Float16 *>(p.y);
  // (K = 128 -- two super-stages -- keeps the 64-byte rows: block 2's conv3 2.48 against 2.55 ms)
  d.arows = g_wide16_arows && p.wt_packed_a != nullptr && (g_wide16_arows == 2 || p.ksize * p.ksize * p.Cin >= 256);
  d.wtp = d.arows ? static_cast<const _Float16 *>(p.wt_packed_a) : g_wide16_packed ? static_cast<const _Float16 *>(p.wt_packed) : nullptr;
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride; d.pad = p.pad;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.M = (int)M;
  d.K = p.ksize * p.ksize * p.Cin;
  d.mtiles = (int)((M + WBM - 1) / WBM);
  d.ntiles = p.wsplit ? p.Cout / 64 : p.Cout / 128;
  const bool hreuse = g_wide16_hreuse && p.wt_packed_h != nullptr && p.ksize == 3 && p.stride == 1 && p.pad == 1 && p.H == p.Ho && p.W == p.Wo;
  if (hreuse) {
    d.wtp = static_cast<const _Float16 *>(p.wt_packed_h);
    d.mtiles = (int)((M + WHM - 1) / WHM);
    const int resh = !p.res ? 0 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
    return p.wsplit ? launch_h<true>(d, p.relu != 0, resh, s) : launch_h<false>(d, p.relu != 0, resh, s);
  }
#ifdef DVSG_STAMPS
  d.stamp = g_w16_stamp_sel == p.ksize * 100000000 + p.Cin * 10000 + p.Cout;
  if (d.stamp) std::fprintf(stderr, "wide16 stamps: ks %d Cin %d Cout %d M %ld tiles %d x %d res %d relu %d\n", p.ksize, p.Cin, p.Cout, M, d.mtiles, d.ntiles, p.res != nullptr, p.relu);
#endif
  const int res = !p.res ? 0 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  if (p.wsplit) return p.ksize == 3 ? launch_ks<3, true>(d, p.relu != 0, res, s) : launch_ks<1, true>(d, p.relu != 0, res, s);
  DVSG_REQUIRE(p.Cout % 128 == 0, "conv_wide16: plain weights need Cout %% 128 == 0, got %d", p.Cout);
  return p.ksize == 3 ? launch_ks<3, false>(d, p.relu != 0, res, s) : launch_ks<1, false>(d, p.relu != 0, res, s);
}

}  // namespace dvsg
