// Diagnostic: what HBM gives the epilogue traffic of a 1x1 "expand" layer (read a residual row piece, write an output row
// piece, float16 NHWC with C = 512 / 1024 / 2048 channels) depending on the ORDER in which a workgroup visits its pieces:
//   P0  contiguous: thread t copies 16 bytes at 16 t (the ceiling: a streaming read + write)
//   P1  group-major, the conv kernels' order: a workgroup owns 192 pixels and walks the channel groups of 64; per group
//       every wave moves 128 bytes of each of its 32 pixels (16 bytes per lane, 4 passes of 16 pixels x 32 channels)
//   P2  the same with groups of 128 channels (256 bytes per pixel and visit)
//   P3  pixel-major: a wave finishes all channels of 4 pixels (whole rows) before it moves on
//   P4  group-major like P1, but one instruction covers 8 pixels x 128 bytes (whole lines) instead of 16 pixels x 64 bytes
//   P5  group-major, 128 channels per visit, one instruction = 4 pixels x 256 bytes
//   P6  group-major, 64 channels per visit, 8 BYTES per lane: one instruction = 4 pixels x 128 bytes (conv_wide16_kernel's
//       epilogue: 16 lanes x 4 float16 channels per pixel)
// Each also with the read only (R) and the write only (W).  Prints TB/s of bytes moved.
//   hipcc -O3 --offload-arch=gfx950 tools/hbm_pattern_bench.hip -o build/hbm_pattern_bench && build/hbm_pattern_bench
#include <hip/hip_runtime.h>
#include <cstdio>

typedef unsigned uintx4 __attribute__((ext_vector_type(4)));

template <int MODE, bool RD, bool WR>
__global__ __launch_bounds__(384) void pattern_kernel(const uintx4 *__restrict__ res, uintx4 *__restrict__ y, int M, int C, int gw) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = C / 8;   // 16-byte pieces per pixel row
  if (MODE == 0) {
    const size_t n = (size_t)M * c16;
    for (size_t i = (size_t)blockIdx.x * 384 + tid; i < n; i += (size_t)gridDim.x * 384) {
      uintx4 v = {1, 2, 3, 4};
      if (RD) v = res[i];
      v[0] += 1;
      if (WR) y[i] = v;
      else if (v[0] == 0x12345678u) y[0] = v;
    }
    return;
  }
  const int m0 = blockIdx.x * 192 + wave * 32;
  if (MODE == 1 || MODE == 2) {   // gw channels per visit: gw / 8 pieces per pixel; lanes: pixel lane / 4, piece lane % 4 (+ 4 per pass)
    const int ppv = gw / 8;       // pieces per pixel and visit: 8 or 16
    for (int g = 0; g < C / gw; ++g)
      for (int pass = 0; pass < 2 * ppv / 4; ++pass) {   // 16 pixels x 4 pieces per pass
        const int ph = pass / (ppv / 4), pc = pass % (ppv / 4);
        const int m = m0 + ph * 16 + (lane >> 2);
        if (m >= M) continue;
        const size_t i = (size_t)m * c16 + g * ppv + pc * 4 + (lane & 3);
        uintx4 v = {1, 2, 3, 4};
        if (RD) v = res[i];
        v[0] += 1;
        if (WR) y[i] = v;
        else if (v[0] == 0x12345678u) y[0] = v;
      }
  } else if (MODE == 4 || MODE == 5) {   // one instruction = (64 / lpp) pixels x lpp lanes x 16 bytes, lpp = gw / 8
    const int lpp = gw / 8;                // lanes per pixel: 8 (a 128-byte line) or 16
    const int ppi = 64 / lpp;              // pixels per instruction: 8 or 4
    for (int g = 0; g < C / gw; ++g)
      for (int pass = 0; pass < 32 / ppi; ++pass) {
        const int m = m0 + pass * ppi + lane / lpp;
        if (m >= M) continue;
        const size_t i = (size_t)m * c16 + g * lpp + lane % lpp;
        uintx4 v = {1, 2, 3, 4};
        if (RD) v = res[i];
        v[0] += 1;
        if (WR) y[i] = v;
        else if (v[0] == 0x12345678u) y[0] = v;
      }
  } else if (MODE == 6) {
    typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
    const uintx2 *res2 = reinterpret_cast<const uintx2 *>(res);
    uintx2 *y2 = reinterpret_cast<uintx2 *>(y);
    const int c8 = C / 4;   // 8-byte pieces per row
    for (int g = 0; g < C / 64; ++g)
      for (int pass = 0; pass < 8; ++pass) {
        const int m = m0 + pass * 4 + (lane >> 4);
        if (m >= M) continue;
        const size_t i = (size_t)m * c8 + g * 16 + (lane & 15);
        uintx2 v = {1, 2};
        if (RD) v = res2[i];
        v[0] += 1;
        if (WR) y2[i] = v;
        else if (v[0] == 0x12345678u) y2[0] = v;
      }
  } else {   // whole rows: 4 pixels at a time, 16 lanes per pixel, c16 / 16 pieces per lane
    for (int p4 = 0; p4 < 8; ++p4) {
      const int m = m0 + 4 * p4 + (lane >> 4);
      if (m >= M) continue;
      for (int k = (lane & 15); k < c16; k += 16) {
        const size_t i = (size_t)m * c16 + k;
        uintx4 v = {1, 2, 3, 4};
        if (RD) v = res[i];
        v[0] += 1;
        if (WR) y[i] = v;
        else if (v[0] == 0x12345678u) y[0] = v;
      }
    }
  }
}

template <int MODE, bool RD, bool WR>
void run(const char *name, const uintx4 *res, uintx4 *y, int M, int C, int gw) {
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int grid = MODE == 0 ? 256 * 16 : (M + 191) / 192;
  float best = 1e30f;
  for (int rep = 0; rep < 8; ++rep) {
    (void)hipEventRecord(e0, 0);
    hipLaunchKernelGGL((pattern_kernel<MODE, RD, WR>), dim3(grid), dim3(384), 0, 0, res, y, M, C, gw);
    (void)hipEventRecord(e1, 0);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep >= 2 && ms < best) best = ms;
  }
  const double bytes = (double)M * C * 2 * ((RD ? 1 : 0) + (WR ? 1 : 0));
  std::printf("  %-44s %8.3f ms  %6.2f TB/s  (%s)\n", name, best, bytes / best / 1e9, hipGetErrorString(hipGetLastError()));
}

int main() {
  const size_t bytes = (size_t)5 << 30;
  uintx4 *res, *y;
  if (hipMalloc(&res, bytes) != hipSuccess || hipMalloc(&y, bytes) != hipSuccess) return 1;
  (void)hipMemset(res, 1, bytes);
  (void)hipMemset(y, 0, bytes);
  for (int C : {512, 1024, 2048}) {
    const int M = (int)(((size_t)4 << 30) / ((size_t)C * 2));   // 4 GiB per tensor
    std::printf("C = %d channels, M = %d pixels (4 GiB read + 4 GiB written)\n", C, M);
    run<0, true, true>("P0 contiguous copy", res, y, M, C, 64);
    run<1, true, true>("P1 group-major, 64 channels per visit", res, y, M, C, 64);
    run<2, true, true>("P2 group-major, 128 channels per visit", res, y, M, C, 128);
    run<3, true, true>("P3 pixel-major (whole rows)", res, y, M, C, 64);
    run<4, true, true>("P4 group-major, 8 pixels x 128 B per instr", res, y, M, C, 64);
    run<5, true, true>("P5 group-major, 4 pixels x 256 B per instr", res, y, M, C, 128);
    run<6, true, true>("P6 group-major, 4 pixels x 128 B, 8 B per lane", res, y, M, C, 64);
    run<0, true, false>("P0 read only", res, y, M, C, 64);
    run<1, true, false>("P1 read only", res, y, M, C, 64);
    run<3, true, false>("P3 read only", res, y, M, C, 64);
    run<0, false, true>("P0 write only", res, y, M, C, 64);
    run<1, false, true>("P1 write only", res, y, M, C, 64);
    run<3, false, true>("P3 write only", res, y, M, C, 64);
  }
  return 0;
}
