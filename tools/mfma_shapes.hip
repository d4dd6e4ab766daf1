// Diagnostic: does the f32 MFMA SHAPE change what this device sustains?  (MI355X_MICROARCH.md, DVFS
// give-back item 7: for bf16 the 16x16 shape holds a higher clock than the 32x32 one.)  Bare loops of
// v_mfma_f32_32x32x2_f32 and v_mfma_f32_16x16x4_f32 on pseudo-random operands, 1 / 2 / 4 waves per
// SIMD, long enough (>= 0.3 s each) for the clock governor to settle; reports TFLOP/s and the
// in-kernel shader clock (delta s_memtime / delta s_memrealtime x 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_shapes.hip -o build/mfma_shapes && build/mfma_shapes
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float rnd(unsigned &s) {
  s = s * 1664525u + 1013904223u;
  return (float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void mfma_loop(float *out, unsigned long long *stamps, int iters, unsigned seed) {
  unsigned s = seed ^ (threadIdx.x * 2654435761u) ^ (blockIdx.x * 40503u);
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = rnd(s);
    b[i] = rnd(s);
  }
  floatx16 acc32[4];
  floatx4 acc16[16];
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) acc32[j][q] = 0.f;
  for (int j = 0; j < 16; ++j)
    for (int q = 0; q < 4; ++q) acc16[j][q] = 0.f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
    if (SHAPE == 32) {
      // 4 accumulators x 8 k-steps = 32 MFMAs of 4096 FLOP
#pragma unroll
      for (int k = 0; k < 8; ++k)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc32[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[k], b[(k + j) & 7], acc32[j], 0, 0, 0);
    } else {
      // 16 accumulators x 4 k-steps = 64 MFMAs of 2048 FLOP
#pragma unroll
      for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < 16; ++j) acc16[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[(k + j) & 7], b[(k + 2 * j) & 7], acc16[j], 0, 0, 0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float sum = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) sum += acc32[j][q];
  for (int j = 0; j < 16; ++j)
    for (int q = 0; q < 4; ++q) sum += acc16[j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  float *out;
  unsigned long long *stamps;
  hipMalloc(&out, 256 * 4096 * sizeof(float));
  hipMalloc(&stamps, 2 * 4096 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int shape : {32, 16, 32, 16}) {
    for (int blocks_per_cu : {1, 2, 4}) {
      const int blocks = 256 * blocks_per_cu;
      const int iters = 60000 / blocks_per_cu;   // ~0.3-0.4 s per launch
      float ms = 0;
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (shape == 32)
          hipLaunchKernelGGL(mfma_loop<32>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 12345u + rep);
        else
          hipLaunchKernelGGL(mfma_loop<16>, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 12345u + rep);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
      }
      std::vector<unsigned long long> h(2 * blocks);
      hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double clk = 0;
      for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
      clk /= blocks;
      const double flops = (double)blocks * 4 /*waves*/ * iters * 32.0 * 4096.0;
      printf("shape %2d  waves/SIMD=%d  %.1f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles per 4096 FLOP per SIMD %.1f\n",
             shape, blocks_per_cu, ms, flops / ms / 1e9, clk, (double)h[0] / ((double)iters * 32 * blocks_per_cu));
    }
  }
  return 0;
}
