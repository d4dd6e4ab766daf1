// Diagnostic: what does THIS device sustain on v_mfma_f32_32x32x2_f32 (the instruction the
// conv kernels are priced against)?  Bare MFMA loop, operands in registers, random-ish data,
// 1 / 2 / 4 waves per SIMD; reports TFLOP/s and the in-kernel shader clock
// (delta s_memtime / delta s_memrealtime x 100 MHz).
//   hipcc -O3 --offload-arch=gfx950 tools/mfma_peak.hip -o build/mfma_peak && build/mfma_peak
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float floatx16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(256) void mfma_loop(float *out, unsigned long long *stamps, int iters, float seed) {
  floatx16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;
  float a = seed + threadIdx.x * 0.001f, b = seed - threadIdx.x * 0.002f;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    a += 1e-6f;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int q = 0; q < 16; ++q) s += acc[j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) {
    stamps[2 * blockIdx.x] = t1 - t0;
    stamps[2 * blockIdx.x + 1] = r1 - r0;
  }
}

int main() {
  const int iters = 20000;
  float *out;
  unsigned long long *stamps;
  hipMalloc(&out, 256 * 4096 * sizeof(float));
  hipMalloc(&stamps, 2 * 4096 * sizeof(unsigned long long));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int blocks_per_cu : {1, 2, 4}) {
    const int blocks = 256 * blocks_per_cu;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(256), 0, 0, out, stamps, iters, 0.37f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms = 0;
      hipEventElapsedTime(&ms, e0, e1);
      std::vector<unsigned long long> h(2 * blocks);
      hipMemcpy(h.data(), stamps, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
      double clk = 0;
      for (int i = 0; i < blocks; ++i) clk += (double)h[2 * i] / (double)h[2 * i + 1] * 100.0;
      clk /= blocks;
      const double flops = (double)blocks * 4 /*waves*/ * iters * 4.0 * 4096.0;
      if (rep == 2)
        printf("waves/SIMD=%d  %.3f ms  %.1f TFLOP/s  in-kernel clock %.0f MHz  cycles/MFMA/SIMD %.1f\n",
               blocks_per_cu, ms, flops / ms / 1e9, clk,
               (double)h[0] / ((double)iters * 4 * blocks_per_cu));
    }
  }
  return 0;
}
