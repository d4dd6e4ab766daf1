// Diagnostic: does v_mfma_f32_32x32x16_f16 keep float16 SUBNORMAL inputs (or flush them to zero)?
//   hipcc -O3 --offload-arch=gfx950 tools/f16_subnormal_probe.hip -o build/f16_subnormal_probe && build/f16_subnormal_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef float floatx16 __attribute__((ext_vector_type(16)));
__global__ void probe(float *out, float aval, float bval) {
  halfx8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (_Float16)aval; b[i] = (_Float16)bval; }
  floatx16 c;
  for (int q = 0; q < 16; ++q) c[q] = 0.f;
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
  float *d, h;
  hipMalloc(&d, 4);
  const float vals[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1024.0f}, {5.96e-8f, 16384.0f}, {1.0e-6f, 1.0e-6f * 0 + 1000.f}};
  for (auto &v : vals) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, v[0], v[1]);
    hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("a = %.3e (f16 %s), b = %.3e: sum of 16 products = %.6e, expected %.6e\n", v[0], v[0] < 6.1e-5f ? "subnormal" : "normal", v[1],
           h, 16.0 * (double)(float)(_Float16)v[0] * (double)(float)(_Float16)v[1]);
  }
  return 0;
}
