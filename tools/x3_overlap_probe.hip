// Does the VALU work of the f32x3 activation split run in the shadow of the bf16 MFMAs?  One wave per SIMD-slot loop, no
// memory: per iteration 24 x v_mfma_f32_32x32x16_bf16 (a 64 x 64 wave tile's fragment pair) and the split of the NEXT
// fragments (84 VALU), in three program orders: MFMAs only; split first, then the MFMAs; one MFMA, then 4 VALU of the split.
//   hipcc -O3 --offload-arch=gfx950 tools/x3_overlap_probe.hip -o build/x3_overlap_probe && build/x3_overlap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef float floatx2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void split8(const floatx4 lo, const floatx4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
  const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const floatx2 v = {x[i], x[i + 1]};
    const bf16x2 a = __builtin_convertvector(v, bf16x2);
    const floatx2 r = v - __builtin_convertvector(a, floatx2);
    const bf16x2 b = __builtin_convertvector(r, bf16x2);
    const floatx2 r2 = r - __builtin_convertvector(b, floatx2);
    const bf16x2 c = __builtin_convertvector(r2, bf16x2);
    p1[i] = a.x; p1[i + 1] = a.y; p2[i] = b.x; p2[i + 1] = b.y; p3[i] = c.x; p3[i + 1] = c.y;
  }
}

template <int MODE>  // 0: MFMAs only, 1: split then MFMAs, 2: interleaved
__global__ __launch_bounds__(256) void loop(float *out, int iters, float seed) {
  const int lane = threadIdx.x & 63;
  floatx16 acc[2][2] = {};
  bf16x8 b[3][2];
  for (int p = 0; p < 3; ++p)
    for (int n = 0; n < 2; ++n)
      for (int i = 0; i < 8; ++i) b[p][n][i] = (__bf16)(0.5f + 0.01f * lane + p + n);
  floatx4 rlo[2], rhi[2];
  for (int m = 0; m < 2; ++m) {
    rlo[m] = floatx4{seed + lane, seed * 2, seed * 3, seed * 5 + m};
    rhi[m] = floatx4{seed * 7, seed + m, seed * 11, seed * 13};
  }
  bf16x8 p[2][3];
  for (int m = 0; m < 2; ++m) split8(rlo[m], rhi[m], p[m][0], p[m][1], p[m][2]);
  for (int it = 0; it < iters; ++it) {
    bf16x8 q[2][3];
    if (MODE != 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        rlo[m] += acc[m][0][0] * 1e-30f;   // next fragments depend on nothing but old accumulator values
        split8(rlo[m], rhi[m], q[m][0], q[m][1], q[m][2]);
      }
    }
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int c = 0; c < 6; ++c)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const int pa = c == 0 ? 2 : c == 1 || c == 4 || c == 5 ? 0 : 1, pb = c == 0 || c == 3 || c == 5 ? 0 : c == 1 ? 2 : 1;
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p[m][pa], b[pb][n], acc[m][n], 0, 0, 0);
        }
    if (MODE != 0) {
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int k = 0; k < 3; ++k) p[m][k] = q[m][k];
    }
    if (MODE == 1) {
      __builtin_amdgcn_sched_group_barrier(0x002, 200, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 24, 0);
    }
    if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 24; ++i) {
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
  for (int m = 0; m < 2; ++m)
    for (int n = 0; n < 2; ++n)
      for (int q = 0; q < 16; ++q) s += acc[m][n][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out) {
  const int iters = 10000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int per_cu : {1, 2, 4}) {
    const int blocks = 256 * per_cu;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.37f);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double mfma = (double)blocks * 4 * iters * 24;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.0f executed bf16 TFLOP/s (%.0f float32-equivalent)\n", name, per_cu, ms,
           mfma * 32768 / ms / 1e9, mfma * 32768 / 6 / ms / 1e9);
  }
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 4096 * sizeof(float));
  run<0>("24 MFMA", out);
  run<1>("split, then 24 MFMA", out);
  run<2>("1 MFMA + 4 VALU, x 24", out);
  return 0;
}
