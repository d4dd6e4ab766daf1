// Diagnostic for DESIGN.md section 8 (next steps): inner-loop rate of float32-equivalent products
// on the bf16 matrix cores.  Per iteration a wave does what one 16-k group of a 64 x 32 wave tile
// would cost: 16 float32 A values per lane come out of LDS (4 x ds_read_b128), are split into three
// bfloat16 pieces on the VALU, and meet pre-split B pieces in 6 x 2 v_mfma_f32_32x32x16_bf16.
// The float32 reference does the same k range with 16 v_mfma_f32_32x32x2_f32.
//   hipcc -O3 --offload-arch=gfx950 tools/split_mfma_peak.hip -o build/split_mfma_peak && build/split_mfma_peak
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
    p1[i] = a.x; p1[i + 1] = a.y;
    p2[i] = b.x; p2[i + 1] = b.y;
    p3[i] = c.x; p3[i + 1] = c.y;
  }
}

template <int MODE>  // 0: float32 MFMA, 1: split + 6 bf16 MFMAs, 2: 6 bf16 MFMAs without the split (upper bound)
__global__ __launch_bounds__(256) void loop(float *out, int iters) {
  __shared__ __attribute__((aligned(16))) float tile[64 * 36];
  for (int i = threadIdx.x; i < 64 * 36; i += 256) tile[i] = 0.001f * (float)((i * 37) % 101);
  __syncthreads();
  const int lane = threadIdx.x & 63;
  floatx16 acc[2] = {};
  bf16x8 b1, b2, b3;
  for (int i = 0; i < 8; ++i) {
    b1[i] = (__bf16)(0.5f + 0.01f * lane);
    b2[i] = (__bf16)(0.001f * i);
    b3[i] = (__bf16)(1e-5f);
  }
  const float bf = 0.5f + 0.01f * lane;
  bf16x8 q1 = b1, q2 = b2, q3 = b3;
  for (int it = 0; it < iters; ++it) {
    const float *row = tile + (lane & 31) * 36 + 8 * (lane >> 5) + 16 * (it & 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
      const floatx4 lo = *reinterpret_cast<const floatx4 *>(row + mi * 32 * 36);
      const floatx4 hi = *reinterpret_cast<const floatx4 *>(row + mi * 32 * 36 + 4);
      if (MODE == 0) {
        const float x[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[mi] = __builtin_amdgcn_mfma_f32_32x32x2f32(x[k], bf, acc[mi], 0, 0, 0);
      } else {
        bf16x8 p1, p2, p3;
        if (MODE == 1) {
          split8(lo, hi, p1, p2, p3);
        } else {
          p1 = q1; p2 = q2; p3 = q3;
          q1[0] = (__bf16)lo.x;  // keep the LDS reads alive
        }
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, b1, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, b2, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, b1, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p2, b2, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p1, b3, acc[mi], 0, 0, 0);
        acc[mi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(p3, b1, acc[mi], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 2; ++j)
    for (int q = 0; q < 16; ++q) s += acc[j][q];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char *name, float *out) {
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  for (int per_cu : {2, 4}) {
    const int blocks = 256 * per_cu;
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(loop<MODE>, dim3(blocks), dim3(256), 0, 0, out, iters);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    // algorithmic flops: 2 blocks of 32 x 32 outputs x 16 k x 2 per wave and iteration
    const double flops = (double)blocks * 4 * iters * 2.0 * 32 * 32 * 16 * 2;
    printf("%-34s waves/SIMD=%d  %.3f ms  %.1f float32-equivalent TFLOP/s\n", name, per_cu, ms, flops / ms / 1e9);
  }
}

int main() {
  float *out;
  hipMalloc(&out, 256 * 4096 * sizeof(float));
  run<0>("v_mfma_f32_32x32x2_f32 x 16", out);
  run<1>("split on VALU + 6 x bf16 MFMA", out);
  run<2>("6 x bf16 MFMA, pieces given", out);
  return 0;
}
