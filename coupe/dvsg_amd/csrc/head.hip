// Head of localizationNet: global average pool (slim `reduce_mean([1,2])`) + the four
// tensorlayer DenseLayers 2048 -> 2048 -> 1024 -> 512 -> 50 with leaky-ReLU 0.2
// (networks.py:31,36-44).  Tiny (6.8 MMAC per frame) and latency-bound: every stage writes
// deterministic split partial sums and the next stage folds "sum the partials, add bias,
// leaky-ReLU" into its operand load -- no atomics, bitwise reproducible.
#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

template <typename T>
__global__ __launch_bounds__(256) void avgpool_partial_kernel(const T *__restrict__ x,
                                                             float *__restrict__ part, int B, int HW,
                                                             int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, s = blockIdx.z;
  if (c >= C) return;
  const int per = (HW + kPoolSplits - 1) / kPoolSplits;
  const int i0 = s * per, i1 = min(HW, i0 + per);
  const T *px = x + ((size_t)b * HW) * C + c;
  // eight loads in flight, adds in the original order (bitwise the same sum)
  float acc = 0.f;
  int i = i0;
  for (; i + 8 <= i1; i += 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = (float)px[(size_t)(i + u) * C];
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; i < i1; ++i) acc += (float)px[(size_t)i * C];
  part[((size_t)b * kPoolSplits + s) * C + c] = acc;
}

// the same sums over a P-format tensor ("f32s": float16 pieces, cnn_device.h); channel c of pixel i is flat element i C + c
__global__ __launch_bounds__(256) void avgpool_partial_p_kernel(const void *__restrict__ x, float *__restrict__ part, int B,
                                                               int HW, int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, s = blockIdx.z;
  if (c >= C) return;
  const int per = (HW + kPoolSplits - 1) / kPoolSplits;
  const int i0 = s * per, i1 = min(HW, i0 + per);
  const char *base = static_cast<const char *>(x);
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) {
    const size_t e = ((size_t)b * HW + i) * C + c;
    const char *q = base + (e >> 5) * 128 + (e & 31) * 2;
    acc += (float)*reinterpret_cast<const _Float16 *>(q) + (float)*reinterpret_cast<const _Float16 *>(q + 64);
  }
  part[((size_t)b * kPoolSplits + s) * C + c] = acc;
}

// P format <-> float32 (parity taps, layer-level tests); n elements, n % 4 == 0
__global__ __launch_bounds__(256) void p_to_f32_kernel(const void *__restrict__ x, float *__restrict__ y, size_t n4) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256)
    *reinterpret_cast<float4 *>(y + 4 * e) = load4_p(x, 4 * e);
}
__global__ __launch_bounds__(256) void f32_to_p_kernel(const float *__restrict__ x, void *__restrict__ y, size_t n4) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n4; e += (size_t)gridDim.x * 256)
    store4_p(y, 4 * e, *reinterpret_cast<const float4 *>(x + 4 * e));
}

constexpr int kDenseMaxB = 16;
constexpr int kDenseKC = 256;

// out_part[b][ks][n] = sum_{k in slice ks} f(x)[b][k] * W[k][n],
// f(x)[b][k] = act( scale_in * sum_s xin[b][s][k] + bias_in[k] ).
__global__ __launch_bounds__(256) void dense_kernel(const float *__restrict__ xin, int s_in,
                                                   const float *__restrict__ bias_in, float scale_in,
                                                   int lrelu_in, const float *__restrict__ Wm,
                                                   float *__restrict__ out_part, int B, int K, int N) {
  __shared__ float xs[kDenseMaxB][kDenseKC];
  __shared__ float red[4][kDenseMaxB][64];
  const int tid = threadIdx.x;
  const int col = blockIdx.x * 64 + (tid & 63);
  const int kg = tid >> 6;
  const int ks = blockIdx.y;
  const int kper = K / kDenseSplits;
  float acc[kDenseMaxB];
#pragma unroll
  for (int b = 0; b < kDenseMaxB; ++b) acc[b] = 0.f;
  for (int kc = 0; kc < kper; kc += kDenseKC) {
    const int kbase = ks * kper + kc;
    const int kn = min(kDenseKC, kper - kc);
    __syncthreads();
    for (int e = tid; e < kDenseMaxB * kDenseKC; e += 256) {
      const int b = e / kDenseKC, k = e % kDenseKC;
      float v = 0.f;
      if (b < B && k < kn) {
        // loads four at a time (independent, in flight together), adds in the original order
        const float *xp = xin + (size_t)b * s_in * K + kbase + k;
        int s = 0;
        for (; s + 4 <= s_in; s += 4) {
          const float t0 = xp[(size_t)s * K], t1 = xp[(size_t)(s + 1) * K], t2 = xp[(size_t)(s + 2) * K],
                      t3 = xp[(size_t)(s + 3) * K];
          v += t0; v += t1; v += t2; v += t3;
        }
        for (; s < s_in; ++s) v += xp[(size_t)s * K];
        v *= scale_in;
        if (bias_in) v += bias_in[kbase + k];
        if (lrelu_in) v = v >= 0.f ? v : 0.2f * v;  // networks.py:31
      }
      xs[b][k] = v;
    }
    __syncthreads();
    if (col < N) {
      // eight weight loads in flight per thread (one at a time made the layer latency-bound:
      // 64 dependent L2 round trips); the fmaf chain per output keeps its k order
      for (int k0 = kg; k0 < kn; k0 += 32) {
        float w[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 4 * u;
          w[u] = k < kn ? Wm[(size_t)(kbase + k) * N + col] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int k = k0 + 4 * u;
          if (k < kn) {
#pragma unroll
            for (int b = 0; b < kDenseMaxB; ++b) acc[b] = fmaf(xs[b][k], w[u], acc[b]);
          }
        }
      }
    }
  }
#pragma unroll
  for (int b = 0; b < kDenseMaxB; ++b) red[kg][b][tid & 63] = acc[b];
  __syncthreads();
  for (int e = tid; e < kDenseMaxB * 64; e += 256) {
    const int b = e >> 6, c = e & 63;
    const int n = blockIdx.x * 64 + c;
    if (b < B && n < N)
      out_part[((size_t)b * kDenseSplits + ks) * N + n] =
          (red[0][b][c] + red[1][b][c]) + (red[2][b][c] + red[3][b][c]);
  }
}

__global__ __launch_bounds__(256) void dense_finalize_kernel(const float *__restrict__ part, int s_in,
                                                            float scale, const float *__restrict__ bias,
                                                            float *__restrict__ out, int B, int N) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= B * N) return;
  const int b = e / N, n = e % N;
  float v = 0.f;
  for (int s = 0; s < s_in; ++s) v += part[((size_t)b * s_in + s) * N + n];
  v *= scale;
  out[e] = bias ? v + bias[n] : v;
}

__global__ __launch_bounds__(256) void f16_to_f32_kernel(const _Float16 *__restrict__ x, float *__restrict__ y,
                                                        size_t n) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < n; e += (size_t)gridDim.x * 256) y[e] = (float)x[e];
}

}  // namespace

int launch_f16_to_f32(const void *x, float *y, size_t n, hipStream_t s) {
  const size_t want = (n + 255) / 256;
  hipLaunchKernelGGL(f16_to_f32_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s,
                     static_cast<const _Float16 *>(x), y, n);
  return check_launch("f16_to_f32_kernel");
}

int launch_p_to_f32(const void *x, float *y, size_t n, hipStream_t s) {
  DVSG_REQUIRE(n % 32 == 0, "p_to_f32: %zu elements are not whole 32-element groups", n);
  if (n == 0) return DVSG_OK;
  const size_t want = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(p_to_f32_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, x, y, n / 4);
  return check_launch("p_to_f32_kernel");
}

int launch_f32_to_p(const float *x, void *y, size_t n, hipStream_t s) {
  DVSG_REQUIRE(n % 32 == 0, "f32_to_p: %zu elements are not whole 32-element groups", n);
  if (n == 0) return DVSG_OK;
  const size_t want = (n / 4 + 255) / 256;
  hipLaunchKernelGGL(f32_to_p_kernel, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, x, y, n / 4);
  return check_launch("f32_to_p_kernel");
}

int launch_avgpool_partial(int prec, const void *x, float *part, int B, int HW, int C, hipStream_t s) {
  ProfScope prof(kClsHead, s, 0.0, (double)elem_size(prec) * B * HW * C);
  const dim3 grid(ceil_div(C, 256), B, kPoolSplits);
  if (prec == kF32S) {
    hipLaunchKernelGGL(avgpool_partial_p_kernel, grid, dim3(256), 0, s, x, part, B, HW, C);
    return check_launch("avgpool_partial_p_kernel");
  }
  if (prec == kF16)
    hipLaunchKernelGGL(avgpool_partial_kernel<_Float16>, grid, dim3(256), 0, s, static_cast<const _Float16 *>(x),
                       part, B, HW, C);
  else
    hipLaunchKernelGGL(avgpool_partial_kernel<float>, grid, dim3(256), 0, s, static_cast<const float *>(x), part,
                       B, HW, C);
  return check_launch("avgpool_partial_kernel");
}

int launch_dense(const float *xin, int s_in, const float *bias_in, float scale_in, int lrelu_in,
                 const float *W, float *out_part, int B, int K, int N, hipStream_t s) {
  DVSG_REQUIRE(B >= 1 && B <= kDenseMaxB, "dense: B=%d outside [1,%d]", B, kDenseMaxB);
  DVSG_REQUIRE(K % kDenseSplits == 0, "dense: K=%d must be a multiple of %d", K, kDenseSplits);
  ProfScope prof(kClsHead, s, 2.0 * B * (double)K * N, 4.0 * (double)K * N);
  hipLaunchKernelGGL(dense_kernel, dim3(ceil_div(N, 64), kDenseSplits), dim3(256), 0, s, xin, s_in, bias_in,
                     scale_in, lrelu_in, W, out_part, B, K, N);
  return check_launch("dense_kernel");
}

int launch_dense_finalize(const float *part, int s_in, float scale, const float *bias, float *out, int B,
                          int N, hipStream_t s) {
  hipLaunchKernelGGL(dense_finalize_kernel, dim3(ceil_div((long)B * N, 256)), dim3(256), 0, s, part, s_in,
                     scale, bias, out, B, N);
  return check_launch("dense_finalize_kernel");
}

}  // namespace dvsg
