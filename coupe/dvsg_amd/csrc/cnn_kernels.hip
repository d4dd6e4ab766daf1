// localizationNet (networks.py:30-46; slim resnet_v1_50 + 4 dense layers) as gfx950 kernels.
//
// Precision: float32 storage, exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered
// fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).  Every convolution is an implicit
// GEMM  Y[m, n] = sum_k A[m, k] Wt[n, k]  with m = (b, ho, wo) and k = (kh, kw, c): NHWC
// activations make each (kh, kw) tap of a pixel a contiguous run of C_in floats, so A tiles
// are staged HBM -> registers -> LDS in full 128-byte rows and never exist as an im2col
// matrix.
//
// Fragment scheme (both kernels): the 32x32x2 MFMA takes ONE f32 per lane per operand --
// lane (r = l & 31, h = l >> 5) supplies A[r][k_h] and B[k_h][r].  The order in which the k
// values of a tile are fed is free as long as A and B agree, so each lane reads a short
// run of CONSECUTIVE k (a float4 / float2 from a K-contiguous LDS row) and feeds them over
// consecutive MFMAs; lane half h takes the second half of the run.  That turns the operand
// fetch into conflict-free ds_read_b128 / ds_read_b64 with no transposes anywhere.
#include "cnn_kernels.h"

namespace dvsg {
namespace {

typedef float floatx16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// XCD-aware block remap (bijective for any grid size): the hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs; give each XCD a contiguous range of logical
// tiles so that tiles sharing an A panel / weight panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = id & 7, slot = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

// ----------------------------------------------------------------------------------------
// Generic conv (1x1 / 3x3) implicit GEMM.  Tile 128 (m) x BN (n) x 32 (k), 4 waves as 2x2,
// wave tile 64 x BN/2, LDS rows padded to 36 floats (ds_read_b128 conflict-free: 9r mod 16
// is a bijection over every 16-lane read group), two LDS stages with register prefetch.
// ----------------------------------------------------------------------------------------
constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDT = 36;

struct ConvGemmDev {
  const float *x, *wt, *bias, *res;
  float *y;
  int H, W, Cin, Ho, Wo, Cout;
  int stride, pad;
  int res_H, res_W, res_stride;
  int M, K, mtiles, ntiles;
};

template <int BN, int KS, bool RELU, int RES>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv_gemm_kernel(ConvGemmDev p) {
  constexpr int NI = BN / 64;      // 32-col sub-tiles per wave
  constexpr int BROWS = BN / 32;   // weight rows staged per thread
  __shared__ __attribute__((aligned(16))) float lds[2 * (BM + BN) * LDT];
  float *As = lds;
  float *Bs = lds + 2 * BM * LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates: rows lrow + 32 i, float4 column lcol
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  long a_off[4];
  unsigned a_mask[4];  // bit kh: input row valid, bit 4+kw: input col valid
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + lrow + 32 * i;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + lcol;
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
  const float *wrow = p.wt + (size_t)(n0 + lrow) * p.K + lcol;

  // named registers (not an array): hipcc keeps a 4-entry float4 array of plain loads in scratch
  float4 a_reg[4], b_reg0, b_reg1, b_reg2, b_reg3;
  auto load_stage = [&](int k0) __attribute__((always_inline)) {
    int kh = 0, kw = 0, c0 = k0;
    if (KS > 1) {
      const int tap = k0 / p.Cin;
      c0 = k0 - tap * p.Cin;
      kh = tap / KS;
      kw = tap - kh * KS;
    }
    const long tap_off = ((long)kh * p.W + kw) * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const bool ok = ((a_mask[i] >> kh) & (a_mask[i] >> (4 + kw)) & 1u) != 0;
      a_reg[i] = ok ? *reinterpret_cast<const float4 *>(p.x + a_off[i] + tap_off)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    b_reg0 = *reinterpret_cast<const float4 *>(wrow + k0);
    b_reg1 = *reinterpret_cast<const float4 *>(wrow + (size_t)32 * p.K + k0);
    if (BROWS == 4) {
      b_reg2 = *reinterpret_cast<const float4 *>(wrow + (size_t)64 * p.K + k0);
      b_reg3 = *reinterpret_cast<const float4 *>(wrow + (size_t)96 * p.K + k0);
    }
  };
  auto store_stage = [&](int buf) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      *reinterpret_cast<float4 *>(As + (buf * BM + lrow + 32 * i) * LDT + lcol) = a_reg[i];
    *reinterpret_cast<float4 *>(Bs + (buf * BN + lrow) * LDT + lcol) = b_reg0;
    *reinterpret_cast<float4 *>(Bs + (buf * BN + lrow + 32) * LDT + lcol) = b_reg1;
    if (BROWS == 4) {
      *reinterpret_cast<float4 *>(Bs + (buf * BN + lrow + 64) * LDT + lcol) = b_reg2;
      *reinterpret_cast<float4 *>(Bs + (buf * BN + lrow + 96) * LDT + lcol) = b_reg3;
    }
  };

  floatx16 acc[2][NI];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

  auto compute_stage = [&](int buf) __attribute__((always_inline)) {
    const float *a_base = As + (buf * BM + wm * 64 + r) * LDT + 4 * h;
    const float *b_base = Bs + (buf * BN + wn * (BN / 2) + r) * LDT + 4 * h;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      float4 a4[2], b4[NI];
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
        a4[mi] = *reinterpret_cast<const float4 *>(a_base + mi * 32 * LDT + kb * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        b4[ni] = *reinterpret_cast<const float4 *>(b_base + ni * 32 * LDT + kb * 8);
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) {
          acc[mi][ni] = mfma32(a4[mi].x, b4[ni].x, acc[mi][ni]);
          acc[mi][ni] = mfma32(a4[mi].y, b4[ni].y, acc[mi][ni]);
          acc[mi][ni] = mfma32(a4[mi].z, b4[ni].z, acc[mi][ni]);
          acc[mi][ni] = mfma32(a4[mi].w, b4[ni].w, acc[mi][ni]);
        }
    }
  };

  // Software pipeline: the global loads of stage kt+1 are in flight while stage kt is
  // multiplied out of LDS; one barrier per stage (the two LDS buffers alternate).
  const int KT = p.K / BK;
  load_stage(0);
  store_stage(0);
  __syncthreads();
  for (int kt = 0; kt < KT - 1; ++kt) {
    load_stage((kt + 1) * BK);
    __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of the MFMAs (hipcc sinks it otherwise)
    compute_stage(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    store_stage((kt + 1) & 1);
    __syncthreads();
  }
  compute_stage((KT - 1) & 1);

  // ---- epilogue: C/D map of the 32x32 MFMA: col = lane & 31, row = (q&3) + 8 (q>>2) + 4 h
#pragma unroll
  for (int ni = 0; ni < NI; ++ni) {
    const int n = n0 + wn * (BN / 2) + ni * 32 + r;
    const float bias = p.bias[n];
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int m = m0 + wm * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
        if (m < p.M) {
          float v = acc[mi][ni][q] + bias;
          if (RES == 1) {
            v += p.res[(size_t)m * p.Cout + n];
          } else if (RES == 2) {  // slim `subsample`: shortcut = x[:, ::s, ::s, :]
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int b = t / p.Ho;
            v += p.res[(((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W +
                        (size_t)wo * p.res_stride) * p.Cout + n];
          }
          if (RELU) v = fmaxf(v, 0.f);
          p.y[(size_t)m * p.Cout + n] = v;
        }
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// conv1: 7x7 / stride 2 / pad 3, 21 -> 64, scale_RGB fused into the load stage.
//
// A workgroup owns 128 consecutive output pixels of one output row and all 64 channels.
// For kernel row kh the 7*21 = 147 taps of an output pixel are ONE contiguous run of the
// input row, and neighbouring output pixels start 42 floats apart: the raw input row
// segment (261 px * 21 ch = 5481 floats, 22 KB) is staged once in LDS and every A fragment
// is read from it in place as lds[42 * pixel + k] -- overlapping windows, no im2col, each
// input byte fetched from HBM once per kernel row.  42 r mod 64 visits every even bank
// once over r = 0..31, so the ds_read_b64 fragment reads are conflict-free as they stand.
// ----------------------------------------------------------------------------------------
constexpr int C1_TILE = 128;
constexpr int C1_SEG = (2 * (C1_TILE - 1) + 7) * kConv1Cin;  // 5481
constexpr int C1_SEG_PAD = 5488;
constexpr int C1_INLOADS = (C1_SEG_PAD + 255) / 256;         // 22
constexpr int C1_WELEMS = 64 * kConv1Ld;                      // 9600 floats per kernel row
constexpr int C1_WLOADS = (C1_WELEMS / 4 + 255) / 256;        // 10 float4

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) void conv1_kernel(const float *__restrict__ x,
                                                      const float *__restrict__ wt1,
                                                      const float *__restrict__ bias,
                                                      float *__restrict__ y, int H, int W, int Ho,
                                                      int Wo, int wtiles) {
  __shared__ __attribute__((aligned(16))) float w_s[C1_WELEMS];
  __shared__ __attribute__((aligned(16))) float in_s[C1_SEG_PAD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  int blk = blockIdx.x;
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho = blk % Ho;
  const int b = blk / Ho;
  const int wo0 = wt_i * C1_TILE;

  // element e of the staged segment is input-row float (2 wo0 - 3) * 21 + e
  const long seg0 = (long)(2 * wo0 - 3) * kConv1Cin;
  const long row_elems = (long)W * kConv1Cin;
  float mean_i[C1_INLOADS];
#pragma unroll
  for (int i = 0; i < C1_INLOADS; ++i) {
    // scale_RGB (networks.py:6-16): raw channel c of group g = c / 7 lands in output group
    // 2 - g and gets that group's mean; the channel permutation itself is folded into wt1.
    const int c = (tid + 256 * i) % kConv1Cin;
    const int g = c / (kConv1Cin / 3);
    mean_i[i] = g == 0 ? 123.68f : (g == 1 ? 116.779f : 103.939f);
  }

  float in_reg[C1_INLOADS];
  float4 w_reg[C1_WLOADS];
  auto load_stage = [&](int kh) __attribute__((always_inline)) {
    const int hi = 2 * ho + kh - 3;
    const bool row_ok = hi >= 0 && hi < H;
    const float *xrow = x + ((long)b * H + hi) * row_elems;
#pragma unroll
    for (int i = 0; i < C1_INLOADS; ++i) {
      const int e = tid + 256 * i;
      const long ge = seg0 + e;
      float v = 0.f;  // zero padding lives in the SCALED domain (pad happens after scale_RGB)
      if (row_ok && e < C1_SEG && ge >= 0 && ge < row_elems)
        v = __fsub_rn(__fmul_rn(xrow[ge], 255.0f), mean_i[i]);
      in_reg[i] = v;
    }
    const float4 *wsrc = reinterpret_cast<const float4 *>(wt1 + (size_t)kh * C1_WELEMS);
#pragma unroll
    for (int i = 0; i < C1_WLOADS; ++i) {
      const int q = tid + 256 * i;
      w_reg[i] = q < C1_WELEMS / 4 ? wsrc[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto store_stage = [&]() __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < C1_INLOADS; ++i) {
      const int e = tid + 256 * i;
      if (e < C1_SEG_PAD) in_s[e] = in_reg[i];
    }
#pragma unroll
    for (int i = 0; i < C1_WLOADS; ++i) {
      const int q = tid + 256 * i;
      if (q < C1_WELEMS / 4) reinterpret_cast<float4 *>(w_s)[q] = w_reg[i];
    }
  };

  floatx16 acc[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[mi][q] = 0.f;

  load_stage(0);
  for (int kh = 0; kh < 7; ++kh) {
    __syncthreads();  // everyone is done reading the previous kernel row
    store_stage();
    __syncthreads();
    if (kh + 1 < 7) load_stage(kh + 1);
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA loop
    const float *a0 = in_s + 2 * kConv1Cin * (wm * 64 + r) + 2 * h;
    const float *a1 = a0 + 2 * kConv1Cin * 32;
    const float *bp = w_s + (wn * 32 + r) * kConv1Ld + 2 * h;
#pragma unroll 4
    for (int u = 0; u < kConv1Kpad / 4; ++u) {
      const float2 va0 = *reinterpret_cast<const float2 *>(a0 + 4 * u);
      const float2 va1 = *reinterpret_cast<const float2 *>(a1 + 4 * u);
      const float2 vb = *reinterpret_cast<const float2 *>(bp + 4 * u);
      acc[0] = mfma32(va0.x, vb.x, acc[0]);
      acc[1] = mfma32(va1.x, vb.x, acc[1]);
      acc[0] = mfma32(va0.y, vb.y, acc[0]);
      acc[1] = mfma32(va1.y, vb.y, acc[1]);
    }
  }

  const int n = wn * 32 + r;
  const float bs = bias[n];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int wo = wo0 + wm * 64 + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      if (wo < Wo) y[(((size_t)b * Ho + ho) * Wo + wo) * 64 + n] = fmaxf(acc[mi][q] + bs, 0.f);
    }
  }
}

// ----------------------------------------------------------------------------------------
// 3x3 / stride 2 max pool with TF 'SAME' padding (pad_before = pad_total / 2: nothing on
// the top/left for even sizes).  One thread = one output pixel x 4 channels.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void maxpool_kernel(const float *__restrict__ x, float *__restrict__ y,
                                                     int H, int W, int C4, int Ho, int Wo, int pad_top,
                                                     int pad_left, size_t total) {
  for (size_t e = (size_t)blockIdx.x * 256 + threadIdx.x; e < total; e += (size_t)gridDim.x * 256) {
    const int c4 = (int)(e % C4);
    size_t t = e / C4;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const size_t b = t / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int hi = 2 * ho - pad_top + i;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int wi = 2 * wo - pad_left + j;
        if (wi < 0 || wi >= W) continue;
        const float4 v = reinterpret_cast<const float4 *>(x)[((b * H + hi) * W + wi) * C4 + c4];
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    reinterpret_cast<float4 *>(y)[e] = m;
  }
}

// ----------------------------------------------------------------------------------------
// Head: global average pool + 4 dense layers (networks.py:36-44).  Tiny (6.8 MMAC per
// frame) and latency-bound: every stage writes deterministic split partial sums and the
// next stage folds "sum the partials, add bias, leaky-ReLU" into its operand load.
// ----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void avgpool_partial_kernel(const float *__restrict__ x,
                                                             float *__restrict__ part, int B, int HW,
                                                             int C) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  const int b = blockIdx.y, s = blockIdx.z;
  if (c >= C) return;
  const int per = (HW + kPoolSplits - 1) / kPoolSplits;
  const int i0 = s * per, i1 = min(HW, i0 + per);
  const float *px = x + ((size_t)b * HW) * C + c;
  float acc = 0.f;
  for (int i = i0; i < i1; ++i) acc += px[(size_t)i * C];
  part[((size_t)b * kPoolSplits + s) * C + c] = acc;
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
        for (int s = 0; s < s_in; ++s) v += xin[((size_t)b * s_in + s) * K + kbase + k];
        v *= scale_in;
        if (bias_in) v += bias_in[kbase + k];
        if (lrelu_in) v = v >= 0.f ? v : 0.2f * v;  // networks.py:31
      }
      xs[b][k] = v;
    }
    __syncthreads();
    if (col < N) {
      for (int k = kg; k < kn; k += 4) {
        const float w = Wm[(size_t)(kbase + k) * N + col];
#pragma unroll
        for (int b = 0; b < kDenseMaxB; ++b) acc[b] = fmaf(xs[b][k], w, acc[b]);
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
      out_part[((size_t)b * kDenseSplits + ks) * N + n] = (red[0][b][c] + red[1][b][c]) + (red[2][b][c] + red[3][b][c]);
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

template <int BN, int KS>
int launch_conv_gemm_t(const ConvGemmDev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid(d.mtiles * d.ntiles), block(256);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_gemm_kernel<BN, KS, R, Q>), grid, block, 0, s, d)
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
  return check_launch("conv_gemm_kernel");
}

}  // namespace

int launch_conv_gemm(const ConvGemm &p, hipStream_t s) {
  DVSG_REQUIRE(p.ksize == 1 || p.ksize == 3, "conv_gemm: kernel size %d unsupported", p.ksize);
  DVSG_REQUIRE(p.Cin % BK == 0 && p.Cout % 64 == 0, "conv_gemm: Cin=%d must be a multiple of 32 and Cout=%d of 64",
               p.Cin, p.Cout);
  const long M = (long)p.B * p.Ho * p.Wo;
  DVSG_REQUIRE(M > 0 && M < (1L << 31) - BM, "conv_gemm: M=%ld out of range", M);
  ConvGemmDev d;
  d.x = p.x; d.wt = p.wt; d.bias = p.bias; d.res = p.res; d.y = p.y;
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride; d.pad = p.pad;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.M = (int)M;
  d.K = p.ksize * p.ksize * p.Cin;
  d.mtiles = (int)((M + BM - 1) / BM);
  const int res = !p.res ? 0 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  // algorithmic work: 2 M N K flops; bytes = input + weights + output (+ residual) once each
  ProfScope prof(p.ksize == 3 ? kClsConv3x3 : kClsConv1x1, s, 2.0 * (double)M * p.Cout * d.K,
                 4.0 * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * d.K +
                        (double)M * p.Cout * (p.res ? 2.0 : 1.0)));
  // 128-wide n tiles when there are enough of them to fill the chip, else 64-wide.
  const bool wide = p.Cout % 128 == 0 && (long)d.mtiles * (p.Cout / 128) >= 512;
  if (wide) {
    d.ntiles = p.Cout / 128;
    return p.ksize == 1 ? launch_conv_gemm_t<128, 1>(d, p.relu != 0, res, s)
                        : launch_conv_gemm_t<128, 3>(d, p.relu != 0, res, s);
  }
  d.ntiles = p.Cout / 64;
  return p.ksize == 1 ? launch_conv_gemm_t<64, 1>(d, p.relu != 0, res, s)
                      : launch_conv_gemm_t<64, 3>(d, p.relu != 0, res, s);
}

int launch_conv1(const float *x, const float *wt1, const float *bias, float *y, int B, int H, int W,
                 int Ho, int Wo, hipStream_t s) {
  const int wtiles = ceil_div(Wo, C1_TILE);
  const long blocks = (long)wtiles * Ho * B;
  DVSG_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv1: grid of %ld workgroups out of range", blocks);
  ProfScope prof(kClsConv1, s, 2.0 * (double)B * Ho * Wo * 64 * 49 * kConv1Cin,
                 4.0 * ((double)B * H * W * kConv1Cin + (double)B * Ho * Wo * 64));
  hipLaunchKernelGGL(conv1_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, wt1, bias, y, H, W, Ho, Wo,
                     wtiles);
  return check_launch("conv1_kernel");
}

int launch_maxpool(const float *x, float *y, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                   int pad_left, hipStream_t s) {
  DVSG_REQUIRE(C % 4 == 0, "maxpool: C=%d must be a multiple of 4", C);
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  const size_t want = (total + 255) / 256;
  const int blocks = (int)(want < 16384 ? want : 16384);
  ProfScope prof(kClsMaxpool, s, 0.0, 4.0 * C * ((double)B * H * W + (double)B * Ho * Wo));
  hipLaunchKernelGGL(maxpool_kernel, dim3(blocks), dim3(256), 0, s, x, y, H, W, C / 4, Ho, Wo, pad_top,
                     pad_left, total);
  return check_launch("maxpool_kernel");
}

int launch_avgpool_partial(const float *x, float *part, int B, int HW, int C, hipStream_t s) {
  ProfScope prof(kClsHead, s, 0.0, 4.0 * (double)B * HW * C);
  hipLaunchKernelGGL(avgpool_partial_kernel, dim3(ceil_div(C, 256), B, kPoolSplits), dim3(256), 0, s, x,
                     part, B, HW, C);
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
