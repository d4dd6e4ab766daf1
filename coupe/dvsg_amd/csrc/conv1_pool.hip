// Root of resnet_v1_50 (slim `conv2d_same(64, 7, stride=2)` + BatchNorm + ReLU, then
// `max_pool2d(3x3, stride 2, SAME)`) with scale_RGB (networks.py:6-16) fused into conv1's
// load stage.
#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

// ----------------------------------------------------------------------------------------
// conv1: 7x7 / stride 2 / pad 3, 21 -> 64.
//
// A workgroup owns 128 consecutive output pixels of one output row and all 64 channels.
// For kernel row kh the 7*21 = 147 taps of an output pixel are ONE contiguous run of the
// input row, and neighbouring output pixels start 42 floats apart: the raw input row
// segment (261 px * 21 ch = 5481 floats, 22 KB) is staged once in LDS and every A fragment
// is read from it in place as lds[42 * pixel + k] -- overlapping windows, no im2col, each
// input byte fetched once per kernel row.  42 r mod 64 visits every even bank once over
// r = 0..31, so the ds_read_b64 fragment reads are conflict-free as they stand; the weight
// rows use a stride of 150 floats (2 x odd) for the same reason.
//
// scale_RGB: y = 255 x - mean, channel groups reversed.  The zero padding of conv2d_same
// happens AFTER scale_RGB, so pad taps must be 0 in the scaled domain: the scale is applied
// per element while staging (valid elements only), and the group reversal is a permutation
// of conv1's input channels that is folded into the weights at load time (locnet.hip).
//
// NW waves per workgroup: NW/2 along the pixels x 2 along the channels.
// ----------------------------------------------------------------------------------------
constexpr int C1_TILE = 128;
constexpr int C1_SEG = (2 * (C1_TILE - 1) + 7) * kConv1Cin;  // 5481
constexpr int C1_SEG_PAD = 5488;
constexpr int C1_WELEMS = 64 * kConv1Ld;                      // 9600 floats per kernel row
constexpr int C1_LDC = 68;                                    // epilogue tile row stride

template <int NW, typename TO>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
void conv1_kernel(const float *__restrict__ x, const float *__restrict__ wt1, const float *__restrict__ bias,
                  TO *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles) {
  constexpr int NT = 64 * NW;
  constexpr int MI = 4 / (NW / 2);                       // 32-pixel MFMA blocks per wave: 2 or 1
  constexpr int INLOADS = (C1_SEG_PAD + NT - 1) / NT;    // 22 or 11 dwords per thread per kernel row
  constexpr int WLOADS = (C1_WELEMS / 4 + NT - 1) / NT;  // 10 or 5 float4
  static_assert(C1_TILE * C1_LDC <= C1_WELEMS, "epilogue tile must fit in the weight stage");
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

  // element e of the staged segment is input-row float (2 wo0 - 3) * 21 + e.  Loads are issued
  // UNCONDITIONALLY from a clamped index (a predicated load whose value feeds arithmetic makes
  // hipcc wait for each load in turn: serialised L2 round trips); validity is a per-thread
  // bit mask applied when the value is written to LDS.
  const long seg0 = (long)(2 * wo0 - 3) * kConv1Cin;
  const long row_elems = (long)W * kConv1Cin;
  float mean_i[INLOADS];
  int idx_i[INLOADS];
  unsigned col_ok = 0;
#pragma unroll
  for (int i = 0; i < INLOADS; ++i) {
    const int e = tid + NT * i;
    const long ge = seg0 + e;
    // raw channel c of group g = c / 7 lands in output group 2 - g and gets that group's mean
    const int c = e % kConv1Cin;
    const int g = c / (kConv1Cin / 3);
    mean_i[i] = g == 0 ? 123.68f : (g == 1 ? 116.779f : 103.939f);
    const bool ok = e < C1_SEG && ge >= 0 && ge < row_elems;
    if (ok) col_ok |= 1u << i;
    idx_i[i] = ok ? (int)ge : 0;
  }

  float in_reg[INLOADS];
  floatx4 w_reg[WLOADS];
  bool row_ok = false;  // validity of the input row whose values sit in in_reg
  auto load_stage = [&](int kh) __attribute__((always_inline)) {
    const int hi = 2 * ho + kh - 3;
    row_ok = hi >= 0 && hi < H;
    const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
    const float *xrow = x + ((long)b * H + hc) * row_elems;
#pragma unroll
    for (int i = 0; i < INLOADS; ++i) in_reg[i] = xrow[idx_i[i]];
    const floatx4 *wsrc = reinterpret_cast<const floatx4 *>(wt1 + (size_t)kh * C1_WELEMS);
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      w_reg[i] = wsrc[q < C1_WELEMS / 4 ? q : 0];
    }
  };
  auto store_stage = [&]() __attribute__((always_inline)) {
    const unsigned ok = row_ok ? col_ok : 0u;
#pragma unroll
    for (int i = 0; i < INLOADS; ++i) {
      const int e = tid + NT * i;
      // zero padding lives in the SCALED domain; x*255 and the subtraction round separately,
      // as the two TF ops do (this file is compiled with -ffp-contract=off)
      const float v = in_reg[i] * 255.0f - mean_i[i];
      if (e < C1_SEG_PAD) in_s[e] = ((ok >> i) & 1u) ? v : 0.f;
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      if (q < C1_WELEMS / 4) reinterpret_cast<floatx4 *>(w_s)[q] = w_reg[i];
    }
  };

  floatx16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[mi][q] = 0.f;

  load_stage(0);
  for (int kh = 0; kh < 7; ++kh) {
    __syncthreads();  // everyone is done reading the previous kernel row
    store_stage();
    __syncthreads();
    if (kh + 1 < 7) load_stage(kh + 1);
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA loop
    const float *a0 = in_s + 2 * kConv1Cin * (wm * 32 * MI + r) + 2 * h;
    const float *bp = w_s + (wn * 32 + r) * kConv1Ld + 2 * h;
#pragma unroll 4
    for (int u = 0; u < kConv1Kpad / 4; ++u) {
      float2 va[MI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        va[mi] = *reinterpret_cast<const float2 *>(a0 + mi * (2 * kConv1Cin * 32) + 4 * u);
      const float2 vb = *reinterpret_cast<const float2 *>(bp + 4 * u);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma32(va[mi].x, vb.x, acc[mi]);
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma32(va[mi].y, vb.y, acc[mi]);
    }
  }

  // ---- epilogue: transpose through the (idle) weight stage so stores are float4s along channels
  float *Cs = w_s;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wm * 32 * MI + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + wn * 32 + r] = acc[mi][q];
  __syncthreads();
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
  TO *yrow = y + (((size_t)b * Ho + ho) * Wo + wo0) * 64 + 4 * col4;
#pragma unroll 4
  for (int row = row0; row < C1_TILE; row += NT / 16) {
    if (wo0 + row >= Wo) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * C1_LDC + 4 * col4);
    v.x = fmaxf(v.x + b4.x, 0.f);
    v.y = fmaxf(v.y + b4.y, 0.f);
    v.z = fmaxf(v.z + b4.z, 0.f);
    v.w = fmaxf(v.w + b4.w, 0.f);
    store4(yrow + (size_t)row * 64, v);
  }
}

// ----------------------------------------------------------------------------------------
// 3x3 / stride 2 max pool with TF 'SAME' padding (pad_before = pad_total / 2: nothing on
// the top/left for even sizes).  One thread = one output pixel x 4 channels.
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T *__restrict__ x, T *__restrict__ y, int H, int W,
                                                     int C4, int Ho, int Wo, int pad_top, int pad_left,
                                                     size_t total) {
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
        const float4 v = load4(x + (((b * H + hi) * W + wi) * C4 + c4) * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    store4(y + e * 4, m);
  }
}

int g_conv1_variant = 0;  // dvsg_debug_set_option("conv1_variant", v): 0 = 4 waves (measured equal or better), 1 = 8

}  // namespace

void set_conv1_variant(int v) { g_conv1_variant = v; }

int launch_conv1(int out_prec, const float *x, const float *wt1, const float *bias, void *y, int B, int H,
                 int W, int Ho, int Wo, hipStream_t s) {
  const int wtiles = ceil_div(Wo, C1_TILE);
  const long blocks = (long)wtiles * Ho * B;
  DVSG_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv1: grid of %ld workgroups out of range", blocks);
  DVSG_REQUIRE((long)W * kConv1Cin < (1L << 31), "conv1: input row too long");
  ProfScope prof(kClsConv1, s, 2.0 * (double)B * Ho * Wo * 64 * 49 * kConv1Cin,
                 4.0 * (double)B * H * W * kConv1Cin + (double)elem_size(out_prec) * B * Ho * Wo * 64);
  const dim3 grid((unsigned)blocks);
  if (out_prec == kF16) {
    hipLaunchKernelGGL((conv1_kernel<4, _Float16>), grid, dim3(256), 0, s, x, wt1, bias,
                       static_cast<_Float16 *>(y), H, W, Ho, Wo, wtiles);
  } else if (g_conv1_variant == 0) {
    hipLaunchKernelGGL((conv1_kernel<4, float>), grid, dim3(256), 0, s, x, wt1, bias, static_cast<float *>(y), H,
                       W, Ho, Wo, wtiles);
  } else {
    hipLaunchKernelGGL((conv1_kernel<8, float>), grid, dim3(512), 0, s, x, wt1, bias, static_cast<float *>(y), H,
                       W, Ho, Wo, wtiles);
  }
  return check_launch("conv1_kernel");
}

int launch_maxpool(int prec, const void *x, void *y, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                   int pad_left, hipStream_t s) {
  DVSG_REQUIRE(C % 4 == 0, "maxpool: C=%d must be a multiple of 4", C);
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  const size_t want = (total + 255) / 256;
  const int blocks = (int)(want < 16384 ? want : 16384);
  ProfScope prof(kClsMaxpool, s, 0.0, (double)elem_size(prec) * C * ((double)B * H * W + (double)B * Ho * Wo));
  if (prec == kF16)
    hipLaunchKernelGGL(maxpool_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16 *>(x),
                       static_cast<_Float16 *>(y), H, W, C / 4, Ho, Wo, pad_top, pad_left, total);
  else
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float *>(x),
                       static_cast<float *>(y), H, W, C / 4, Ho, Wo, pad_top, pad_left, total);
  return check_launch("maxpool_kernel");
}

}  // namespace dvsg
