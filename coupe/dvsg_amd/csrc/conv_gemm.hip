// slim conv2d (1x1, and 3x3 conv2d_same) + folded BatchNorm + residual + ReLU of
// resnet_v1_50 (networks.py:33-34 -> tf.contrib.slim bottleneck_v1) as ONE implicit-GEMM
// kernel on the exact-f32 matrix cores:
//
//     Y[m, n] = act( sum_k A[m, k] Wt[n, k] + bias[n] (+ res[m', n]) ),
//     m = (b, ho, wo), k = (kh, kw, c).
//
// NHWC activations make each (kh, kw) tap of a pixel a contiguous run of C_in floats, so A
// tiles are staged HBM -> registers -> LDS in full 128-byte rows (zero-filled where the tap
// falls outside the image) and never exist as an im2col matrix.
//
// Tile 128 (m) x BN (n) x 32 (k); WM x WN waves, each owning a (128/WM) x (BN/WN) patch of
// 32x32 MFMA blocks; LDS rows padded to 36 floats (9r mod 16 is a bijection over every 16-lane
// ds_read_b128 group: conflict-free); two LDS stages, next stage prefetched into registers
// while the current one is multiplied; accumulators transposed through the idle staging LDS
// in the epilogue so that global stores / residual loads are float4s along the channel axis.
#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int LDT = 36;

// what a tap outside the image reads (conv2d_same zero padding)
__device__ const floatx4 g_zero4 = {0.f, 0.f, 0.f, 0.f};

struct ConvGemmDev {
  const float *x, *wt, *bias, *res;
  float *y;
  int H, W, Cin, Ho, Wo, Cout;
  int stride, pad;
  int res_H, res_W, res_stride;
  int M, K, mtiles, ntiles;
};

template <int BN, int WM, int WN, int KS, bool RELU, int RES>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN / 2, WM * WN / 2)))
void conv_gemm_kernel(ConvGemmDev p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int MI = BM / WM / 32;  // 32-row MFMA blocks per wave
  constexpr int NI = BN / WN / 32;  // 32-col MFMA blocks per wave
  constexpr int RS = NT / 8;        // tile rows staged per pass (8 float4 lanes per 32-float row)
  constexpr int AL = BM / RS;       // A float4 loads per thread per stage
  constexpr int BL = BN / RS;       // B float4 loads per thread per stage
  static_assert(MI >= 1 && NI >= 1 && AL >= 1 && BL >= 1 && AL <= 4 && BL <= 4, "bad tile configuration");
  constexpr int LDS_STAGE = 2 * (BM + BN) * LDT;
  constexpr int LDS_EPI = BM * (BN + 4);
  __shared__ __attribute__((aligned(16))) float lds[LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI];
  float *As = lds;
  float *Bs = lds + 2 * BM * LDT;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- per-thread staging coordinates: rows lrow + RS i, float4 column lcol
  const int lrow = tid >> 3, lcol = (tid & 7) * 4;
  long a_off[AL];
  unsigned a_mask[AL];  // bit kh: input row valid, bit 4+kw: input col valid
#pragma unroll
  for (int i = 0; i < AL; ++i) {
    const int m = m0 + lrow + RS * i;
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

  // Staging registers are ext_vector types (hipcc keeps arrays of HIP's float4 struct in
  // scratch and splits a predicated float4 load into four dword loads).  Every load is
  // UNCONDITIONAL: a tap outside the image reads a 16-byte zero constant instead, so there is
  // no branch and no select between the load and its LDS store.
  floatx4 a_reg[AL], b_reg[BL];
  auto load_stage = [&](int k0) __attribute__((always_inline)) {
    int kh = 0, kw = 0, c0 = k0;
    if (KS > 1) {
      const int tap = k0 / p.Cin;
      c0 = k0 - tap * p.Cin;
      kh = tap / KS;
      kw = tap - kh * KS;
    }
    const float *xa = p.x + ((long)kh * p.W + kw) * p.Cin + c0;
#pragma unroll
    for (int i = 0; i < AL; ++i) {
      const bool ok = ((a_mask[i] >> kh) & (a_mask[i] >> (4 + kw)) & 1u) != 0;
      const floatx4 *src = ok ? reinterpret_cast<const floatx4 *>(xa + a_off[i]) : &g_zero4;
      a_reg[i] = *src;
    }
#pragma unroll
    for (int i = 0; i < BL; ++i)
      b_reg[i] = *reinterpret_cast<const floatx4 *>(wrow + (size_t)(RS * i) * p.K + k0);
  };
  auto store_stage = [&](int buf) __attribute__((always_inline)) {
    float *ap = As + (buf * BM + lrow) * LDT + lcol;
    float *bp = Bs + (buf * BN + lrow) * LDT + lcol;
#pragma unroll
    for (int i = 0; i < AL; ++i) *reinterpret_cast<floatx4 *>(ap + i * RS * LDT) = a_reg[i];
#pragma unroll
    for (int i = 0; i < BL; ++i) *reinterpret_cast<floatx4 *>(bp + i * RS * LDT) = b_reg[i];
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

  auto compute_part = [&](int buf, int kb0, int kb1) __attribute__((always_inline)) {
    const float *a_base = As + (buf * BM + wm * (BM / WM) + r) * LDT + 4 * h;
    const float *b_base = Bs + (buf * BN + wn * (BN / WN) + r) * LDT + 4 * h;
#pragma unroll
    for (int kb = kb0; kb < kb1; ++kb) {
      floatx4 a4[MI], b4[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
        a4[mi] = *reinterpret_cast<const floatx4 *>(a_base + mi * 32 * LDT + kb * 8);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
        b4[ni] = *reinterpret_cast<const floatx4 *>(b_base + ni * 32 * LDT + kb * 8);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(a4[mi][s], b4[ni][s], acc[mi][ni]);
    }
  };
  auto compute_stage = [&](int buf) __attribute__((always_inline)) { compute_part(buf, 0, BK / 8); };

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

  // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane & 31, row = (q&3) + 8 (q>>2) + 4 h, i.e.
  // a lane owns ONE output channel: stored straight from registers that is 16 MI NI dword
  // stores (and as many residual loads) per lane.  The accumulators are transposed through the
  // (now idle) staging LDS instead, so every thread moves float4s along the channel axis.
  constexpr int LDC = BN + 4;
  float *Cs = lds;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        Cs[(wm * (BM / WM) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * LDC + wn * (BN / WN) + ni * 32 + r] =
            acc[mi][ni][q];
  __syncthreads();
  constexpr int C4 = BN / 4;       // float4 columns per tile row
  constexpr int RSTEP = NT / C4;   // tile rows covered per pass
  const int col4 = tid % C4, row0 = tid / C4;
  const int n = n0 + 4 * col4;
  const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll 4
  for (int row = row0; row < BM; row += RSTEP) {
    const int m = m0 + row;
    if (m >= p.M) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDC + 4 * col4);
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (RES != 0) {
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
      const float4 rv = *reinterpret_cast<const float4 *>(p.res + roff);
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    if (RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4 *>(p.y + (size_t)m * p.Cout + n) = v;
  }
}

// ----------------------------------------------------------------------------------------
// Same GEMM with direct-to-LDS staging (global_load_lds_dwordx4): tiles go L2 -> LDS without
// passing through VGPRs, so the stage has no ds_write and no staging registers.  An LDS-DMA
// wave-instruction writes 64 x 16 B = 8 tile rows LINEARLY (wave-uniform base + lane x 16), so
// rows cannot be padded; bank conflicts are avoided by an XOR swizzle applied on the per-lane
// GLOBAL source address and again on the fragment read: row R keeps its k-chunk c (16 B) at
// position c ^ ((R >> 1) & 7).  Over the 16 rows of every ds_read_b128 lane group the pair
// (R & 1, (R >> 1) & 7) is distinct, i.e. the 16 reads hit the 16 distinct 16-byte slots of
// the 256-byte bank row: conflict-free without padding (LDS 64 KB instead of 72 KB).
// ----------------------------------------------------------------------------------------
template <int BN, int WM, int WN, int KS, bool RELU, int RES>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN / 2, WM * WN / 2)))
void conv_gemm_glds_kernel(ConvGemmDev p) {
  constexpr int NW = WM * WN;
  constexpr int NT = 64 * NW;
  constexpr int MI = BM / WM / 32;
  constexpr int NI = BN / WN / 32;
  constexpr int AG = BM / 8 / NW;  // 8-row groups (one LDS-DMA instruction each) per wave, A tile
  constexpr int BG = BN / 8 / NW;  // same for the weight tile
  static_assert(AG >= 1 && BG >= 1, "bad tile configuration");
  constexpr int LDS_STAGE = 2 * (BM + BN) * BK;
  constexpr int LDS_EPI = BM * (BN + 4);
  __shared__ __attribute__((aligned(16))) float lds[LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI];
  float *As = lds;
  float *Bs = lds + 2 * BM * BK;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;

  const int tile = xcd_remap(blockIdx.x, p.mtiles * p.ntiles);
  const int mt = tile / p.ntiles, nt = tile - mt * p.ntiles;
  const int m0 = mt * BM, n0 = nt * BN;

  // ---- staging: wave `wave` fills row groups g = wave + NW i; lane -> row 8 g + lane / 8,
  // LDS chunk position lane % 8, which must receive global chunk pos ^ ((row >> 1) & 7).
  const int lrow8 = lane >> 3, lpos = lane & 7;
  long a_off[AG];
  unsigned a_mask[AG];
#pragma unroll
  for (int i = 0; i < AG; ++i) {
    const int row = 8 * (wave + NW * i) + lrow8;
    const int chunk = lpos ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + 4 * chunk;
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
  const float *wsrc[BG];
#pragma unroll
  for (int i = 0; i < BG; ++i) {
    const int row = 8 * (wave + NW * i) + lrow8;
    wsrc[i] = p.wt + (size_t)(n0 + row) * p.K + 4 * (lpos ^ ((row >> 1) & 7));
  }

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  // (kh, kw, c0) of the stage being issued advance incrementally: no integer division in the loop
  int s_kh = 0, s_kw = 0, s_c0 = 0, s_k0 = 0;
  auto issue_stage = [&](int buf) __attribute__((always_inline)) {
    const float *xa = p.x + ((long)s_kh * p.W + s_kw) * p.Cin + s_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
      const float *src = ok ? xa + a_off[i] : reinterpret_cast<const float *>(&g_zero4);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * BM + 8 * (wave + NW * i)) * BK), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < BG; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + s_k0),
                                       (lptr_t)(Bs + (buf * BN + 8 * (wave + NW * i)) * BK), 16, 0, 0);
    s_k0 += BK;
    s_c0 += BK;
    if (KS > 1 && s_c0 == p.Cin) {
      s_c0 = 0;
      if (++s_kw == KS) {
        s_kw = 0;
        ++s_kh;
      }
    }
  };

  floatx16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;

  const int sw = (r >> 1) & 7;  // fragment rows are 32-aligned + r, so (row >> 1) & 7 == (r >> 1) & 7
  auto compute_stage = [&](int buf) __attribute__((always_inline)) {
    const float *a_base = As + (buf * BM + wm * (BM / WM) + r) * BK;
    const float *b_base = Bs + (buf * BN + wn * (BN / WN) + r) * BK;
#pragma unroll
    for (int kb = 0; kb < BK / 8; ++kb) {
      const int co = 4 * ((2 * kb + h) ^ sw);
      floatx4 a4[MI], b4[NI];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi) a4[mi] = *reinterpret_cast<const floatx4 *>(a_base + mi * 32 * BK + co);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) b4[ni] = *reinterpret_cast<const floatx4 *>(b_base + ni * 32 * BK + co);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = mfma32(a4[mi][s], b4[ni][s], acc[mi][ni]);
    }
  };

  // The DMA of stage kt+1 is in flight while stage kt is multiplied; __syncthreads() carries
  // the vmcnt(0) that retires it (and orders everyone's reads of the buffer about to be refilled).
  const int KT = p.K / BK;
  issue_stage(0);
  __syncthreads();
  for (int kt = 0; kt < KT - 1; ++kt) {
    issue_stage((kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
  compute_stage((KT - 1) & 1);

  constexpr int LDC = BN + 4;
  float *Cs = lds;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        Cs[(wm * (BM / WM) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * LDC + wn * (BN / WN) + ni * 32 + r] =
            acc[mi][ni][q];
  __syncthreads();
  constexpr int C4 = BN / 4;
  constexpr int RSTEP = NT / C4;
  const int col4 = tid % C4, row0 = tid / C4;
  const int n = n0 + 4 * col4;
  const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
#pragma unroll 4
  for (int row = row0; row < BM; row += RSTEP) {
    const int m = m0 + row;
    if (m >= p.M) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * LDC + 4 * col4);
    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
    if (RES != 0) {
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
      const float4 rv = *reinterpret_cast<const float4 *>(p.res + roff);
      v.x += rv.x; v.y += rv.y; v.z += rv.z; v.w += rv.w;
    }
    if (RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    }
    *reinterpret_cast<float4 *>(p.y + (size_t)m * p.Cout + n) = v;
  }
}

int g_conv_variant = 5;  // dvsg_debug_set_option("conv_variant", v), see launch_ks

template <int BN, int WM, int WN, int KS>
int launch_cfg(const ConvGemmDev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid(d.mtiles * d.ntiles), block(64 * WM * WN);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_gemm_kernel<BN, WM, WN, KS, R, Q>), grid, block, 0, s, d)
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

template <int BN, int WM, int WN, int KS>
int launch_cfg_glds(const ConvGemmDev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid(d.mtiles * d.ntiles), block(64 * WM * WN);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_gemm_glds_kernel<BN, WM, WN, KS, R, Q>), grid, block, 0, s, d)
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
  return check_launch("conv_gemm_glds_kernel");
}

template <int KS>
int launch_ks(const ConvGemmDev &d, bool wide, bool relu, int res, hipStream_t s) {
  // 8 waves per workgroup (4 per SIMD at 2 workgroups per CU) when the K loop is short: those
  // launches are prologue / epilogue bound and want more waves in flight; long K loops run
  // slightly better with 4 fat waves (bigger register tiles, fewer LDS fragment reads per MFMA).
  // conv_variant: 0 = always 4 waves, 1 = always 8, 2 (default) = by K.
  if (g_conv_variant == 3)   // LDS-DMA staging, 4 waves
    return wide ? launch_cfg_glds<128, 2, 2, KS>(d, relu, res, s) : launch_cfg_glds<64, 2, 2, KS>(d, relu, res, s);
  if (g_conv_variant == 4)   // LDS-DMA staging, 8 waves
    return wide ? launch_cfg_glds<128, 2, 4, KS>(d, relu, res, s) : launch_cfg_glds<64, 4, 2, KS>(d, relu, res, s);
  if (g_conv_variant == 5) {  // default: LDS-DMA staging; fat 4-wave workgroups when one round covers the grid
    const bool four = (long)d.mtiles * d.ntiles <= 512;
    if (four)
      return wide ? launch_cfg_glds<128, 2, 2, KS>(d, relu, res, s) : launch_cfg_glds<64, 2, 2, KS>(d, relu, res, s);
    return wide ? launch_cfg_glds<128, 2, 4, KS>(d, relu, res, s) : launch_cfg_glds<64, 4, 2, KS>(d, relu, res, s);
  }
  const bool eight = g_conv_variant == 1 || (g_conv_variant == 2 && d.K <= 256);
  if (!eight)
    return wide ? launch_cfg<128, 2, 2, KS>(d, relu, res, s) : launch_cfg<64, 2, 2, KS>(d, relu, res, s);
  return wide ? launch_cfg<128, 2, 4, KS>(d, relu, res, s) : launch_cfg<64, 4, 2, KS>(d, relu, res, s);
}

}  // namespace

void set_conv_variant(int v) { g_conv_variant = v; }

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
  d.ntiles = p.Cout / (wide ? 128 : 64);
  return p.ksize == 1 ? launch_ks<1>(d, wide, p.relu != 0, res, s) : launch_ks<3>(d, wide, p.relu != 0, res, s);
}

}  // namespace dvsg
