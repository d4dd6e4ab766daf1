// The float16 mode's EXPAND layers -- 1x1, stride 1, K = Cin of 128 or 256, Cout = 4 K (a bottleneck unit's conv3 and
// block 2's shortcut; resnet_utils / resnet_v1.bottleneck, networks.py:20-33) -- with the workgroup's activation tile
// RESIDENT in LDS.
//
// Why.  In conv_gemm_wide16.hip's geometry (256 pixels x 64 output channels per workgroup) such a layer is 8 or 16
// n-tiles per pixel tile, each re-reading the pixel tile out of L2 (x8 / x16: 3 MB of L2 -> LDS traffic per 256 pixels
// against 1.15 MB of HBM traffic at K = 256), each paying index setup, a first-stage round trip, a K loop of only 4 or
// 8 stages that never gets its pipeline full, and an epilogue of four workgroup barriers: per-tile stamps
// (tools/stamp_probe_wide16.py) put 49 % of a tile's lifetime outside the K loop at K = 128, 36 % at K = 256,
// and the class runs at 2.8-3.7 TB/s of HBM.  Here a workgroup loads its pixels ONCE (32 NW pixels x K float16, rows
// swizzled by 16-byte chunk so that fragment reads are conflict-free) and walks the output-channel groups of 64 itself:
//   * the stacked [hi | lo] weight rows of the groups stream through a ring of four 8 KB stages (32 k each) as ONE
//     sequence over all groups -- the first three stages of the next group are requested under the current group's last
//     stages, so the pipeline never restarts; a stage is waited for with a counted s_waitcnt (two younger stages stay in
//     flight) followed by the stage barrier;
//   * wave w owns pixels [32 w, 32 w + 32) x all 64 channels of the group (hi and lo products in separate accumulators,
//     8 MFMAs per stage), so its epilogue is PRIVATE: 8 pixels x 64 channels at a time through its own 2 KB of LDS, no
//     workgroup barrier, then bias + residual + ReLU and one 16-byte float16 store per lane -- 8 lanes per pixel, so every
//     residual load and every store instruction moves whole 128-byte lines;
//   * the residual rows and the bias of a group are requested before its K loop.
// K = 128: 4 waves, 128 pixels, 72 KB -- two workgroups per CU, each multiplying under the other's epilogue.
// K = 256: 6 waves, 192 pixels, 140 KB -- one workgroup per CU (the 128 KB of a 256-pixel tile would leave no room for
// the ring).
#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

constexpr int XROWB = 64;              // bytes of k per weight row and stage (32 float16)
constexpr int XNS = 4;                 // weight stages in the ring
constexpr int XSTAGE = 128 * XROWB;    // 128 stacked rows: 8 KiB

#ifdef DVSG_STAMPS  // diagnostic build (tools/stamp_probe_expand16.py): phase times of wave 0 of every workgroup
__device__ unsigned long long g_x16_stamps[8 * 65536];
#define X16_T() __builtin_amdgcn_s_memtime()
#define X16_ADD(i, t0) st[i] += X16_T() - (t0)
#else
#define X16_T() 0ull
#define X16_ADD(i, t0) (void)(t0)
#endif

struct ConvExpand16Dev {
  const _Float16 *x, *wt, *res;
  const float *bias;
  _Float16 *y;
  int M, Cout, ngroups;
  int Ho, Wo, res_H, res_W, res_stride;
};

template <int K, int NW, bool RELU, int RES>
__global__ __launch_bounds__(64 * NW) void conv_expand16_kernel(ConvExpand16Dev p) {
  constexpr int MT = 32 * NW;                 // pixels per workgroup
  constexpr int AROWB = 2 * K;                // bytes of a resident pixel row: 256 or 512
  constexpr int ABYTES = MT * AROWB;
  constexpr int CH = AROWB / 16;              // 16-byte chunks per row: 16 or 32
  constexpr int KT = K / 32;                  // weight stages per group: 4 or 8
  constexpr int A_DMAS = ABYTES / 1024 / NW;  // LDS-DMA instructions (1 KiB) per wave for the tile
  constexpr int B_MAX = (8 + NW - 1) / NW;    // ... per wave and weight stage, at most
  constexpr float kLoScale = 1.0f / 2048.0f;
  static_assert(ABYTES % (1024 * NW) == 0 && KT >= 4 && XNS == 4, "tile / ring geometry");
  __shared__ __attribute__((aligned(16))) char lds[ABYTES + XNS * XSTAGE + NW * 2048];
  char *As = lds;
  char *Bs = lds + ABYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  float *Cw = reinterpret_cast<float *>(lds + ABYTES + XNS * XSTAGE) + wave * 512;   // this wave's [8][64] float32
  const int r = lane & 31, h = lane >> 5;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int m0 = blockIdx.x * MT;
#ifdef DVSG_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = X16_T();
#endif

  // ---- the pixel tile: row `row`, 16-byte chunk position c' of the LDS image holds chunk c' ^ (row & 15) of the pixel
#pragma unroll
  for (int i = 0; i < A_DMAS; ++i) {
    const int a = wave + NW * i;
    const int row = a * (1024 / AROWB) + lane / CH;
    const int cpos = lane % CH;
    const int m = m0 + row < p.M ? m0 + row : p.M - 1;
    __builtin_amdgcn_global_load_lds((gptr_t)(p.x + (size_t)m * K + 8 * (cpos ^ (row & 15))), (lptr_t)(As + a * 1024), 16, 0, 0);
  }
  // ---- weight stages: one sequence g = group * KT + kt over all groups
  const int total = p.ngroups * KT;
  const int n_b = (8 - wave + NW - 1) / NW;   // this wave's instructions per stage (uniform): 2 or 1
  const int brow = lane >> 2, bpos = lane & 3;
  auto issue_b = [&](int g) __attribute__((always_inline)) {
    const int grp = g / KT, kt = g - grp * KT;
    const _Float16 *wg = p.wt + (size_t)grp * 128 * K + kt * 32;
    char *dst = Bs + (g & (XNS - 1)) * XSTAGE;
#pragma unroll
    for (int i = 0; i < B_MAX; ++i) {
      const int j = wave + NW * i;   // 16 stacked rows each
      if (j < 8) {
        const int row = 16 * j + brow;
        __builtin_amdgcn_global_load_lds((gptr_t)(wg + (size_t)row * K + 8 * (bpos ^ ((row >> 2) & 3))), (lptr_t)(dst + j * 1024), 16, 0,
                                         0);
      }
    }
  };
  issue_b(0);
  issue_b(1);
  issue_b(2);
  int gi = 3;   // next stage to request

  floatx16 acc_hi[2], acc_lo[2];   // [32-channel block]
  const char *a_row = As + (wave * 32 + r) * AROWB;
  const int a_sw = r & 15;
  const int b_sw = (r >> 2) & 3;
  // One k step (16 k) of a stage: the pixel fragment and the four weight fragments (hi / lo x two channel blocks).  With
  // two waves per SIMD nobody else covers the LDS round trip, so the fragments of step u + 1 are requested before the
  // MFMAs of step u are issued (stamps: 820 -> ticks per stage against 512 of MFMA issue for the two waves of a SIMD).
  struct Frags {
    halfx8 a, bh[2], bl[2];
  };
  auto read_frags = [&](Frags &f, int buf, int kt, int t) __attribute__((always_inline)) {
    const char *b_base = Bs + buf * XSTAGE + r * XROWB;
    f.a = *reinterpret_cast<const halfx8 *>(a_row + 16 * ((4 * kt + 2 * t + h) ^ a_sw));
    const int co = 16 * ((2 * t + h) ^ b_sw);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      f.bh[cb] = *reinterpret_cast<const halfx8 *>(b_base + cb * 32 * XROWB + co);
      f.bl[cb] = *reinterpret_cast<const halfx8 *>(b_base + (64 + cb * 32) * XROWB + co);
    }
  };
  auto mma = [&](const Frags &f) __attribute__((always_inline)) {
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
      acc_hi[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a, f.bh[cb], acc_hi[cb], 0, 0, 0);
      acc_lo[cb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a, f.bl[cb], acc_lo[cb], 0, 0, 0);
    }
  };
  auto compute_stage = [&](int buf, int kt) __attribute__((always_inline)) {
    Frags f0, f1;
    read_frags(f0, buf, kt, 0);
    read_frags(f1, buf, kt, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma(f0);
    __builtin_amdgcn_sched_barrier(0);
    mma(f1);
  };

  // epilogue thread map: a pass is 8 pixels x all 64 channels of the group; lane -> pixel lane / 8, channels 8 (lane % 8) ..,
  // so every memory instruction of the epilogue moves WHOLE 128-byte lines (8 pixels x 8 lanes x 16 bytes): with 64 bytes
  // per pixel and instruction -- 16 pixels x 32 channels -- the read + write mix of this epilogue gets 3.7 TB/s out of
  // HBM, with whole lines 5.1 (tools/hbm_pattern_bench.hip, P1 against P4)
  const int epx = lane >> 3, ec8 = lane & 7;
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  int g = 0;
  for (int grp = 0; grp < p.ngroups; ++grp) {
    // Everything requested so far has landed past this wait: the tile (first group), this group's first three stages
    // (requested under the previous group's last stages) -- and the previous group's stores have been acknowledged, so
    // the counted waits below see loads only (loads retire in order; stores and loads do not order with each other).
    const unsigned long long t_top = X16_T();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    X16_ADD(0, t_top);   // group top: the tile / the previous group's stores
    const int nbase = grp * 64 + ec8 * 8;   // the lane's 8 channels
    uintx4 rv[4];        // residual of the lane's pixel in each of the four passes: 8 float16
    size_t orow[4];      // output row offsets (elements)
    bool ok[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int mr = m0 + wave * 32 + ps * 8 + epx;
      ok[ps] = mr < p.M;
      const int m = ok[ps] ? mr : p.M - 1;
      orow[ps] = (size_t)m * p.Cout;
      if (RES != 0) {
        size_t rrow;
        if (RES == 1) {
          rrow = orow[ps];
        } else {  // slim `subsample`: shortcut = x[:, ::s, ::s, :]
          const int wo = m % p.Wo;
          const int t = m / p.Wo;
          const int ho = t % p.Ho;
          const int b = t / p.Ho;
          rrow = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.Cout;
        }
        rv[ps] = *reinterpret_cast<const uintx4 *>(p.res + rrow + nbase);
      }
    }
    const float4 bias_a = *reinterpret_cast<const float4 *>(p.bias + nbase);
    const float4 bias_b = *reinterpret_cast<const float4 *>(p.bias + nbase + 4);
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc_hi[cb][q] = acc_lo[cb][q] = 0.f;
    const bool last = grp + 1 == p.ngroups;
#pragma unroll
    for (int kt = 0; kt < KT; ++kt, ++g) {
      const unsigned long long t_s0 = X16_T();
      if (kt >= 3) {   // stage g was requested three stages ago; the two requested since may still be in flight
        if (last) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (fewer than two follow: wait for all)
        else if (n_b == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      }
      X16_ADD(1, t_s0);   // own part of stage g not landed
      const unsigned long long t_s1 = X16_T();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();   // stage g is complete and visible; everyone has read stage g - 1
      asm volatile("" ::: "memory");
      X16_ADD(2, t_s1);   // at the stage barrier
      const unsigned long long t_s2 = X16_T();
      if (gi < total) issue_b(gi);    // into the buffer of stage g - 1
      ++gi;
      __builtin_amdgcn_sched_barrier(0);
      compute_stage(g & (XNS - 1), kt);
      __builtin_amdgcn_sched_barrier(0);
#ifdef DVSG_STAMPS
      asm volatile("s_nop 0" ::: "memory");
#endif
      X16_ADD(3, t_s2);   // DMA issue + fragment reads + 8 MFMAs (issue time)
    }
    const unsigned long long t_ep = X16_T();

    // ---- the wave's own epilogue: 4 passes of 8 pixels x 64 channels through Cw ([8][64] float32)
    const float bb[8] = {bias_a.x, bias_a.y, bias_a.z, bias_a.w, bias_b.x, bias_b.y, bias_b.z, bias_b.w};
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {   // accumulator registers 4 ps .. 4 ps + 3: pixels 8 ps + 4 h + (0..3) of the wave
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          Cw[(4 * h + j) * 64 + cb * 32 + r] = acc_hi[cb][4 * ps + j] + acc_lo[cb][4 * ps + j] * kLoScale;
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its own stores are enough)
      const float4 v0 = *reinterpret_cast<const float4 *>(Cw + epx * 64 + ec8 * 8);
      const float4 v1 = *reinterpret_cast<const float4 *>(Cw + epx * 64 + ec8 * 8 + 4);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // read before the next pass overwrites
      const float o[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      halfx8 rh;
      if (RES != 0) rh = __builtin_bit_cast(halfx8, rv[ps]);
      halfx8 out;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = o[j] + bb[j];
        if (RES != 0) t += (float)rh[j];
        if (RELU) t = fmaxf(t, 0.f);
        out[j] = (_Float16)t;
      }
      if (ok[ps]) *reinterpret_cast<halfx8 *>(p.y + orow[ps] + nbase) = out;
    }
    X16_ADD(4, t_ep);   // epilogue
  }
#ifdef DVSG_STAMPS
  if (tid == 0 && blockIdx.x < 65536) {
    unsigned long long *o = g_x16_stamps + (size_t)blockIdx.x * 8;
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = st[4];
    o[5] = X16_T() - t_begin;
  }
#endif
}

int g_expand16 = 1;   // dvsg_debug_set_option("expand16", v): 0 = these layers through conv_wide16_kernel, as before; 2 = also
                      // for launches too small to fill the chip (tests)

template <int K, int NW>
int launch_k(const ConvExpand16Dev &d, bool relu, int res, hipStream_t s) {
  const dim3 grid((d.M + 32 * NW - 1) / (32 * NW)), block(64 * NW);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_expand16_kernel<K, NW, R, Q>), grid, block, 0, s, d)
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
  return check_launch("conv_expand16_kernel");
}

}  // namespace

void set_expand16(int v) { g_expand16 = v; }

#ifdef DVSG_STAMPS
extern "C" int dvsg_debug_read_expand16_stamps(void *host, size_t bytes) {
  if (hipDeviceSynchronize() != hipSuccess) return -3;
  const int rc = hipMemcpyFromSymbol(host, HIP_SYMBOL(g_x16_stamps), bytes) == hipSuccess ? 0 : -3;
  void *sym = nullptr;
  if (hipGetSymbolAddress(&sym, HIP_SYMBOL(g_x16_stamps)) == hipSuccess) (void)hipMemset(sym, 0, sizeof(unsigned long long) * 8 * 65536);
  return rc;
}
#endif

// Does launch_conv_gemm hand this float16 layer (stacked hi / lo weights) to the resident-tile kernel?
bool conv_expand16_takes(const ConvGemm &p) {
  const long M = (long)p.B * p.Ho * p.Wo;
  return g_expand16 != 0 && p.prec == kF16 && p.wsplit != 0 && p.ksize == 1 && p.stride == 1 && p.pad == 0 &&
         (p.Cin == 128 || p.Cin == 256) && p.Cout >= 2 * p.Cin && p.Cout % 64 == 0 &&
         // three rounds of resident workgroups at least (512 of 128 pixels / 256 of 192): a part-filled last round
         // costs a whole workgroup lifetime, which here is all the layer's output groups
         (g_expand16 == 2 || M >= 3L * 256 * (p.Cin == 128 ? 2 * 128 : 192));
}

// the caller has opened the ProfScope
int launch_conv_expand16(const ConvGemm &p, hipStream_t s) {
  const long M = (long)p.B * p.Ho * p.Wo;
  ConvExpand16Dev d;
  d.x = static_cast<const _Float16 *>(p.x); d.wt = static_cast<const _Float16 *>(p.wt);
  d.res = static_cast<const _Float16 *>(p.res); d.bias = p.bias; d.y = static_cast<_Float16 *>(p.y);
  d.M = (int)M; d.Cout = p.Cout; d.ngroups = p.Cout / 64;
  d.Ho = p.Ho; d.Wo = p.Wo; d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  const int res = !p.res ? 0 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  return p.Cin == 128 ? launch_k<128, 4>(d, p.relu != 0, res, s) : launch_k<256, 6>(d, p.relu != 0, res, s);
}

}  // namespace dvsg
