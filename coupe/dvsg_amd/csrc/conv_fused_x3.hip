// Block 1's conv2 (3x3, 64 -> 64, stride 1 / 2, + BN + ReLU) and conv3 (1x1, 64 -> C_out, + BN + residual + ReLU) as ONE
// kernel in the "f32x3" precision (conv_fused.hip's fusion; conv_gemm_tile.h's X3 arithmetic): float32 tensors, every
// product from three bfloat16 pieces per operand, six bfloat16 MFMAs with the five small cross terms in their own
// accumulators.  conv3 with K = 64 is HBM-bound -- it reads the [M,64] tensor conv2 has just written, a residual, and
// writes four times as many channels -- so as two f32x3 GEMMs a stride-1 unit takes 428 + 390 us at batch 16, 720p; fused,
// the [M,64] intermediate never exists and conv3's residual loads / output stores run under conv2's matrix-core time of
// the CU's other workgroup.
//
// Four waves per workgroup (the f32x3 GEMM's reason: an activation fragment is split in registers, ~42 VALU that the
// matrix pipe does not hide, so a wave's tile is wide along N -- one split serves 2 weight blocks in phase 1, 4 in phase 2).
//   phase 1  conv2 tile 128 pixels x 64 channels, K = 9 Cin: conv_gemm_kernel<float, 64, 4, 1, 3, ..., X3>'s loop -- float32
//            activation rows by LDS-DMA (XOR-swizzled) through a ring of three 16 KB slots, two stages ahead; packed weight piece
//            stages (2 x 12 KB behind them), one ahead; channel chunk outer / taps inner; wave w owns pixels [32 w, 32 w + 32) x
//            all 64 channels.
//   hand-over  bias + ReLU, the tile goes to LDS as float32 in the layout a staged activation stage has (two 32-k stages at 0).
//   phase 2  the wave's four activation fragments (32 pixels x 64 mid channels) are read back and split ONCE -- 48 registers
//            of pieces that serve every output channel; then per 64 output channels: conv3's packed piece stages (24 KB,
//            double-buffered at 32 KB / 56 KB, the next group's in flight under this one's work), 4 steps x 12 MFMAs per wave
//            with no VALU, the result through a 128-row float32 transpose at 0 (the conv2 tile's place), bias + residual +
//            ReLU, 16-byte stores.
// 80 KB of LDS, two workgroups per CU.  The opening unit's shortcut conv is NOT fused here (its accumulators would not fit next
// to the doubled ones of this arithmetic): that unit runs shortcut as an f32x3 GEMM and this kernel with the tensor as residual.
#include <algorithm>

#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

constexpr int BM = 128;
constexpr int ROWB = 128;   // bytes of k per activation row and stage (32 float32)
constexpr int CMID = 64;
constexpr int NW = 4;
constexpr int A_STAGE = BM * ROWB;        // 16 KB
constexpr int W2_STAGE = 12 * 1024;       // conv2: one 64-row group's 32-k piece stage
constexpr int W3_GROUP = 24 * 1024;       // conv3: one group of 64 output channels, both 32-k stages of its K = 64
constexpr int OFF_W = 2 * A_STAGE;        // 32 KB
constexpr int LDS_BYTES = OFF_W + 2 * W3_GROUP;   // 80 KB

__device__ const floatx4 g_zero16x = {0.f, 0.f, 0.f, 0.f};

struct FusedX3Dev {
  const float *x;                 // [B,H,W,Cin]
  const unsigned short *wt2x;     // conv2, packed f32x3 stages: [1 group][9 Cin / 32][3][64][32]
  const float *bias2;             // [64]
  const unsigned short *wt3x;     // conv3: [Cout / 64][2][3][64][32]
  const float *bias3;             // [Cout]
  const float *res;               // residual [B,res_H,res_W,Cout]
  float *y;                       // [B,Ho,Wo,Cout]
  int H, W, Cin, Ho, Wo, Cout;
  int stride;
  int res_H, res_W, res_stride;
  int M, mtiles;
};

// one 16-k step of two weight blocks (rows [0, 64) of a piece-plane group at b_row's lane row): the six cross terms, the large
// one and the five small ones into separate accumulators
__device__ __forceinline__ void step_x3(const bf16x8 a1, const bf16x8 a2, const bf16x8 a3, const char *b_row, int g, int swb,
                                        floatx16 (&acc)[2], floatx16 (&accs)[2]) {
  bf16x8 b1[2], b2[2], b3[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const char *bq = b_row + ni * (32 * 64) + 16 * (g ^ swb);
    b1[ni] = *reinterpret_cast<const bf16x8 *>(bq);
    b2[ni] = *reinterpret_cast<const bf16x8 *>(bq + 4096);
    b3[ni] = *reinterpret_cast<const bf16x8 *>(bq + 8192);
  }
#define DVSG_FX3_TERM(C, A, B) \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) C[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A, B[ni], C[ni], 0, 0, 0)
  DVSG_FX3_TERM(accs, a3, b1);
  DVSG_FX3_TERM(accs, a1, b3);
  DVSG_FX3_TERM(accs, a2, b2);
  DVSG_FX3_TERM(accs, a2, b1);
  DVSG_FX3_TERM(accs, a1, b2);
  DVSG_FX3_TERM(acc, a1, b1);
#undef DVSG_FX3_TERM
}

// RES: 1 residual has the output's shape, 2 subsampled shortcut x[:, ::s, ::s, :]
template <int RES>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv3x3_1x1_x3_kernel(FusedX3Dev p) {
  constexpr int AG = BM / 8 / NW;  // 4 LDS-DMA instructions per wave and stage for the activation tile
  __shared__ __attribute__((aligned(16))) char lds[LDS_BYTES];
  char *As = lds;
  char *Ws = lds + OFF_W;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;
  const int lrow8 = lane >> 3, lpos = lane & 7;
  const int sw = (r >> 1) & 7;      // activation rows: chunk c of row R at position c ^ ((R >> 1) & 7); R = 32 wave + r
  const int swb = (r >> 2) & 3;     // weight piece rows: chunk c of row R at position c ^ ((R >> 2) & 3); R = 32 ni + r
  const int K1 = 9 * p.Cin;
  const int KT = K1 / 32;

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  const int m0 = xcd_remap(blockIdx.x, p.mtiles) * BM;
  // (Measured, B=16 720p: 793 us per stride-1 unit where conv2 and conv3 as two f32x3 GEMMs take 428 + 390 -- the [M,64]
  // round trip is gone, but the phases do not hide each other: a 32-k stage is 0.6 us of work per wave here, less than a
  // memory round trip, so with two stages in flight phase 1 waits for its loads, and phase 2 waits once per 64 channels for
  // the residual rows it asked for a group's MFMAs earlier; 2.7 TB/s of HBM, 0.5 of the matrix time.  Starting the second
  // workgroup of every CU a phase late -- so that one workgroup's loads would run under the other's MFMAs -- changed
  // nothing (+-0.3 % of the step for 8 k ... 49 k cycles of delay).  What the numbers above led to: the three-slot activation
  // ring of phase 1 below.)

  // ---------------------------------------------------------------- phase 1: conv2 tile 128 x 64
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
    const int hi0 = ho * p.stride - 1, wi0 = wo * p.stride - 1;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + 4 * chunk;
    unsigned mk = 0;
    if (m < p.M) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (hi0 + q >= 0 && hi0 + q < p.H) mk |= 1u << q;
        if (wi0 + q >= 0 && wi0 + q < p.W) mk |= 16u << q;
      }
    }
    a_mask[i] = mk;
  }
  // The activation stages (HBM, the long round trip) go through a ring of THREE 16 KB slots, two stages ahead of the one being
  // multiplied; the weight piece stages (12 KB, L2-resident) through two, one ahead: 72 KB.  A stage is 0.6 us of work per wave
  // in this arithmetic, less than a memory round trip: with one stage of lead the loop waited for its loads.  Issue order per
  // iteration: weights of stage kt + 1, then activations of stage kt + 2 -- LDS-DMA lands in issue order, so "at most the
  // youngest AG instructions outstanding" means stage kt + 1 is complete while stage kt + 2 stays in flight across the barrier.
  char *Ws2 = lds + 3 * A_STAGE;
  int a_kh = 0, a_kw = 0, a_c0 = 0;
  auto issue_a = [&](int slot) __attribute__((always_inline)) {
    const float *xa = p.x + ((long)a_kh * p.W + a_kw) * p.Cin + a_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> a_kh) & (a_mask[i] >> (4 + a_kw)) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16x);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (slot * BM + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
    }
    if (++a_kw == 3) {
      a_kw = 0;
      if (++a_kh == 3) {
        a_kh = 0;
        a_c0 += 32;
      }
    }
  };
  // K order: channel chunk outer, the 9 taps inner (conv_gemm_tile.h); the weight stage of (tap, c0) is stage (tap Cin + c0) / 32
  // of the packed matrix.  12 KB, linearly: 1 KB runs wave, wave + 4, wave + 8
  int w_tap = 0, w_c0 = 0;
  auto issue_w = [&](int buf) __attribute__((always_inline)) {
    const int wk = w_tap * p.Cin + w_c0;
    const char *wsrc = reinterpret_cast<const char *>(p.wt2x) + (size_t)(wk >> 5) * W2_STAGE + lane * 16;
#pragma unroll
    for (int i = 0; i < 3; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + (wave + NW * i) * 1024),
                                       (lptr_t)(Ws2 + buf * W2_STAGE + (wave + NW * i) * 1024), 16, 0, 0);
    if (++w_tap == 9) {
      w_tap = 0;
      w_c0 += 32;
    }
  };

  floatx16 acc1[2], acc1s[2];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc1[ni][q] = acc1s[ni][q] = 0.f;
  auto compute_stage = [&](int slot, int buf) __attribute__((always_inline)) {
    const char *a_row = As + (slot * BM + wave * 32 + r) * ROWB;
    const char *b_row = Ws2 + buf * W2_STAGE + r * 64;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = 2 * t + h;
      const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_row + 16 * ((2 * g) ^ sw));
      const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_row + 16 * ((2 * g + 1) ^ sw));
      bf16x8 a1, a2, a3;
      split_bf16x3(lo, hi, a1, a2, a3);
      step_x3(a1, a2, a3, b_row, g, swb, acc1, acc1s);
    }
  };

  issue_a(0);
  issue_w(0);
  issue_a(1);   // KT >= 18
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AG) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  int slot = 0, slot2 = 2;   // ring slots of stages kt and kt + 2
  for (int kt = 0; kt < KT; ++kt) {
    if (kt + 1 < KT) issue_w((kt + 1) & 1);
    if (kt + 2 < KT) issue_a(slot2);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(slot, kt & 1);
    __builtin_amdgcn_sched_barrier(0);
    if (kt + 2 < KT) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AG) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    slot = slot == 2 ? 0 : slot + 1;
    slot2 = slot2 == 2 ? 0 : slot2 + 1;
  }

  // conv3's weight pieces of output channels [64 q, 64 q + 64): the group's two 32-k stages are 24 KB in a row in the packed
  // matrix, 24 runs of 1 KB over the four waves, into weight buffer q & 1
  auto w3_issue = [&](int q) __attribute__((always_inline)) {
    const char *src = reinterpret_cast<const char *>(p.wt3x) + (size_t)q * W3_GROUP + lane * 16;
#pragma unroll
    for (int i = 0; i < 6; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (wave + NW * i) * 1024),
                                       (lptr_t)(Ws + (q & 1) * W3_GROUP + (wave + NW * i) * 1024), 16, 0, 0);
  };

  // ---------------------------------------------------------------- the tile becomes conv3's left operand
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (q&3) + 8 (q>>2) + 4 h.  Lane (r, h) of wave w holds mid channels
  // 32 ni + r of 16 pixels: stage ni, 16-byte chunk r >> 2, swizzled like an LDS-DMA'd activation row.
  __syncthreads();  // everyone is done with the phase-1 stage buffers
  w3_issue(0);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni) {
    const float bmid = p.bias2[32 * ni + r];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int R = wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
      const float v = fmaxf(acc1[ni][q] + acc1s[ni][q] + bmid, 0.f);
      *reinterpret_cast<float *>(lds + (ni * BM + R) * ROWB + 16 * ((r >> 2) ^ ((R >> 1) & 7)) + 4 * (r & 3)) = v;
    }
  }
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  lds_barrier();
  // the wave's four fragments (stage s2, step t), split once for all output channels
  bf16x8 ap1[4], ap2[4], ap3[4];
#pragma unroll
  for (int st = 0; st < 4; ++st) {
    const int g = 2 * (st & 1) + h;
    const char *a_row = lds + ((st >> 1) * BM + wave * 32 + r) * ROWB;
    const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_row + 16 * ((2 * g) ^ sw));
    const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_row + 16 * ((2 * g + 1) ^ sw));
    split_bf16x3(lo, hi, ap1[st], ap2[st], ap3[st]);
  }

  // ---------------------------------------------------------------- phase 2: conv3, 64 channels at a time
  float *Cs = reinterpret_cast<float *>(lds);  // 128 rows x 64 channels transpose buffer where the conv2 tile was
  const int col4 = tid & 15, row0 = tid >> 4;  // 16 float4 per row, 16 rows per pass
  const int nq = p.Cout / 64;
  for (int q = 0; q < nq; ++q) {
    // this group's weights have landed (issued a group ago), everyone has read the conv2 tile / the previous group's rows
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // residual rows first, then the next group's weights: waiting for the former leaves the latter in flight
    const int n = 64 * q + 4 * col4;
    float4 rv[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int mr = m0 + row0 + 16 * i;
      const int m = mr < p.M ? mr : p.M - 1;
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
      rv[i] = load4(p.res + roff);
    }
    if (q + 1 < nq) w3_issue(q + 1);   // (its buffer was read a group ago: everyone is past that since the barrier above)
    floatx16 acc2[2], acc2s[2];
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc2[ni][e] = acc2s[ni][e] = 0.f;
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const char *b_row = Ws + (q & 1) * W3_GROUP + (st >> 1) * (12 * 1024) + r * 64;
      step_x3(ap1[st], ap2[st], ap3[st], b_row, 2 * (st & 1) + h, swb, acc2, acc2s);
    }
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int e = 0; e < 16; ++e)
        Cs[(wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * h) * 64 + ni * 32 + r] = acc2[ni][e] + acc2s[ni][e];
    const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias3 + n);
    if (q + 1 < nq) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // the residual rows; the next group's 6 DMA stay in flight
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int row = row0 + 16 * i;
      const int m = m0 + row;
      if (m < p.M) {
        float4 v = *reinterpret_cast<const float4 *>(Cs + row * 64 + 4 * col4);
        v.x = fmaxf(v.x + bias4.x + rv[i].x, 0.f);
        v.y = fmaxf(v.y + bias4.y + rv[i].y, 0.f);
        v.z = fmaxf(v.z + bias4.z + rv[i].z, 0.f);
        v.w = fmaxf(v.w + bias4.w + rv[i].w, 0.f);
        store4(p.y + (size_t)m * p.Cout + n, v);
      }
    }
  }
}

}  // namespace

int launch_conv3x3_1x1_x3(const ConvFused &p, hipStream_t s) {
  DVSG_REQUIRE(p.Cin % 32 == 0 && p.Cin >= 64 && p.Cout % 64 == 0, "conv3x3_1x1 (f32x3): Cin=%d must be a multiple of 32 (>= 64), Cout=%d of 64",
               p.Cin, p.Cout);
  DVSG_REQUIRE(p.res && !p.sc_x, "conv3x3_1x1 (f32x3): the residual is a tensor (the opening unit's shortcut conv runs as its own launch)");
  const long M = (long)p.B * p.Ho * p.Wo;
  DVSG_REQUIRE(M > 0 && M < (1L << 31) - BM, "conv3x3_1x1: M=%ld out of range", M);
  FusedX3Dev d;
  d.x = p.x;
  d.wt2x = reinterpret_cast<const unsigned short *>(p.wt2); d.bias2 = p.bias2;
  d.wt3x = reinterpret_cast<const unsigned short *>(p.wt3); d.bias3 = p.bias3;
  d.res = p.res; d.y = p.y;
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.M = (int)M;
  d.mtiles = (int)((M + BM - 1) / BM);
  const int res = p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2;
  ProfScope prof(kClsFused, s, 2.0 * (double)M * CMID * (9.0 * p.Cin) + 2.0 * (double)M * p.Cout * CMID,
                 4.0 * ((double)p.B * p.H * p.W * p.Cin + (double)CMID * 9 * p.Cin + (double)p.Cout * CMID + 2.0 * (double)M * p.Cout));
  const dim3 grid(d.mtiles), block(64 * NW);
  if (res == 1) hipLaunchKernelGGL((conv3x3_1x1_x3_kernel<1>), grid, block, 0, s, d);
  else hipLaunchKernelGGL((conv3x3_1x1_x3_kernel<2>), grid, block, 0, s, d);
  return check_launch("conv3x3_1x1_x3_kernel");
}

}  // namespace dvsg
