// conv2 (3x3, 64 -> 64, stride 1 or 2, + BN + ReLU) and conv3 (1x1, 64 -> C_out, + BN + residual +
// ReLU) of a block-1 bottleneck unit (networks.py:33-34 -> slim bottleneck_v1) as ONE kernel: float32 (and
// its "f32s" pieces form) first, the float16 mode's version (conv3x3_1x1_f16_kernel) further down.
//
// Why: conv3 with K = 64 is an HBM-bound layer -- it reads the [M,64] tensor conv2 has just written,
// a residual and writes 4x as many channels, with 2 K stages of matrix-core work per tile -- and
// conv2 with N = 64 is matrix-core bound with almost no HBM traffic.  A workgroup's 128 x 64 conv2
// tile is a COMPLETE left operand for conv3 (its K is conv2's N), so the tile goes from the
// accumulators through bias + ReLU straight into LDS, in the layout the fragment reads expect, and
// is multiplied by conv3's weights there: the [M,64] intermediate (2 x 236 MB per unit at batch 16,
// 720p) never exists, and conv3's residual loads / output stores run under conv2's matrix-core time
// of the other workgroup on the CU.
//
// RES == 3 (the unit that opens the block, whose residual is itself a convolution, the 1x1 `shortcut`
// of the unit's input): phase 0 computes that convolution for the tile -- the input tile and the
// shortcut weights go through the still empty LDS -- straight into phase 2's accumulators, which
// conv3 then continues.  The [M,256] shortcut tensor (944 MB written and read back per step at batch
// 16, 720p) and its launch disappear; phase 0 costs what conv3 costs.
//
// Phase 1 is conv_gemm_kernel's 64-wide configuration unchanged (8 waves = 4 x 2 tiles of 32 x 32,
// LDS-DMA staging, XOR-swizzled K-contiguous rows, two stages, channel chunk outer / taps inner).
// Phase 2 keeps the 128 x 64 tile in the first 32 KiB of LDS as two 32-k stages and streams conv3's
// weights through the other 32 KiB, 128 output channels at a time (wave tile 32 x 64); each 128-wide
// result leaves through a 64-row LDS transpose in the weight area (two rounds), 16 bytes per lane.
#include <algorithm>

#include "cnn_device.h"
#include "cnn_kernels.h"

#include <type_traits>

namespace dvsg {
namespace {

constexpr int BM = 128;
constexpr int ROWB = 128;   // bytes of k per tile row and stage (32 float32)
constexpr int CMID = 64;    // conv2's output channels = conv3's K

__device__ const floatx4 g_zero16f = {0.f, 0.f, 0.f, 0.f};

struct ConvFusedDev {
  const float *x;      // [B,H,W,Cin]
  const float *wt2;    // conv2 [64][9*Cin]
  const float *bias2;  // [64]
  const float *wt3;    // conv3 [Cout][64]
  const float *bias3;  // [Cout]
  const float *res;    // residual [B,res_H,res_W,Cout]; RES == 3: the unit's input [M,Csc] (shortcut's operand)
  const float *wts;    // RES == 3: shortcut weights [Cout][Csc]
  const float *biass;  // RES == 3: shortcut bias [Cout]
  float *y;            // [B,Ho,Wo,Cout]
  int H, W, Cin, Ho, Wo, Cout;
  int stride;
  int res_H, res_W, res_stride;
  int M, mtiles;
};

constexpr int CSC = 64;     // RES == 3: channels of the shortcut's input (block 1: the pooled conv1 output)

// One 32-k stage of A-row x B-row products into `acc`: exact float32 (four 16-byte chunks, 16 MFMAs of
// 32x32x2) or, PS ("f32s": both rows hold float16 pieces, 32 hi halves then 32 lo halves), two 16-k steps
// of three 32x32x16 float16 MFMAs -- a1 w1 + a2 w1 + a1 w2 (conv_gemm.hip).
template <bool PS, int NI>
__device__ __forceinline__ void stage_mma(const char *a_row, const char *b_row, int sw, int h, floatx16 (&acc)[NI]) {
  if constexpr (PS) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int g = 2 * t + h;
      const halfx8 ahi = *reinterpret_cast<const halfx8 *>(a_row + 16 * (g ^ sw));
      const halfx8 alo = *reinterpret_cast<const halfx8 *>(a_row + 16 * ((4 + g) ^ sw));
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const halfx8 bhi = *reinterpret_cast<const halfx8 *>(b_row + ni * 32 * ROWB + 16 * (g ^ sw));
        const halfx8 blo = *reinterpret_cast<const halfx8 *>(b_row + ni * 32 * ROWB + 16 * ((4 + g) ^ sw));
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, bhi, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo, bhi, acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi, blo, acc[ni], 0, 0, 0);
      }
    }
  } else {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int co = 16 * ((2 * kb + h) ^ sw);
      const floatx4 a4 = *reinterpret_cast<const floatx4 *>(a_row + co);
#pragma unroll
      for (int ni = 0; ni < NI; ++ni) {
        const floatx4 b4 = *reinterpret_cast<const floatx4 *>(b_row + ni * 32 * ROWB + co);
        acc[ni] = Frag<float>::mma(a4, b4, acc[ni]);
      }
    }
  }
}

// RES: 1 residual has the output's shape, 2 subsampled shortcut x[:, ::s, ::s, :], 3 shortcut conv fused.
// PS ("f32s" precision): every tensor and weight matrix holds float16 pieces in 128-byte groups (cnn_device.h,
// conv_gemm.hip): the staging is byte for byte the float32 kernel's, the products come from the f16 matrix cores.
template <int RES, bool PS = false>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv3x3_1x1_kernel(ConvFusedDev p) {
  constexpr int NW = 8;
  constexpr int AG = BM / 8 / NW;  // 2 LDS-DMA instructions per wave and stage for the activation tile
  __shared__ __attribute__((aligned(16))) char lds[65536];
  char *As = lds;             // phase 1: 2 x 16 KiB activation stages; phase 2: the conv2 tile, 2 x 32 k
  char *Bs = lds + 32768;     // phase 1: 2 x 8 KiB weight stages
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves
  const int r = lane & 31, h = lane >> 5;
  const int lrow8 = lane >> 3, lpos = lane & 7;
  const int sw = (r >> 1) & 7;
  const int K1 = 9 * p.Cin;
  const int KT = K1 / 32;

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  const int m0 = xcd_remap(blockIdx.x, p.mtiles) * BM;

  // ---------------------------------------------------------------- phase 0 (RES == 3): shortcut conv of the tile
  // acc_sc[half][ni]: the 128 x 256 shortcut tile in phase 2's accumulator layout (wave tile 32 x 64 per
  // 128-channel half).  LDS: input tile as two 32-k stages at 0 / 16 KiB, shortcut weights (128 rows x 32 k
  // = 16 KiB per (half, stage) piece) alternating between 32 KiB and 48 KiB.
  floatx16 acc_sc[RES == 3 ? 2 : 1][2];
  if (RES == 3) {
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc_sc[hf][ni][q] = 0.f;
    auto xs_issue = [&](int s2) __attribute__((always_inline)) {   // rows of the unit's input: dense [M, CSC]
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const int row = 8 * (wave + NW * i) + lrow8;
        const int m = m0 + row < p.M ? m0 + row : p.M - 1;
        const float *src = p.res + (size_t)m * CSC + 32 * s2 + 4 * (lpos ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + s2 * 16384 + 8 * (wave + NW * i) * ROWB), 16, 0, 0);
      }
    };
    auto ws_issue = [&](int step) __attribute__((always_inline)) {  // step = 2 half + stage -> buffer step & 1
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + NW * i) + lrow8;
        const float *src = p.wts + (size_t)(128 * (step >> 1) + row) * CSC + 32 * (step & 1) + 4 * (lpos ^ ((row >> 1) & 7));
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + 32768 + (step & 1) * 16384 + 8 * (wave + NW * i) * ROWB),
                                         16, 0, 0);
      }
    };
    xs_issue(0);
    ws_issue(0);
    xs_issue(1);
    ws_issue(1);
#pragma unroll
    for (int step = 0; step < 4; ++step) {
      // pieces land in issue order: all but the youngest two (the next step's) have landed
      if (step == 0) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else if (step == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      const char *a_base = lds + (step & 1) * 16384 + (wm * 32 + r) * ROWB;
      const char *b_base = lds + 32768 + (step & 1) * 16384 + (wn * 64 + r) * ROWB;
      stage_mma<PS, 2>(a_base, b_base, sw, h, acc_sc[step >> 1]);
      if (step < 2) {  // the buffer just read takes the piece two steps ahead
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        ws_issue(step + 2);
      }
    }
    __syncthreads();  // phase 1 restages all of LDS
  }

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
  const float *wsrc;
  {
    const int row = 8 * wave + lrow8;  // 64 weight rows: one DMA instruction per wave
    wsrc = p.wt2 + (size_t)row * K1 + 4 * (lpos ^ ((row >> 1) & 7));
  }
  int s_kh = 0, s_kw = 0, s_c0 = 0;
  auto issue_stage = [&](int buf) __attribute__((always_inline)) {
    const float *xa = p.x + ((long)s_kh * p.W + s_kw) * p.Cin + s_c0;
    const int wk = (s_kh * 3 + s_kw) * p.Cin + s_c0;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16f);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * BM + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
    }
    __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + wk), (lptr_t)(Bs + (buf * CMID + 8 * wave) * ROWB), 16, 0, 0);
    if (++s_kw == 3) {
      s_kw = 0;
      if (++s_kh == 3) {
        s_kh = 0;
        s_c0 += 32;
      }
    }
  };
  floatx16 acc1v[1];
  floatx16 &acc1 = acc1v[0];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc1[q] = 0.f;
  auto compute_stage = [&](int buf) __attribute__((always_inline)) {
    const char *a_base = As + (buf * BM + wm * 32 + r) * ROWB;
    const char *b_base = Bs + (buf * CMID + wn * 32 + r) * ROWB;
    stage_mma<PS, 1>(a_base, b_base, sw, h, acc1v);
  };
  auto w3_issue = [&](int half, int s2) __attribute__((always_inline)) {
    // conv3 weight rows n = 128 half + row, 32-k stage s2: stage 0 -> lds + 48 KiB, stage 1 -> lds + 32 KiB
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (wave + NW * i) + lrow8;
      const float *src = p.wt3 + (size_t)(128 * half + row) * CMID + 32 * s2 + 4 * (lpos ^ ((row >> 1) & 7));
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + (s2 == 0 ? 49152 : 32768) + 8 * (wave + NW * i) * ROWB),
                                       16, 0, 0);
    }
  };
  issue_stage(0);
  issue_stage(1);  // KT >= 18
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(AG + 1) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  compute_stage(0);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 1; kt < KT - 1; ++kt) {
    __syncthreads();
    issue_stage((kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  w3_issue(0, 0);  // lds + 48 KiB is not used by phase 1: conv3's first weight stage lands under the last conv2 stage
  __builtin_amdgcn_sched_barrier(0);
  compute_stage((KT - 1) & 1);

  // ---------------------------------------------------------------- the tile becomes conv3's left operand
  // C/D map of the 32x32 MFMA: col = lane & 31, row = (q&3) + 8 (q>>2) + 4 h.  Lane (r, h) of wave
  // (wm, wn) holds mid channel k2 = 32 wn + r of 16 pixels: stage wn, 16-byte chunk r >> 2, swizzled like
  // an LDS-DMA'd row so compute2 below reads it exactly as compute_stage reads an activation stage.
  const float bmid = p.bias2[32 * wn + r];
  __syncthreads();  // everyone is done with the phase-1 stage buffers
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int R = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    const float v = fmaxf(acc1[q] + bmid, 0.f);
    if (PS) {  // the value's two float16 pieces: hi in 16-byte chunk r / 8 of the row, lo in chunk 4 + r / 8
      const _Float16 hi = (_Float16)v, lo = (_Float16)(v - (float)hi);
      char *row = lds + (wn * BM + R) * ROWB + 2 * (r & 7);
      *reinterpret_cast<_Float16 *>(row + 16 * ((r >> 3) ^ ((R >> 1) & 7))) = hi;
      *reinterpret_cast<_Float16 *>(row + 16 * ((4 + (r >> 3)) ^ ((R >> 1) & 7))) = lo;
    } else {
      *reinterpret_cast<float *>(lds + (wn * BM + R) * ROWB + 16 * ((r >> 2) ^ ((R >> 1) & 7)) + 4 * (r & 3)) = v;
    }
  }
  w3_issue(0, 1);

  // ---------------------------------------------------------------- phase 2: conv3, 128 channels at a time
  float *Cs = reinterpret_cast<float *>(lds + 32768);  // 64 rows x 128 channels transpose buffer
  const int col4 = tid & 31, row0 = tid >> 5;          // 32 float4 per row, 16 rows per pass
  auto lds_barrier = [&]() __attribute__((always_inline)) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  };
  auto compute2 = [&](int s2, floatx16 (&acc2)[2]) __attribute__((always_inline)) {
    const char *a_base = lds + (s2 * BM + wm * 32 + r) * ROWB;
    const char *b_base = lds + (s2 == 0 ? 49152 : 32768) + (wn * 64 + r) * ROWB;
    stage_mma<PS, 2>(a_base, b_base, sw, h, acc2);
  };
  // one 128-channel half of the output: conv3 on top of `acc2` (zeros, or the half's shortcut tile)
  const int nhalves = p.Cout / 128;
  auto do_half = [&](int half, floatx16 (&acc2)[2]) __attribute__((always_inline)) {
    // weight stage 0 has landed once at most stage 1's two DMA instructions are outstanding
    asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // (first half: the conv2 tile written above is visible too)
    asm volatile("" ::: "memory");
    compute2(0, acc2);
    __syncthreads();  // weight stage 1 landed
    compute2(1, acc2);
    // epilogue of this half: two rounds of 64 rows through the (now idle) weight area
    const int n = 128 * half + 4 * col4;
    float4 bias4 = *reinterpret_cast<const float4 *>(p.bias3 + n);
    if (RES == 3) {
      const float4 bs = *reinterpret_cast<const float4 *>(p.biass + n);
      bias4.x += bs.x; bias4.y += bs.y; bias4.z += bs.z; bias4.w += bs.w;
    }
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      // residual of this round's rows is fetched before the transpose (its latency runs under it)
      float4 rv[4];
      if (RES != 3) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int mr = m0 + 64 * rho + row0 + 16 * i;
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
          rv[i] = PS ? load4_p_pair(p.res, roff, col4 & 1) : load4(p.res + roff);
        }
      }
      lds_barrier();  // weights of this half consumed / previous round's rows read
      if ((wm >> 1) == rho) {
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
#pragma unroll
          for (int q = 0; q < 16; ++q)
            Cs[((wm & 1) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 128 + wn * 64 + ni * 32 + r] = acc2[ni][q];
      }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row = row0 + 16 * i;
        const int m = m0 + 64 * rho + row;
        if (m < p.M) {
          float4 v = *reinterpret_cast<const float4 *>(Cs + row * 128 + 4 * col4);
          if (RES == 3) {  // the shortcut is already in the accumulators; its bias in bias4
            v.x = fmaxf(v.x + bias4.x, 0.f);
            v.y = fmaxf(v.y + bias4.y, 0.f);
            v.z = fmaxf(v.z + bias4.z, 0.f);
            v.w = fmaxf(v.w + bias4.w, 0.f);
          } else {
            v.x = fmaxf(v.x + bias4.x + rv[i].x, 0.f);
            v.y = fmaxf(v.y + bias4.y + rv[i].y, 0.f);
            v.z = fmaxf(v.z + bias4.z + rv[i].z, 0.f);
            v.w = fmaxf(v.w + bias4.w + rv[i].w, 0.f);
          }
          if (PS) store4_p_pair(p.y, (size_t)m * p.Cout + n, v, col4 & 1, true);
          else store4(p.y + (size_t)m * p.Cout + n, v);
        }
      }
    }
    if (half + 1 < nhalves) {
      lds_barrier();  // the transpose buffer has been read: the next 128 weight rows may land in it
      w3_issue(half + 1, 0);
      w3_issue(half + 1, 1);
    }
  };
  if (RES == 3) {  // exactly two halves (launch_conv3x3_1x1 checks Cout == 256): static accumulator indices
    do_half(0, acc_sc[0]);
    do_half(1, acc_sc[1]);
  } else {
    for (int half = 0; half < nhalves; ++half) {
      floatx16 acc2[2];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc2[ni][q] = 0.f;
      do_half(half, acc2);
    }
  }
}

// ----------------------------------------------------------------------------------------
// The same fusion for the float16 mode (float16 activations, [hi | lo] float16 weight rows stacked along N,
// conv_gemm.hip): there every block-1 layer is HBM-bound, and the [M,64] conv2 output, conv3's read of it and, in
// the opening unit, the [M,256] shortcut tensor are a third of the block's traffic.
// A 128-byte row is 64 float16 k: one K stage of conv2 is one 3x3 tap of all 64 input channels (9 stages), and
// conv3 / the shortcut (K = 64) are ONE stage each.  Phase 1: A stages 2 x 16 KiB at 0, stacked weight stages
// (128 rows: 64 hi, 64 lo) 2 x 16 KiB at 32 KiB; wave (wm, wn) owns 32 pixels x channels [32 wn, 32 wn + 32), hi and
// lo products in two accumulators that are folded (hi + 2^-11 lo) with the bias and ReLU into the float16 conv2 tile
// at 0.  Phase 2, per 64 output channels: conv3's stacked weights at 32 KiB (and the shortcut's at 48 KiB, its
// operand -- the unit's input tile, re-read rather than multiplied in a phase 0: four groups of accumulators would
// not fit the registers -- at 16 KiB), 8 (+ 8) MFMAs per wave, fold, a 128 x 64 float32 transpose through the
// weight area, bias + residual + ReLU, 8-byte float16 stores.
// ----------------------------------------------------------------------------------------
struct ConvFusedF16Dev {
  const _Float16 *x;     // [B,H,W,Cin]
  const _Float16 *wt2;   // conv2, stacked [128][9*Cin]
  const float *bias2;    // [64]
  const _Float16 *wt3;   // conv3, stacked [Cout/64][128][64]
  const float *bias3;    // [Cout]
  const _Float16 *res;   // residual [B,res_H,res_W,Cout]; RES == 3: the unit's input [M,64]
  const _Float16 *wts;   // RES == 3: shortcut, stacked [Cout/64][128][64]
  const float *biass;    // RES == 3: [Cout]
  _Float16 *y;           // [B,Ho,Wo,Cout]
  int H, W, Cin, Ho, Wo, Cout;
  int stride;
  int res_H, res_W, res_stride;
  int M, mtiles;
};

template <int RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv3x3_1x1_f16_kernel(ConvFusedF16Dev p) {
  constexpr int NW = 8;
  constexpr float kLoScale = 1.0f / 2048.0f;
  __shared__ __attribute__((aligned(16))) char lds[65536];
  char *As = lds;           // phase 1: 2 x 16 KiB activation stages
  char *Bs = lds + 32768;   // phase 1: 2 x 16 KiB stacked weight stages
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves
  const int r = lane & 31, h = lane >> 5;
  const int lrow8 = lane >> 3, lpos = lane & 7;
  const int sw = (r >> 1) & 7;
  const int K1 = 9 * p.Cin;
  const int KT = K1 / 64;   // >= 9

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  const int m0 = xcd_remap(blockIdx.x, p.mtiles) * BM;

  // ---------------------------------------------------------------- phase 1: conv2 tile 128 pixels x 64 channels
  long a_off[2];
  unsigned a_mask[2];
  const _Float16 *wsrc[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + NW * i) + lrow8;
    const int chunk = lpos ^ ((row >> 1) & 7);
    const int m = m0 + row;
    const int mm = m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    const int hi0 = ho * p.stride - 1, wi0 = wo * p.stride - 1;
    a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.Cin + 8 * chunk;
    unsigned mk = 0;
    if (m < p.M) {
#pragma unroll
      for (int q = 0; q < 3; ++q) {
        if (hi0 + q >= 0 && hi0 + q < p.H) mk |= 1u << q;
        if (wi0 + q >= 0 && wi0 + q < p.W) mk |= 16u << q;
      }
    }
    a_mask[i] = mk;
    wsrc[i] = p.wt2 + (size_t)row * K1 + 8 * chunk;   // stacked weight row `row` (0..127)
  }
  int s_kh = 0, s_kw = 0, s_c0 = 0;
  auto issue_stage = [&](int buf) __attribute__((always_inline)) {
    const _Float16 *xa = p.x + ((long)s_kh * p.W + s_kw) * p.Cin + s_c0;
    const int wk = (s_kh * 3 + s_kw) * p.Cin + s_c0;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16f);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * BM + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + wk), (lptr_t)(Bs + (buf * 128 + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
    if (++s_kw == 3) {
      s_kw = 0;
      if (++s_kh == 3) {
        s_kh = 0;
        s_c0 += 64;
      }
    }
  };
  floatx16 acc_hi, acc_lo;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc_hi[q] = acc_lo[q] = 0.f;
  // one 64-k stage: A rows at a_row, stacked weight rows at b_row (hi) and b_row + 64 rows (lo)
  auto mma_stage = [&](const char *a_row, const char *b_row) __attribute__((always_inline)) {
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const int co = 16 * ((2 * kb + h) ^ sw);
      const halfx8 a = *reinterpret_cast<const halfx8 *>(a_row + co);
      const halfx8 bh = *reinterpret_cast<const halfx8 *>(b_row + co);
      const halfx8 bl = *reinterpret_cast<const halfx8 *>(b_row + 64 * ROWB + co);
      acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc_hi, 0, 0, 0);
      acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc_lo, 0, 0, 0);
    }
  };
  auto compute_stage = [&](int buf) __attribute__((always_inline)) {
    mma_stage(As + (buf * BM + wm * 32 + r) * ROWB, Bs + (buf * 128 + wn * 32 + r) * ROWB);
  };
  // 16 KiB of 128 rows x 128 bytes (a K = 64 operand), rows at src + row x row_elems, into lds + dst
  auto issue_rows = [&](const _Float16 *src, size_t row_elems, int dst, bool by_pixel) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (wave + NW * i) + lrow8;
      size_t rr = row;
      if (by_pixel) rr = m0 + row < p.M ? m0 + row : p.M - 1;   // the unit's input tile: dense [M,64]
      __builtin_amdgcn_global_load_lds((gptr_t)(src + rr * row_elems + 8 * (lpos ^ ((row >> 1) & 7))),
                                       (lptr_t)(lds + dst + 8 * (wave + NW * i) * ROWB), 16, 0, 0);
    }
  };
  issue_stage(0);
  issue_stage(1);
  asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
  compute_stage(0);
  __builtin_amdgcn_sched_barrier(0);
  for (int kt = 1; kt < KT - 1; ++kt) {
    __syncthreads();
    issue_stage((kt + 1) & 1);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(kt & 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  __syncthreads();
  // the stage buffers not used by the last stage take phase 2's first operands
  const int last = (KT - 1) & 1;
  if (RES == 3) {
    issue_rows(p.res, 64, last == 0 ? 16384 : 0, true);                    // (KT = 9 for block 1: last == 0, the tile at 16 KiB)
    issue_rows(p.wts, 64, last == 0 ? 49152 : 32768, false);
  }
  __builtin_amdgcn_sched_barrier(0);
  compute_stage(last);

  // ---------------------------------------------------------------- the tile becomes conv3's left operand (float16)
  const float bmid = p.bias2[32 * wn + r];
  __syncthreads();  // everyone is done with the phase-1 stage buffers
  // LDS map of phase 2 (KT odd, last == 0): conv2 tile at 0, the unit's input tile at 16 KiB, conv3 weights of the
  // group at 32 KiB, shortcut weights at 48 KiB.  (KT even would swap the roles of the halves; block 1 has KT = 9.)
  const int T2 = last == 0 ? 0 : 16384, SX = last == 0 ? 16384 : 0;
  const int W3 = last == 0 ? 32768 : 49152, WS = last == 0 ? 49152 : 32768;
  issue_rows(p.wt3, 64, W3, false);
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int R = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    const int k2 = 32 * wn + r;
    const float v = fmaxf(acc_hi[q] + acc_lo[q] * kLoScale + bmid, 0.f);
    *reinterpret_cast<_Float16 *>(lds + T2 + R * ROWB + 16 * ((k2 >> 3) ^ ((R >> 1) & 7)) + 2 * (k2 & 7)) = (_Float16)v;
  }

  // ---------------------------------------------------------------- phase 2: conv3 (+ shortcut), 64 channels at a time
  float *Cs = reinterpret_cast<float *>(lds + 32768);   // 128 rows x 64 channels
  const int col4 = tid & 15, row0 = tid >> 4;            // 16 float4 per row, 32 rows per pass
  const int ngroups = p.Cout / 64;
  for (int g = 0; g < ngroups; ++g) {
    // residual of the thread's rows: in flight under the weight DMA and the MFMAs
    float4 rv[4];
    const int n = 64 * g + 4 * col4;
    if (RES != 3) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int mr = m0 + row0 + 32 * i;
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
    }
    float4 bias4 = *reinterpret_cast<const float4 *>(p.bias3 + n);
    if (RES == 3) {
      const float4 bs = *reinterpret_cast<const float4 *>(p.biass + n);
      bias4.x += bs.x; bias4.y += bs.y; bias4.z += bs.z; bias4.w += bs.w;
    }
    // the group's weights (and, first group, the conv2 tile written above) have landed: only the residual loads,
    // issued after them, may still be outstanding
    if (RES != 3) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_hi[q] = acc_lo[q] = 0.f;
    mma_stage(lds + T2 + (wm * 32 + r) * ROWB, lds + W3 + (wn * 32 + r) * ROWB);
    if (RES == 3) mma_stage(lds + SX + (wm * 32 + r) * ROWB, lds + WS + (wn * 32 + r) * ROWB);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // the weights have been read: their area becomes the transpose buffer
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 64 + wn * 32 + r] = acc_hi[q] + acc_lo[q] * kLoScale;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    float4 v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const float4 *>(Cs + (row0 + 32 * i) * 64 + 4 * col4);
    if (g + 1 < ngroups) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();  // the transpose buffer has been read: the next group's weights may land in it
      asm volatile("" ::: "memory");
      issue_rows(p.wt3 + (size_t)(g + 1) * 128 * 64, 64, W3, false);
      if (RES == 3) issue_rows(p.wts + (size_t)(g + 1) * 128 * 64, 64, WS, false);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = m0 + row0 + 32 * i;
      if (m < p.M) {
        float4 o = v[i];
        o.x += bias4.x; o.y += bias4.y; o.z += bias4.z; o.w += bias4.w;
        if (RES != 3) {
          o.x += rv[i].x; o.y += rv[i].y; o.z += rv[i].z; o.w += rv[i].w;
        }
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
        store4(p.y + (size_t)m * p.Cout + n, o);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// The float16 fusion again for the stride-1 units of block 1 (Cin = 64: a pixel's channels are one 128-byte line), built
// around what bounds the kernel above: it is latency-bound -- nine tap stages of 32 KB, each requested one stage (256
// cycles of MFMA per wave) before it is needed, then per 64 output channels a weight fetch and four barriers; 2.4-2.8 TB/s.
//   * A kernel ROW's three taps come from ONE staged run of pixels (conv_gemm_wide16.hip, conv_wide16h_kernel): LDS rows
//     0..127 are pixels m0 .. m0 + 127 of input row ho + kh - 1 at their own column, output pixel i reads rows i - 1, i,
//     i + 1; a tile computes the 126 pixels in between (tiles overlap by two), edge-column lanes are zeroed.  Three
//     activation stages of 16 KB per tile instead of nine, each requested three tap stages ahead.
//   * The weights are one STREAM through a ring of three 16 KB slots, requested two stages ahead and waited for with
//     counted s_waitcnt (loads retire in order): the nine conv2 taps, then conv3's (and the shortcut's) first groups
//     while conv2 is still multiplying.
//   * The opening unit's shortcut operand (the unit's input tile) is fetched under the last kernel row into the activation
//     buffer the middle row has left.
//   * 80 KB per workgroup (two activation buffers, three ring slots), two workgroups per CU as before; the transpose runs
//     in 64-row rounds through the second activation buffer (RES 1) or the third ring slot (RES 3).
// ----------------------------------------------------------------------------------------
constexpr int FHM = BM - 2;   // output pixels per tile

template <int RES>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4)))
void conv3x3_1x1_f16h_kernel(ConvFusedF16Dev p) {
  static_assert(RES == 1 || RES == 3, "stride-1 units only");
  constexpr int NW = 8;
  constexpr float kLoScale = 1.0f / 2048.0f;
  constexpr int A0 = 0, A1 = 16384, RING = 32768, SLOT = 16384;
  __shared__ __attribute__((aligned(16))) char lds[81920];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;   // 4 x 2 waves: 32 pixels x 32 channels (hi + lo)
  const int r = lane & 31, h = lane >> 5;
  const int lrow8 = lane >> 3, lpos = lane & 7;
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  const int m0 = xcd_remap(blockIdx.x, p.mtiles) * FHM - 1;   // LDS row i <-> pixel m0 + i; outputs for i = 1 .. 126

  // ---- staging of an activation stage (kernel row kh): row `row` is pixel m0 + row of the image row above / at / below
  int a_pix[2], a_chunk[2];
  unsigned a_mask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave + NW * i) + lrow8;
    a_chunk[i] = 8 * (lpos ^ ((row >> 1) & 7));
    const int m = m0 + row;
    const bool in = m >= 0 && m < p.M;
    const int mm = in ? m : 0;
    const int wo = mm % p.Wo;
    const int t = mm / p.Wo;
    const int ho = t % p.Ho;
    const int b = t / p.Ho;
    a_pix[i] = (b * p.H + ho - 1) * p.W + wo;
    unsigned mk = 0;
    if (in) {
#pragma unroll
      for (int q = 0; q < 3; ++q)
        if (ho - 1 + q >= 0 && ho - 1 + q < p.H) mk |= 1u << q;
    }
    a_mask[i] = mk;
  }
  auto issue_a = [&](int kh, int dst) __attribute__((always_inline)) {
    const _Float16 *xa = p.x + (long)kh * p.W * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const bool ok = ((a_mask[i] >> kh) & 1u) != 0;
      const void *src = ok ? static_cast<const void *>(xa + ((long)a_pix[i] * 64 + a_chunk[i])) : static_cast<const void *>(&g_zero16f);
      __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + dst + 8 * (wave + NW * i) * ROWB), 16, 0, 0);
    }
  };
  // ---- a 16 KB weight stage: 128 stacked rows x 64 k, rows `row_elems` apart in memory, into ring slot `slot`
  auto issue_w = [&](const _Float16 *src, int row_elems, int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int row = 8 * (wave + NW * i) + lrow8;
      __builtin_amdgcn_global_load_lds((gptr_t)(src + (size_t)row * row_elems + 8 * (lpos ^ ((row >> 1) & 7))),
                                       (lptr_t)(lds + RING + slot * SLOT + 8 * (wave + NW * i) * ROWB), 16, 0, 0);
    }
  };
  // stream element q: 0..8 conv2's taps, then conv3's (RES 3: alternating with the shortcut's) groups
  auto issue_q = [&](int q) __attribute__((always_inline)) {
    if (q < 9) issue_w(p.wt2 + q * 64, 576, q % 3);
    else if (RES == 1) issue_w(p.wt3 + (size_t)(q - 9) * 128 * 64, 64, (q - 9) & 1);
    else if (q == 9) issue_w(p.wt3, 64, 0);
    else issue_w(p.wts, 64, 1);
  };

  floatx16 acc_hi, acc_lo;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc_hi[q] = acc_lo[q] = 0.f;
  // the lane's output pixel and whether its left / right neighbour exists
  const int pr = wm * 32 + r;
  bool no_left, no_right;
  {
    const int m = m0 + pr;
    const int mm = m >= 0 && m < p.M ? m : 0;
    const int wo = mm % p.Wo;
    no_left = wo == 0;
    no_right = wo == p.W - 1;
  }
  // 64 k of one tap: activation rows shifted by kw - 1 in buffer `abuf`, weights in ring slot `slot`
  auto mma_tap = [&](int abuf, int kw, int slot) __attribute__((always_inline)) {
    int rr = pr + kw - 1;
    asm volatile("" : "+v"(rr));
    rr = rr < 0 ? 0 : (rr > BM - 1 ? BM - 1 : rr);
    const bool zero = (kw == 0 && no_left) || (kw == 2 && no_right);
    const char *a_row = lds + abuf + rr * ROWB;
    const int swa = (rr >> 1) & 7;
    const char *b_row = lds + RING + slot * SLOT + (wn * 32 + r) * ROWB;
    const int swb = (r >> 1) & 7;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      halfx8 a = *reinterpret_cast<const halfx8 *>(a_row + 16 * ((2 * kb + h) ^ swa));
      if (zero) a = halfx8{0, 0, 0, 0, 0, 0, 0, 0};
      const halfx8 bh = *reinterpret_cast<const halfx8 *>(b_row + 16 * ((2 * kb + h) ^ swb));
      const halfx8 bl = *reinterpret_cast<const halfx8 *>(b_row + 64 * ROWB + 16 * ((2 * kb + h) ^ swb));
      acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bh, acc_hi, 0, 0, 0);
      acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bl, acc_lo, 0, 0, 0);
    }
  };

  // ---------------------------------------------------------------- phase 1: conv2, nine tap stages
  issue_a(0, A0);
  issue_q(0);
  issue_q(1);
  auto tap_stage = [&](auto T) __attribute__((always_inline)) {
    constexpr int t = decltype(T)::value;
    constexpr int kh = t / 3, kw = t % 3;
    // the stage's weights (and its activation stage) have landed when only what was requested after them is outstanding:
    // one weight stage (2 instructions per wave), plus the next activation stage behind taps 1 and 4 (NOT 7: kernel row 2
    // is the last -- a vmcnt(4) there let tap 7 start on weights still in flight, ~10 tiles in 16 000 at 3840x2160)
    // (RES 3: and the shortcut's operand behind tap 7)
    if (t == 1 || t == 4 || (RES == 3 && t == 7)) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // visible to all; everyone has read stage t - 1
    asm volatile("" ::: "memory");
    if (kw == 0 && kh < 2) issue_a(kh + 1, (kh + 1) & 1 ? A1 : A0);
    if (RES == 3 && t == 6) {   // the unit's input tile (rows = this tile's pixels) into A1, free since tap 5
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave + NW * i) + lrow8;
        const int mr = m0 + row;
        const int m = mr < 0 ? 0 : (mr < p.M ? mr : p.M - 1);
        __builtin_amdgcn_global_load_lds((gptr_t)(p.res + (size_t)m * CSC + 8 * (lpos ^ ((row >> 1) & 7))),
                                         (lptr_t)(lds + A1 + 8 * (wave + NW * i) * ROWB), 16, 0, 0);
      }
    }
    issue_q(t + 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_tap(kh & 1 ? A1 : A0, kw, t % 3);
    __builtin_amdgcn_sched_barrier(0);
  };
  tap_stage(std::integral_constant<int, 0>{});
  tap_stage(std::integral_constant<int, 1>{});
  tap_stage(std::integral_constant<int, 2>{});
  tap_stage(std::integral_constant<int, 3>{});
  tap_stage(std::integral_constant<int, 4>{});
  tap_stage(std::integral_constant<int, 5>{});
  tap_stage(std::integral_constant<int, 6>{});
  tap_stage(std::integral_constant<int, 7>{});
  tap_stage(std::integral_constant<int, 8>{});

  // ---------------------------------------------------------------- the tile becomes conv3's left operand (float16, in A0)
  const float bmid = p.bias2[32 * wn + r];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();   // everyone is done with kh = 2's activation stage (A0)
  asm volatile("" ::: "memory");
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int R = wm * 32 + (q & 3) + 8 * (q >> 2) + 4 * h;
    const int k2 = 32 * wn + r;
    const float v = fmaxf(acc_hi[q] + acc_lo[q] * kLoScale + bmid, 0.f);
    *reinterpret_cast<_Float16 *>(lds + A0 + R * ROWB + 16 * ((k2 >> 3) ^ ((R >> 1) & 7)) + 2 * (k2 & 7)) = (_Float16)v;
  }

  // ---------------------------------------------------------------- phase 2: conv3 (+ shortcut), 64 channels at a time
  float *Cs = reinterpret_cast<float *>(lds + (RES == 1 ? A1 : RING + 2 * SLOT));   // [64][64] float32
  const int col4 = tid & 15, row0 = tid >> 4;   // 16 float4 per row, 32 rows per pass
  const int ngroups = p.Cout / 64;
  for (int g = 0; g < ngroups; ++g) {
    const int n = 64 * g + 4 * col4;
    float4 bias4 = *reinterpret_cast<const float4 *>(p.bias3 + n);
    if (RES == 3) {
      const float4 bs = *reinterpret_cast<const float4 *>(p.biass + n);
      bias4.x += bs.x; bias4.y += bs.y; bias4.z += bs.z; bias4.w += bs.w;
    }
    float4 rv[2][2];
    if (RES == 1) {   // residual rows of both rounds: in flight under the MFMAs and the transposes
#pragma unroll
      for (int rho = 0; rho < 2; ++rho)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int mr = m0 + 64 * rho + row0 + 32 * i;
          const int m = mr < 0 ? 0 : (mr < p.M ? mr : p.M - 1);
          rv[rho][i] = load4(p.res + (size_t)m * p.Cout + n);
        }
      asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");   // all but the four residual loads: the group's weights are in
    } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#pragma unroll
    for (int q = 0; q < 16; ++q) acc_hi[q] = acc_lo[q] = 0.f;
    {
      const int slot = RES == 1 ? (g & 1) : 0;
      const char *a_row = lds + A0 + (wm * 32 + r) * ROWB;
      const char *b_row = lds + RING + slot * SLOT + (wn * 32 + r) * ROWB;
      const int sw = (r >> 1) & 7;
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int co = 16 * ((2 * kb + h) ^ sw);
        const halfx8 a = *reinterpret_cast<const halfx8 *>(a_row + co);
        acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, *reinterpret_cast<const halfx8 *>(b_row + co), acc_hi, 0, 0, 0);
        acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, *reinterpret_cast<const halfx8 *>(b_row + 64 * ROWB + co), acc_lo, 0, 0, 0);
      }
      if (RES == 3) {   // + the shortcut conv of the unit's input tile, in A1
        const char *x_row = lds + A1 + (wm * 32 + r) * ROWB;
        const char *s_row = lds + RING + SLOT + (wn * 32 + r) * ROWB;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const int co = 16 * ((2 * kb + h) ^ sw);
          const halfx8 a = *reinterpret_cast<const halfx8 *>(x_row + co);
          acc_hi = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, *reinterpret_cast<const halfx8 *>(s_row + co), acc_hi, 0, 0, 0);
          acc_lo = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, *reinterpret_cast<const halfx8 *>(s_row + 64 * ROWB + co), acc_lo, 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();   // the group's weight slot(s) have been read; so has the previous group's transpose
    asm volatile("" ::: "memory");
    if (RES == 1) {
      if (g + 2 < ngroups) issue_w(p.wt3 + (size_t)(g + 2) * 128 * 64, 64, g & 1);
    } else if (g + 1 < ngroups) {
      issue_w(p.wt3 + (size_t)(g + 1) * 128 * 64, 64, 0);
      issue_w(p.wts + (size_t)(g + 1) * 128 * 64, 64, 1);
    }
#pragma unroll
    for (int rho = 0; rho < 2; ++rho) {
      if (rho == 1) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // the first round's rows have been read
        asm volatile("" ::: "memory");
      }
      if ((wm >> 1) == rho) {
#pragma unroll
        for (int q = 0; q < 16; ++q)
          Cs[((wm & 1) * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * 64 + wn * 32 + r] = acc_hi[q] + acc_lo[q] * kLoScale;
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row = row0 + 32 * i;
        const int prow = 64 * rho + row;
        const int m = m0 + prow;
        if (prow >= 1 && prow <= FHM && m < p.M) {
          float4 o = *reinterpret_cast<const float4 *>(Cs + row * 64 + 4 * col4);
          o.x += bias4.x; o.y += bias4.y; o.z += bias4.z; o.w += bias4.w;
          if (RES == 1) {
            o.x += rv[rho][i].x; o.y += rv[rho][i].y; o.z += rv[rho][i].z; o.w += rv[rho][i].w;
          }
          o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
          store4(p.y + (size_t)m * p.Cout + n, o);
        }
      }
    }
  }
}

int g_fused_hreuse = 1;   // dvsg_debug_set_option("fused_hreuse", 0): block 1's stride-1 units through conv3x3_1x1_f16_kernel

int g_fuse_conv = 1;  // dvsg_debug_set_option("fuse_conv", 0) turns the fused block-1 path off

}  // namespace

void set_fuse_conv(int v) { g_fuse_conv = v; }
int get_fuse_conv() { return g_fuse_conv; }
void set_fused_hreuse(int v) { g_fused_hreuse = v; }
bool conv_fusable(int prec, int Cin, int Cmid, int Cout, int ksize) {
  if (prec == kF16) return g_fuse_conv != 0 && ksize == 3 && Cmid == CMID && Cin % 64 == 0 && Cout % 64 == 0;
  return g_fuse_conv != 0 && (prec == kF32 || prec == kF32S) && ksize == 3 && Cmid == CMID && Cin % 32 == 0 && Cin >= 64 && Cout % 128 == 0;
}

static int launch_conv3x3_1x1_f16(const ConvFused &p, hipStream_t s) {
  DVSG_REQUIRE(p.Cin % 64 == 0 && p.Cin >= 64 && p.Cout % 64 == 0, "conv3x3_1x1 (float16): Cin=%d and Cout=%d must be multiples of 64",
               p.Cin, p.Cout);
  const bool sc = p.sc_x != nullptr;
  DVSG_REQUIRE(p.res || sc, "conv3x3_1x1: the unit output needs its residual");
  DVSG_REQUIRE(!sc || (p.sc_wt && p.sc_bias && p.sc_cin == CSC && p.stride == 1),
               "conv3x3_1x1 (float16): fused shortcut needs %d input channels and stride 1 (got %d, %d)", CSC, p.sc_cin, p.stride);
  const long M = (long)p.B * p.Ho * p.Wo;
  DVSG_REQUIRE(M > 0 && M < (1L << 31) - BM, "conv3x3_1x1: M=%ld out of range", M);
  ConvFusedF16Dev d;
  d.x = reinterpret_cast<const _Float16 *>(p.x);
  d.wt2 = reinterpret_cast<const _Float16 *>(p.wt2); d.bias2 = p.bias2;
  d.wt3 = reinterpret_cast<const _Float16 *>(p.wt3); d.bias3 = p.bias3;
  d.res = reinterpret_cast<const _Float16 *>(sc ? p.sc_x : p.res);
  d.wts = reinterpret_cast<const _Float16 *>(p.sc_wt); d.biass = p.sc_bias;
  d.y = reinterpret_cast<_Float16 *>(p.y);
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.M = (int)M;
  d.mtiles = (int)((M + BM - 1) / BM);
  const int res = sc ? 3 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  ProfScope prof(kClsFused, s,
                 2.0 * (double)M * CMID * (9.0 * p.Cin) + 2.0 * (double)M * p.Cout * CMID + (sc ? 2.0 * (double)M * p.Cout * CSC : 0.0),
                 2.0 * ((double)p.B * p.H * p.W * p.Cin + 2.0 * CMID * 9 * p.Cin + 2.0 * p.Cout * CMID +
                        (sc ? (double)M * CSC + 2.0 * p.Cout * CSC + (double)M * p.Cout : 2.0 * (double)M * p.Cout)));
  if (g_fused_hreuse && p.stride == 1 && p.Cin == 64 && p.Cout >= 128 && p.H == p.Ho && p.W == p.Wo && (res == 1 || res == 3)) {
    d.mtiles = (int)((M + FHM - 1) / FHM);
    const dim3 gridh(d.mtiles), blockh(512);
    if (res == 1) hipLaunchKernelGGL((conv3x3_1x1_f16h_kernel<1>), gridh, blockh, 0, s, d);
    else hipLaunchKernelGGL((conv3x3_1x1_f16h_kernel<3>), gridh, blockh, 0, s, d);
    return check_launch("conv3x3_1x1_f16h_kernel");
  }
  const dim3 grid(d.mtiles), block(512);
  if (res == 1) hipLaunchKernelGGL((conv3x3_1x1_f16_kernel<1>), grid, block, 0, s, d);
  else if (res == 2) hipLaunchKernelGGL((conv3x3_1x1_f16_kernel<2>), grid, block, 0, s, d);
  else hipLaunchKernelGGL((conv3x3_1x1_f16_kernel<3>), grid, block, 0, s, d);
  return check_launch("conv3x3_1x1_f16_kernel");
}

int launch_conv3x3_1x1(const ConvFused &p, hipStream_t s) {
  if (p.f16) return launch_conv3x3_1x1_f16(p, s);
  if (p.x3) return launch_conv3x3_1x1_x3(p, s);
  DVSG_REQUIRE(p.Cin % 32 == 0 && p.Cin >= 64 && p.Cout % 128 == 0, "conv3x3_1x1: Cin=%d must be a multiple of 32 (>= 64), Cout=%d of 128",
               p.Cin, p.Cout);
  const bool sc = p.sc_x != nullptr;  // the residual is the 1x1 shortcut conv of sc_x, computed in the kernel
  DVSG_REQUIRE(p.res || sc, "conv3x3_1x1: the unit output needs its residual");
  DVSG_REQUIRE(!sc || (p.sc_wt && p.sc_bias && p.sc_cin == CSC && p.Cout == 256 && p.stride == 1),
               "conv3x3_1x1: fused shortcut needs %d input channels, 256 output channels and stride 1 (got %d, %d, %d)",
               CSC, p.sc_cin, p.Cout, p.stride);
  const long M = (long)p.B * p.Ho * p.Wo;
  DVSG_REQUIRE(M > 0 && M < (1L << 31) - BM, "conv3x3_1x1: M=%ld out of range", M);
  ConvFusedDev d;
  d.x = p.x; d.wt2 = p.wt2; d.bias2 = p.bias2; d.wt3 = p.wt3; d.bias3 = p.bias3; d.y = p.y;
  d.res = sc ? p.sc_x : p.res; d.wts = p.sc_wt; d.biass = p.sc_bias;
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.M = (int)M;
  d.mtiles = (int)((M + BM - 1) / BM);
  const int res = sc ? 3 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  // algorithmic work: the contractions; bytes = input + weight sets + residual (or the shortcut's input) + output, once each
  ProfScope prof(kClsFused, s,
                 2.0 * (double)M * CMID * (9.0 * p.Cin) + 2.0 * (double)M * p.Cout * CMID + (sc ? 2.0 * (double)M * p.Cout * CSC : 0.0),
                 4.0 * ((double)p.B * p.H * p.W * p.Cin + (double)CMID * 9 * p.Cin + (double)p.Cout * CMID +
                        (sc ? (double)M * CSC + (double)p.Cout * CSC + (double)M * p.Cout : 2.0 * (double)M * p.Cout)));
  const dim3 grid(d.mtiles), block(512);
  if (p.pieces) {
    if (res == 1) hipLaunchKernelGGL((conv3x3_1x1_kernel<1, true>), grid, block, 0, s, d);
    else if (res == 2) hipLaunchKernelGGL((conv3x3_1x1_kernel<2, true>), grid, block, 0, s, d);
    else hipLaunchKernelGGL((conv3x3_1x1_kernel<3, true>), grid, block, 0, s, d);
  } else {
    if (res == 1) hipLaunchKernelGGL((conv3x3_1x1_kernel<1>), grid, block, 0, s, d);
    else if (res == 2) hipLaunchKernelGGL((conv3x3_1x1_kernel<2>), grid, block, 0, s, d);
    else hipLaunchKernelGGL((conv3x3_1x1_kernel<3>), grid, block, 0, s, d);
  }
  return check_launch("conv3x3_1x1_kernel");
}

}  // namespace dvsg
