// slim conv2d (1x1, and 3x3 conv2d_same) + folded BatchNorm + residual + ReLU of
// resnet_v1_50 (networks.py:33-34 -> tf.contrib.slim bottleneck_v1) as ONE implicit-GEMM
// kernel on the matrix cores:
//
//     Y[m, n] = act( sum_k A[m, k] Wt[n, k] + bias[n] (+ res[m', n]) ),
//     m = (b, ho, wo), k = (kh, kw, c).
//
// NHWC activations make each (kh, kw) tap of a pixel a contiguous run of C_in elements, so an
// A tile row is one 128-byte run of global memory (or 128 bytes of zeros where the tap falls
// outside the image) and the im2col matrix never exists.
//
// Tile 128 (m) x BN (n) x 128 bytes of k (32 f32 / 64 f16); WM x WN waves, each owning a
// (128/WM) x (BN/WN) patch of 32x32 MFMA blocks.
//
// Staging is direct-to-LDS (global_load_lds_dwordx4): tiles go L2 -> LDS without passing through
// VGPRs, so a stage has no ds_write and no staging registers.  An LDS-DMA wave-instruction
// writes 64 x 16 B = 8 tile rows LINEARLY (wave-uniform base + lane x 16), so rows cannot be
// padded; bank conflicts are avoided by an XOR swizzle applied on the per-lane GLOBAL source
// address and again on the fragment read: row R keeps its 16-byte k-chunk c at position
// c ^ ((R >> 1) & 7).  Over the 16 rows of every ds_read_b128 lane group the pair
// (R & 1, (R >> 1) & 7) is distinct, i.e. the 16 reads hit the 16 distinct 16-byte slots of the
// 256-byte bank row: conflict-free without padding.  Two LDS stages; the DMA of stage kt+1 is in
// flight while stage kt is multiplied; one barrier per stage.
//
// Epilogue: a lane of the 32x32 C/D layout owns ONE output channel, which would mean 16 dword
// stores (and residual loads) per MFMA block per lane; instead the accumulators are transposed
// through the idle staging LDS so every thread moves 4 consecutive channels (float4 / 4 x f16).
//
// (Measured again in round 2 with sustained clocks, tools/conv_bench.py --variants 0,7: a resident grid
// of 512 / 768 workgroups that walk the tiles at a static stride, nothing else changed, is 1.3 % slower
// over the 52 launches of a step -- +2 % on the K = 64 layers, -2..-17 % on the others: workgroup
// turnover is not what small-K tiles lose, and a static assignment gives up the dispatcher's balancing.
// Four fat waves (64 x 64 wave tiles: a third fewer fragment reads per MFMA) in every mode, stream-K
// included: 17.83 vs 17.76 ms, i.e. nothing -- LDS read volume is not a lever either.)
//
// (Measured and rejected alternatives to this epilogue, batch 16 at 720p: storing straight from the
// accumulators -- 16 dword stores per block, or, with the MFMA operands swapped so a lane holds 4
// consecutive channels of one pixel, 4 x 16-byte stores of 32 bytes per pixel per instruction --
// was 3-17 % slower per layer; persistent workgroups that put the next tile's first K stage in
// flight before the current tile's last one, with the transpose squeezed into the one free stage
// buffer in two 64-row rounds, gained 7 % on the K = 64 layers and lost 2-5 % everywhere else.)
//
// (float16 mode, round 2: its loop is bound by the L2 -> LDS stream -- tools/pieces_loop_bench.hip: every 128 x 128
// configuration, 4 or 8 waves, fragment reads pipelined or not, runs at the same ~15 TB/s of LDS-DMA -- and a 128 x 256
// tile of two [hi | lo] channel groups with a ring of three LDS stages, one workgroup per CU, is 20 % faster in that
// loop.  Built into this kernel (BN = 256, plain tiles) it LOST 4.5 % end to end at 720p and 6 % at 4K (3x3 class 2.72
// vs 2.54 ms): with a single workgroup per CU nobody multiplies while a tile's prologue, first-stage wait and 133 KB
// transpose run.  Removed again; what keeps two workgroups per CU at the same bytes per product is a 256 x 128 tile
// with 64-byte K stages: conv_gemm_wide16.hip, which launch_conv_gemm picks for float16 launches of >= 256 such tiles.
// The same geometry for EXACT float32 -- there to spread a short-K tile's prologue, first-stage wait and epilogue over
// twice the products -- was 5-10 % slower on exactly those layers (block 2's 128 -> 512: 333 vs 318 us, block 3's
// 256 -> 1024: 295 vs 268 us; tools/conv_bench.py) and 4 % over the 52 launches of a step.)
//
// Where the main loop's time goes (ablations at batch 16, 720p, plain tiles): with the LDS-DMA
// removed the 52 launches of a step take 17.8 ms instead of 19.6, and exactly the same with the DMA
// left in but the activation addresses folded into a 256 KB (L2-resident) window; removing the
// per-stage barriers as well changes nothing.  So instruction issue and barriers are free and the
// 9 % is the service of L2 misses on the activation stream (one stage of lead does not cover it).
// An L2 prefetch of the stage after next (4-byte LDS-DMA per 8 rows into a scratch corner, counted
// vmcnt waits so nobody waits for it) made the launches 4 % SLOWER, and a third LDS stage for the
// 64-wide tiles (two stages of lead, still two workgroups per CU) changed nothing: it is not the
// lead that is short.
//
// Work decomposition (MODE):
//   0  one workgroup per output tile.
//   1  split-K: a launch with too few tiles for the chip (batch 1-2) gives each tile to `ksplit`
//      workgroups, each running a contiguous slice of the K stages.
//   2  stream-K tail: blocks [0, tile_begin) run the tiles of the full rounds as in mode 0; in the
//      SAME launch 512 more blocks share the tiles of the last, partly filled round (or a whole
//      launch of 256-511 tiles) as equal runs of (tile, K-stage) units, so that round costs its share
//      of the work instead of a full tile time.
// Modes 1 and 2 reduce in the launch, last-arriver form: a contributor publishes its raw partial
// tile, takes a ticket, and whoever draws the last ticket of a tile sums all partials in K order
// (bitwise reproducible) and runs the epilogue.  No workgroup ever waits for another one.
#pragma once
#include <algorithm>

#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {

// launch-policy switches (conv_gemm.hip)
extern int g_conv_variant;
extern long g_wide16_min_tiles;

struct ConvGemmDev {
  const void *x, *wt, *res;
  const float *bias;
  void *y;
  float *slabs;    // partial tiles of modes 1 / 2
  int *counters;   // arrival tickets, zeroed before the launch
  int ksplit;      // mode 1: slices per tile
  int tile_begin, tile_count;  // tiles [tile_begin, tile_begin + tile_count) belong to this launch
  int H, W, Cin, Ho, Wo, Cout;
  int stride, pad;
  int res_H, res_W, res_stride;
  int M, K, mtiles, ntiles;
  int mt_fast;     // tile order, see run_segment
  int ldx;         // elements between two pixels of x (Cin, or more when x is a column range of a wider tensor)
  int res_ld;      // the same for the residual (Cout by default)
  int relu_from;   // RELU = false instantiations: output channels >= relu_from get the ReLU all the same (INT_MAX: none)
};

// the f32x3 instantiations (conv_gemm_x3.hip), called by launch_conv_gemm with its tile / split-K / stream-K decisions made
int launch_conv_gemm_x3(const ConvGemmDev &d, int ksize, bool wide, int streamk_tail, bool relu, int res, hipStream_t s);

namespace {

constexpr int BM = 128;
constexpr int ROWB = 128;        // bytes of k per tile row and stage
static_assert(kSplitKMaxTiles >= 512, "a stream-K launch has up to 511 tiles, one ticket each");
constexpr int kResident = 512;   // workgroups of the 128-wide 8-wave configuration resident on the chip (2 per CU)

// what a tap outside the image reads (conv2d_same zero padding)
__device__ const floatx4 g_zero16 = {0.f, 0.f, 0.f, 0.f};


// SPLIT, T = _Float16 ("f16" precision): the weight matrix holds, for every group of 64 output channels,
// 128 rows -- the float16 weights of the group (hi) followed by their rounding residuals (w - hi) x 2^11,
// again float16 (lo).  A 128-wide tile is then [hi | lo] of ONE 64-channel group and the epilogue folds it
// to 64 output channels, hi + 2^-11 lo: float16 activations against effectively float32 weights.  (A
// float16 weight carries a FIXED relative error of up to 2^-12 that is the same at every pixel, so it
// survives the global average pool; it was 9/10 of the float16 mode's error in F_t.  The scale keeps lo a
// normal float16.)
//
// SPLIT, T = float ("f32s" precision): float32 storage everywhere, float32 accumulation, but every
// PRODUCT is formed on the float16 matrix cores from two float16 pieces per operand, x ~ x1 + x2 with
// x1 = rtz_f16(x), x2 = f16(x - x1): 22 significant bits for |x| >= 2^-3 (x2 unscaled: absolute step 2^-24 below).  a w ~ a1 w1 + a2 w1 + a1 w2 (the dropped a2 w2
// is 2^-22 relative), each float16 product exact in float32 -- three v_mfma_f32_32x32x16_f16 (96 cycles)
// where the exact path issues eight v_mfma_f32_32x32x2_f32 (512 cycles).  Activations are split in
// once by their producer (an epilogue stores the two pieces of each value: cnn_device.h, P format), weights
// once at load: an activation or weight row stage is 128 bytes like a float32 one -- 32 hi halves, then 32
// lo halves -- so tiles, LDS-DMA and swizzle are the float32 kernel's, byte for byte, and the main loop
// has no arithmetic but the MFMAs.  (Splitting the float32 activations in registers after the fragment
// read instead -- ~24 VALU per 8 values and wave -- ran the same layers at 270 instead of ... TFLOP/s and
// held the shader clock at 1.79 GHz.)  float16 subnormals are inputs
// the matrix cores keep (tools/f16_subnormal_probe.hip), so nothing is scaled.  Against the exact path:
// stage activations within 2e-6 relative, F_t within 1e-7 (a CPU emulation of the arithmetic and the GPU
// tests agree), i.e. at the level at which two float32 GEMMs with different summation orders differ.
//
// X3, T = float ("f32x3" precision): float32 tensors exactly as in the exact path, float32 accumulation, and every
// PRODUCT formed on the bfloat16 matrix cores from THREE bfloat16 pieces per operand, x = x1 + x2 + x3 with
// x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2), each rounded to nearest: bfloat16 has float32's exponent
// range and 8 significant bits, so the three pieces hold all 24 bits of ANY float32 (no range condition, unlike the
// float16 pieces above) and every piece product is exact in float32.  a w = sum of 9 cross terms; the six largest --
// a1 w1, a1 w2, a2 w1, a2 w2, a1 w3, a3 w1 -- are six v_mfma_f32_32x32x16_bf16 (192 cycles per 16 k where the exact
// path's eight v_mfma_f32_32x32x2_f32 take 512); the dropped a2 w3 + a3 w2 + a3 w3 are at most 2^-23 |a w| (|x2| <= 2^-8 |x|,
// |x3| <= 2^-16 |x|), the size of ONE rounding of a float32 multiply, typically 2^-27.  The activations stay float32 in
// HBM and LDS (rows, LDS-DMA, swizzle, epilogue: the exact kernel's) and are split in registers behind the fragment
// read -- ~40 VALU per 8 values, which shares the issue port with the MFMAs: wave tiles are 64 wide along N so that a
// split serves two weight blocks (3.3 VALU per MFMA).  The weights are split once at load and packed stage by stage in
// the LDS image's order (pack_x3 in locnet.hip): per group of 64 output channels and 32-k stage three 4 KB piece
// planes [64 rows][32 k] whose row R keeps its 16-byte chunk c at position c ^ ((R >> 2) & 3) (16 consecutive rows of a
// ds_read_b128 lane group hit 16 distinct slots), 12 KB that the LDS-DMA copies linearly.
template <typename T, int BN, int WM, int WN, int KS, bool RELU, int RES, int MODE, bool SPLIT = false, bool X3 = false>
__global__ __launch_bounds__(64 * WM * WN) __attribute__((amdgpu_waves_per_eu(WM * WN / 2, WM * WN / 2)))
void conv_gemm_kernel(ConvGemmDev p) {
  constexpr bool WSTACK = SPLIT && sizeof(T) == 2;  // f16: [hi | lo] weight rows stacked along N
  constexpr bool PSPLIT = SPLIT && sizeof(T) == 4;  // f32s: products from float16 pieces
  static_assert(!WSTACK || BN == 128, "stacked hi / lo weights: 128-wide tiles");
  static_assert(!X3 || (sizeof(T) == 4 && !SPLIT && BN % 64 == 0), "f32x3: float32 tensors");
  constexpr int BROWB = X3 ? 192 : ROWB;    // bytes of weights per tile row and stage
  constexpr int NW = WM * WN;
  constexpr int NT = 64 * NW;
  constexpr int MI = BM / WM / 32;          // 32-row MFMA blocks per wave
  constexpr int NI = BN / WN / 32;          // 32-col MFMA blocks per wave
  constexpr int AG = BM / 8 / NW;           // 8-row groups (one LDS-DMA instruction each) per wave, A tile
  constexpr int BG = X3 ? BN / 64 * 12 / NW : BN / 8 / NW;   // same for the weight tile (f32x3: 1 KB runs of the packed stage)
  static_assert(!X3 || BN / 64 * 12 % NW == 0, "f32x3: the 12 KB of a 64-row group's stage split evenly over the waves");
  constexpr int BKE = ROWB / sizeof(T);     // k elements per stage: 32 (f32) / 64 (f16)
  constexpr int CHE = 16 / sizeof(T);       // elements per 16-byte chunk
  static_assert(MI >= 1 && NI >= 1 && AG >= 1 && BG >= 1, "bad tile configuration");
  constexpr int LDS_STAGE = 2 * (BM * ROWB + BN * BROWB);
  constexpr int LDS_EPI = BM * (BN + 4) * 4;
  constexpr int LDS_MAIN = LDS_STAGE > LDS_EPI ? LDS_STAGE : LDS_EPI;
  // ONE shared object (a second one makes hipcc drain vmcnt before every fragment read); the last
  // 16 bytes carry the ticket of modes 1 / 2
  // (f32x3, 128-wide: 80 KB of stages, two workgroups per CU fill the 160 KB exactly -- the ticket sits in the stage area,
  // behind the epilogue's transpose image, which is all that is live when it is drawn)
  constexpr int TICKET_OFF = X3 ? LDS_MAIN - 16 : LDS_MAIN;
  static_assert(!X3 || LDS_MAIN - LDS_EPI >= 16, "f32x3: the ticket needs 16 bytes behind the transpose image");
  __shared__ __attribute__((aligned(16))) char lds[TICKET_OFF + 16];
  char *As = lds;
  char *Bs = lds + 2 * BM * ROWB;
  const T *px = static_cast<const T *>(p.x);
  const T *pw = static_cast<const T *>(p.wt);
  const T *pres = static_cast<const T *>(p.res);
  T *py = static_cast<T *>(p.y);

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int r = lane & 31, h = lane >> 5;
  const int lrow8 = lane >> 3, lpos = lane & 7;
  const int KT_all = p.K / BKE;

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  typedef typename Frag<T>::type frag_t;
#ifdef DVSG_STAMPS  // diagnostic build (tools/stamp_probe.py): per-workgroup phase times, mode 0 only
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  st[0] = __builtin_amdgcn_s_memtime();
  st[6] = __builtin_amdgcn_s_memrealtime();
#define DVSG_STAMP(i) st[i] = __builtin_amdgcn_s_memtime()
#else
#define DVSG_STAMP(i)
#endif

  // One (tile, K-stage range) segment.  `n_contrib` contributors share the tile; this one is
  // number `own` (in K order), its partial goes to slab `slab_of(own)`, the tile's ticket is
  // `ticket_idx`.  n_contrib == 1: plain tile.
  auto run_segment = [&](int tile, int kt0, int kt1, int n_contrib, int own, int ticket_idx,
                         auto slab_of) __attribute__((always_inline)) {
    // tile -> (mt, nt): nt fastest by default (neighbouring tiles share the activation panel);
    // mt fastest when one weight panel is a large part of an XCD's L2 (see launch_conv_gemm)
    int mt, nt;
    if (p.mt_fast) {
      nt = tile / p.mtiles;
      mt = tile - nt * p.mtiles;
    } else {
      mt = tile / p.ntiles;
      nt = tile - mt * p.ntiles;
    }
    const int m0 = mt * BM, n0 = nt * BN;

    // ---- staging: wave `wave` fills row groups g = wave + NW i; lane -> row 8 g + lane / 8,
    // LDS chunk position lane % 8, which must receive global chunk pos ^ ((row >> 1) & 7).
    long a_off[AG];       // element offset of the pixel's (kh=0, kw=0, c=chunk) tap
    unsigned a_mask[AG];  // bit kh: input row valid, bit 4+kw: input col valid
    // A 1x1 convolution of stride 1 -- every 1x1 layer of resnet_v1_50 -- is a plain row-major GEMM:
    // row m starts at x + m Cin.  No pixel decomposition (three integer divisions per row group, ~100
    // instructions, a third of the prologue of a K = 64 tile whose whole main loop is 64 MFMAs).
    const bool dense = KS == 1 && p.stride == 1;
#pragma unroll
    for (int i = 0; i < AG; ++i) {
      const int row = 8 * (wave + NW * i) + lrow8;
      const int chunk = lpos ^ ((row >> 1) & 7);
      const int m = m0 + row;
      const int mm = m < p.M ? m : 0;
      if (dense) {
        a_off[i] = (long)mm * p.ldx + CHE * chunk;
        a_mask[i] = m < p.M ? 0x11u : 0u;
        continue;
      }
      const int wo = mm % p.Wo;
      const int t = mm / p.Wo;
      const int ho = t % p.Ho;
      const int b = t / p.Ho;
      const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
      a_off[i] = (((long)b * p.H + hi0) * p.W + wi0) * p.ldx + CHE * chunk;
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
    const T *wsrc[BG];
#pragma unroll
    for (int i = 0; i < BG; ++i) {
      if constexpr (X3) {   // 1 KB run number (wave + NW i) of the tile's packed stage: group run / 12, 16 bytes per lane
        const int run = wave + NW * i;
        wsrc[i] = reinterpret_cast<const T *>(reinterpret_cast<const char *>(p.wt) +
                                             ((size_t)(n0 / 64 + run / 12) * KT_all * 12 + run % 12) * 1024 + lane * 16);
      } else {
        const int row = 8 * (wave + NW * i) + lrow8;
        wsrc[i] = pw + (size_t)(n0 + row) * p.K + CHE * (lpos ^ ((row >> 1) & 7));
      }
    }

    // K order of a 3x3 conv: channel chunk OUTER, the 9 taps INNER (k order is free as long as A and
    // the weights agree: the weight stage of (tap, c0) is just column tap * Cin + c0 of the same
    // matrix).  Consecutive stages then re-read the same input lines shifted by one pixel / one row,
    // i.e. out of L1 / L2, instead of coming back to them one tap (Cin / 32 stages x 64 workgroups x
    // 16 KB, more than an XCD's L2) later.  (kh, kw, c0) advance incrementally: no division in the loop.
    int s_kh = 0, s_kw = 0, s_c0 = kt0 * BKE;
    if (KS > 1) {
      const int chunk = kt0 / (KS * KS), tap = kt0 - chunk * (KS * KS);
      s_c0 = chunk * BKE;
      s_kh = tap / KS;
      s_kw = tap - s_kh * KS;
    }
    auto issue_stage = [&](int buf) __attribute__((always_inline)) {
      const T *xa = px + ((long)s_kh * p.W + s_kw) * p.ldx + s_c0;
      const int wk = KS > 1 ? (s_kh * KS + s_kw) * p.Cin + s_c0 : s_c0;
#pragma unroll
      for (int i = 0; i < AG; ++i) {
        const bool ok = ((a_mask[i] >> s_kh) & (a_mask[i] >> (4 + s_kw)) & 1u) != 0;
        const void *src = ok ? static_cast<const void *>(xa + a_off[i]) : static_cast<const void *>(&g_zero16);
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(As + (buf * BM + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < BG; ++i) {
        if constexpr (X3)   // stage wk / 32 of the group: 12 KB further per stage
          __builtin_amdgcn_global_load_lds((gptr_t)(reinterpret_cast<const char *>(wsrc[i]) + (size_t)(wk >> 5) * (12 * 1024)),
                                           (lptr_t)(Bs + buf * BN * BROWB + (wave + NW * i) * 1024), 16, 0, 0);
        else
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc[i] + wk),
                                           (lptr_t)(Bs + (buf * BN + 8 * (wave + NW * i)) * ROWB), 16, 0, 0);
      }
      if (KS > 1) {
        if (++s_kw == KS) {
          s_kw = 0;
          if (++s_kh == KS) {
            s_kh = 0;
            s_c0 += BKE;
          }
        }
      } else {
        s_c0 += BKE;
      }
    };

    floatx16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[mi][ni][q] = 0.f;
    // f32x3: the five small cross terms of a product accumulate apart from the large one.  An MFMA adds its 16 products to the
    // accumulator with bits below the accumulator's last place cut off (a bias towards -inf that survives the network's global
    // average pool: measured as a mean error of -5e-8 of the pooled features with ONE accumulator, four times the exact
    // path's); kept apart, the small terms meet an accumulator 2^-8 the size, and the large one is rounded once per 16 k.
    floatx16 accs[X3 ? MI : 1][X3 ? NI : 1];
    if constexpr (X3) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
          for (int q = 0; q < 16; ++q) accs[mi][ni][q] = 0.f;
    }

    const int sw = (r >> 1) & 7;  // fragment rows are 32-aligned + r, so (row >> 1) & 7 == (r >> 1) & 7
    auto compute_stage = [&](int buf) __attribute__((always_inline)) {
      const char *a_base = As + (buf * BM + wm * (BM / WM) + r) * ROWB;
      const char *b_base = Bs + (buf * BN + wn * (BN / WN) + r) * ROWB;
      if constexpr (X3) {
        // two 16-k MFMA steps per 32-k stage; lane half h owns the k-run of 8 number g = 2 t + h: its float32 activations
        // are chunks 2 g, 2 g + 1 of the row, its weight pieces chunk g of the row's 64 bytes in each piece plane.
        // (The split's ~3.5 VALU instructions per MFMA do NOT run in the matrix pipe's shadow on this chip: a loop of 24 MFMAs
        // and 84 VALU takes the same time split-first or as "one MFMA, four VALU" -- 259-274 against 350 float32-equivalent
        // TFLOP/s without the VALU, tools/x3_overlap_probe.hip -- and so does this kernel, software-pipelined or not.)
        const int wrow = wn * (BN / WN) + r;   // + 32 ni: rows of ONE 64-row group when the wave's N range is 64 wide or less
        const char *bx = Bs + buf * BN * BROWB + (wrow >> 6) * (12 * 1024) + (wrow & 63) * 64;
        const int swb = (wrow >> 2) & 3;       // (32 ni does not change bits 2-3 of the row)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int g = 2 * t + h;
          bf16x8 a1[MI], a2[MI], a3[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            const floatx4 lo = *reinterpret_cast<const floatx4 *>(a_base + mi * 32 * ROWB + 16 * ((2 * g) ^ sw));
            const floatx4 hi = *reinterpret_cast<const floatx4 *>(a_base + mi * 32 * ROWB + 16 * ((2 * g + 1) ^ sw));
            split_bf16x3(lo, hi, a1[mi], a2[mi], a3[mi]);
          }
          bf16x8 b1[NI], b2[NI], b3[NI];
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const char *bq = bx + (ni >> 1) * (12 * 1024) + (ni & 1) * (32 * 64) + 16 * (g ^ swb);
            b1[ni] = *reinterpret_cast<const bf16x8 *>(bq);
            b2[ni] = *reinterpret_cast<const bf16x8 *>(bq + 4096);
            b3[ni] = *reinterpret_cast<const bf16x8 *>(bq + 8192);
          }
          // one piece pair at a time over the wave's blocks (consecutive MFMAs write different accumulators); the small
          // terms into their own accumulators
#define DVSG_X3_TERM(C, A, B)                                                                       \
  _Pragma("unroll") for (int mi = 0; mi < MI; ++mi) _Pragma("unroll") for (int ni = 0; ni < NI; ++ni) \
      C[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[mi], B[ni], C[mi][ni], 0, 0, 0)
          DVSG_X3_TERM(accs, a3, b1);
          DVSG_X3_TERM(accs, a1, b3);
          DVSG_X3_TERM(accs, a2, b2);
          DVSG_X3_TERM(accs, a2, b1);
          DVSG_X3_TERM(accs, a1, b2);
          DVSG_X3_TERM(acc, a1, b1);
#undef DVSG_X3_TERM
        }
        return;
      }
      if constexpr (PSPLIT) {
        // two 16-k MFMA steps per 32-k stage; lane half h owns the k-run of 8 number g = 2 t + h: its hi
        // pieces are 16-byte chunk g of the row, its lo pieces chunk 4 + g -- activations and weights alike
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const int g = 2 * t + h;
          halfx8 ahi[MI], alo[MI];
#pragma unroll
          for (int mi = 0; mi < MI; ++mi) {
            ahi[mi] = *reinterpret_cast<const halfx8 *>(a_base + mi * 32 * ROWB + 16 * (g ^ sw));
            alo[mi] = *reinterpret_cast<const halfx8 *>(a_base + mi * 32 * ROWB + 16 * ((4 + g) ^ sw));
          }
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) {
            const halfx8 bhi = *reinterpret_cast<const halfx8 *>(b_base + ni * 32 * ROWB + 16 * (g ^ sw));
            const halfx8 blo = *reinterpret_cast<const halfx8 *>(b_base + ni * 32 * ROWB + 16 * ((4 + g) ^ sw));
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], bhi, acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(alo[mi], bhi, acc[mi][ni], 0, 0, 0);
              acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ahi[mi], blo, acc[mi][ni], 0, 0, 0);
            }
          }
        }
        return;
      }
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const int co = 16 * ((2 * kb + h) ^ sw);
        frag_t a4[MI], b4[NI];
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) a4[mi] = *reinterpret_cast<const frag_t *>(a_base + mi * 32 * ROWB + co);
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) b4[ni] = *reinterpret_cast<const frag_t *>(b_base + ni * 32 * ROWB + co);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = Frag<T>::mma(a4[mi], b4[ni], acc[mi][ni]);
      }
    };

    // Both stage buffers are free when a tile starts, so stages 0 AND 1 go in flight together and the
    // first wait is a counted one (LDS-DMA completes in issue order: "at most one stage's worth
    // outstanding" means stage 0 has landed).  Issuing stage 1 only after stage 0 had arrived exposed
    // most of a second L2 / HBM round trip per tile -- a third of the time of a K = 64 tile.
    // From then on __syncthreads() carries the vmcnt(0) that retires the DMA of the next stage and
    // orders everyone's reads of the buffer about to be refilled.
    const int KT = kt1 - kt0;
    constexpr int PER = AG + BG;  // LDS-DMA instructions per wave and stage
    // ---- epilogue geometry (C/D map of the 32x32 MFMA: col = lane & 31, row = (q&3) + 8 (q>>2) + 4 h)
    constexpr int LDC = BN + 4;
    constexpr int BNO = WSTACK ? BN / 2 : BN;  // output channels of the tile
    constexpr int C4 = BNO / 4;      // 4-channel groups per output row
    constexpr int RSTEP = NT / C4;   // tile rows covered per pass
    constexpr int NROW = BM / RSTEP; // rows per thread
    constexpr float kLoScale = 1.0f / 2048.0f;
    const int col4 = tid % C4, row0 = tid / C4;
    const int n = (WSTACK ? nt * BNO : n0) + 4 * col4;
    const bool reduce = MODE != 0 && n_contrib > 1;
    // The residual of a plain tile is fetched right behind the first two stages' DMA (EARLY_RES): its
    // HBM round trip and its share of the layer's traffic run under the main loop instead of in front of
    // the epilogue -- a K = 128 layer with a residual moves 4 bytes of residual and output per byte of
    // input.  (Other tiles fetch it before the epilogue's transpose, where its latency still runs under
    // the two barriers and the LDS round trip.)  Rows past M read row M-1: unconditional loads, the
    // store is what is guarded.
    constexpr bool EARLY_RES = RES != 0 && MODE == 0 && sizeof(T) == 4 && !PSPLIT && NROW <= 8;
    const float4 bias4 = *reinterpret_cast<const float4 *>(p.bias + n);
    float4 rv[NROW];
    auto load_residual = [&]() __attribute__((always_inline)) {
#pragma unroll
      for (int i = 0; i < NROW; ++i) {
        const int mr = m0 + row0 + i * RSTEP;
        const int m = mr < p.M ? mr : p.M - 1;
        size_t roff;
        if (RES == 1) {
          roff = (size_t)m * p.res_ld + n;
        } else {  // slim `subsample`: shortcut = x[:, ::s, ::s, :]
          const int wo = m % p.Wo;
          const int t = m / p.Wo;
          const int ho = t % p.Ho;
          const int b = t / p.Ho;
          roff = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.res_ld + n;
        }
        rv[i] = PSPLIT ? load4_p_pair(p.res, roff, col4 & 1) : load4(pres + roff);
      }
    };
    DVSG_STAMP(1);
    issue_stage(0);
    if (KT > 1) {
      issue_stage(1);
      if (EARLY_RES) {
        load_residual();  // NROW more loads behind the DMA of stage 1: stage 0 has landed once PER + NROW are outstanding
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + (EARLY_RES ? NROW : 0)) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
      }
    } else {
      if (EARLY_RES) load_residual();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DVSG_STAMP(2);
    __builtin_amdgcn_sched_barrier(0);
    compute_stage(0);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 1; kt < KT - 1; ++kt) {
#ifdef DVSG_STAMPS
      const unsigned long long b0 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long b1 = __builtin_amdgcn_s_memtime();
      __syncthreads();
      const unsigned long long b2 = __builtin_amdgcn_s_memtime();
      st[5] += b1 - b0;   // own DMA not landed yet
      st[7] += b2 - b1;   // waiting for the other waves
#else
      __syncthreads();
#endif
      issue_stage((kt + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      compute_stage(kt & 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (KT > 1) {
      __syncthreads();
      compute_stage((KT - 1) & 1);
    }

    if constexpr (X3) {
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni) acc[mi][ni] += accs[mi][ni];
    }
    DVSG_STAMP(3);
    // ---- epilogue
    float *Cs = reinterpret_cast<float *>(lds);
    if (RES != 0 && !reduce && !EARLY_RES) load_residual();
    auto lds_barrier = [&]() __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    lds_barrier();  // everyone is done reading the stage buffers
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int q = 0; q < 16; ++q)
          Cs[(wm * (BM / WM) + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * LDC + wn * (BN / WN) + ni * 32 + r] =
              acc[mi][ni][q];
    lds_barrier();
    if (reduce) {
      float *mine = slab_of(own);
      for (int row = row0; row < BM; row += RSTEP) {
        *reinterpret_cast<float4 *>(mine + row * BN + 4 * col4) = *reinterpret_cast<const float4 *>(Cs + row * LDC + 4 * col4);
        if (WSTACK)
          *reinterpret_cast<float4 *>(mine + row * BN + BNO + 4 * col4) =
              *reinterpret_cast<const float4 *>(Cs + row * LDC + BNO + 4 * col4);
      }
      int &s_ticket = *reinterpret_cast<int *>(lds + TICKET_OFF);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        s_ticket = __hip_atomic_fetch_add(p.counters + ticket_idx, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      __syncthreads();
      if (s_ticket != n_contrib - 1) return;  // someone else will finish this tile
      if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __syncthreads();
    }
    // The finisher sums the partial tiles contributor by contributor (K order), all of the thread's
    // rows per step: NROW independent loads in flight instead of one (the row-by-row form made a
    // batch-1 split-K launch wait for NROW x n_contrib L2 round trips in sequence).
    floatx4 vsum[NROW];  // (an ext-vector type: arrays of the float4 struct spill)
    floatx4 vlo[WSTACK ? NROW : 1];
    if (reduce) {
#pragma unroll
      for (int i = 0; i < NROW; ++i) vsum[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      if (WSTACK) {
#pragma unroll
        for (int i = 0; i < NROW; ++i) vlo[i] = floatx4{0.f, 0.f, 0.f, 0.f};
      }
      for (int j = 0; j < n_contrib; ++j) {
        const float *src = j == own ? Cs + 4 * col4 : slab_of(j) + 4 * col4;
        const int ld = j == own ? LDC : BN;
#pragma unroll
        for (int i = 0; i < NROW; ++i) {
          vsum[i] += *reinterpret_cast<const floatx4 *>(src + (row0 + i * RSTEP) * ld);
          if (WSTACK) vlo[i] += *reinterpret_cast<const floatx4 *>(src + (row0 + i * RSTEP) * ld + BNO);
        }
      }
      if (WSTACK) {
#pragma unroll
        for (int i = 0; i < NROW; ++i) vsum[i] += vlo[i] * kLoScale;
      }
    }
#pragma unroll
    for (int i = 0; i < NROW; ++i) {
      const int row = row0 + i * RSTEP;
      const int m = m0 + row;
      if (m >= p.M) break;
      float4 v;
      if (reduce) {
        v = make_float4(vsum[i][0], vsum[i][1], vsum[i][2], vsum[i][3]);
      } else {
        v = *reinterpret_cast<const float4 *>(Cs + row * LDC + 4 * col4);
        if (WSTACK) {
          const float4 lo = *reinterpret_cast<const float4 *>(Cs + row * LDC + BNO + 4 * col4);
          v.x += lo.x * kLoScale; v.y += lo.y * kLoScale; v.z += lo.z * kLoScale; v.w += lo.w * kLoScale;
        }
      }
      v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
      if (RES != 0) {
        float4 r4;
        if (reduce) {  // the finishing contributor of a split tile loads it here
          size_t roff;
          if (RES == 1) {
            roff = (size_t)m * p.res_ld + n;
          } else {
            const int wo = m % p.Wo;
            const int t = m / p.Wo;
            const int ho = t % p.Ho;
            const int b = t / p.Ho;
            roff = (((size_t)b * p.res_H + (size_t)ho * p.res_stride) * p.res_W + (size_t)wo * p.res_stride) * p.res_ld + n;
          }
          r4 = PSPLIT ? load4_p_pair(p.res, roff, col4 & 1) : load4(pres + roff);
        } else {
          r4 = rv[i];
        }
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      if (RELU || n >= p.relu_from) {   // (the thread's four channels sit on one side of relu_from: it is a multiple of 64)
        v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
      }
      if (PSPLIT) store4_p_pair(p.y, (size_t)m * p.Cout + n, v, col4 & 1, true);  // lanes 2j, 2j+1: one pixel, 8 channels
      else store4(py + (size_t)m * p.Cout + n, v);
    }
  };

  if (MODE == 0) {
    const int tile = p.tile_begin + xcd_remap(blockIdx.x, p.tile_count);
    run_segment(tile, 0, KT_all, 1, 0, 0, [&](int) -> float * { return nullptr; });
#ifdef DVSG_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    st[4] = __builtin_amdgcn_s_memtime();
    if (p.slabs && tid == 0) {
      unsigned long long *o = reinterpret_cast<unsigned long long *>(p.slabs) + (size_t)blockIdx.x * 8;
      unsigned xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      unsigned hwid;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
      o[0] = __builtin_amdgcn_s_memrealtime() - st[6]; o[1] = st[1] - st[0]; o[2] = st[2] - st[1]; o[3] = st[3] - st[2]; o[4] = st[4] - st[3];
      o[5] = st[5]; o[6] = st[6]; o[7] = st[7] | ((unsigned long long)(xcc & 0xf) << 56) | ((unsigned long long)(hwid & 0xffff) << 40);
    }
#endif
  } else if (MODE == 1) {
    // the slices of one tile are adjacent logical ids (same XCD: the reducer reads its siblings'
    // slabs out of its own L2)
    const int logical = xcd_remap(blockIdx.x, p.tile_count * p.ksplit);
    const int tile = logical / p.ksplit, slice = logical - tile * p.ksplit;
    const int kt0 = (int)((long)slice * KT_all / p.ksplit), kt1 = (int)((long)(slice + 1) * KT_all / p.ksplit);
    run_segment(p.tile_begin + tile, kt0, kt1, p.ksplit, slice, tile,
                [&](int j) -> float * { return p.slabs + ((size_t)tile * p.ksplit + j) * (BM * BN); });
  } else {
    // Blocks [0, tile_begin) run the plain tiles of the full rounds; the G blocks after them share
    // the rest: workgroup w of G owns units [w U / G, (w + 1) U / G) of the U = tile_count x KT
    // (tile, stage) units, i.e. the end of one tile and / or the beginning of the next.  One launch:
    // the shares are dealt as the last full round drains, with no idle gap between two kernels.
    if ((int)blockIdx.x < p.tile_begin) {
      run_segment(xcd_remap(blockIdx.x, p.tile_begin), 0, KT_all, 1, 0, 0, [&](int) -> float * { return nullptr; });
      return;
    }
    const long G = (long)gridDim.x - p.tile_begin;
    const long U = (long)p.tile_count * KT_all;
    const int w = xcd_remap((int)blockIdx.x - p.tile_begin, (int)G);
    auto first_unit = [&](long wg) -> long { return wg * U / G; };
    auto owner = [&](long u) -> int { return (int)(((u + 1) * G - 1) / U); };  // largest wg with first_unit(wg) <= u
    long u = first_unit(w);
    const long u_end = first_unit(w + 1);
    while (u < u_end) {
      const int t_local = (int)(u / KT_all);
      const int kt0 = (int)(u - (long)t_local * KT_all);
      const long kt_to = kt0 + (u_end - u);
      const int kt1 = kt_to < KT_all ? (int)kt_to : KT_all;
      const int w_first = owner((long)t_local * KT_all), w_last = owner((long)t_local * KT_all + KT_all - 1);
      // contributor wg keeps this tile's partial in its slot 0 if the tile is where its range begins, else slot 1
      auto slab_of = [&](int j) -> float * {
        const long wg = w_first + j;
        const int slot = first_unit(wg) / KT_all == t_local ? 0 : 1;
        return p.slabs + ((size_t)wg * 2 + slot) * (BM * BN);
      };
      run_segment(p.tile_begin + t_local, kt0, kt1, w_last - w_first + 1, w - w_first, t_local, slab_of);
      u += kt1 - kt0;
      __syncthreads();  // the next segment restages the LDS this one's epilogue was reading
    }
  }
}


template <typename T, int BN, int WM, int WN, int KS, int MODE, bool SPLIT = false, bool X3 = false>
int launch_cfg(const ConvGemmDev &d, int blocks, bool relu, int res, hipStream_t s) {
  const dim3 grid(blocks), block(64 * WM * WN);
#define DVSG_LAUNCH(R, Q) hipLaunchKernelGGL((conv_gemm_kernel<T, BN, WM, WN, KS, R, Q, MODE, SPLIT, X3>), grid, block, 0, s, d)
  if constexpr (MODE == 2) {
    // stream-K launches never carry a residual (launch_conv_gemm): with one, the finisher's partial sums
    // and residual rows together do not fit the 128-VGPR budget of 4 waves per SIMD (8-13 spills), and
    // no layer of resnet_v1_50 needs it (a residual layer's K is the unit's narrow width)
    if (res != 0) return fail(DVSG_ERR_UNSUPPORTED, "conv_gemm: stream-K launch with a residual");
    if (relu) DVSG_LAUNCH(true, 0);
    else DVSG_LAUNCH(false, 0);
  } else if (relu) {
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

template <typename T, int KS, bool SPLIT = false>
int launch_ks(ConvGemmDev d, bool wide, int streamk_tail, bool relu, int res, hipStream_t s) {
  const int tiles = d.mtiles * d.ntiles;
  d.tile_begin = 0;
  d.tile_count = tiles;
  if constexpr (SPLIT && sizeof(T) == 2) {  // always 128-wide: a tile is [hi | lo] of one 64-channel group
    if (d.ksplit > 1) return launch_cfg<T, 128, 2, 4, KS, 1, SPLIT>(d, tiles * d.ksplit, relu, res, s);
    if (streamk_tail > 0) {
      d.tile_begin = tiles - streamk_tail;
      d.tile_count = streamk_tail;
      return launch_cfg<T, 128, 2, 4, KS, 2, SPLIT>(d, d.tile_begin + kResident, relu, res, s);
    }
    if (tiles <= 512) return launch_cfg<T, 128, 2, 2, KS, 0, SPLIT>(d, tiles, relu, res, s);
    return launch_cfg<T, 128, 2, 4, KS, 0, SPLIT>(d, tiles, relu, res, s);
  } else {
  if (d.ksplit > 1)  // few tiles (small batch): 64-wide tiles, K split over several workgroups per tile
    return launch_cfg<T, 64, 2, 2, KS, 1, SPLIT>(d, tiles * d.ksplit, relu, res, s);
  if (wide && streamk_tail > 0) {
    // full rounds as plain tiles, then the rest (the partly filled last round, or a launch that is
    // less than one round) as equal shares of (tile, K-stage) units
    d.tile_begin = tiles - streamk_tail;
    d.tile_count = streamk_tail;
    return launch_cfg<T, 128, 2, 4, KS, 2, SPLIT>(d, d.tile_begin + kResident, relu, res, s);
  }
  // Fat 4-wave workgroups when a single (partial) round of tiles covers the launch, else 8 waves
  // (4 per SIMD at 2 workgroups per CU): short K loops are prologue / epilogue bound and want more
  // waves in flight.  (A 4-stage LDS-DMA ring, 96 KB of LDS and one workgroup per CU, was measured
  // for the small launches and lost 3-10 % to two 2-stage workgroups per CU.  Round 4 measured it again where that
  // argument does not apply -- the split-K launches of batch 1-2, at most one workgroup per CU, three stages in flight
  // behind counted waits and a barrier that leaves them in flight, same bits: 1.120 against 1.086 ms per 512x288 frame,
  // 2.214 against 2.110 ms at 720p, profiles/r04_latency_ab_ring4_rejected.log.  A slice's stages do not wait for memory.
  // Nor is the slice's MFMA chain what a split-K launch costs: 8 waves of 32 x 32 per workgroup instead of 4 of 64 x 32 --
  // half the matrix-core work per wave, twice the waves -- 1.074 against 1.089 ms at 512x288, 2.111 = 2.112 at 720p,
  // +0.5 % at batch 2 (profiles/r04_latency_ab_8waves.log).  What is left per launch is fixed: ramp, prologue, first
  // round trip, transpose, partial-tile exchange, drain -- 13-25 us where an empty kernel takes 5.)
  // f32s (SPLIT, float): a tile's matrix-core time is a fifth of the exact path's, so the 4 fat waves win at every
  // tile count -- half the fragment reads per MFMA -- (measured per layer, tools/conv_bench.py --precision f32s)
  const bool four = g_conv_variant == 1 || (SPLIT && sizeof(T) == 4 && g_conv_variant == 0) ||
                    ((g_conv_variant == 0 || g_conv_variant == 6) && tiles <= 512);
  if (four)
    return wide ? launch_cfg<T, 128, 2, 2, KS, 0, SPLIT>(d, tiles, relu, res, s)
                : launch_cfg<T, 64, 2, 2, KS, 0, SPLIT>(d, tiles, relu, res, s);
  if (!wide) return launch_cfg<T, 64, 4, 2, KS, 0, SPLIT>(d, tiles, relu, res, s);
  return launch_cfg<T, 128, 2, 4, KS, 0, SPLIT>(d, tiles, relu, res, s);
  }
}

}  // namespace
}  // namespace dvsg
