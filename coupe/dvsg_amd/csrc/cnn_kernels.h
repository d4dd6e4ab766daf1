// Launch interface of the localizationNet kernels (conv_gemm.hip, conv1_pool.hip, head.hip),
// used by locnet.hip.
//
// Three precisions share the kernels through template parameters:
//   kF32   float32 activations / weights, exact-f32 matrix cores (v_mfma_f32_32x32x2_f32): the path of record
//   kF32S  4 bytes per value and float32 accumulation, but every operand enters the float16 matrix cores as two
//          float16 pieces (22 significant bits for |x| >= 2^-3, absolute step 2^-24 below: the second piece is unscaled)
//          and the activation tensors between layers hold those pieces;
//          values must stay inside float16's range (|x| < 65504), which post-BatchNorm activations and BN-folded
//          weights do by orders of magnitude
//   kF16   float16 activations, conv weights as float16 hi / lo pairs, v_mfma_f32_32x32x16_f16 with float32
//          accumulation; bias, accumulators and the dense head stay float32.
#pragma once
#include "common.h"

namespace dvsg {

// kF32S: 4 bytes per value and float32 accumulation like kF32, but every product is formed on the float16
// matrix cores from two float16 pieces per operand (22 significant bits down to |x| = 2^-3; conv_gemm.hip), and the activation
// tensors between the layers hold those two pieces (P format, cnn_device.h) instead of one float32.
// kF32X ("f32x3"): float32 tensors everywhere, exactly as kF32; inside the 52 conv GEMMs every product is formed on the
// bfloat16 matrix cores from THREE bfloat16 pieces per operand -- all 24 significant bits of both float32 operands, at any
// magnitude -- as six v_mfma_f32_32x32x16_bf16 with float32 accumulation (conv_gemm_tile.h, X3).
enum Precision { kF32 = 0, kF16 = 1, kF32S = 2, kF32X = 3 };
inline size_t elem_size(int prec) { return prec == kF16 ? 2 : 4; }

// Implicit-GEMM convolution (1x1 or 3x3, NHWC, C_in % 64 == 0 (f16) / 32 (f32), C_out % 64 == 0):
//   y[m, n] = act( sum_k A[m, k] * wt[n, k] + bias[n] (+ res[...]) )
// with m = (b, ho, wo), k = (kh, kw, c).  wt is [C_out][K] (K contiguous, BN scale folded in).
struct ConvGemm {
  int prec;           // kF32 / kF16: element type of x, wt, res, y
  const void *x;      // [B,H,W,Cin]
  const void *wt;     // [Cout][ksize*ksize*Cin]
  const float *bias;  // [Cout]
  const void *res;    // optional residual [B,res_H,res_W,Cout], sampled at (ho*res_stride, wo*res_stride)
  void *y;            // [B,Ho,Wo,Cout]
  int B, H, W, Cin, Ho, Wo, Cout;
  int ksize, stride, pad;
  int res_H, res_W, res_stride;
  int relu;
  // float32 / f32s only (0 / -1 = the dense defaults): x and res may be column ranges of wider tensors -- ldx, res_ld elements
  // between two pixels -- and, with relu == 0, output channels >= relu_from get the ReLU all the same: a unit's `shortcut`
  // (no ReLU) and `conv1` (ReLU) read the same input and run as ONE launch over their concatenated weight rows (locnet.hip)
  int ldx = 0, res_ld = 0, relu_from = -1;
  const void *wt_packed = nullptr;   // kF16, optional: the rows of `wt` packed stage by stage for conv_gemm_wide16.hip (launch_pack_wide16, order 0)
  const void *wt_packed_a = nullptr; // ... in order 1, for its 128-byte-activation-row kernel
  const void *wt_packed_h = nullptr; // ... in order 2, for its 3x3 stride-1 kernel (a kernel row's taps from one staged run)
  int x3 = 0;         // kF32 only: wt is the layer's packed bfloat16 piece stages (launch_pack_x3) and the products are formed
                      // from three bfloat16 pieces per operand (conv_gemm_tile.h, X3); x, res, y stay float32
  int wsplit = 0;     // kF16: wt is the stacked layout [Cout/64][128][K] (64 hi rows, then 64 lo rows); kF32: wt is
                      // [Cout][K/32][32 hi halves | 32 lo halves] and products are formed from float16 pieces (conv_gemm.hip)
  // optional split-K scratch (small batches): partial-tile slabs and kSplitKMaxTiles zeroed int tickets
  void *splitk_scratch = nullptr;
  size_t splitk_scratch_bytes = 0;
  int *splitk_counters = nullptr;
};
constexpr int kSplitKMaxTiles = 512;                                   // tickets one launch may use
constexpr size_t kSplitKSlabBytes = (size_t)512 * 2 * 128 * 128 * 4;   // 64 MiB: 512 workgroups x 2 partial 128x128 f32 tiles (stream-K tail)
int launch_conv_gemm(const ConvGemm &p, hipStream_t s);
// f32x3 weights: float32 wt [rows][K] (rows % 64 == 0, K % 32 == 0) -> per group of 64 rows and 32-k stage three 4 KB planes
// of bfloat16 pieces [64 rows][32 k] (p1 = bf16(w), p2 = bf16(w - p1), p3 = bf16(w - p1 - p2)), row R's 16-byte chunk c at
// position c ^ ((R >> 2) & 3): 6 bytes per weight, [rows / 64][K / 32][3][64][32]
inline size_t x3_packed_bytes(int rows, int K) { return (size_t)rows * K * 6; }
int launch_pack_x3(const float *wt, void *out, int rows, int K, hipStream_t s);
// float16 mode, big launches: 256 x 128 tiles with 64-byte K stages (conv_gemm_wide16.hip); called by launch_conv_gemm
int launch_conv_wide16(const ConvGemm &p, hipStream_t s);
void set_wide16_min_tiles(int v);
// weights packed for that kernel: every 32-k stage of a 128-row tile contiguous, in the LDS image's chunk order
size_t wide16_packed_bytes(int rows, int Cin, int ksize);
int launch_pack_wide16(const void *wt, void *out, int rows, int Cin, int ksize, int order, hipStream_t s);
void set_wide16_arows(int v);       // diagnostic (dvsg_debug_set_option "wide16_arows")
void set_wide16_hreuse(int v);      // diagnostic (dvsg_debug_set_option "wide16_hreuse")
void set_wide16_packed(int v);      // diagnostic (dvsg_debug_set_option "wide16_packed")
// Zeroes n split-K / stream-K tickets with a KERNEL: a hipMemsetAsync captured into a HIP graph (memset node) did not
// take effect on the second and later replays of the graph on ROCm 7.2 (tests/test_gpu_cnn.py::test_a_step_replays_...).
int launch_zero_tickets(int *tickets, size_t n, hipStream_t s);
// conv2 (3x3, Cin -> 64, stride 1/2, pad 1, + bias + ReLU) followed by conv3 (1x1, 64 -> Cout, + bias +
// residual + ReLU) of a bottleneck unit whose middle width is 64 (block 1), fused: the [M,64]
// intermediate stays in LDS.  float32, "f32s" pieces, or the float16 mode (stacked hi / lo weights).
struct ConvFused {
  const float *x;      // [B,H,W,Cin]
  const float *wt2;    // [64][9*Cin]
  const float *bias2;  // [64]
  const float *wt3;    // [Cout][64]
  const float *bias3;  // [Cout]
  const float *res;    // [B,res_H,res_W,Cout], sampled at (ho*res_stride, wo*res_stride); NULL with sc_x
  float *y;            // [B,Ho,Wo,Cout]
  int B, H, W, Cin, Ho, Wo, Cout;
  int stride;
  int res_H, res_W, res_stride;
  // the unit that opens a block: residual = 1x1 `shortcut` conv (+ BN) of the unit's input, computed in the
  // kernel from sc_x [B,Ho,Wo,sc_cin] (stride 1), sc_wt [Cout][sc_cin], sc_bias [Cout]; the tensor never exists
  const float *sc_x = nullptr;
  const float *sc_wt = nullptr;
  const float *sc_bias = nullptr;
  int sc_cin = 0;
  // "f32s": x, res / sc_x, y hold float16 pieces (P format) and wt2, wt3, sc_wt are the layers' piece matrices
  // ([rows][K/32][32 hi | 32 lo]); same bytes per value, same addressing (pass them through the float pointers)
  int pieces = 0;
  // float16 mode: x, res / sc_x, y are float16 tensors and wt2, wt3, sc_wt the layers' stacked [hi | lo] float16 matrices
  // ([cout/64][128][K], conv_gemm.hip SPLIT), passed through the float pointers
  int f16 = 0;
  // "f32x3": x, res, y are float32 tensors as in the float32 form; wt2, wt3 are the layers' packed bfloat16 piece stages
  // (launch_pack_x3), passed through the float pointers; the residual is a tensor (no fused shortcut): conv_fused_x3.hip
  int x3 = 0;
};
int launch_conv3x3_1x1_x3(const ConvFused &p, hipStream_t s);
bool conv_fusable(int prec, int Cin, int Cmid, int Cout, int ksize);
int launch_conv3x3_1x1(const ConvFused &p, hipStream_t s);
void set_fuse_conv(int v);
int get_fuse_conv();
void set_fused_hreuse(int v);        // diagnostic (dvsg_debug_set_option "fused_hreuse")
void set_conv_variant(int v);   // diagnostic A/B switches (dvsg_debug_set_option)
void set_conv1_variant(int v);

// conv1: 7x7 stride 2, explicit pad 3, C_in = 21 -> 64, with scale_RGB fused into the LDS
// load stage (networks.py:6-16 + slim conv2d_same root).  wt1 is [7][64][kConv1Ld] float32:
// per kernel row kh, per output channel, a zero tap and then the 7*21 (kw, c) taps in memory order
// of the input row (tap k at index k + 1: the kernels stage the input row from one element before the
// window, which is a 16-byte boundary of the row), BN scale folded, channel-group reversal folded.
constexpr int kConv1Cin = 21;
constexpr int kConv1K = 7 * kConv1Cin;   // 147 taps per kernel row
constexpr int kConv1Kpad = 148;          // zero tap + 147, a whole number of 4-k MFMA steps
constexpr int kConv1Ld = 150;            // LDS / global row stride (2*odd: conflict-free ds_read_b64)
// wt1h (float16 precision only): the same row as [7][64][kConv1LdH] float16, zero-padded to 160.
constexpr int kConv1LdH = 168;
// wt1s ("f32s" precision only): float16 pieces [7][2][64][kConv1LdH], same tap positions: hi = f16(w), lo = f16((w - hi) * 2^11).
// conv1's input: the window tensor [B,H,W,21] float32 the reference feeds (eval.py:106-110), or -- SURVEY.md 8f-1/-2 --
// a pool of RGB frames [n_pool,H,W,3] (float32 in [0,1], or raw uint8 whose / 255. is fused) plus the table [B,7] of
// the pool frame in each window slot, oldest to newest: the np.concatenate of eval.py:103-104 happens in the load stage.
// wt1x ("f32x3" precision only): bfloat16 pieces [14][3][64][kConv1X3Ld]: stage 2 kh + half holds taps [80 half, 80 half + 80)
// of kernel row kh (position p = tap + 1 behind the zero tap, as in wt1h; positions 148..159 zero) as three piece planes,
// p1 = bf16(w), p2 = bf16(w - p1), p3 = bf16(w - p1 - p2); rows zero-padded from 80 to 88.
constexpr int kConv1X3Ld = 88;
constexpr int kConv1X3StageElems = 3 * 64 * kConv1X3Ld;   // 16 896 bfloat16 per half kernel row
enum Conv1SrcKind { kSrcWindow = 0, kSrcRingF32 = 1, kSrcRingU8 = 2 };
struct Conv1Src {
  const void *base;   // window tensor, or frame pool
  const int *table;   // ring: [B,7] pool indices (device); an index outside [0, n_pool) stages zeros
  int n_pool;
  const float *mask = nullptr;   // optional [B,H,W]: eval_train.py:43-45,53-64 -- the 18 history channels of window b are multiplied
                                 // by mask[b] (one plane: the warp of an all-ones image is the same in every channel) in the load stage
};
int launch_conv1(int out_prec, const Conv1Src &src, int src_kind, const float *wt1, const void *wt1h, const void *wt1s,
                 const void *wt1x, const float *bias, void *y, int B, int H, int W, int Ho, int Wo, hipStream_t s);

// 3x3 stride-2 TF-SAME max pool (slim resnet root), C % 8 == 0.
int launch_maxpool(int prec, const void *x, void *y, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                   int pad_left, hipStream_t s);

// Global average pool as float32 partial sums: part[b][s][c] = sum over the s-th slice of HW.
constexpr int kPoolSplits = 8;
int launch_avgpool_partial(int prec, const void *x, float *part, int B, int HW, int C, hipStream_t s);

// Element-wise float16 -> float32 (parity taps of a float16 run).
int launch_f16_to_f32(const void *x, float *y, size_t n, hipStream_t s);
// P format ("f32s" activations: float16 pieces, cnn_device.h) <-> float32; n % 32 == 0.
int launch_p_to_f32(const void *x, float *y, size_t n, hipStream_t s);
int launch_f32_to_p(const float *x, void *y, size_t n, hipStream_t s);

// Dense layer on split partial sums (see head.hip); float32 throughout.
constexpr int kDenseSplits = 8;
int launch_dense(const float *xin, int s_in, const float *bias_in, float scale_in, int lrelu_in,
                 const float *W, float *out_part, int B, int K, int N, hipStream_t s);
// out[b][n] = scale * sum_s part[b][s][n] + bias[n] (bias may be NULL)
int launch_dense_finalize(const float *part, int s_in, float scale, const float *bias, float *out, int B,
                          int N, hipStream_t s);

}  // namespace dvsg
