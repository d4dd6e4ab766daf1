/*
 * dvsg_amd.h -- C ABI of libdvsg_amd.so: the MI355X (gfx950) implementation of the
 * coupe.DVSG per-frame inference hot path.
 *
 * The reference has no FFI / plugin interface: its boundary is a set of Python callables
 * taking TF tensors.  Each entry point below replaces the TF
 * sub-graph built by the reference function it cites; the Python facade
 * (the coupe.dvsg_amd Python package) keeps the reference's names / argument order / return tuples and
 * forwards to these.  INTEGRATION.md shows the ctypes stub a reference maintainer would add.
 *
 * Conventions
 *   - every function returns 0 (DVSG_OK) or a negative dvsg_status; it never throws or
 *     aborts across the ABI.  dvsg_last_error_string() describes the calling thread's last
 *     failure.
 *   - every tensor argument is a CALLER-OWNED DEVICE pointer to dense float32 data in the
 *     reference's layout (NHWC images, values as the reference feeds them) unless a
 *     parameter is documented as "host".
 *   - `stream` is a hipStream_t passed as void* (NULL = the legacy default stream).  All
 *     calls are asynchronous and stream-ordered; none allocates, frees or synchronises.
 *     Calls on distinct streams are re-entrant as long as their output / workspace buffers
 *     are distinct.
 *   - optional outputs may be NULL.
 */
#ifndef DVSG_AMD_H
#define DVSG_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DVSG_ABI_VERSION 1

typedef enum dvsg_status {
  DVSG_OK = 0,
  DVSG_ERR_INVALID_ARG = -1, /* NULL pointer, non-positive size, unsupported shape        */
  DVSG_ERR_HIP = -2,         /* a HIP runtime call failed; see dvsg_last_error_string()    */
  DVSG_ERR_WORKSPACE = -3,   /* workspace too small or misaligned                          */
  DVSG_ERR_WEIGHTS = -4,     /* checkpoint array missing / wrong shape                     */
  DVSG_ERR_UNSUPPORTED = -5  /* valid request this build does not implement                */
} dvsg_status;

int dvsg_abi_version(void);
const char *dvsg_last_error_string(void);
/* "gfx950" -- the only code object in the library. */
const char *dvsg_target_arch(void);

/* ---------------------------------------------------------------------------------------
 * Thin-plate-spline transformer: ThinPlateSpline.py:4-170, ThinPlateSpline2.py:4-170.
 * ------------------------------------------------------------------------------------- */

/* `_solve_system` (ThinPlateSpline.py:143-166).  coord [B,P,2]; rhs [B,P,2] is `vector`
 * when rhs_is_vector != 0 (T solves for coord+vector, ThinPlateSpline.py:161) or `target`
 * (ThinPlateSpline2.py:160).  T [B,2,P+3].  3 <= P <= 61.  The 28x28 system is assembled in
 * float32 exactly as the reference does and solved in float64 (partial pivoting), which
 * lands inside the reference's own float32 LU noise. */
int dvsg_tps_solve_f32(const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                       float *T, void *stream);

/* As above, for callers that want tf.matrix_inverse's error behaviour (ThinPlateSpline.py:159
 * raises InvalidArgument "Input is not invertible" on repeated / collinear control points): the
 * number of samples whose system has a pivot below 1e-13 x max|W| is ADDED to the device int
 * *n_singular (zero it first; read it after the stream has run).  Either way such a sample's T is
 * NaN, never a finite garbage map. */
int dvsg_tps_solve_checked_f32(const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                               float *T, int *n_singular, void *stream);

/* `_transform` + `_meshgrid` + `_interpolate` fused (ThinPlateSpline.py:30-141): for every
 * output pixel evaluate [1, x_t, y_t, r_0..r_{P-1}], (x_s, y_s) = T . basis, then sampler A
 * (coords scaled by W/2, indices clipped BEFORE the weights).  U [B,H,W,C]; coord [B,P,2];
 * T [B,2,P+3]; out [B,out_h,out_w,C]; x_s, y_s [B*out_h*out_w] (optional).  The [B,P+3,H*W]
 * basis the reference materialises never exists.  U == NULL && out == NULL: grid only. */
int dvsg_tps_warp_f32(const float *U, const float *coord, const float *T, int B, int H, int W,
                      int C, int P, int out_h, int out_w, float *out, float *x_s, float *y_s,
                      void *stream);

/* ---------------------------------------------------------------------------------------
 * Optical-flow warp: warp_with_optical_flow.py:96-176 `tf_warp` (sampler C).
 * im [B,H,W,C], flow [B,H,W,2] in pixels (dx, dy), out [B,H,W,C].
 * ------------------------------------------------------------------------------------- */
int dvsg_flow_warp_f32(const float *im, const float *flow, int B, int H, int W, int C,
                       float *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * Spatial transformers: spatial_transformer.py.
 * ------------------------------------------------------------------------------------- */

/* `bilinear_interp` (spatial_transformer.py:496-563, sampler B) from explicit normalised
 * coordinates x_s, y_s [B*out_h*out_w]. */
int dvsg_stn_sample_f32(const float *im, const float *x_s, const float *y_s, int B, int H,
                        int W, int C, int out_h, int out_w, float *out, void *stream);

/* AffineTransformer._transform (spatial_transformer.py:74-91), theta [B,6]; when im and out
 * are non-NULL the sampler B gather is fused.  x_s / y_s optional. */
int dvsg_grid_affine_f32(const float *theta, const float *im, int B, int H, int W, int C,
                         int out_h, int out_w, float *out, float *x_s, float *y_s, void *stream);

/* ProjectiveTransformer._transform (spatial_transformer.py:423-452), theta [B,8] (the 9th
 * entry is 1), tf.div_no_nan by z. */
int dvsg_grid_projective_f32(const float *theta, const float *im, int B, int H, int W, int C,
                             int out_h, int out_w, float *out, float *x_s, float *y_s,
                             void *stream);

/* ElasticTransformer constants (spatial_transformer.py:313-362): HOST outputs
 * source_points [2,n] and L_inv [n,n+3] (= transpose(inverse(L)[:,3:])) for an
 * grid_size x grid_size control grid, n = grid_size^2 <= 61. */
int dvsg_elastic_constants_f32(int grid_size, float *source_points_host, float *L_inv_host);

/* ElasticTransformer._transform (spatial_transformer.py:276-296): theta_abs [B,2,n] are the
 * absolute control point positions (source_points + theta), L_inv [n,n+3] and
 * source_points [2,n] are DEVICE copies of the constants above.  The
 * [n+3, H*W] `right_mat` is evaluated on the fly, basis order [x; y; 1; U_0..U_{n-1}]. */
int dvsg_grid_elastic_f32(const float *theta_abs, const float *L_inv, const float *source_points,
                          int n, const float *im, int B, int H, int W, int C, int out_h,
                          int out_w, float *out, float *x_s, float *y_s, void *stream);

/* `scale_RGB` (networks.py:6-16): y = 255 x - mean with the three channel GROUPS reversed.
 * C must be a multiple of 3. */
int dvsg_scale_rgb_f32(const float *rgb, int B, int H, int W, int C, float *out, void *stream);

/* ---------------------------------------------------------------------------------------
 * localizationNet: networks.py:30-46 (slim resnet_v1_50 + 4 dense layers).
 * ------------------------------------------------------------------------------------- */
typedef struct dvsg_locnet dvsg_locnet_t;

/* Build the immutable device-side network from HOST arrays named like the reference's
 * checkpoint (ckpt_manager.py:42; `stabNet/localizationNet/...`, with or without `:0`).
 * dims is [n_arrays][4] (unused trailing entries ignored), ndims[i] in 1..4.  Unknown names
 * are ignored; a missing or mis-shaped array is DVSG_ERR_WEIGHTS (the reference silently
 * runs on random weights instead, ckpt_manager.py:21-22).  BatchNorm (eps 1e-5, moving
 * statistics) is folded into per-channel scale/shift here.  Allocates device memory. */
int dvsg_locnet_create(int n_arrays, const char *const *names, const float *const *host_data,
                       const int *ndims, const int64_t *dims, dvsg_locnet_t **net);
int dvsg_locnet_destroy(dvsg_locnet_t *net);
/* input channels of the loaded conv1 (21 for the 7-frame window) */
int dvsg_locnet_in_channels(const dvsg_locnet_t *net);
/* bytes of scratch `forward` needs for a [B,H,W,c_in] batch (256-byte aligned). */
int dvsg_locnet_workspace_bytes(const dvsg_locnet_t *net, int B, int H, int W, size_t *bytes);

/* patches [B,H,W,c_in] in [0,1] -> F_t [B,25,2].  float32 storage, exact-f32 MFMA
 * (v_mfma_f32_32x32x2_f32) accumulation. */
int dvsg_locnet_forward_f32(const dvsg_locnet_t *net, const float *patches, int B, int H, int W,
                            float *F_t, void *workspace, size_t workspace_bytes, void *stream);

/* Debug / parity tap: run the network up to and including `stage` and copy that stage's
 * activation (NHWC float32) to act_out.  Stages: 0 conv1, 1 pool1, 2..17 the 16 bottleneck
 * units in order, 18 pool5 [B,2048].  act_dims receives (h, w, c) on the HOST. */
int dvsg_locnet_forward_tap_f32(const dvsg_locnet_t *net, const float *patches, int B, int H,
                                int W, int stage, float *act_out, size_t act_out_bytes,
                                int *act_dims_host, void *workspace, size_t workspace_bytes,
                                void *stream);

/* The evaluation graph of model.py:98-123 in one call: F_t = localizationNet(patches_t);
 * T = solve(V_src, F_t); s_t_pred = TPS warp of u_t.  V_src is the constant 5x5 grid of
 * model.py:105-110.  s_t_pred [B,H,W,3]; F_t [B,25,2], x_s, y_s [B*H*W] optional. */
int dvsg_stabilize_f32(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B,
                       int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                       void *workspace, size_t workspace_bytes, void *stream);

/* Building block of localizationNet, exposed for layer-level parity tests and micro-benchmarks:
 * slim conv2d (1x1, or 3x3 with pad 1 = conv2d_same) + folded BatchNorm + optional residual +
 * optional ReLU as one implicit-GEMM launch.  x [B,H,W,Cin]; wt [Cout][ksize*ksize*Cin] (k order
 * kh, kw, c; BatchNorm scale already folded in); bias [Cout]; res (optional) is sampled at
 * (ho*res_stride, wo*res_stride) of a [B, (Ho-1)*res_stride+1, (Wo-1)*res_stride+1, Cout] tensor;
 * y [B,Ho,Wo,Cout] with Ho = (H-1)/stride+1.  Cin % 32 == 0, Cout % 64 == 0.
 * scratch (optional, 256-byte aligned; 17 MiB serves split-K, 65 MiB also the stream-K tail): lets
 * launches with few output tiles (batch 1-2) split K over several workgroups per tile, and lets
 * large launches cut the partly filled last round of tiles into equal (tile, K-stage) shares; the
 * partial tiles are summed in K order by the last workgroup to arrive, so results stay bitwise
 * reproducible from run to run. */
int dvsg_conv_gemm_f32(const float *x, const float *wt, const float *bias, const float *res, float *y,
                       int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                       int res_stride, void *scratch, size_t scratch_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * float16 variants (BASELINE.json configs[4]: "fp16 MFMA convs").  Same arguments and the same
 * float32 inputs / outputs as the _f32 entry points; inside, activations and conv weights are
 * stored as float16 and the 1x1 / 3x3 convolutions run on v_mfma_f32_32x32x16_f16 with float32
 * accumulation, their weights kept as float16 hi / lo pairs (see dvsg_conv_gemm_f16s: a plain float16
 * weight is off by up to 2^-12 relative at EVERY pixel alike, which the global average pool does not
 * average away -- it was 9/10 of this mode's error in F_t); conv1 multiplies float16 copies of the
 * scaled frames and writes f16; BatchNorm shift, the dense head, the TPS solve and the warp stay float32.  Not bit-compatible with the
 * float32 reference path: tests/test_gpu_f16.py states the measured F_t / pixel error.
 * The workspace of dvsg_locnet_workspace_bytes is sufficient (half of it is used).
 * ------------------------------------------------------------------------------------- */
int dvsg_locnet_forward_f16(const dvsg_locnet_t *net, const float *patches, int B, int H, int W,
                            float *F_t, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_locnet_forward_tap_f16(const dvsg_locnet_t *net, const float *patches, int B, int H,
                                int W, int stage, float *act_out, size_t act_out_bytes,
                                int *act_dims_host, void *workspace, size_t workspace_bytes,
                                void *stream);
int dvsg_stabilize_f16(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B,
                       int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                       void *workspace, size_t workspace_bytes, void *stream);
/* Calibration of the float16 mode (optional; round 4).  The hi / lo weight pairs cost twice the matrix-core work to remove
 * ONE thing: the bias a rounded float16 weight leaves in its output channel through the MEAN of its input channel,
 * sum_k (q_k - w_k) mu_k -- the same at every pixel, so the global average pool keeps it.  Given calibration windows
 * `patches` [B,H,W,c_in] (a few frames of the clip about to be stabilised), this call measures mu per convolution input
 * in one float32 pass and re-rounds the PLAIN float16 copy of every weight of blocks 2-4 to one of its two float16
 * neighbours so that the running sum of (q_k - w_k) mu_k along K stays within half a step ("error feedback"); from then
 * on the float16 entry points of this handle run blocks 2-4 on those plain weights (half the MFMAs) and keep the pairs in
 * block 1.  Measured on MI355X (synthetic checkpoint, 720p, against a float64 CPU evaluation): F_t 2.0-2.6e-6 where the
 * pairs give 1.4-1.9e-6 and round-to-nearest plain weights 1.9e-5; 22 % less time per step (tests/test_gpu_f16.py,
 * tools/f16_ef_masks.py).  The means are those of the calibration windows: frames of other statistics bring part of the
 * bias back (bounded by the plain mode's).  Deterministic (fixed-order sums); synchronous; mutates the handle, so not
 * while another thread runs it.  patches == NULL restores the uncalibrated state (pairs everywhere). */
int dvsg_locnet_calibrate_f16(dvsg_locnet_t *net, const float *patches, int B, int H, int W, void *workspace,
                              size_t workspace_bytes, void *stream);
/* ---------------------------------------------------------------------------------------
 * "f32s": float32 storage, float32 accumulation, float32-equivalent PRODUCTS from the float16 matrix
 * cores.  Same tensors, same workspace, same launch sequence as the *_f32 entry points (the dense head, the
 * TPS solve and the warp are the float32 kernels themselves; the max and average pools join the two pieces of
 * a value, which is exact); in conv1 (conv1_split_kernel) and the 52 1x1 / 3x3
 * convolutions every operand enters the matrix cores as two float16 pieces, x ~ f16(x) + f16(x - f16(x))
 * (22 significant bits while |x| >= 2^-3;
 * the second piece is NOT scaled, so below that it falls into float16's subnormal range -- absolute step 2^-24 --
 * and the pair carries about log2(|x|) + 25 bits: 20 at |x| = 0.03, 15 at 1e-3, plain float16's 11 at 1e-4), and a product is three
 * v_mfma_f32_32x32x16_f16 -- a1 w1 + a2 w1 + a1 w2, each
 * exact in float32 -- instead of eight v_mfma_f32_32x32x2_f32.  The activation tensors between the layers
 * (library workspace) hold the two pieces of each value instead of one float32, so they are split once,
 * by the layer that produces them.  Measured against the
 * exact path: stage activations within 2e-6 relative, F_t within 1e-7, i.e. what two float32 GEMMs with
 * different summation orders differ by; every float32 parity test of tests/ also passes in this mode
 * (tests/test_gpu_f32s.py) -- with operands of magnitude 0.03 .. 1 (the synthetic checkpoint, band-limited frames);
 * a checkpoint whose BatchNorm-folded weights or activations are much smaller loses bits as stated above
 * (tests/test_gpu_f32s.py::test_small_operands_lose_bits_as_documented) and is unpinned in this mode.
 * It is NOT the exact float32 arithmetic of the reference and is therefore a
 * separately named precision; dvsg_*_f32 stays the path of record.  Values must stay inside float16's range
 * (|x| < 65504; beyond it a piece is infinite), which BatchNorm-folded weights and post-BatchNorm activations
 * do by orders of magnitude.
 * ------------------------------------------------------------------------------------- */
int dvsg_locnet_forward_f32s(const dvsg_locnet_t *net, const float *patches, int B, int H, int W,
                             float *F_t, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_locnet_forward_tap_f32s(const dvsg_locnet_t *net, const float *patches, int B, int H,
                                 int W, int stage, float *act_out, size_t act_out_bytes,
                                 int *act_dims_host, void *workspace, size_t workspace_bytes,
                                 void *stream);
int dvsg_stabilize_f32s(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B,
                        int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                        void *workspace, size_t workspace_bytes, void *stream);
/* One layer in that mode.  x, res, y are activation tensors in the mode's INTERNAL format: 4 bytes per
 * value, but as the value's two float16 pieces -- flat element e of the dense NHWC tensor (channels % 32
 * == 0) lives in 128-byte group e / 32: 32 hi halves, then 32 lo halves (dvsg_f32_to_pieces /
 * dvsg_pieces_to_f32 convert).  wt_pieces is [Cout][K/32][32 hi halves | 32 lo halves] float16 (hi = f16(w),
 * lo = f16(w - hi)).  A 32-k row stage of either operand is 128 bytes, like a float32 one. */
int dvsg_conv_gemm_f32s(const void *x, const void *wt_pieces, const float *bias, const void *res,
                        void *y, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                        int relu, int res_stride, void *scratch, size_t scratch_bytes, void *stream);
int dvsg_f32_to_pieces(const float *x, void *y, size_t n, void *stream);   /* n % 32 == 0 */
int dvsg_pieces_to_f32(const void *x, float *y, size_t n, void *stream);

/* ---------------------------------------------------------------------------------------
 * "f32x3": float32 tensors everywhere -- boundary, workspace, every activation between the layers, exactly as the *_f32
 * entry points -- and float32 accumulation; inside conv1 and the 52 1x1 / 3x3 convolutions every PRODUCT is formed on the bfloat16
 * matrix cores from THREE bfloat16 pieces per operand, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1),
 * x3 = bf16(x - x1 - x2) (round to nearest).  bfloat16 has float32's exponent range and 8 significant bits, so the three
 * pieces hold all 24 significant bits of ANY finite float32 operand above 2^-110 -- no magnitude condition, unlike the two
 * float16 pieces of "f32s" -- and each piece product is exact in float32.  Of the nine cross terms of a w the six largest
 * (a1 w1, a1 w2, a2 w1, a2 w2, a1 w3, a3 w1) are accumulated, six v_mfma_f32_32x32x16_bf16 where the exact path issues
 * eight v_mfma_f32_32x32x2_f32 at 2.7 x the matrix-core time; the dropped a2 w3 + a3 w2 + a3 w3 are bounded by 2^-23 |a w|
 * -- one rounding of a float32 multiply; 2^-27 typically.  The large term a1 w1 and the five small ones accumulate in
 * SEPARATE float32 accumulators, joined once per tile: an MFMA adds its 16 products to the accumulator with the bits below the
 * accumulator's last place cut off, a bias that a shared accumulator lets through the network's average pool.
 * Activations are split in registers behind the LDS fragment read (conv1: once while its input row is staged; block 1's
 * fused conv2 + conv3: conv3's operand once per tile); weights once at load (dvsg_locnet_create; dvsg_pack_weights_f32x3 for
 * a single layer).  The pools, the dense head, the TPS solve and the warp are the float32 kernels themselves.  Measured
 * against float64 evaluations it is as close as the exact path is, layer by layer and end to end (tests/test_gpu_f32x3.py,
 * tests/test_f32x3_numerics_cpu.py); bitwise reproducible.  A separately named precision: dvsg_*_f32 stays the path whose
 * matrix instructions are float32 ones.
 * ------------------------------------------------------------------------------------- */
int dvsg_locnet_forward_f32x3(const dvsg_locnet_t *net, const float *patches, int B, int H, int W,
                              float *F_t, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_locnet_forward_tap_f32x3(const dvsg_locnet_t *net, const float *patches, int B, int H,
                                  int W, int stage, float *act_out, size_t act_out_bytes,
                                  int *act_dims_host, void *workspace, size_t workspace_bytes,
                                  void *stream);
int dvsg_stabilize_f32x3(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B,
                         int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                         void *workspace, size_t workspace_bytes, void *stream);
/* One layer in that mode: x, res, y float32 as for dvsg_conv_gemm_f32 (Cin % 32 == 0, Cout % 64 == 0); wt_packed is what
 * dvsg_pack_weights_f32x3 makes of the float32 matrix wt [Cout][K] (K = ksize^2 Cin, the layout dvsg_conv_gemm_f32 takes):
 * 6 Cout K bytes, per group of 64 output channels and 32-k stage three 4 KB planes of bfloat16 pieces. */
int dvsg_conv_gemm_f32x3(const float *x, const void *wt_packed, const float *bias, const float *res,
                         float *y, int B, int H, int W, int Cin, int Cout, int ksize, int stride,
                         int relu, int res_stride, void *scratch, size_t scratch_bytes, void *stream);
int dvsg_pack_weights_f32x3(const float *wt, void *wt_packed, int Cout, int K, void *stream);
/* dvsg_conv3x3_1x1_f32 (below) in that precision: conv2 (3x3, Cin -> 64, + bias + ReLU) and conv3 (1x1, 64 -> Cout, + bias +
 * residual + ReLU) of a block-1 unit in one kernel, float32 tensors; wt2_packed / wt3_packed are dvsg_pack_weights_f32x3 of the
 * float32 matrices [64][9 Cin] and [Cout][64].  Cin % 32 == 0 (>= 64), Cout % 64 == 0. */
int dvsg_conv3x3_1x1_f32x3(const float *x, const void *wt2_packed, const float *bias2, const void *wt3_packed,
                           const float *bias3, const float *res, float *y, int B, int H, int W, int Cin, int Cout,
                           int stride, int res_stride, void *stream);

/* x, wt, res, y are float16 (Cin % 64 == 0); bias float32. */
int dvsg_conv_gemm_f16(const void *x, const void *wt, const float *bias, const void *res, void *y,
                       int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                       int res_stride, void *scratch, size_t scratch_bytes, void *stream);
/* The layer as the float16 network runs it: float16 activations against weights kept as float16
 * hi / lo PAIRS, w ~ hi + 2^-11 lo (hi = f16(w), lo = f16((w - hi) * 2048)), i.e. effectively float32
 * weights at twice the matrix-core work and unchanged activation traffic.  wt_split is
 * [Cout/64][128][K]: for each group of 64 output channels 64 rows of hi, then 64 rows of lo.
 * scratch: as for dvsg_conv_gemm_f32 (tickets + 64 MiB of partial-tile slabs); with 3 x (2 Cout K) x 2 more bytes
 * behind them the call also makes the stage-packed weight copies the network's big launches run from
 * (dvsg_locnet_create keeps them per layer); otherwise it reads the weights from wt_split as they are: same results. */
int dvsg_conv_gemm_f16s(const void *x, const void *wt_split, const float *bias, const void *res, void *y,
                        int B, int H, int W, int Cin, int Cout, int ksize, int stride, int relu,
                        int res_stride, void *scratch, size_t scratch_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Data formats either side of the path (the reference's eval.py driver, which keeps frames in
 * host NumPy arrays; here they stay in HBM).  channel_flip != 0 reverses the three channels of
 * every pixel (cv2's BGR <-> RGB, eval.py:79,113).
 * ------------------------------------------------------------------------------------- */
/* eval.py:80 `frame / 255.` (float64) as TF receives it (float32): dst = (float)((double)src / 255.0).
 * src [n_pixels*3] uint8, dst [n_pixels*3] float32 (16-byte aligned). */
int dvsg_frames_u8_to_f32(const uint8_t *src, size_t n_pixels, int channel_flip, float *dst, void *stream);
/* eval.py:80 with a size change: cv2.resize(frame / 255., (dst_W, dst_H)), default INTER_LINEAR,
 * on float64, result as float32.  src [n,src_H,src_W,3] uint8 -> dst [n,dst_H,dst_W,3] float32.
 * OpenCV is neither in the reference tree nor in this image: restated from its published
 * algorithm, parity unpinned.  u8_dst (optional) [n,dst_H,u8_W,3] receives np.uint8(resized * 255.)
 * -- rendered from the float64 value, as eval.py:112 does for the unstable half of its output
 * video -- in columns [u8_x0, u8_x0 + dst_W), in the channel order of src. */
int dvsg_frames_resize_u8_f32(const uint8_t *src, int n, int src_H, int src_W, int channel_flip, float *dst,
                              int dst_H, int dst_W, uint8_t *u8_dst, int u8_W, int u8_x0, void *stream);
/* eval.py:103-104 `np.concatenate(total_frames[sample_idx], axis=2)` for B windows at once:
 * patches[b, y, x, 3 s + c] = pool[idx[b*S + s], y, x, c].  pool [n_pool,H,W,3] float32, idx [B*S]
 * int32 ON THE DEVICE, patches [B,H,W,3S] float32 (16-byte aligned).  An index outside
 * [0, n_pool) reads as a frame of zeros. */
int dvsg_window_gather_f32(const float *pool, int n_pool, int H, int W, const int32_t *idx, int B, int S,
                           float *patches, void *stream);

/* ---------------------------------------------------------------------------------------
 * The evaluation graph fed straight from a FRAME RING: what eval.py:76-81,103-110 does on the
 * host -- frames / 255., np.concatenate of the 7 window frames on the channel axis, u_t = the newest of them --
 * happens inside conv1's load stage and the warp's tap loads, so the [B,H,W,21] window tensor never exists.
 *   pool   [n_pool,H,W,3] RGB frames on the device: float32 in [0,1] (dvsg_stabilize_ring_f32; e.g. the history of
 *          eval.py:93-124, whose write-back of s_t_pred is a float frame), or the raw uint8 frames
 *          (dvsg_stabilize_ring_u8: eval_train.py's clips, independent windows; 4x less input traffic).  uint8:
 *          float32(v / 255.) * 255 == v exactly for every byte value, so conv1's scaled input float(v) - mean is
 *          bit-identical to the float path's on the converted frame.
 *   table  [B,7] int32 ON THE DEVICE: pool frame of window slot s of window b, oldest to newest (the `sample_idx`
 *          of eval.py:103; coupe.dvsg_amd.clip.window_index_table builds it for a whole clip).  u_t of window b is
 *          frame table[b][6].  An index outside [0, n_pool) reads as a frame of zeros (dvsg_window_gather_f32).
 *   precision  DVSG_PRECISION_F32 (the reference's arithmetic: bit-identical to dvsg_window_gather_f32 +
 *          dvsg_stabilize_f32 on the same frames), DVSG_PRECISION_F16, DVSG_PRECISION_F32S (same as the *_f16 / *_f32s
 *          entry points on the gathered window).
 * Outputs, workspace and stream as dvsg_stabilize_f32; s_t_pred may be a frame of `pool` itself only if no window
 * of this call reads it (eval.py:116 writes the result back into the history AFTER the step).
 * dvsg_locnet_forward_ring: the CNN alone from a ring -- F_t [B,25,2] into `out` (stage = -1), or the parity tap of
 * `stage` (0..18, see dvsg_locnet_forward_tap_f32) with its [h,w,c] in act_dims_host.
 * ------------------------------------------------------------------------------------- */
#define DVSG_PRECISION_F32 0
#define DVSG_PRECISION_F16 1
#define DVSG_PRECISION_F32S 2
#define DVSG_PRECISION_F32X3 3
int dvsg_stabilize_ring_f32(const dvsg_locnet_t *net, int precision, const float *pool, int n_pool,
                            const int32_t *table, int B, int H, int W, float *s_t_pred, float *F_t, float *x_s,
                            float *y_s, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_stabilize_ring_u8(const dvsg_locnet_t *net, int precision, const uint8_t *pool, int n_pool,
                           const int32_t *table, int B, int H, int W, float *s_t_pred, float *F_t, float *x_s,
                           float *y_s, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_locnet_forward_ring(const dvsg_locnet_t *net, int precision, const void *pool, int pool_is_u8, int n_pool,
                             const int32_t *table, int B, int H, int W, int stage, float *out, size_t out_bytes,
                             int *act_dims_host, void *workspace, size_t workspace_bytes, void *stream);
/* ---------------------------------------------------------------------------------------
 * eval_train.py's evaluation graph (eval_train.py:25-51): unlike model.py's, its CNN input is
 * `patches_masked_t = patches_t * mask` (:43-45), where `random_mask` (:53-64, = model.py:156-167) warps an all-ones
 * image of the 18 history channels with ProjectiveTransformer and a near-identity homography H and leaves the newest
 * frame's 3 channels unmasked; the TPS warp still samples the UNMASKED u_t (:48).
 *   dvsg_random_mask_plane_f32  theta [B,8] (the homography of :55-57 AFTER its scale and identity offset; the
 *          reference draws it with tf.random_uniform inside the graph, here the caller supplies it) -> mask [B,H,W]:
 *          ProjectiveTransformer(out_size).transform(ones, theta) -- grid (spatial_transformer.py:423-452) and
 *          sampler B's blend (:545-562) on taps that read 1 inside the image and 0 on the zero ring.  The warp of an
 *          all-ones image is the same in every channel, so ONE plane stands for `random_masks_t[..., :18]`.
 *   dvsg_stabilize_masked_f32 / dvsg_stabilize_ring_masked_{f32,u8}: dvsg_stabilize_* / dvsg_stabilize_ring_* with that
 *          plane multiplied into the 18 history channels INSIDE conv1's load stage ((x * m) * 255 - mean, three
 *          roundings like the TF ops); neither a [B,H,W,21] product tensor nor a multiply launch exists.  A plane of
 *          exact ones reproduces the unmasked entry points bit for bit.  precision: DVSG_PRECISION_*.
 *   dvsg_locnet_forward_masked: the CNN alone (stage -1: F_t into `out`) or a parity tap (stage 0..18) from a masked
 *          source; src_kind 0 = window tensor [B,H,W,21], 1 = float32 frame pool + table, 2 = uint8 frame pool + table.
 * ------------------------------------------------------------------------------------- */
int dvsg_random_mask_plane_f32(const float *theta, int B, int H, int W, float *mask, void *stream);
int dvsg_stabilize_masked_f32(const dvsg_locnet_t *net, int precision, const float *patches_t, const float *u_t,
                              const float *mask, int B, int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                              void *workspace, size_t workspace_bytes, void *stream);
int dvsg_stabilize_ring_masked_f32(const dvsg_locnet_t *net, int precision, const float *pool, int n_pool,
                                   const int32_t *table, const float *mask, int B, int H, int W, float *s_t_pred, float *F_t,
                                   float *x_s, float *y_s, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_stabilize_ring_masked_u8(const dvsg_locnet_t *net, int precision, const uint8_t *pool, int n_pool,
                                  const int32_t *table, const float *mask, int B, int H, int W, float *s_t_pred, float *F_t,
                                  float *x_s, float *y_s, void *workspace, size_t workspace_bytes, void *stream);
int dvsg_locnet_forward_masked(const dvsg_locnet_t *net, int precision, const void *src, int src_kind, int n_pool,
                               const int32_t *table, const float *mask, int B, int H, int W, int stage, float *out,
                               size_t out_bytes, int *act_dims_host, void *workspace, size_t workspace_bytes, void *stream);
/* eval.py:112 `np.uint8(x * 255.)`: float64 product, truncation toward zero (values outside
 * [0, 256) saturate; NumPy leaves them undefined).  src [n,H,W,3] float32 is written into columns
 * [dst_x0, dst_x0 + W) of dst [n,H,dst_W,3] uint8 -- dst_W = 2 W and dst_x0 = 0 / W give the
 * reference's side-by-side layout, dst_W = W and dst_x0 = 0 a plain frame. */
int dvsg_frames_f32_to_u8(const float *src, int n, int H, int W, int channel_flip, uint8_t *dst, int dst_W,
                          int dst_x0, void *stream);
/* same for frames the caller holds as float64 (the dtype of eval.py's history list) */
int dvsg_frames_f64_to_u8(const double *src, int n, int H, int W, int channel_flip, uint8_t *dst, int dst_W,
                          int dst_x0, void *stream);

/* Second building block: conv2 (3x3, pad 1, stride 1 or 2, Cin -> 64, + bias + ReLU) and conv3 (1x1,
 * 64 -> Cout, + bias + residual + ReLU) of a block-1 bottleneck unit as one kernel; the [M,64]
 * intermediate never leaves the chip.  float32.  x [B,H,W,Cin] (Cin % 32 == 0, >= 64); wt2 [64][9*Cin]
 * (k order kh, kw, c); wt3 [Cout][64] (Cout % 128 == 0); res / y as in dvsg_conv_gemm_f32. */
int dvsg_conv3x3_1x1_f32(const float *x, const float *wt2, const float *bias2, const float *wt3, const float *bias3,
                         const float *res, float *y, int B, int H, int W, int Cin, int Cout, int stride, int res_stride,
                         void *stream);

/* Diagnostic A/B switches for kernel experiments, process-global: "conv_variant", "conv1_variant", "fuse_conv",
 * "fuse_shortcut", "f16_split", "f16_pair_mask", and for the float16 mode's big launches "wide16_min_tiles" (tiles from
 * which the 256 x 128 geometry runs: 128), "wide16_packed" / "wide16_arows" / "wide16_hreuse" / "fused_hreuse" (1: weight
 * stages from the packed copy, 128-byte activation rows, a 3x3 kernel row's taps from one staged run, the same in block 1's
 * fused kernel; 0 selects the kernel each replaced), "warp_xcd" (0: the sampler kernels' workgroups in plain dispatch order).
 * Results do not depend on them beyond float32 re-association. */
int dvsg_debug_set_option(const char *name, int value);
/* The A/B form of dvsg_locnet_calibrate_f16 (tools/f16_ef_sweep.py): re-rounds the plain float16 weight copies and leaves
 * the pair policy to "f16_pair_mask".  mode 0: round to nearest (what dvsg_locnet_create makes); 1: error feedback with
 * mu = 1 (measured: useless -- channel means are far from uniform); 2: calibrated channel means. */
int dvsg_debug_calibrate_f16_weights(dvsg_locnet_t *net, const float *patches, int B, int H, int W, int mode, void *workspace,
                                     size_t workspace_bytes, void *stream);

/* ---------------------------------------------------------------------------------------
 * Measurement hook (bench.py's roofline leg; not part of the reference surface).  While
 * armed for one kernel class, every launch of that class made by this library is bracketed
 * by a hipEvent pair on the launch stream.  dvsg_prof_end synchronises on those events and
 * returns the summed durations, the launch count and the algorithmic FLOPs / bytes of those
 * launches.  Classes: 0 conv1 (7x7/2 + scale_RGB), 1 conv 3x3, 2 conv 1x1, 3 max pool,
 * 4 head (avg pool + dense), 5 TPS solve, 6 TPS grid + sampler A, 7 flow / STN samplers,
 * 8 block 1's fused conv2 + conv3.
 * Process-global and not thread-safe: arm it only around single-threaded benchmark code.
 * ------------------------------------------------------------------------------------- */
/* 1 if DVSG_ROCTX=1 was set when the library first looked and a roctx library could be loaded: every stage of the
 * evaluation graph (dvsg/conv1, dvsg/pool1, dvsg/block<b>/unit_<u>, dvsg/head, dvsg/tps) then opens a roctx range
 * around its launches, for `rocprofv3 --kernel-trace --marker-trace` (tracing). */
int dvsg_markers_enabled(void);
int dvsg_prof_begin(int kernel_class);
int dvsg_prof_end(double *total_ms, int *launches, double *flops, double *bytes);

#ifdef __cplusplus
}
#endif
#endif /* DVSG_AMD_H */
