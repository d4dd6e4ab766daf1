// Geometric warps of the DVSG hot path as gfx950 HIP kernels: thin-plate-spline solve,
// fused TPS grid generation + sampler A, optical-flow warp (sampler C), and the affine /
// projective / elastic spatial transformers over sampler B.
//
// All of these are HBM-bound gathers (24-32 algorithmic bytes per output pixel); none is
// GEMM-shaped, so there is no MFMA here.  What matters: every store is a fully coalesced
// 768-byte wave write (lane l owns pixel base+l, 12 B each), the per-sample transform
// coefficients are staged once per workgroup in LDS and read back as wave-uniform
// (broadcast) ds_read_b128, and the [B,P+3,H*W] basis the reference materialises
// (ThinPlateSpline.py:92-111) lives only in registers.
//
// This translation unit is compiled with -ffp-contract=off: the reference graph is a chain
// of separately rounded float32 mul / add ops, and keeping the same roundings lets the
// parity tests compare source coordinates to the oracle within 1-2 ulp (only logf differs).
#include <vector>

#include <cstdint>

#include "common.h"
#include "cnn_device.h"

namespace dvsg {
namespace {

constexpr int kThreads = 256;
constexpr int kMaxPts = 61;  // P + 3 <= 64

// ----------------------------------------------------------------------------------------
// TPS system solve: ThinPlateSpline.py:143-166.  One wave per batch sample, thread = row of
// the (P+3)x(P+3) system, Gauss-Jordan with partial pivoting in float64 on the float32-built
// matrix, two right-hand sides (x and y of the displaced control points).
// ----------------------------------------------------------------------------------------
constexpr int kSolveLd = 67;  // 64 + 2 rhs columns, +1 pad

__global__ __launch_bounds__(64) void tps_solve_kernel(const float *__restrict__ coord,
                                                       long coord_bstride,
                                                       const float *__restrict__ rhs,
                                                       int rhs_is_vector, int P,
                                                       float *__restrict__ T, double *__restrict__ Td,
                                                       int *__restrict__ n_singular) {
  __shared__ double A[64 * kSolveLd];
  __shared__ float cx[64], cy[64];
  __shared__ int pivrow[64];
  const int b = blockIdx.x;
  const int t = threadIdx.x;
  const int n = P + 3;
  if (t < P) {
    cx[t] = coord[b * coord_bstride + t * 2 + 0];
    cy[t] = coord[b * coord_bstride + t * 2 + 1];
  }
  __syncthreads();
  double *row = A + t * kSolveLd;
  if (t < P) {
    // W_0 = [p, r] (:156), p = [1, cx, cy] (:148), r = d2 * log(d2 + 1e-6) (:152-153)
    row[0] = 1.0;
    row[1] = (double)cx[t];
    row[2] = (double)cy[t];
    for (int j = 0; j < P; ++j) {
      float dx = cx[t] - cx[j];
      float dy = cy[t] - cy[j];
      float d2 = dx * dx + dy * dy;
      float r = d2 * logf(d2 + 1e-6f);
      row[3 + j] = (double)r;
    }
    float rx = rhs[((size_t)b * P + t) * 2 + 0];
    float ry = rhs[((size_t)b * P + t) * 2 + 1];
    if (rhs_is_vector) {  // coord + vector in float32 (:161)
      rx = cx[t] + rx;
      ry = cy[t] + ry;
    }
    row[n] = (double)rx;
    row[n + 1] = (double)ry;
  } else if (t < n) {
    // W_1 = [0_3x3, p^T] (:157); rhs padded with three zero rows (:161-162)
    const int q = t - P;
    row[0] = row[1] = row[2] = 0.0;
    for (int j = 0; j < P; ++j) row[3 + j] = q == 0 ? 1.0 : (q == 1 ? (double)cx[j] : (double)cy[j]);
    row[n] = 0.0;
    row[n + 1] = 0.0;
  }
  __syncthreads();
  // scale of the system, for the singularity test below (the p columns hold 1.0, so amax >= 1)
  double amax = 0.0;
  if (t < n)
    for (int c = 0; c < n; ++c) amax = fmax(amax, fabs(row[c]));
  for (int off = 32; off > 0; off >>= 1) amax = fmax(amax, __shfl_xor(amax, off));
  double minpiv = amax;
  bool used = false;
  for (int k = 0; k < n; ++k) {
    double v = (t < n && !used) ? fabs(row[k]) : -1.0;
    int idx = t;
    for (int off = 32; off > 0; off >>= 1) {
      double ov = __shfl_xor(v, off);
      int oi = __shfl_xor(idx, off);
      if (ov > v || (ov == v && oi < idx)) {
        v = ov;
        idx = oi;
      }
    }
    const int p = idx;
    const double *prow = A + p * kSolveLd;
    const double piv = prow[k];
    minpiv = fmin(minpiv, fabs(piv));
    if (t < n && t != p) {
      const double f = row[k] / piv;
      for (int c = k; c < n + 2; ++c) row[c] -= f * prow[c];
    }
    if (t == p) {
      used = true;
      pivrow[k] = p;
    }
    __syncthreads();
  }
  // tf.matrix_inverse (:159) raises InvalidArgument ("Input is not invertible") when the LU has a
  // zero pivot: repeated or collinear control points.  A launch cannot raise, so such a sample gets
  // T = NaN (never a finite garbage map) and is counted in *n_singular for the host to turn into an
  // error (dvsg_tps_solve_checked_f32).  Well-posed systems (cond(W) ~ 4e2) are nowhere near the bound.
  const bool singular = !(minpiv > 1e-13 * amax);
  if (singular && t == 0 && n_singular) atomicAdd(n_singular, 1);
  // T = (W^-1 tp)^T (:163-164): T[b][c][k]
  if (t < n) {
    const double *prow = A + pivrow[t] * kSolveLd;
    const double piv = singular ? 0.0 : prow[t];
    if (singular) {
      const double qnan = __builtin_nan("");
      if (Td) {
        Td[((size_t)b * 2 + 0) * n + t] = qnan;
        Td[((size_t)b * 2 + 1) * n + t] = qnan;
      } else {
        T[((size_t)b * 2 + 0) * n + t] = (float)qnan;
        T[((size_t)b * 2 + 1) * n + t] = (float)qnan;
      }
    } else if (Td) {  // float64 solution (columns of W^-1 for tps_apply_kernel)
      Td[((size_t)b * 2 + 0) * n + t] = prow[n] / piv;
      Td[((size_t)b * 2 + 1) * n + t] = prow[n + 1] / piv;
    } else {
      T[((size_t)b * 2 + 0) * n + t] = (float)(prow[n] / piv);
      T[((size_t)b * 2 + 1) * n + t] = (float)(prow[n + 1] / piv);
    }
  }
}

// The evaluation graph solves the SAME 28x28 system for every frame: its control points are the
// constant V_src grid (model.py:105-111), only the right-hand side (V_src + F_t) changes.  With the
// first P columns of W^-1 computed once (float64, by the solver above on unit right-hand sides; the
// last three right-hand-side rows are always zero), T = (W^-1 [coord + vector; 0])^T is 56 dot products
// of length P per sample.  winv_cols is [P][n]: column j of W^-1.
__global__ __launch_bounds__(64) void tps_apply_kernel(const double *__restrict__ winv_cols,
                                                       const float *__restrict__ coord,
                                                       const float *__restrict__ rhs, int rhs_is_vector,
                                                       int P, float *__restrict__ T) {
  __shared__ double yx[64], yy[64];
  const int b = blockIdx.x, t = threadIdx.x, n = P + 3;
  if (t < P) {
    float rx = rhs[((size_t)b * P + t) * 2 + 0];
    float ry = rhs[((size_t)b * P + t) * 2 + 1];
    if (rhs_is_vector) {  // coord + vector in float32 (:161)
      rx = coord[t * 2 + 0] + rx;
      ry = coord[t * 2 + 1] + ry;
    }
    yx[t] = (double)rx;
    yy[t] = (double)ry;
  }
  __syncthreads();
  if (t < n) {
    double ax = 0.0, ay = 0.0;
    for (int j = 0; j < P; ++j) {
      const double w = winv_cols[(size_t)j * n + t];
      ax += w * yx[j];
      ay += w * yy[j];
    }
    T[((size_t)b * 2 + 0) * n + t] = (float)ax;
    T[((size_t)b * 2 + 1) * n + t] = (float)ay;
  }
}

// ----------------------------------------------------------------------------------------
// Samplers.
// ----------------------------------------------------------------------------------------
template <int C>
struct Pix {
  float v[C];
};

template <int C>
__device__ __forceinline__ Pix<C> load_pix(const float *__restrict__ p) {
  Pix<C> r;
#pragma unroll
  for (int c = 0; c < C; ++c) r.v[c] = p[c];
  return r;
}
// a uint8 frame: the pixel as eval.py:80 hands it to the graph, float32(v / 255.) -- the float64 quotient rounded
// once.  A correctly rounded float32 division gives the same value for all 256 bytes (exhaustive:
// tests/test_frames_cpu.py); hipcc's `/` is correctly rounded (no -ffast-math in this build).
template <int C>
__device__ __forceinline__ Pix<C> load_pix(const uint8_t *__restrict__ p) {
  Pix<C> r;
#pragma unroll
  for (int c = 0; c < C; ++c) r.v[c] = (float)p[c] / 255.0f;
  return r;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// Guarded float -> int conversion (floor already applied): out-of-range source coordinates
// are garbage in the reference too (tf.cast saturates to INT_MIN on x86); keep it defined.
__device__ __forceinline__ int f2i(float f) {
  f = f < -1073741824.f ? -1073741824.f : (f > 1073741824.f ? 1073741824.f : f);
  return (int)f;
}

// Sampler A (ThinPlateSpline.py:30-90): normalised (xs, ys) -> (x+1)*W/2, indices clipped to
// the image BEFORE the weights are formed, so out-of-range taps coincide and cancel.
template <int C>
struct TapsA {  // loaded taps (x0,y0) (x0,y1) (x1,y0) (x1,y1) and their weights
  Pix<C> a, b, c, d;
  float wa, wb, wc, wd;
};
template <>
struct TapsA<0> {  // generic channel count: the blend reads through the tap pointers
  const float *pa, *pb, *pc, *pd;
  float wa, wb, wc, wd;
};

// Address + weight computation and the four tap loads; the loads are only ISSUED here, so the
// caller can do other work before sample_a_blend() needs them.
template <int C, typename TU = float>
__device__ __forceinline__ void sample_a_load(const TU *__restrict__ img /* [H,W,C] of this sample */,
                                              int H, int W, int Cn, float xs, float ys, TapsA<C> &t) {
  const float x = ((xs + 1.0f) * (float)W) / 2.0f;  // :48
  const float y = ((ys + 1.0f) * (float)H) / 2.0f;  // :49
  int x0 = f2i(floorf(x));
  int y0 = f2i(floorf(y));
  int x1 = x0 + 1;
  int y1 = y0 + 1;
  x0 = clampi(x0, 0, W - 1);  // :57-60
  x1 = clampi(x1, 0, W - 1);
  y0 = clampi(y0, 0, H - 1);
  y1 = clampi(y1, 0, H - 1);
  const float x0f = (float)x0, x1f = (float)x1, y0f = (float)y0, y1f = (float)y1;
  t.wa = (x1f - x) * (y1f - y);  // :85-88
  t.wb = (x1f - x) * (y - y0f);
  t.wc = (x - x0f) * (y1f - y);
  t.wd = (x - x0f) * (y - y0f);
  const TU *pa = img + ((size_t)y0 * W + x0) * Cn;  // (x0,y0)
  const TU *pb = img + ((size_t)y1 * W + x0) * Cn;  // (x0,y1)
  const TU *pc = img + ((size_t)y0 * W + x1) * Cn;  // (x1,y0)
  const TU *pd = img + ((size_t)y1 * W + x1) * Cn;  // (x1,y1)
  if constexpr (C > 0) {
    t.a = load_pix<C>(pa);
    t.b = load_pix<C>(pb);
    t.c = load_pix<C>(pc);
    t.d = load_pix<C>(pd);
  } else {
    t.pa = pa; t.pb = pb; t.pc = pc; t.pd = pd;
  }
}

template <int C>
__device__ __forceinline__ void sample_a_blend(const TapsA<C> &t, int Cn, float *__restrict__ dst) {
  if constexpr (C > 0) {
#pragma unroll
    for (int c = 0; c < C; ++c) dst[c] = ((t.wa * t.a.v[c] + t.wb * t.b.v[c]) + t.wc * t.c.v[c]) + t.wd * t.d.v[c];  // :89
  } else {
    for (int c = 0; c < Cn; ++c) dst[c] = ((t.wa * t.pa[c] + t.wb * t.pb[c]) + t.wc * t.pc[c]) + t.wd * t.pd[c];
  }
}

// Samplers B and C share this tail (spatial_transformer.py:517-562,
// warp_with_optical_flow.py:128-173): (x, y) in unpadded pixel units, clamped to [-1,W] /
// [-1,H], +1 into the zero-ringed image, floor, upper index min()-ed, weights from the
// UNclamped x0+1.  The ring is never materialised: a tap on it reads as 0.
template <int C>
struct TapsB {  // loaded taps (x0,y0) (x1,y0) (x0,y1) (x1,y1), weights, and which taps are on the ring
  Pix<C> a, b, c, d;
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
};
template <>
struct TapsB<0> {  // generic channel count: the blend reads through the tap pointers
  const float *p00, *p01, *p10, *p11;
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
};

// Address / weight computation and the four tap loads (only ISSUED here, so a thread can put the taps
// of several pixels in flight before it blends the first one).  A tap on the ring reads as 0: the
// loads are unconditional, from indices clamped into the image, and the ring is applied in the blend
// as a select -- a predicated load that feeds arithmetic makes the compiler wait for each load in turn.
// weights, ring flags and the (clamped) tap coordinates of one sample; shared by the image samplers and the mask plane
struct PadGeom {
  float w00, w01, w10, w11;
  bool v00, v01, v10, v11;
  int xa, xb, ya, yb;   // tap coordinates clamped into the image (what an unconditional global load may touch)
  int x0, y0;           // the lower tap in the zero-ringed image's coordinates (image pixel x0 - 1, y0 - 1), in [0, W + 1] / [0, H + 1]
};
__device__ __forceinline__ PadGeom padded_geom(int H, int W, float x, float y) {
  PadGeom g;
  const float wf = (float)W, hf = (float)H;
  x = fminf(fmaxf(x, -1.0f), wf);  // (W-1)+1
  y = fminf(fmaxf(y, -1.0f), hf);
  x = x + 1.0f;
  y = y + 1.0f;
  const float x0f = floorf(x), y0f = floorf(y);
  const float x1f = x0f + 1.0f, y1f = y0f + 1.0f;
  const int x0 = (int)x0f, y0 = (int)y0f;
  const int x1 = (int)fminf(x1f, wf + 1.0f);
  const int y1 = (int)fminf(y1f, hf + 1.0f);
  g.w00 = (x1f - x) * (y1f - y);
  g.w01 = (x - x0f) * (y1f - y);
  g.w10 = (x1f - x) * (y - y0f);
  g.w11 = (x - x0f) * (y - y0f);
  const bool vx0 = x0 >= 1 && x0 <= W, vx1 = x1 >= 1 && x1 <= W;
  const bool vy0 = y0 >= 1 && y0 <= H, vy1 = y1 >= 1 && y1 <= H;
  g.v00 = vx0 && vy0; g.v01 = vx1 && vy0; g.v10 = vx0 && vy1; g.v11 = vx1 && vy1;
  g.xa = clampi(x0 - 1, 0, W - 1); g.xb = clampi(x1 - 1, 0, W - 1);
  g.ya = clampi(y0 - 1, 0, H - 1); g.yb = clampi(y1 - 1, 0, H - 1);
  g.x0 = x0; g.y0 = y0;
  return g;
}

template <int C>
__device__ __forceinline__ void sample_padded_load(const float *__restrict__ img, int H, int W, int Cn, float x,
                                                   float y, TapsB<C> &t) {
  const PadGeom g = padded_geom(H, W, x, y);
  t.w00 = g.w00; t.w01 = g.w01; t.w10 = g.w10; t.w11 = g.w11;
  t.v00 = g.v00; t.v01 = g.v01; t.v10 = g.v10; t.v11 = g.v11;
  const float *p00 = img + ((long)g.ya * W + g.xa) * Cn;
  const float *p01 = img + ((long)g.ya * W + g.xb) * Cn;
  const float *p10 = img + ((long)g.yb * W + g.xa) * Cn;
  const float *p11 = img + ((long)g.yb * W + g.xb) * Cn;
  if constexpr (C > 0) {
    t.a = load_pix<C>(p00);
    t.b = load_pix<C>(p01);
    t.c = load_pix<C>(p10);
    t.d = load_pix<C>(p11);
  } else {
    t.p00 = p00; t.p01 = p01; t.p10 = p10; t.p11 = p11;
  }
}

template <int C>
__device__ __forceinline__ void sample_padded_blend(const TapsB<C> &t, int Cn, float *__restrict__ dst) {
  if constexpr (C > 0) {
#pragma unroll
    for (int c = 0; c < C; ++c)
      dst[c] = ((t.w00 * (t.v00 ? t.a.v[c] : 0.f) + t.w01 * (t.v01 ? t.b.v[c] : 0.f)) + t.w10 * (t.v10 ? t.c.v[c] : 0.f)) +
               t.w11 * (t.v11 ? t.d.v[c] : 0.f);
  } else {
    for (int c = 0; c < Cn; ++c) {
      const float a = t.v00 ? t.p00[c] : 0.f;
      const float bq = t.v01 ? t.p01[c] : 0.f;
      const float cq = t.v10 ? t.p10[c] : 0.f;
      const float d = t.v11 ? t.p11[c] : 0.f;
      dst[c] = ((t.w00 * a + t.w01 * bq) + t.w10 * cq) + t.w11 * d;
    }
  }
}

template <int C>
__device__ __forceinline__ void store_pix(float *__restrict__ out, size_t pix, int Cn,
                                          const float *__restrict__ v) {
  if constexpr (C == 3) {
    // one 12-byte store per lane (global_store_dwordx3 needs only 4-byte alignment): a wave writes
    // 768 contiguous bytes with one instruction instead of three stride-12 dword stores
    typedef float floatx3 __attribute__((ext_vector_type(3)));
    typedef floatx3 floatx3_a4 __attribute__((aligned(4)));
    *reinterpret_cast<floatx3_a4 *>(out + pix * 3) = floatx3{v[0], v[1], v[2]};
  } else if constexpr (C > 0) {
#pragma unroll
    for (int c = 0; c < C; ++c) out[pix * C + c] = v[c];
  } else {
    for (int c = 0; c < Cn; ++c) out[pix * Cn + c] = v[c];
  }
}

constexpr int kMaxGenericC = 64;

// ----------------------------------------------------------------------------------------
// Fused TPS grid generation + sampler A (ThinPlateSpline.py:92-141).
// Thread = one output column, PPT consecutive rows: (x_t - px)^2 is shared by the rows, and
// each of the P control points costs one broadcast ds_read_b128 {px, py, T0, T1}.
// ----------------------------------------------------------------------------------------
typedef float floatx2 __attribute__((ext_vector_type(2)));

// The 25 basis terms per pixel make this kernel VALU-bound, so the inner loop is written for the
// packed-float32 pipe -- one v_pk_* instruction per PAIR of rows -- and kept to 5 packed ops + 2
// v_log_f32 per control point and pair:
//   * (y_t - py)^2 -- the same for every column of a row -- is computed once per workgroup and read
//     back as a broadcast ds_read_b128; (x_t - px)^2 is shared by the thread's rows;
//   * r = d2 ln(d2 + 1e-6) enters the map only through T . r, so the ln 2 of ln = ln 2 x log2 is
//     folded into the two T entries of the control point when they are staged in LDS, and the loop
//     uses v_log_f32 (log2, 1 ulp) as it stands -- no product by a hi / lo ln 2 per term;
//   * T . basis accumulates with fused multiply-adds.  The reference's T @ grid is a GEMM (Eigen:
//     blocked, FMA where the CPU has it), so no summation order or rounding of it is pinned; the k
//     order of the oracle is kept.
// Against the previous form (device-library ln, separately rounded mul / add: 11 packed ops + 2 logs)
// the source grid moves by < 1e-3 px at 720p, inside the float32 evaluation noise of the map itself
// (3e-3 px against the oracle either way); tests/test_gpu_warps.py bounds grid and pixel error.
// A thread owns one column and 4 rows (two pairs); a variant with 8 rows per thread that issued one
// group's tap loads under the other group's basis loop was no faster, so the simple form stays.
constexpr int kTpsRows = 4;
constexpr float kLn2 = 0x1.62e43p-1f;

// U is [B,H,W,C], or -- u_index given -- a pool of frames [n_pool,H,W,C] of which sample b reads frame
// u_index[b * u_stride] (the frame ring of dvsg_stabilize_ring_*: u_t is the newest frame of the window; an index
// outside the pool reads as a zero frame, like dvsg_window_gather_f32).  TU = uint8_t: raw frames, / 255. fused.
template <int C, typename TU = float>
__global__ __launch_bounds__(kThreads) void tps_warp_kernel(
    const TU *__restrict__ U, const float *__restrict__ coord, long coord_bstride,
    const float *__restrict__ T, int H, int W, int Cn, int P, int out_h, int out_w, float step_x,
    float step_y, float *__restrict__ out,
    float *__restrict__ xs_out, float *__restrict__ ys_out, const int *__restrict__ u_index = nullptr, int u_stride = 0,
    int n_pool = 0) {
  __shared__ float4 sp[64];      // {px, py, T[0][3+k], T[1][3+k]}
  __shared__ float4 sdy[64];     // (y_t[r] - py)^2 for the 4 rows of this workgroup
  __shared__ float sa[6];        // T[0][0..2], T[1][0..2]
  const int b = blockIdx.z;
  const int n = P + 3;
  const int t = threadIdx.x;
  const int i0 = blockIdx.y * kTpsRows;
  if (t < P) {
    const float px = coord[b * coord_bstride + t * 2], py = coord[b * coord_bstride + t * 2 + 1];
    sp[t] = make_float4(px, py, T[((size_t)b * 2) * n + 3 + t] * kLn2, T[((size_t)b * 2 + 1) * n + 3 + t] * kLn2);
    float dy2[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float dy = (-1.0f + step_y * (float)(i0 + r)) - py;  // :96
      dy2[r] = dy * dy;
    }
    sdy[t] = make_float4(dy2[0], dy2[1], dy2[2], dy2[3]);
  } else if (t >= 64 && t < 70) {
    const int q = t - 64;
    sa[q] = T[((size_t)b * 2 + q / 3) * n + q % 3];
  }
  __syncthreads();
  const int j = blockIdx.x * kThreads + t;
  if (j >= out_w) return;
  const float x_t = -1.0f + step_x * (float)j;  // tf.linspace: start + step * i (:94)
  int frame = b;
  bool frame_ok = true;
  if (u_index) {
    frame = u_index[(size_t)b * u_stride];
    frame_ok = frame >= 0 && frame < n_pool;
    if (!frame_ok) frame = 0;
  }
  const TU *img = U ? U + (size_t)frame * H * W * Cn : nullptr;

  floatx2 xs2[2], ys2[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // T . [1, x_t, y_t, ...] accumulated in k order (:129)
    const floatx2 yy = {-1.0f + step_y * (float)(i0 + 2 * h), -1.0f + step_y * (float)(i0 + 2 * h + 1)};
    const float ax = sa[0] + sa[1] * x_t, ay = sa[3] + sa[4] * x_t;
    xs2[h] = floatx2{ax, ax} + sa[2] * yy;
    ys2[h] = floatx2{ay, ay} + sa[5] * yy;
  }
  for (int k = 0; k < P; ++k) {
    const float4 c = sp[k];
    const float4 q = sdy[k];
    const float dx = x_t - c.x;
    const float dx2 = dx * dx;
    const floatx2 dxx = {dx2, dx2};
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const floatx2 dyy = h == 0 ? floatx2{q.x, q.y} : floatx2{q.z, q.w};
      const floatx2 d2 = dxx + dyy;                        // :104
      const floatx2 e = d2 + floatx2{1e-6f, 1e-6f};
      const floatx2 l2 = {__builtin_amdgcn_logf(e.x), __builtin_amdgcn_logf(e.y)};
      const floatx2 rk = d2 * l2;                          // :105 up to the factor ln 2 carried by c.z / c.w
      xs2[h] = __builtin_elementwise_fma(floatx2{c.z, c.z}, rk, xs2[h]);
      ys2[h] = __builtin_elementwise_fma(floatx2{c.w, c.w}, rk, ys2[h]);
    }
  }
  const float xs[4] = {xs2[0].x, xs2[0].y, xs2[1].x, xs2[1].y};
  const float ys[4] = {ys2[0].x, ys2[0].y, ys2[1].x, ys2[1].y};
  TapsA<C> taps[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + r;
    if (i >= out_h) break;
    const size_t pix = ((size_t)b * out_h + i) * out_w + j;
    if (xs_out) xs_out[pix] = xs[r];
    if (ys_out) ys_out[pix] = ys[r];
    if (img) sample_a_load<C, TU>(img, H, W, Cn, xs[r], ys[r], taps[r]);  // all 16 tap loads in flight
  }
  if (!img) return;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = i0 + r;
    if (i >= out_h) break;
    float v[C > 0 ? C : kMaxGenericC];
    sample_a_blend<C>(taps[r], Cn, v);
    if constexpr (C > 0) {
      if (!frame_ok) {
#pragma unroll
        for (int c = 0; c < C; ++c) v[c] = 0.f;
      }
    }
    store_pix<C>(out, ((size_t)b * out_h + i) * out_w + j, Cn, v);
  }
}

// ----------------------------------------------------------------------------------------
// Sampler B / C front ends: flow (warp_with_optical_flow.py:106-120), explicit coords,
// affine (spatial_transformer.py:74-91), projective (:423-452), elastic (:276-296).
// ----------------------------------------------------------------------------------------
enum Src { kFlow = 0, kCoords = 1, kAffine = 2, kProjective = 3, kElastic = 4 };

struct StnParams {
  const float *im;      // [B,H,W,C] or null (grid only)
  const float *a;       // flow [B,H,W,2] | x_s | theta
  const float *b;       // y_s | L_inv [n,n+3]
  const float *c;       // source_points [2,n]
  float *out;           // [B,out_h,out_w,C]
  float *xs_out, *ys_out;
  int H, W, Cn, out_h, out_w, n;
  float step_x, step_y;
  int nbx, nby;         // column blocks / row groups per image (the grid is one-dimensional: see stn_kernel)
  int xcd;              // 1: each XCD takes a contiguous range of (image, row group, column block) -- diagnostic switch
};

template <int SRC, int C, int PPT>
__global__ __launch_bounds__(kThreads) void stn_kernel(StnParams p) {
  __shared__ float sc[2 * 64];   // elastic coefficients [2][n+3]
  __shared__ float ssrc[2 * 64]; // elastic source points [2][n]
  __shared__ float sth[9];
  // One-dimensional grid, decoded as (image, row group, column block) with the column block fastest.  The hardware deals
  // consecutive workgroup ids round-robin over the 8 XCDs, and a row group's taps reach +- the flow's range into the rows
  // of its neighbours: dealt that way every source line was fetched by ~3 different L2s (PMC, configs[2]: 2.65 GB fetched
  // per launch for 1.18 GB of frames + flow; 6.1 TB/s of fabric traffic -- the kernel was bound by it).  xcd_remap gives
  // each XCD a contiguous range of ids, i.e. whole bands of rows of whole images: neighbouring row groups share an L2.
  const int id = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int bx = id % p.nbx, by = (id / p.nbx) % p.nby;
  const int b = id / (p.nbx * p.nby);
  const int t = threadIdx.x;
  if (SRC == kAffine) {
    if (t < 6) sth[t] = p.a[(size_t)b * 6 + t];
    __syncthreads();
  } else if (SRC == kProjective) {
    if (t < 9) sth[t] = t < 8 ? p.a[(size_t)b * 8 + t] : 1.0f;  // :430
    __syncthreads();
  } else if (SRC == kElastic) {
    const int n = p.n, m = p.n + 3;
    if (t < 2 * m) {
      // coefficients = theta @ L_inv (:284), k order
      const int r = t / m, q = t % m;
      const float *th = p.a + ((size_t)b * 2 + r) * n;
      float acc = th[0] * p.b[q];
      for (int k = 1; k < n; ++k) acc = acc + th[k] * p.b[(size_t)k * m + q];
      sc[r * 64 + q] = acc;
    }
    if (t < 2 * n) ssrc[(t / n) * 64 + t % n] = p.c[t];
    __syncthreads();
  }
  const int j = bx * kThreads + t;
  const int i0 = by * PPT;
  if (j >= p.out_w) return;
  const float *img = p.im ? p.im + (size_t)b * p.H * p.W * p.Cn : nullptr;
  const float x_t = -1.0f + p.step_x * (float)j;
  // Three passes over the thread's PPT rows so that memory operations of all rows are in flight
  // together: coordinates (the flow / coordinate loads), then the 4 x PPT tap loads, then the blends.
  float xx[PPT], yy[PPT];
  float2 fl[PPT];
  if (SRC == kFlow) {
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int i = i0 + r < p.out_h ? i0 + r : p.out_h - 1;
      fl[r] = reinterpret_cast<const float2 *>(p.a)[((size_t)b * p.out_h + i) * p.out_w + j];
    }
  } else if (SRC == kCoords) {
#pragma unroll
    for (int r = 0; r < PPT; ++r) {
      const int i = i0 + r < p.out_h ? i0 + r : p.out_h - 1;
      const size_t pix = ((size_t)b * p.out_h + i) * p.out_w + j;
      fl[r] = make_float2(p.a[pix], p.b[pix]);
    }
  }
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const int i = i0 + r;
    const size_t pix = ((size_t)b * p.out_h + i) * p.out_w + j;
    const float y_t = -1.0f + p.step_y * (float)i;
    float xs, ys, x, y;
    if (SRC == kFlow) {
      x = (float)j + fl[r].x;  // :117-119
      y = (float)i + fl[r].y;
      xs = x;
      ys = y;
    } else {
      if (SRC == kCoords) {
        xs = fl[r].x;
        ys = fl[r].y;
      } else if (SRC == kAffine) {
        xs = (sth[0] * x_t + sth[1] * y_t) + sth[2];  // theta . [x_t; y_t; 1] in k order (:85)
        ys = (sth[3] * x_t + sth[4] * y_t) + sth[5];
      } else if (SRC == kProjective) {
        const float xq = (sth[0] * x_t + sth[1] * y_t) + sth[2];  // :438
        const float yq = (sth[3] * x_t + sth[4] * y_t) + sth[5];
        const float zq = (sth[6] * x_t + sth[7] * y_t) + sth[8];
        xs = zq != 0.f ? xq / zq : 0.f;  // tf.div_no_nan (:446-447)
        ys = zq != 0.f ? yq / zq : 0.f;
      } else {  // elastic: coefficients . [x; y; 1; U_0..U_{n-1}] (:288-290)
        xs = (sc[0] * x_t + sc[1] * y_t) + sc[2];
        ys = (sc[64] * x_t + sc[65] * y_t) + sc[66];
        for (int k = 0; k < p.n; ++k) {
          const float dx = x_t - ssrc[k];
          const float dy = y_t - ssrc[64 + k];
          const float rsq = dx * dx + dy * dy;              // :301
          const float u = rsq == 0.f ? 0.f : rsq * logf(rsq);  // log(0) -> 0 via is_inf (:302-304)
          xs = xs + sc[3 + k] * u;
          ys = ys + sc[64 + 3 + k] * u;
        }
      }
      // scale to [0, W-1] (:515-516)
      x = ((xs + 1.0f) / 2.0f) * ((float)p.W - 1.0f);
      y = ((ys + 1.0f) / 2.0f) * ((float)p.H - 1.0f);
    }
    if (SRC != kFlow && SRC != kCoords && i < p.out_h) {
      if (p.xs_out) p.xs_out[pix] = xs;
      if (p.ys_out) p.ys_out[pix] = ys;
    }
    xx[r] = x;
    yy[r] = y;
  }
  if (!img) return;
  TapsB<C> taps[PPT];
#pragma unroll
  for (int r = 0; r < PPT; ++r) sample_padded_load<C>(img, p.H, p.W, p.Cn, xx[r], yy[r], taps[r]);
#pragma unroll
  for (int r = 0; r < PPT; ++r) {
    const int i = i0 + r;
    if (i >= p.out_h) break;
    float v[C > 0 ? C : kMaxGenericC];
    sample_padded_blend<C>(taps[r], p.Cn, v);
    store_pix<C>(p.out, ((size_t)b * p.out_h + i) * p.out_w + j, p.Cn, v);
  }
}

// eval_train.py:53-64 / model.py:156-167 `random_mask`: mask = ProjectiveTransformer(out_size).transform(ones, H).  The warp of an
// all-ones image is the same in every channel, so ONE plane [B,H,W] is produced: the projective grid of stn_kernel<kProjective>
// (:423-452) and sampler B's blend (:545-562) with every tap inside the image reading 1 and every tap on the zero ring 0 --
// the same operations in the same order as the image path, no image.
__global__ __launch_bounds__(kThreads) void mask_plane_kernel(const float *__restrict__ theta, int H, int W, float step_x,
                                                             float step_y, float *__restrict__ out) {
  const int b = blockIdx.z, i = blockIdx.y;
  const int j = blockIdx.x * kThreads + threadIdx.x;
  if (j >= W) return;
  const float *th = theta + (size_t)b * 8;
  const float x_t = -1.0f + step_x * (float)j;
  const float y_t = -1.0f + step_y * (float)i;
  const float xq = (th[0] * x_t + th[1] * y_t) + th[2];  // :438
  const float yq = (th[3] * x_t + th[4] * y_t) + th[5];
  const float zq = (th[6] * x_t + th[7] * y_t) + 1.0f;   // :430: the ninth entry is 1
  const float xs = zq != 0.f ? xq / zq : 0.f;            // tf.div_no_nan (:446-447)
  const float ys = zq != 0.f ? yq / zq : 0.f;
  const float x = ((xs + 1.0f) / 2.0f) * ((float)W - 1.0f);  // :515-516
  const float y = ((ys + 1.0f) / 2.0f) * ((float)H - 1.0f);
  const PadGeom g = padded_geom(H, W, x, y);
  out[((size_t)b * H + i) * W + j] =
      ((g.w00 * (g.v00 ? 1.f : 0.f) + g.w01 * (g.v01 ? 1.f : 0.f)) + g.w10 * (g.v10 ? 1.f : 0.f)) + g.w11 * (g.v11 ? 1.f : 0.f);
}

// ----------------------------------------------------------------------------------------
// tf_warp (warp_with_optical_flow.py:96-176) on RGB frames with the source rows of a column strip STREAMED THROUGH LDS.
//
// stn_kernel<kFlow> gathers its 4 taps x 12 B per pixel straight from global memory: 16 `global_load_dwordx3` per
// thread whose lanes are 12 bytes apart and -- the flow differs from pixel to pixel -- scattered over 2-4 image rows per
// wave instruction.  The texture addresser works such an instruction off in 40 (constant flow) to 65 cycles (BASELINE
// configs[2]'s flow): SQ_WAIT_INST_ANY, waves stalled at the ISSUE of memory instructions, is 56 % of the wave cycles
// while the HBM-side traffic is the algorithmic one -- bound by the address path, not by HBM (3.4 TB/s of algorithmic
// bytes; 5.5 with a constant flow).  (A first LDS form, round 4 -- 64 x 16 tiles that reduce the bounding box of their
// taps and then stage it -- gave the same bits 25 % SLOWER: flow load -> reduction -> window load -> gather is three
// dependent memory round trips per tile with 12 waves per CU, and the box of a 64 x 16 tile under this flow is 3.2 tiles.)
//
// Here a workgroup owns a strip of 128 output columns and marches down a band of rows, 16 at a time.  The source window
// is FIXED relative to the step -- the step's rows and columns +- 12 pixels -- so nothing about it depends on the flow:
// its rows stream through a ring of 57 LDS rows (106 KB, one workgroup of 16 waves per CU), each fetched once per strip
// as whole 16-byte chunks per lane (1 KB per wave instruction), the 16 new rows and the flow of step s + 2 requested
// before step s is computed.  Taps inside the
// window come from LDS; a pixel with a tap outside it (a flow beyond +- 12 px: 0.3 % of the pixels at sigma = 4) takes
// its four taps from global memory as before; taps on the zero ring are never read.  Geometry (`padded_geom`) and blend
// are the functions stn_kernel uses on the same values in the same order: the output is bit-identical.
// ----------------------------------------------------------------------------------------
constexpr int kFsThreads = 1024;                           // 8 waves: 128 columns x 4 row pairs
constexpr int kFsW = 128, kFsStep = 16, kFsPPT = 2;        // strip width, rows per step, rows per thread
constexpr int kFsMX = 12, kFsMY = 12;                     // window margins, pixels
constexpr int kFsCols = kFsW + 2 * kFsMX + 1;             // 153 pixels: taps x .. x + 1
constexpr int kFsCpr = (kFsCols * 12 + 12 + 15) / 16;     // 116 chunks of 16 bytes per window row (up to 12 bytes of lead-in)
constexpr int kFsLive = kFsStep + 2 * kFsMY + 1;          // 41 rows feed a step
constexpr int kFsRing = kFsLive + kFsStep;                // 57: the next step's 16 new rows land beside them
constexpr int kFsNew = (kFsStep * kFsCpr + kFsThreads - 1) / kFsThreads;   // 2 chunks per thread and step
// (steps of 8 rows with 512 threads and two workgroups per CU -- the same 16 waves -- were 3 % slower: twice the barriers;
// steps of 24 rows, 135 KB, three rows per thread: 530 us against 390)

struct FlowStripParams {
  const float *im;      // [B,H,W,3], 16-byte aligned
  const float *flow;    // [B,H,W,2]
  float *out;           // [B,H,W,3]
  int H, W, B;
  int nstrips, nbands, band_rows;   // per image; band_rows is a multiple of kFsStep
  int xcd;
};

// a workgroup barrier that leaves global loads and stores in flight: __syncthreads() is s_waitcnt vmcnt(0) lgkmcnt(0) + s_barrier,
// which put every step's store acknowledgements and prefetch latency back in front of the barrier (SQ_WAIT_ANY 65 %)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(kFsThreads) __attribute__((amdgpu_waves_per_eu(4, 4)))
void flow_warp_strip_kernel(FlowStripParams p) {
  __shared__ __attribute__((aligned(16))) float ring[kFsRing * kFsCpr * 4];   // 106 KB: one workgroup (16 waves) per CU
  const int t = threadIdx.x;
  const int id = p.xcd ? xcd_remap((int)blockIdx.x, (int)gridDim.x) : (int)blockIdx.x;
  const int strip = id % p.nstrips, band = (id / p.nstrips) % p.nbands, b = id / (p.nstrips * p.nbands);
  const int H = p.H, W = p.W;
  const int c0 = strip * kFsW, wx0 = c0 - kFsMX;          // first window column (may be negative)
  const int i_begin = band * p.band_rows, i_end = min(H, i_begin + p.band_rows);
  const long img_pix = (long)b * H * W;
  const long total_bytes = (long)p.B * H * W * 12;
  const char *base = reinterpret_cast<const char *>(p.im);
  const float *img = p.im + img_pix * 3;
  const int j = c0 + (t & (kFsW - 1));
  const int rg = t >> 7;
  const bool col_ok = j < W;
  const int jc = col_ok ? j : W - 1;
  // lead-in (floats) of image row y's first chunk: ((img_pix + y W + wx0) * 3) mod 4, in 32-bit arithmetic
  const int a4 = (int)((img_pix + wx0) & 3), w4 = W & 3;
  const float xf = (float)j;

  // chunk c of image row y of the window: the 16 bytes at floor16(byte offset of pixel (y, wx0)) + 16 c.  UNCONDITIONAL:
  // a load under a branch makes hipcc wait for it at the end of the branch (`s_waitcnt vmcnt(0)` a dozen lines behind
  // every prefetch: the first builds of this kernel waited for each chunk in turn).  The offset is clamped into the
  // tensor instead: bytes in front of it are columns < 0 of its first row, never used; an aligned 16-byte chunk that holds
  // the tensor's last valid byte ends at most 12 bytes behind it, inside the same page; chunks wholly behind it are not used.
  const long last_chunk = (total_bytes - 1) & ~15L;
  auto load_chunk = [&](int y, int c) __attribute__((always_inline)) -> floatx4 {
    long off = (((img_pix + (long)y * W + wx0) * 12) & ~15L) + 16L * c;
    off = off < 0 ? 0 : (off > last_chunk ? last_chunk : off);
    return *reinterpret_cast<const floatx4 *>(base + off);
  };
  auto ring_slot = [&](int s) __attribute__((always_inline)) -> int { return s >= kFsRing ? s - kFsRing : s; };

  // the whole ring starts finite: slots of rows / columns outside the image are read (and dropped by the blend's select)
  for (int e = t; e < kFsRing * kFsCpr; e += kFsThreads) reinterpret_cast<floatx4 *>(ring)[e] = floatx4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  // prologue: the 33 rows of the first step, straight into ring slots 0..32
  for (int e = t; e < kFsLive * kFsCpr; e += kFsThreads) {
    const int r = e / kFsCpr, c = e - r * kFsCpr;
    const int y = i_begin - kFsMY + r;
    if (y >= 0 && y < H) reinterpret_cast<floatx4 *>(ring)[r * kFsCpr + c] = load_chunk(y, c);
  }
  // Two steps of prefetch, in registers: at the head of step s the 8 new source rows and the flow of step s + 2 are
  // requested; the rows of step s + 1 (requested a step earlier) go into the ring at the end of step s.  Register sets
  // rotate by compile-time index (nw[s & 1], fl[s & 3]; the loop is unrolled by four), so no copy ever waits on a load.
  floatx4 nw[2][kFsNew];
  float2 fl[4][kFsPPT];
  auto issue = [&](int i_of_step, floatx4 (&rows)[kFsNew], float2 (&flw)[kFsPPT]) __attribute__((always_inline)) {
    i_of_step = min(i_of_step, H - 1);                     // behind the band's last step: harmless repeats (no branch)
#pragma unroll
    for (int k = 0; k < kFsNew; ++k) {
      const int e = min(t + kFsThreads * k, kFsStep * kFsCpr - 1);   // (the lanes behind the last chunk repeat it: no branch)
      const int r = e / kFsCpr, c = e - r * kFsCpr;
      const int y = i_of_step + kFsMY + 1 + r;             // the 8 rows that step adds to the window of the one before it
      rows[k] = load_chunk(min(y, H - 1), c);              // rows behind the image: a copy of its last row, never committed
    }
    const int rn = i_of_step + rg * kFsPPT;
    const float2 *fn = reinterpret_cast<const float2 *>(p.flow) + (img_pix + (long)min(rn, H - 1) * W + jc);
#pragma unroll
    for (int r = 0; r < kFsPPT; ++r) flw[r] = fn[rn + r < H ? (long)r * W : 0];
  };
  {
    const int r_first = i_begin + rg * kFsPPT;
    const float2 *fptr = reinterpret_cast<const float2 *>(p.flow) + (img_pix + (long)min(r_first, H - 1) * W + jc);
#pragma unroll
    for (int r = 0; r < kFsPPT; ++r) fl[0][r] = fptr[r_first + r < H ? (long)r * W : 0];
  }
  issue(i_begin + kFsStep, nw[1], fl[1]);
  __syncthreads();

  int s0 = 0;   // ring slot of row i0 - kFsMY
  auto step = [&](int i0, floatx4 (&rows_issue)[kFsNew], float2 (&fl_issue)[kFsPPT], const floatx4 (&rows_commit)[kFsNew],
                  const float2 (&flc)[kFsPPT]) __attribute__((always_inline)) {
    issue(i0 + 2 * kFsStep, rows_issue, fl_issue);
    // ---- step s from the ring.  Branch-free main path: geometry of both rows, then all their LDS reads, then the blends.
    // A pixel with a tap outside the window reads the window's first pixel instead and is redone from global memory in
    // the rare branch behind.
    const int wy0 = i0 - kFsMY;                            // first window row of this step
    PadGeom g[kFsPPT];
    int ia[kFsPPT], ic[kFsPPT];
    bool inwin[kFsPPT];
#pragma unroll
    for (int r = 0; r < kFsPPT; ++r) {
      const int i = i0 + rg * kFsPPT + r;
      const float x = xf + flc[r].x;                       // :117-119
      const float y = (float)i + flc[r].y;
      g[r] = padded_geom(H, W, x, y);
      // The taps are image pixels (xl, yl), (xl + 1, yl), (xl, yl + 1), (xl + 1, yl + 1) with xl = x0 - 1 in [-1, W]; those
      // outside the image are on the zero ring: read from the window like the others (its slots there hold finite
      // bytes) and dropped by the blend's select.  All four inside the window?
      const int xl = g[r].x0 - 1, yl = g[r].y0 - 1;
      int dx = xl - wx0, dy = yl - wy0;
      inwin[r] = (unsigned)dx <= (unsigned)(kFsCols - 2) && (unsigned)dy <= (unsigned)(kFsLive - 2);
      dx = inwin[r] ? dx : 0;
      dy = inwin[r] ? dy : 0;
      const int yy = wy0 + dy;
      const int sa = ring_slot(s0 + dy), sb = ring_slot(s0 + dy + 1);
      ia[r] = sa * (kFsCpr * 4) + (((a4 + yy * w4) * 3) & 3) + dx * 3;
      ic[r] = sb * (kFsCpr * 4) + (((a4 + (yy + 1) * w4) * 3) & 3) + dx * 3;
    }
    TapsB<3> tp[kFsPPT];
#pragma unroll
    for (int r = 0; r < kFsPPT; ++r) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tp[r].a.v[c] = ring[ia[r] + c];
        tp[r].b.v[c] = ring[ia[r] + 3 + c];
        tp[r].c.v[c] = ring[ic[r] + c];
        tp[r].d.v[c] = ring[ic[r] + 3 + c];
      }
    }
#pragma unroll
    for (int r = 0; r < kFsPPT; ++r) {
      const int i = i0 + rg * kFsPPT + r;
      tp[r].w00 = g[r].w00; tp[r].w01 = g[r].w01; tp[r].w10 = g[r].w10; tp[r].w11 = g[r].w11;
      tp[r].v00 = g[r].v00; tp[r].v01 = g[r].v01; tp[r].v10 = g[r].v10; tp[r].v11 = g[r].v11;
      if (!inwin[r] && (g[r].v00 || g[r].v01 || g[r].v10 || g[r].v11) && col_ok && i < i_end) {
        // a flow beyond the margins: global gathers, as stn_kernel
        tp[r].a = load_pix<3>(img + ((long)g[r].ya * W + g[r].xa) * 3);
        tp[r].b = load_pix<3>(img + ((long)g[r].ya * W + g[r].xb) * 3);
        tp[r].c = load_pix<3>(img + ((long)g[r].yb * W + g[r].xa) * 3);
        tp[r].d = load_pix<3>(img + ((long)g[r].yb * W + g[r].xb) * 3);
      }
      float v[3];
      sample_padded_blend<3>(tp[r], 3, v);
      if (col_ok && i < i_end) store_pix<3>(p.out, (size_t)(img_pix + (long)i * W + j), 3, v);
    }
    // ---- the rows of step s + 1 (requested at the head of step s - 1) into the slots beside the live ones (last read in
    // step s - 1, behind that step's barrier)
    if (i0 + kFsStep < i_end) {
      const int snew = ring_slot(s0 + kFsLive);
#pragma unroll
      for (int k = 0; k < kFsNew; ++k) {
        const int e = t + kFsThreads * k;
        const int r = e / kFsCpr, c = e - r * kFsCpr;
        const int y = i0 + kFsStep + kFsMY + 1 + r;
        if (e < kFsStep * kFsCpr && y < H) reinterpret_cast<floatx4 *>(ring)[ring_slot(snew + r) * kFsCpr + c] = rows_commit[k];
      }
    }
    s0 = ring_slot(s0 + kFsStep);
    lds_barrier();
  };
  for (int i0 = i_begin; i0 < i_end; i0 += 4 * kFsStep) {   // (every thread of the workgroup takes the same exits)
    step(i0, nw[0], fl[2], nw[1], fl[0]);
    if (i0 + kFsStep >= i_end) break;
    step(i0 + kFsStep, nw[1], fl[3], nw[0], fl[1]);
    if (i0 + 2 * kFsStep >= i_end) break;
    step(i0 + 2 * kFsStep, nw[0], fl[0], nw[1], fl[2]);
    if (i0 + 3 * kFsStep >= i_end) break;
    step(i0 + 3 * kFsStep, nw[1], fl[1], nw[0], fl[3]);
  }
}

__global__ __launch_bounds__(kThreads) void scale_rgb_kernel(const float *__restrict__ in,
                                                            float *__restrict__ out, size_t npix,
                                                            int C) {
  // networks.py:6-16: out group g' = 2 - g, mean[g'] subtracted after the x255 scale.
  const int G = C / 3;
  const size_t total = npix * C;
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total;
       e += (size_t)gridDim.x * kThreads) {
    const int c = (int)(e % C);
    const size_t pix = e / C;
    const int g = c / G;            // output group
    const int src = (2 - g) * G + c % G;
    const float mean = g == 0 ? 103.939f : (g == 1 ? 116.779f : 123.68f);
    out[e] = in[pix * C + src] * 255.0f - mean;
  }
}

inline float lin_step(int n) { return n > 1 ? (1.0f - (-1.0f)) / (float)(n - 1) : 0.0f; }

int g_flow_tiled = 1;  // dvsg_debug_set_option("flow_tiled", v): 0 = stn_kernel<kFlow> (global gathers), 1 = column strips streamed
                       // through LDS, workgroups in XCD-aware order (default), 2 = the same in plain dispatch order
int g_flow_rounds = 4; // rounds of resident workgroups the strip kernel's bands aim at ("flow_rounds")
int g_warp_xcd = 0;   // dvsg_debug_set_option("warp_xcd", 0): the samplers' workgroups in plain dispatch order (A/B)

template <int SRC>
int launch_stn(StnParams p, int B, hipStream_t s, const char *what) {
  constexpr int PPT = (SRC == kFlow || SRC == kCoords) ? 4 : 2;  // memory-fed coordinates: more loads in flight per thread
  p.nbx = ceil_div(p.out_w, kThreads);
  p.nby = ceil_div(p.out_h, PPT);
  p.xcd = g_warp_xcd;
  DVSG_REQUIRE((long)p.nbx * p.nby * B < (1L << 31), "%s: grid of %ld workgroups out of range", what, (long)p.nbx * p.nby * B);
  dim3 grid((unsigned)(p.nbx * p.nby * B));
  // algorithmic bytes per output pixel: read C + write C floats (+ flow 8 B / coords 8 B)
  const double px = (double)B * p.out_h * p.out_w;
  ProfScope prof(kClsStn, s, 0.0,
                 px * (p.im ? 8.0 * p.Cn : 0.0) + px * ((SRC == kFlow || SRC == kCoords) ? 8.0 : 0.0) +
                     px * ((p.xs_out ? 4.0 : 0.0) + (p.ys_out ? 4.0 : 0.0)));
  if (p.Cn == 3)
    hipLaunchKernelGGL((stn_kernel<SRC, 3, PPT>), grid, dim3(kThreads), 0, s, p);
  else if (p.Cn == 1)
    hipLaunchKernelGGL((stn_kernel<SRC, 1, PPT>), grid, dim3(kThreads), 0, s, p);
  else
    hipLaunchKernelGGL((stn_kernel<SRC, 0, PPT>), grid, dim3(kThreads), 0, s, p);
  return check_launch(what);
}

int check_image_args(const char *fn, int B, int H, int W, int C, int out_h, int out_w) {
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && out_h > 0 && out_w > 0,
               "%s: sizes must be positive (B=%d H=%d W=%d C=%d out=%dx%d)", fn, B, H, W, C, out_h, out_w);
  DVSG_REQUIRE(C <= kMaxGenericC, "%s: C=%d exceeds the supported maximum %d", fn, C, kMaxGenericC);
  DVSG_REQUIRE(B <= 65535, "%s: B=%d exceeds the grid limit 65535", fn, B);
  DVSG_REQUIRE((long)H * W < (1L << 31) && (long)out_h * out_w < (1L << 31), "%s: image too large", fn);
  return DVSG_OK;
}

}  // namespace

void set_warp_xcd(int v) { g_warp_xcd = v != 0; }
void set_flow_tiled(int v) { g_flow_tiled = v; }
void set_flow_rounds(int v) { g_flow_rounds = v > 0 ? v : 1; }

// coord_bstride = 0 broadcasts one set of control points over the batch (model.py:111 tiles
// the constant V_src; the fused evaluation graph does not materialise the tile).
int tps_solve_impl(const float *coord, long coord_bstride, const float *rhs, int rhs_is_vector, int B,
                   int P, float *T, int *n_singular, void *stream) {
  DVSG_REQUIRE(coord && rhs && T, "dvsg_tps_solve_f32: NULL pointer");
  DVSG_REQUIRE(B > 0, "dvsg_tps_solve_f32: B=%d must be positive", B);
  DVSG_REQUIRE(P >= 3 && P <= kMaxPts, "dvsg_tps_solve_f32: P=%d outside [3,%d]", P, kMaxPts);
  ProfScope prof(kClsTpsSolve, as_stream(stream), 0.0, (double)B * (4.0 * P + 2.0 * (P + 3)) * 4.0);
  hipLaunchKernelGGL(tps_solve_kernel, dim3(B), dim3(64), 0, as_stream(stream), coord, coord_bstride, rhs,
                     rhs_is_vector, P, T, static_cast<double *>(nullptr), n_singular);
  return check_launch("tps_solve_kernel");
}

// Columns 0..P-1 of W^-1 for ONE set of control points, float64, winv_cols [P][P+3]; `scratch` holds
// ceil(P/2) x P x 2 floats of unit right-hand sides.  Synchronous helper for handle creation.
int tps_inverse_columns(const float *coord, int P, double *winv_cols, float *scratch, void *stream) {
  DVSG_REQUIRE(coord && winv_cols && scratch, "tps_inverse_columns: NULL pointer");
  DVSG_REQUIRE(P >= 3 && P <= kMaxPts, "tps_inverse_columns: P=%d outside [3,%d]", P, kMaxPts);
  const int pairs = (P + 1) / 2, n = P + 3;
  std::vector<float> eye((size_t)pairs * P * 2, 0.f);
  for (int j = 0; j < P; ++j) eye[((size_t)(j / 2) * P + j) * 2 + (j % 2)] = 1.f;  // sample j/2, rhs column j%2: e_j
  DVSG_HIP(hipMemcpyAsync(scratch, eye.data(), eye.size() * sizeof(float), hipMemcpyHostToDevice, as_stream(stream)));
  // sample b solves for columns 2b and 2b+1: Td[b][c][k] = (W^-1 e_{2b+c})[k], i.e. winv_cols[2b+c][k] -- contiguous
  double *dtmp = nullptr;   // [pairs][2][n] solutions, then one int: the singular-system count
  const size_t sol_bytes = (size_t)pairs * 2 * n * sizeof(double);
  DVSG_HIP(hipMalloc(&dtmp, sol_bytes + sizeof(int)));
  int *flag = reinterpret_cast<int *>(reinterpret_cast<char *>(dtmp) + sol_bytes);
  int n_singular = 0;
  hipError_t e = hipMemsetAsync(flag, 0, sizeof(int), as_stream(stream));
  if (e == hipSuccess) {
    hipLaunchKernelGGL(tps_solve_kernel, dim3(pairs), dim3(64), 0, as_stream(stream), coord, 0L, scratch, 0, P,
                       static_cast<float *>(nullptr), dtmp, flag);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = hipMemcpyAsync(winv_cols, dtmp, (size_t)P * n * sizeof(double), hipMemcpyDeviceToDevice, as_stream(stream));
  if (e == hipSuccess) e = hipMemcpyAsync(&n_singular, flag, sizeof(int), hipMemcpyDeviceToHost, as_stream(stream));
  if (e == hipSuccess) e = hipStreamSynchronize(as_stream(stream));
  (void)hipFree(dtmp);
  if (e != hipSuccess) return fail(DVSG_ERR_HIP, "tps_inverse_columns: %s", hipGetErrorString(e));
  if (n_singular) return fail(DVSG_ERR_INVALID_ARG, "TPS system of the control points is not invertible");
  return DVSG_OK;
}

int tps_apply_impl(const double *winv_cols, const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                   float *T, void *stream) {
  DVSG_REQUIRE(winv_cols && coord && rhs && T, "tps_apply: NULL pointer");
  DVSG_REQUIRE(B > 0 && P >= 3 && P <= kMaxPts, "tps_apply: bad shape B=%d P=%d", B, P);
  ProfScope prof(kClsTpsSolve, as_stream(stream), 0.0, (double)B * (4.0 * P + 2.0 * (P + 3)) * 4.0);
  hipLaunchKernelGGL(tps_apply_kernel, dim3(B), dim3(64), 0, as_stream(stream), winv_cols, coord, rhs, rhs_is_vector, P, T);
  return check_launch("tps_apply_kernel");
}

int tps_warp_impl(const float *U, const float *coord, long coord_bstride, const float *T, int B, int H,
                  int W, int C, int P, int out_h, int out_w, float *out, float *x_s, float *y_s,
                  void *stream) {
  DVSG_REQUIRE(coord && T, "dvsg_tps_warp_f32: NULL coord/T");
  DVSG_REQUIRE((U == nullptr) == (out == nullptr), "dvsg_tps_warp_f32: U and out must both be given or both NULL");
  DVSG_REQUIRE(U || x_s || y_s, "dvsg_tps_warp_f32: nothing to compute");
  DVSG_REQUIRE(P >= 1 && P <= kMaxPts, "dvsg_tps_warp_f32: P=%d outside [1,%d]", P, kMaxPts);
  if (int rc = check_image_args("dvsg_tps_warp_f32", B, U ? H : 1, U ? W : 1, U ? C : 1, out_h, out_w)) return rc;
  dim3 grid(ceil_div(out_w, kThreads), ceil_div(out_h, kTpsRows), B);
  const float sx = lin_step(out_w), sy = lin_step(out_h);
  hipStream_t s = as_stream(stream);
  // algorithmic bytes: read U once (C floats per input pixel) + write out (+ x_s, y_s when asked)
  ProfScope prof(kClsTpsWarp, s, 0.0,
                 (U ? 4.0 * C * ((double)B * H * W + (double)B * out_h * out_w) : 0.0) +
                     (double)B * out_h * out_w * ((x_s ? 4.0 : 0.0) + (y_s ? 4.0 : 0.0)));
  if (!U) C = 3;
  if (C == 3)
    hipLaunchKernelGGL((tps_warp_kernel<3, float>), grid, dim3(kThreads), 0, s, U, coord, coord_bstride, T, H, W,
                       C, P, out_h, out_w, sx, sy, out, x_s, y_s, (const int *)nullptr, 0, 0);
  else if (C == 1)
    hipLaunchKernelGGL((tps_warp_kernel<1, float>), grid, dim3(kThreads), 0, s, U, coord, coord_bstride, T, H, W,
                       C, P, out_h, out_w, sx, sy, out, x_s, y_s, (const int *)nullptr, 0, 0);
  else
    hipLaunchKernelGGL((tps_warp_kernel<0, float>), grid, dim3(kThreads), 0, s, U, coord, coord_bstride, T, H, W,
                       C, P, out_h, out_w, sx, sy, out, x_s, y_s, (const int *)nullptr, 0, 0);
  return check_launch("tps_warp_kernel");
}

// u_t = frame table[b * tstride] of a pool [n_pool,H,W,3] (float32, or uint8 with / 255. fused), same-size output
int tps_warp_ring_impl(const void *pool, int pool_is_u8, int n_pool, const int *table, int tstride, const float *coord,
                       long coord_bstride, const float *T, int B, int H, int W, int P, float *out, float *x_s, float *y_s,
                       void *stream) {
  DVSG_REQUIRE(pool && table && coord && T && out, "tps_warp_ring: NULL pointer");
  DVSG_REQUIRE(n_pool > 0 && tstride > 0 && P >= 1 && P <= kMaxPts, "tps_warp_ring: bad arguments");
  if (int rc = check_image_args("tps_warp_ring", B, H, W, 3, H, W)) return rc;
  dim3 grid(ceil_div(W, kThreads), ceil_div(H, kTpsRows), B);
  const float sx = lin_step(W), sy = lin_step(H);
  hipStream_t s = as_stream(stream);
  ProfScope prof(kClsTpsWarp, s, 0.0, (pool_is_u8 ? 3.0 : 12.0) * B * H * W + 12.0 * B * H * W +
                                          (double)B * H * W * ((x_s ? 4.0 : 0.0) + (y_s ? 4.0 : 0.0)));
  if (pool_is_u8)
    hipLaunchKernelGGL((tps_warp_kernel<3, uint8_t>), grid, dim3(kThreads), 0, s, static_cast<const uint8_t *>(pool), coord,
                       coord_bstride, T, H, W, 3, P, H, W, sx, sy, out, x_s, y_s, table, tstride, n_pool);
  else
    hipLaunchKernelGGL((tps_warp_kernel<3, float>), grid, dim3(kThreads), 0, s, static_cast<const float *>(pool), coord,
                       coord_bstride, T, H, W, 3, P, H, W, sx, sy, out, x_s, y_s, table, tstride, n_pool);
  return check_launch("tps_warp_kernel");
}

}  // namespace dvsg

using namespace dvsg;

extern "C" {

int dvsg_tps_solve_f32(const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                       float *T, void *stream) {
  return tps_solve_impl(coord, (long)P * 2, rhs, rhs_is_vector, B, P, T, nullptr, stream);
}

int dvsg_tps_solve_checked_f32(const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                               float *T, int *n_singular, void *stream) {
  DVSG_REQUIRE(n_singular, "dvsg_tps_solve_checked_f32: NULL n_singular");
  return tps_solve_impl(coord, (long)P * 2, rhs, rhs_is_vector, B, P, T, n_singular, stream);
}

int dvsg_tps_warp_f32(const float *U, const float *coord, const float *T, int B, int H, int W,
                      int C, int P, int out_h, int out_w, float *out, float *x_s, float *y_s,
                      void *stream) {
  return tps_warp_impl(U, coord, (long)P * 2, T, B, H, W, C, P, out_h, out_w, out, x_s, y_s, stream);
}

int dvsg_flow_warp_f32(const float *im, const float *flow, int B, int H, int W, int C, float *out,
                       void *stream) {
  DVSG_REQUIRE(im && flow && out, "dvsg_flow_warp_f32: NULL pointer");
  if (int rc = check_image_args("dvsg_flow_warp_f32", B, H, W, C, H, W)) return rc;
  if (C == 3 && g_flow_tiled && reinterpret_cast<uintptr_t>(im) % 16 == 0) {
    // RGB frames: the source rows of a column strip streamed through LDS (flow_warp_strip_kernel); same bits as stn_kernel<kFlow>
    FlowStripParams q{};
    q.im = im; q.flow = flow; q.out = out;
    q.H = H; q.W = W; q.B = B;
    q.nstrips = ceil_div(W, kFsW);
    // bands: enough workgroups for >= 4 rounds of the 256 resident ones, bands of >= 128 rows (a band primes 41 rows)
    const int steps = ceil_div(H, kFsStep);
    int bands = (int)std::min<long>(std::max<long>(1, ((long)g_flow_rounds * 256 + (long)q.nstrips * B - 1) / ((long)q.nstrips * B)), std::max(1, steps / 8));
    const int band_steps = ceil_div(steps, bands);
    q.band_rows = band_steps * kFsStep;
    q.nbands = ceil_div(H, q.band_rows);
    q.xcd = g_flow_tiled != 2;
    const long wgs = (long)q.nstrips * q.nbands * B;
    DVSG_REQUIRE(wgs < (1L << 31), "dvsg_flow_warp_f32: grid of %ld workgroups out of range", wgs);
    hipStream_t s = as_stream(stream);
    ProfScope prof(kClsStn, s, 0.0, 32.0 * B * H * W);
    hipLaunchKernelGGL(flow_warp_strip_kernel, dim3((unsigned)wgs), dim3(kFsThreads), 0, s, q);
    return check_launch("flow_warp_strip_kernel");
  }
  StnParams p{};
  p.im = im; p.a = flow; p.out = out;
  p.H = H; p.W = W; p.Cn = C; p.out_h = H; p.out_w = W;
  return launch_stn<kFlow>(p, B, as_stream(stream), "flow_warp_kernel");
}

int dvsg_stn_sample_f32(const float *im, const float *x_s, const float *y_s, int B, int H, int W,
                        int C, int out_h, int out_w, float *out, void *stream) {
  DVSG_REQUIRE(im && x_s && y_s && out, "dvsg_stn_sample_f32: NULL pointer");
  if (int rc = check_image_args("dvsg_stn_sample_f32", B, H, W, C, out_h, out_w)) return rc;
  StnParams p{};
  p.im = im; p.a = x_s; p.b = y_s; p.out = out;
  p.H = H; p.W = W; p.Cn = C; p.out_h = out_h; p.out_w = out_w;
  return launch_stn<kCoords>(p, B, as_stream(stream), "stn_sample_kernel");
}

static int grid_common(const char *fn, StnParams &p, const float *im, int B, int H, int W, int C,
                       int out_h, int out_w, float *out, float *x_s, float *y_s) {
  DVSG_REQUIRE((im == nullptr) == (out == nullptr), "%s: im and out must both be given or both NULL", fn);
  DVSG_REQUIRE(im || x_s || y_s, "%s: nothing to compute", fn);
  if (int rc = check_image_args(fn, B, im ? H : 1, im ? W : 1, im ? C : 1, out_h, out_w)) return rc;
  p.im = im; p.out = out; p.xs_out = x_s; p.ys_out = y_s;
  p.H = H; p.W = W; p.Cn = im ? C : 1; p.out_h = out_h; p.out_w = out_w;
  p.step_x = lin_step(out_w);
  p.step_y = lin_step(out_h);
  return DVSG_OK;
}

int dvsg_grid_affine_f32(const float *theta, const float *im, int B, int H, int W, int C, int out_h,
                         int out_w, float *out, float *x_s, float *y_s, void *stream) {
  DVSG_REQUIRE(theta, "dvsg_grid_affine_f32: NULL theta");
  StnParams p{};
  if (int rc = grid_common("dvsg_grid_affine_f32", p, im, B, H, W, C, out_h, out_w, out, x_s, y_s)) return rc;
  p.a = theta;
  return launch_stn<kAffine>(p, B, as_stream(stream), "stn_affine_kernel");
}

int dvsg_grid_projective_f32(const float *theta, const float *im, int B, int H, int W, int C,
                             int out_h, int out_w, float *out, float *x_s, float *y_s, void *stream) {
  DVSG_REQUIRE(theta, "dvsg_grid_projective_f32: NULL theta");
  StnParams p{};
  if (int rc = grid_common("dvsg_grid_projective_f32", p, im, B, H, W, C, out_h, out_w, out, x_s, y_s)) return rc;
  p.a = theta;
  return launch_stn<kProjective>(p, B, as_stream(stream), "stn_projective_kernel");
}

int dvsg_random_mask_plane_f32(const float *theta, int B, int H, int W, float *mask, void *stream) {
  DVSG_REQUIRE(theta && mask, "dvsg_random_mask_plane_f32: NULL pointer");
  if (int rc = check_image_args("dvsg_random_mask_plane_f32", B, H, W, 1, H, W)) return rc;
  DVSG_REQUIRE(H <= 65535, "dvsg_random_mask_plane_f32: H=%d exceeds the grid limit 65535", H);
  hipStream_t s = as_stream(stream);
  ProfScope prof(kClsStn, s, 0.0, 4.0 * B * H * W);
  hipLaunchKernelGGL(mask_plane_kernel, dim3(ceil_div(W, kThreads), H, B), dim3(kThreads), 0, s, theta, H, W, lin_step(W),
                     lin_step(H), mask);
  return check_launch("mask_plane_kernel");
}

int dvsg_grid_elastic_f32(const float *theta_abs, const float *L_inv, const float *source_points,
                          int n, const float *im, int B, int H, int W, int C, int out_h, int out_w,
                          float *out, float *x_s, float *y_s, void *stream) {
  DVSG_REQUIRE(theta_abs && L_inv && source_points, "dvsg_grid_elastic_f32: NULL pointer");
  DVSG_REQUIRE(n >= 1 && n <= kMaxPts, "dvsg_grid_elastic_f32: n=%d outside [1,%d]", n, kMaxPts);
  StnParams p{};
  if (int rc = grid_common("dvsg_grid_elastic_f32", p, im, B, H, W, C, out_h, out_w, out, x_s, y_s)) return rc;
  p.a = theta_abs; p.b = L_inv; p.c = source_points; p.n = n;
  return launch_stn<kElastic>(p, B, as_stream(stream), "stn_elastic_kernel");
}

int dvsg_elastic_constants_f32(int grid_size, float *source_points_host, float *L_inv_host) {
  DVSG_REQUIRE(source_points_host && L_inv_host, "dvsg_elastic_constants_f32: NULL pointer");
  const int n = grid_size * grid_size;
  DVSG_REQUIRE(grid_size >= 2 && n <= kMaxPts, "dvsg_elastic_constants_f32: grid_size=%d unsupported", grid_size);
  const int m = n + 3;
  // source points: meshgrid(linspace(-1,1,g), linspace(-1,1,g)), x fastest (:313-322)
  const float step = lin_step(grid_size);
  for (int i = 0; i < grid_size; ++i)
    for (int j = 0; j < grid_size; ++j) {
      source_points_host[i * grid_size + j] = -1.0f + step * (float)j;
      source_points_host[n + i * grid_size + j] = -1.0f + step * (float)i;
    }
  const float *sx = source_points_host, *sy = source_points_host + n;
  // L (:339-346), built in float32 like the reference, inverted in float64.
  static thread_local double L[64 * 64], Li[64 * 64];
  for (int i = 0; i < m * m; ++i) L[i] = 0.0;
  for (int j = 0; j < n; ++j) {
    L[0 * m + 3 + j] = sx[j];
    L[1 * m + 3 + j] = sy[j];
    L[2 * m + 3 + j] = 1.0;
  }
  L[2 * m + 2] = 1.0;  // L_mid = [0, 0, 1, 1...1]
  for (int i = 0; i < n; ++i) {
    double *row = L + (size_t)(3 + i) * m;
    row[0] = sx[i];
    row[1] = sy[i];
    row[2] = 1.0;
    for (int j = 0; j < n; ++j) {
      const float dx = sx[i] - sx[j], dy = sy[i] - sy[j];
      const float rsq = dx * dx + dy * dy;
      row[3 + j] = rsq == 0.f ? 0.0 : (double)(rsq * logf(rsq));
    }
  }
  for (int i = 0; i < m; ++i)
    for (int j = 0; j < m; ++j) Li[i * m + j] = i == j ? 1.0 : 0.0;
  for (int k = 0; k < m; ++k) {
    int piv = k;
    for (int i = k + 1; i < m; ++i)
      if (fabs(L[i * m + k]) > fabs(L[piv * m + k])) piv = i;
    if (L[piv * m + k] == 0.0) return fail(DVSG_ERR_INVALID_ARG, "dvsg_elastic_constants_f32: singular L");
    if (piv != k)
      for (int c = 0; c < m; ++c) {
        double tmp = L[k * m + c]; L[k * m + c] = L[piv * m + c]; L[piv * m + c] = tmp;
        tmp = Li[k * m + c]; Li[k * m + c] = Li[piv * m + c]; Li[piv * m + c] = tmp;
      }
    const double d = L[k * m + k];
    for (int c = 0; c < m; ++c) {
      L[k * m + c] /= d;
      Li[k * m + c] /= d;
    }
    for (int i = 0; i < m; ++i) {
      if (i == k) continue;
      const double f = L[i * m + k];
      if (f == 0.0) continue;
      for (int c = 0; c < m; ++c) {
        L[i * m + c] -= f * L[k * m + c];
        Li[i * m + c] -= f * Li[k * m + c];
      }
    }
  }
  // L_inv = transpose(inverse(L)[:, 3:]) -> [n, m] (:358)
  for (int j = 0; j < n; ++j)
    for (int i = 0; i < m; ++i) L_inv_host[(size_t)j * m + i] = (float)Li[(size_t)i * m + 3 + j];
  return DVSG_OK;
}

int dvsg_scale_rgb_f32(const float *rgb, int B, int H, int W, int C, float *out, void *stream) {
  DVSG_REQUIRE(rgb && out, "dvsg_scale_rgb_f32: NULL pointer");
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0 && C > 0 && C % 3 == 0,
               "dvsg_scale_rgb_f32: bad shape B=%d H=%d W=%d C=%d (C must be a positive multiple of 3)", B, H, W, C);
  const size_t npix = (size_t)B * H * W;
  const int blocks = (int)((npix * C + kThreads - 1) / kThreads < 8192 ? (npix * C + kThreads - 1) / kThreads : 8192);
  hipLaunchKernelGGL(scale_rgb_kernel, dim3(blocks), dim3(kThreads), 0, as_stream(stream), rgb, out, npix, C);
  return check_launch("scale_rgb_kernel");
}

}  // extern "C"
