// Shared host-side helpers for libdvsg_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../../include/dvsg_amd.h"

namespace dvsg {

// Thread-local last-error text returned by dvsg_last_error_string().
char *error_buffer();
int fail(int status, const char *fmt, ...) __attribute__((format(printf, 2, 3)));

inline hipStream_t as_stream(void *s) { return reinterpret_cast<hipStream_t>(s); }

// Check the launch itself (invalid configuration etc.); execution errors surface at the
// caller's next synchronisation, as for any stream-ordered API.
inline int check_launch(const char *what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(DVSG_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
  return DVSG_OK;
}

#define DVSG_REQUIRE(cond, ...)                                  \
  do {                                                           \
    if (!(cond)) return ::dvsg::fail(DVSG_ERR_INVALID_ARG, __VA_ARGS__); \
  } while (0)

#define DVSG_HIP(call)                                                                     \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return ::dvsg::fail(DVSG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));    \
  } while (0)

// warp_kernels.hip internals shared with locnet.hip (coord_bstride in floats; 0 = broadcast)
int tps_solve_impl(const float *coord, long coord_bstride, const float *rhs, int rhs_is_vector, int B,
                   int P, float *T, int *n_singular, void *stream);
// constant control points (the evaluation graph's V_src): W^-1 once, then a matrix-vector product per frame
int tps_inverse_columns(const float *coord, int P, double *winv_cols, float *scratch, void *stream);
int tps_apply_impl(const double *winv_cols, const float *coord, const float *rhs, int rhs_is_vector, int B, int P,
                   float *T, void *stream);
int tps_warp_impl(const float *U, const float *coord, long coord_bstride, const float *T, int B, int H,
                  int W, int C, int P, int out_h, int out_w, float *out, float *x_s, float *y_s,
                  void *stream);

int tps_warp_ring_impl(const void *pool, int pool_is_u8, int n_pool, const int *table, int tstride, const float *coord,
                       long coord_bstride, const float *T, int B, int H, int W, int P, float *out, float *x_s, float *y_s,
                       void *stream);

void set_flow_tiled(int v); // diagnostic (dvsg_debug_set_option "flow_tiled"): 0 = tf_warp by global gathers (stn_kernel<kFlow>)
void set_flow_rounds(int v);
void set_warp_xcd(int v);   // diagnostic (dvsg_debug_set_option "warp_xcd"): XCD-aware workgroup order of the sampler kernels

inline int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

}  // namespace dvsg

// ---- optional roctx ranges (tracing) ------------------------------------------------
// With DVSG_ROCTX=1 in the environment every stage of the evaluation graph (conv1, pool1, each bottleneck unit, head,
// tps_solve, tps_warp) opens a roctx range around its launches, so `rocprofv3 --kernel-trace --marker-trace` attributes
// the kernels of a step to stages and units instead of to kernel names only.  The roctx library is looked up with dlopen
// at the first range (no link-time dependency); without the variable a range is one relaxed load.
namespace dvsg {
struct MarkerRange {
  explicit MarkerRange(const char *name);
  ~MarkerRange();
  MarkerRange(const MarkerRange &) = delete;
  MarkerRange &operator=(const MarkerRange &) = delete;
  bool open_;
};
}  // namespace dvsg

// ---- optional per-kernel-class timing (bench.py roofline leg; see dvsg_prof_begin) ----------
namespace dvsg {
enum KernelClass {
  kClsConv1 = 0,   // conv1_kernel (7x7/2 + scale_RGB)
  kClsConv3x3 = 1, // conv_gemm_kernel<*,3,..>
  kClsConv1x1 = 2, // conv_gemm_kernel<*,1,..>
  kClsMaxpool = 3,
  kClsHead = 4,    // avgpool + dense
  kClsTpsSolve = 5,
  kClsTpsWarp = 6,
  kClsStn = 7,     // flow / sampler B / affine / projective / elastic
  kClsFused = 8,   // conv3x3_1x1_kernel (block 1's conv2 + conv3)
  kNumCls = 9
};
// RAII: when profiling is armed for `cls`, brackets the launches issued in its lifetime with a
// hipEvent pair on `s` and books their algorithmic FLOPs / bytes.  Otherwise a no-op.
struct ProfScope {
  ProfScope(int cls, hipStream_t s, double flops, double bytes);
  ~ProfScope();
  int idx_;
  hipStream_t s_;
  hipEvent_t stop_ = nullptr;
};
}  // namespace dvsg
