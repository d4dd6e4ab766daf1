// Data formats either side of the hot path (eval.py:76-124): the driver of
// the reference reads uint8 video frames, keeps them as float64 / 255 in a host list, builds each
// 21-channel window with np.concatenate, and writes uint8 frames back.  Here the frames live in
// HBM as float32 and these HBM-bound kernels do the conversions and the window assembly:
//
//   frames_u8_to_f32   eval.py:79-80   uint8 -> (float)(double(u) / 255.0), optional BGR<->RGB flip
//   window_gather      eval.py:103-104 patches[b,y,x,3s+c] = pool[idx[b,s],y,x,c]
//   frames_f32_to_u8   eval.py:112-113 np.uint8(x * 255.) (float64 product, truncation), written
//                                      into a column band of a wider image (side-by-side layout)
//   resize_bilinear    eval.py:80      cv2.resize(frame / 255., (w, h)), INTER_LINEAR on float64
//
// All of them move each byte once: the roofline is HBM bandwidth.
#include <cstdint>

#include "common.h"

namespace dvsg {
namespace {

constexpr int kThreads = 256;

inline int grid_for(size_t items, int cap = 1 << 16) {
  const size_t b = (items + kThreads - 1) / kThreads;
  return (int)(b < (size_t)cap ? (b ? b : 1) : (size_t)cap);
}

// One thread converts 4 consecutive pixels (12 bytes in, 48 bytes out).
__global__ __launch_bounds__(kThreads) void frames_u8_to_f32_kernel(const uint8_t *__restrict__ src,
                                                                   float *__restrict__ dst, size_t npix,
                                                                   int flip) {
  const size_t ngroups = (npix + 3) / 4;
  for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kThreads) {
    const size_t p0 = 4 * g;
    if (p0 + 4 <= npix && (reinterpret_cast<uintptr_t>(src) & 3) == 0) {
      const uint32_t *s4 = reinterpret_cast<const uint32_t *>(src + 3 * p0);
      const uint32_t w[3] = {s4[0], s4[1], s4[2]};
      float v[12];
#pragma unroll
      for (int i = 0; i < 12; ++i) v[i] = (float)((double)((w[i >> 2] >> (8 * (i & 3))) & 255u) / 255.0);
      if (flip) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float t = v[3 * q];
          v[3 * q] = v[3 * q + 2];
          v[3 * q + 2] = t;
        }
      }
      float4 *d4 = reinterpret_cast<float4 *>(dst + 3 * p0);
      d4[0] = make_float4(v[0], v[1], v[2], v[3]);
      d4[1] = make_float4(v[4], v[5], v[6], v[7]);
      d4[2] = make_float4(v[8], v[9], v[10], v[11]);
    } else {
      for (size_t p = p0; p < npix && p < p0 + 4; ++p)
        for (int c = 0; c < 3; ++c) dst[3 * p + c] = (float)((double)src[3 * p + (flip ? 2 - c : c)] / 255.0);
    }
  }
}

// One thread produces 4 consecutive floats of the [B, HW, C] window tensor (one 16-byte store);
// its sources are <= 4 runs of the pool frames.  An index outside the pool yields zeros.
__global__ __launch_bounds__(kThreads) void window_gather_kernel(const float *__restrict__ pool, int n_pool,
                                                                size_t hw, const int *__restrict__ idx, int B,
                                                                int S, float *__restrict__ patches) {
  const int C = 3 * S;
  const size_t per_window = hw * C;
  const size_t total = per_window * B;
  const size_t ngroups = (total + 3) / 4;
  for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kThreads) {
    const size_t e0 = 4 * g;
    int b = (int)(e0 / per_window);
    const size_t rem = e0 - (size_t)b * per_window;
    size_t pix = rem / C;
    int c = (int)(rem - pix * C);
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      v[i] = 0.f;
      if (e0 + i < total) {
        const int s = c / 3;
        const int f = idx[b * S + s];
        if (f >= 0 && f < n_pool) v[i] = pool[((size_t)f * hw + pix) * 3 + (c - 3 * s)];
      }
      if (++c == C) {
        c = 0;
        if (++pix == hw) {
          pix = 0;
          ++b;
        }
      }
    }
    if (e0 + 4 <= total) {
      *reinterpret_cast<float4 *>(patches + e0) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      for (int i = 0; i < 4 && e0 + i < total; ++i) patches[e0 + i] = v[i];
    }
  }
}

// np.uint8(x * 255.) of eval.py:112: the product is float64 and the cast truncates toward zero.
// (NumPy leaves out-of-range casts undefined; here they saturate to 0 / 255, NaN gives 0.)
__device__ __forceinline__ uint8_t to_u8(double x) {
  const double d = x * 255.0;
  return d >= 255.0 ? (uint8_t)255 : (d > 0.0 ? (uint8_t)(int)d : (uint8_t)0);
}

// One thread converts 4 consecutive values of a row (row = W*3 values); rows land at
// dst + (n*H + y) * dst_row_bytes + dst_col_bytes.
template <typename T>
__global__ __launch_bounds__(kThreads) void frames_to_u8_kernel(const T *__restrict__ src, uint8_t *__restrict__ dst,
                                                               size_t rows, int row_vals, size_t dst_row_bytes,
                                                               size_t dst_col_bytes, int flip) {
  const int groups_per_row = (row_vals + 3) / 4;
  const size_t ngroups = rows * groups_per_row;
  for (size_t g = (size_t)blockIdx.x * kThreads + threadIdx.x; g < ngroups; g += (size_t)gridDim.x * kThreads) {
    const size_t row = g / groups_per_row;
    const int v0 = 4 * (int)(g - row * groups_per_row);
    const T *s = src + row * row_vals;
    uint8_t *d = dst + row * dst_row_bytes + dst_col_bytes;
    uint8_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int v = v0 + i;
      if (v < row_vals) {
        const int px = v / 3, c = v - 3 * px;
        o[i] = to_u8((double)s[flip ? 3 * px + 2 - c : v]);
      } else {
        o[i] = 0;
      }
    }
    if (v0 + 4 <= row_vals && ((reinterpret_cast<uintptr_t>(d) + v0) & 3) == 0) {
      *reinterpret_cast<uint32_t *>(d + v0) = (uint32_t)o[0] | ((uint32_t)o[1] << 8) | ((uint32_t)o[2] << 16) | ((uint32_t)o[3] << 24);
    } else {
      for (int i = 0; i < 4 && v0 + i < row_vals; ++i) d[v0 + i] = o[i];
    }
  }
}

// cv2.resize(src_float64, (out_w, out_h)) with the default INTER_LINEAR, restated from OpenCV's
// published resize algorithm (OpenCV is not part of the reference tree nor of this image: parity
// unpinned): pixel centres at (d + 0.5) * scale - 0.5; the fractional weight is computed AND kept
// as float32; taps left of 0 / right of the last column clamp with weight 0; the row pass runs
// first (float64 accumulate), then the column pass.  Input is the uint8 frame (eval.py:80 divides by
// 255. in float64 first), output the float32 TF is fed.
struct ResizeTap {
  int s0, s1;
  float w1;
};
__device__ __forceinline__ ResizeTap resize_tap(int d, double scale, int n_src) {
  float f = (float)(((double)d + 0.5) * scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (s < 0) {
    s = 0;
    f = 0.f;
  }
  if (s >= n_src - 1) {
    s = n_src - 1;
    f = 0.f;
  }
  ResizeTap t;
  t.s0 = s;
  t.s1 = s + 1 < n_src ? s + 1 : n_src - 1;
  t.w1 = f;
  return t;
}

__global__ __launch_bounds__(kThreads) void resize_u8_kernel(const uint8_t *__restrict__ src, int n, int sh, int sw,
                                                            float *__restrict__ dst, int dh, int dw, int flip,
                                                            uint8_t *__restrict__ u8_dst, int u8_w, int u8_x0) {
  const double scale_x = (double)sw / dw, scale_y = (double)sh / dh;
  const size_t total = (size_t)n * dh * dw;
  for (size_t e = (size_t)blockIdx.x * kThreads + threadIdx.x; e < total; e += (size_t)gridDim.x * kThreads) {
    const int dx = (int)(e % dw);
    const size_t t = e / dw;
    const int dy = (int)(t % dh);
    const size_t f = t / dh;
    const ResizeTap tx = resize_tap(dx, scale_x, sw), ty = resize_tap(dy, scale_y, sh);
    const uint8_t *r0 = src + ((f * sh + ty.s0) * sw) * 3;
    const uint8_t *r1 = src + ((f * sh + ty.s1) * sw) * 3;
    const double a1 = (double)tx.w1, a0 = (double)(1.f - tx.w1);
    const double b1 = (double)ty.w1, b0 = (double)(1.f - ty.w1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int cs = flip ? 2 - c : c;
      const double p00 = (double)r0[3 * tx.s0 + cs] / 255.0, p01 = (double)r0[3 * tx.s1 + cs] / 255.0;
      const double p10 = (double)r1[3 * tx.s0 + cs] / 255.0, p11 = (double)r1[3 * tx.s1 + cs] / 255.0;
      const double h0 = __dadd_rn(__dmul_rn(p00, a0), __dmul_rn(p01, a1));
      const double h1 = __dadd_rn(__dmul_rn(p10, a0), __dmul_rn(p11, a1));
      const double v = __dadd_rn(__dmul_rn(h0, b0), __dmul_rn(h1, b1));
      dst[e * 3 + c] = (float)v;
      // the unstable half of the output video (eval.py:112-113) is rendered from the float64 value
      if (u8_dst) u8_dst[((t * u8_w) + u8_x0 + dx) * 3 + cs] = to_u8(v);
    }
  }
}

template <typename T>
int frames_to_u8(const char *what, const T *src, int n, int H, int W, int channel_flip, uint8_t *dst, int dst_W, int dst_x0,
                 void *stream) {
  DVSG_REQUIRE(src && dst, "%s: NULL pointer", what);
  DVSG_REQUIRE(n > 0 && H > 0 && W > 0, "%s: bad shape n=%d H=%d W=%d", what, n, H, W);
  DVSG_REQUIRE(dst_x0 >= 0 && dst_W >= dst_x0 + W, "%s: columns [%d, %d) do not fit a row of %d pixels", what, dst_x0,
               dst_x0 + W, dst_W);
  const size_t rows = (size_t)n * H;
  const size_t groups = rows * ((3 * (size_t)W + 3) / 4);
  hipLaunchKernelGGL(frames_to_u8_kernel<T>, dim3(grid_for(groups)), dim3(kThreads), 0, as_stream(stream), src, dst, rows,
                     3 * W, (size_t)3 * dst_W, (size_t)3 * dst_x0, channel_flip);
  return check_launch("frames_to_u8_kernel");
}

}  // namespace
}  // namespace dvsg

using namespace dvsg;

extern "C" {

int dvsg_frames_u8_to_f32(const uint8_t *src, size_t n_pixels, int channel_flip, float *dst, void *stream) {
  DVSG_REQUIRE(src && dst, "dvsg_frames_u8_to_f32: NULL pointer");
  DVSG_REQUIRE(n_pixels > 0, "dvsg_frames_u8_to_f32: no pixels");
  DVSG_REQUIRE((reinterpret_cast<uintptr_t>(dst) & 15) == 0, "dvsg_frames_u8_to_f32: dst must be 16-byte aligned");
  hipLaunchKernelGGL(frames_u8_to_f32_kernel, dim3(grid_for((n_pixels + 3) / 4)), dim3(kThreads), 0, as_stream(stream),
                     src, dst, n_pixels, channel_flip);
  return check_launch("frames_u8_to_f32_kernel");
}

int dvsg_window_gather_f32(const float *pool, int n_pool, int H, int W, const int32_t *idx, int B, int S,
                           float *patches, void *stream) {
  DVSG_REQUIRE(pool && idx && patches, "dvsg_window_gather_f32: NULL pointer");
  DVSG_REQUIRE(n_pool > 0 && H > 0 && W > 0 && B > 0 && S > 0, "dvsg_window_gather_f32: bad shape n_pool=%d H=%d W=%d B=%d S=%d",
               n_pool, H, W, B, S);
  DVSG_REQUIRE((reinterpret_cast<uintptr_t>(patches) & 15) == 0, "dvsg_window_gather_f32: patches must be 16-byte aligned");
  const size_t hw = (size_t)H * W;
  const size_t groups = (hw * 3 * S * B + 3) / 4;
  hipLaunchKernelGGL(window_gather_kernel, dim3(grid_for(groups)), dim3(kThreads), 0, as_stream(stream), pool, n_pool, hw,
                     idx, B, S, patches);
  return check_launch("window_gather_kernel");
}

int dvsg_frames_f32_to_u8(const float *src, int n, int H, int W, int channel_flip, uint8_t *dst, int dst_W,
                          int dst_x0, void *stream) {
  return frames_to_u8("dvsg_frames_f32_to_u8", src, n, H, W, channel_flip, dst, dst_W, dst_x0, stream);
}

int dvsg_frames_f64_to_u8(const double *src, int n, int H, int W, int channel_flip, uint8_t *dst, int dst_W,
                          int dst_x0, void *stream) {
  return frames_to_u8("dvsg_frames_f64_to_u8", src, n, H, W, channel_flip, dst, dst_W, dst_x0, stream);
}

int dvsg_frames_resize_u8_f32(const uint8_t *src, int n, int src_H, int src_W, int channel_flip, float *dst,
                              int dst_H, int dst_W, uint8_t *u8_dst, int u8_W, int u8_x0, void *stream) {
  DVSG_REQUIRE(src && dst, "dvsg_frames_resize_u8_f32: NULL pointer");
  DVSG_REQUIRE(n > 0 && src_H > 0 && src_W > 0 && dst_H > 0 && dst_W > 0,
               "dvsg_frames_resize_u8_f32: bad shape n=%d src=%dx%d dst=%dx%d", n, src_H, src_W, dst_H, dst_W);
  DVSG_REQUIRE(!u8_dst || (u8_x0 >= 0 && u8_W >= u8_x0 + dst_W),
               "dvsg_frames_resize_u8_f32: columns [%d, %d) do not fit a row of %d pixels", u8_x0, u8_x0 + dst_W, u8_W);
  hipLaunchKernelGGL(resize_u8_kernel, dim3(grid_for((size_t)n * dst_H * dst_W)), dim3(kThreads), 0, as_stream(stream), src,
                     n, src_H, src_W, dst, dst_H, dst_W, channel_flip, u8_dst, u8_W, u8_x0);
  return check_launch("resize_u8_kernel");
}

}  // extern "C"
