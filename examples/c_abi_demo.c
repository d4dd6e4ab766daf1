/* The C ABI of libdvsg_amd.so from plain C: no Python, no torch, no C++ -- device buffers from the HIP runtime,
 * raw pointers and sizes across the boundary, int status codes back (include/dvsg_amd.h).
 *
 * Runs the known-answer tests of SURVEY.md 8c that need no checkpoint:
 *   1. tf_warp(im, 0) == im exactly                      (warp_with_optical_flow.py:135-171)
 *   2. tf_warp with an integer flow == the shifted image, zero beyond the 1-pixel ring
 *   3. TPS with vector = 0: T's affine part is the identity and the source grid is the target grid
 *                                                         (ThinPlateSpline.py:143-166, :92-141)
 *   4. a NULL tensor is refused with a negative status and a message, nothing is launched
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_demo.c \
 *       -Lcoupe/dvsg_amd -ldvsg_amd -L/opt/rocm/lib -lamdhip64 -lm -Wl,-rpath,$PWD/coupe/dvsg_amd -o build/c_abi_demo
 */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dvsg_amd.h"

#define HIP_OK(call)                                                                   \
  do {                                                                                 \
    hipError_t e_ = (call);                                                            \
    if (e_ != hipSuccess) {                                                            \
      fprintf(stderr, "%s failed: %s\n", #call, hipGetErrorString(e_));                \
      return 2;                                                                        \
    }                                                                                  \
  } while (0)
#define DVSG_OK_OR_FAIL(call)                                                          \
  do {                                                                                 \
    int rc_ = (call);                                                                  \
    if (rc_ != 0) {                                                                    \
      fprintf(stderr, "%s -> %d: %s\n", #call, rc_, dvsg_last_error_string());         \
      return 3;                                                                        \
    }                                                                                  \
  } while (0)

enum { B = 2, H = 37, W = 53, C = 3, P = 25 };

int main(void) {
  const size_t npix = (size_t)B * H * W, nim = npix * C;
  float *im = (float *)malloc(nim * sizeof(float)), *out = (float *)malloc(nim * sizeof(float));
  float *flow = (float *)calloc(npix * 2, sizeof(float));
  float *d_im, *d_flow, *d_out, *d_coord, *d_vec, *d_T, *d_xs, *d_ys;
  float coord[B * P * 2], vec[B * P * 2], T[B * 2 * (P + 3)];
  float *xs = (float *)malloc(npix * sizeof(float)), *ys = (float *)malloc(npix * sizeof(float));
  size_t i;
  int b, y, x, c, k, failures = 0;
  unsigned s = 12345u;

  printf("libdvsg_amd ABI %d, code object %s\n", dvsg_abi_version(), dvsg_target_arch());
  for (i = 0; i < nim; ++i) {
    s = s * 1664525u + 1013904223u;
    im[i] = (float)(s >> 8) / 16777216.0f;
  }
  HIP_OK(hipMalloc((void **)&d_im, nim * sizeof(float)));
  HIP_OK(hipMalloc((void **)&d_out, nim * sizeof(float)));
  HIP_OK(hipMalloc((void **)&d_flow, npix * 2 * sizeof(float)));
  HIP_OK(hipMemcpy(d_im, im, nim * sizeof(float), hipMemcpyHostToDevice));

  /* 1. zero flow */
  HIP_OK(hipMemcpy(d_flow, flow, npix * 2 * sizeof(float), hipMemcpyHostToDevice));
  DVSG_OK_OR_FAIL(dvsg_flow_warp_f32(d_im, d_flow, B, H, W, C, d_out, NULL));
  HIP_OK(hipMemcpy(out, d_out, nim * sizeof(float), hipMemcpyDeviceToHost)); /* synchronises with the NULL stream */
  if (memcmp(out, im, nim * sizeof(float)) != 0) {
    printf("FAIL 1: tf_warp(im, 0) != im\n");
    ++failures;
  }

  /* 2. integer flow (dx, dy) = (2, -1): out[y][x] = im[y - 1][x + 2], 0 outside the image */
  for (i = 0; i < npix; ++i) {
    flow[2 * i] = 2.0f;
    flow[2 * i + 1] = -1.0f;
  }
  HIP_OK(hipMemcpy(d_flow, flow, npix * 2 * sizeof(float), hipMemcpyHostToDevice));
  DVSG_OK_OR_FAIL(dvsg_flow_warp_f32(d_im, d_flow, B, H, W, C, d_out, NULL));
  HIP_OK(hipMemcpy(out, d_out, nim * sizeof(float), hipMemcpyDeviceToHost));
  for (b = 0; b < B; ++b)
    for (y = 0; y < H; ++y)
      for (x = 0; x < W; ++x)
        for (c = 0; c < C; ++c) {
          const int sy = y - 1, sx = x + 2;
          const float want = (sy >= 0 && sy < H && sx >= 0 && sx < W) ? im[(((size_t)b * H + sy) * W + sx) * C + c] : 0.0f;
          if (out[(((size_t)b * H + y) * W + x) * C + c] != want) ++failures;
        }
  if (failures) printf("FAIL 2: integer-flow shift (%d mismatches)\n", failures);

  /* 3. TPS, 5 x 5 control grid (model.py:105-110), zero displacement */
  for (b = 0; b < B; ++b)
    for (k = 0; k < P; ++k) {
      coord[(b * P + k) * 2] = -1.0f + 0.5f * (float)(k % 5);
      coord[(b * P + k) * 2 + 1] = -1.0f + 0.5f * (float)(k / 5);
      vec[(b * P + k) * 2] = vec[(b * P + k) * 2 + 1] = 0.0f;
    }
  HIP_OK(hipMalloc((void **)&d_coord, sizeof(coord)));
  HIP_OK(hipMalloc((void **)&d_vec, sizeof(vec)));
  HIP_OK(hipMalloc((void **)&d_T, sizeof(T)));
  HIP_OK(hipMalloc((void **)&d_xs, npix * sizeof(float)));
  HIP_OK(hipMalloc((void **)&d_ys, npix * sizeof(float)));
  HIP_OK(hipMemcpy(d_coord, coord, sizeof(coord), hipMemcpyHostToDevice));
  HIP_OK(hipMemcpy(d_vec, vec, sizeof(vec), hipMemcpyHostToDevice));
  DVSG_OK_OR_FAIL(dvsg_tps_solve_f32(d_coord, d_vec, 1, B, P, d_T, NULL));
  DVSG_OK_OR_FAIL(dvsg_tps_warp_f32(NULL, d_coord, d_T, B, H, W, C, P, H, W, NULL, d_xs, d_ys, NULL)); /* grid only */
  HIP_OK(hipMemcpy(T, d_T, sizeof(T), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(xs, d_xs, npix * sizeof(float), hipMemcpyDeviceToHost));
  HIP_OK(hipMemcpy(ys, d_ys, npix * sizeof(float), hipMemcpyDeviceToHost));
  {
    const float want[6] = {0.f, 1.f, 0.f, 0.f, 0.f, 1.f}; /* rows of T: [1, x, y | r_k ...] -> x_s = x_t, y_s = y_t */
    float worst = 0.f, gworst = 0.f;
    for (b = 0; b < B; ++b)
      for (k = 0; k < 2 * (P + 3); ++k) {
        const int row = k / (P + 3), col = k % (P + 3);
        const float w = col < 3 ? want[row * 3 + col] : 0.f;
        const float e = fabsf(T[b * 2 * (P + 3) + k] - w);
        if (e > worst) worst = e;
      }
    for (b = 0; b < B; ++b)
      for (y = 0; y < H; ++y)
        for (x = 0; x < W; ++x) {
          const float xt = -1.0f + 2.0f * (float)x / (float)(W - 1), yt = -1.0f + 2.0f * (float)y / (float)(H - 1);
          const size_t p = ((size_t)b * H + y) * W + x;
          const float e = fmaxf(fabsf(xs[p] - xt), fabsf(ys[p] - yt));
          if (e > gworst) gworst = e;
        }
    printf("TPS identity: max |T - [0 1 0; 0 0 1 | 0]| = %.2e, max |grid - target grid| = %.2e\n", worst, gworst);
    if (!(worst < 1e-5f) || !(gworst < 1e-5f)) {
      printf("FAIL 3: TPS with zero displacement is not the identity\n");
      ++failures;
    }
  }

  /* 4. error behaviour */
  {
    const int rc = dvsg_flow_warp_f32(NULL, d_flow, B, H, W, C, d_out, NULL);
    if (rc >= 0 || strlen(dvsg_last_error_string()) == 0) {
      printf("FAIL 4: a NULL image was accepted (rc %d)\n", rc);
      ++failures;
    } else {
      printf("NULL image refused: %d (%s)\n", rc, dvsg_last_error_string());
    }
  }
  (void)hipFree(d_im); (void)hipFree(d_out); (void)hipFree(d_flow); (void)hipFree(d_coord); (void)hipFree(d_vec);
  (void)hipFree(d_T); (void)hipFree(d_xs); (void)hipFree(d_ys);
  free(im); free(out); free(flow); free(xs); free(ys);
  printf(failures ? "c_abi_demo: %d FAILURES\n" : "c_abi_demo: all checks passed\n", failures);
  return failures ? 1 : 0;
}
