// Device helpers shared by the localizationNet kernels.
//
// Precision: float32 storage, exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered
// fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).
//
// Fragment scheme: the 32x32x2 MFMA takes ONE f32 per lane per operand -- lane (r = l & 31,
// h = l >> 5) supplies A[r][k_h] and B[k_h][r].  The order in which the k values of a tile are
// fed is free as long as A and B agree, so each lane reads a short run of CONSECUTIVE k (a
// float4 / float2 from a K-contiguous LDS row) and feeds it over consecutive MFMAs; lane half h
// takes the second half of the run.  That turns the operand fetch into conflict-free
// ds_read_b128 / ds_read_b64 with no transposes anywhere.
#pragma once
#include <hip/hip_runtime.h>

namespace dvsg {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// XCD-aware block remap (bijective for any grid size): the hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs; give each XCD a contiguous range of logical
// tiles so that tiles sharing an activation panel / weight panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int id, int n) {
  const int q = n >> 3, r = n & 7;
  const int xcd = id & 7, slot = id >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
}

}  // namespace dvsg
