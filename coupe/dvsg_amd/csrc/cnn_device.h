// Device helpers shared by the localizationNet kernels.
//
// Precision kF32: float32 storage, exact-f32 matrix cores (v_mfma_f32_32x32x2_f32: a k-ordered
// fmaf chain, 64 FLOP/clk/SIMD = 157 TFLOP/s chip peak).  Precision kF16: float16 storage,
// v_mfma_f32_32x32x16_f16 with float32 accumulation (~2.5 PFLOP/s dense peak).
//
// Fragment scheme: lane (r = l & 31, h = l >> 5) of a 32x32 MFMA supplies row r of A and column
// r of B for the k values owned by lane half h.  The order in which the k values of a tile are
// fed is free as long as A and B agree, so each lane reads ONE 16-byte run of consecutive k from
// a K-contiguous LDS row: four f32 fed over four 32x32x2 MFMAs, or eight f16 that are exactly
// one 32x32x16 operand.  Operand fetch is a conflict-free ds_read_b128 with no transposes.
#pragma once
#include <hip/hip_runtime.h>

namespace dvsg {

typedef float floatx16 __attribute__((ext_vector_type(16)));
typedef float floatx4 __attribute__((ext_vector_type(4)));
typedef _Float16 halfx8 __attribute__((ext_vector_type(8)));
typedef _Float16 halfx4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ floatx16 mfma32(float a, float b, floatx16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// One 16-byte fragment chunk (k-run) of A and of B -> accumulate into c.
template <typename T>
struct Frag;

template <>
struct Frag<float> {
  typedef floatx4 type;
  static constexpr int kElems = 4;  // k values per 16-byte chunk
  static __device__ __forceinline__ floatx16 mma(floatx4 a, floatx4 b, floatx16 c) {
    c = mfma32(a[0], b[0], c);
    c = mfma32(a[1], b[1], c);
    c = mfma32(a[2], b[2], c);
    return mfma32(a[3], b[3], c);
  }
};

template <>
struct Frag<_Float16> {
  typedef halfx8 type;
  static constexpr int kElems = 8;
  static __device__ __forceinline__ floatx16 mma(halfx8 a, halfx8 b, floatx16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  }
};

// "f32s" products: eight float32 values (two 16-byte chunks, consecutive k) -> their float16 pieces
// hi = rtz_f16(x) and lo = f16(x - hi).  x - hi is exact in float32 (hi is x truncated to 11 significant
// bits), so hi + lo carries 22 significant bits of x; lo may be a float16 subnormal, which the matrix
// cores keep.  4 v_cvt_pkrtz + 8 v_cvt_f32_f16 + 8 subtractions + 4 v_cvt_pkrtz per 8 values.
__device__ __forceinline__ void split_f16x2(floatx4 f0, floatx4 f1, halfx8 &hi, halfx8 &lo) {
  typedef __fp16 fp16x2 __attribute__((ext_vector_type(2)));
  typedef _Float16 halfx2 __attribute__((ext_vector_type(2)));
  const float f[8] = {f0[0], f0[1], f0[2], f0[3], f1[0], f1[1], f1[2], f1[3]};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const fp16x2 hp = __builtin_amdgcn_cvt_pkrtz(f[2 * i], f[2 * i + 1]);
    const halfx2 h2 = __builtin_bit_cast(halfx2, hp);
    hi[2 * i] = h2[0];
    hi[2 * i + 1] = h2[1];
    const fp16x2 lp = __builtin_amdgcn_cvt_pkrtz(f[2 * i] - (float)h2[0], f[2 * i + 1] - (float)h2[1]);
    const halfx2 l2 = __builtin_bit_cast(halfx2, lp);
    lo[2 * i] = l2[0];
    lo[2 * i + 1] = l2[1];
  }
}

// Four consecutive channels of an activation tensor <-> float4.
__device__ __forceinline__ float4 load4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ float4 load4(const _Float16 *p) {
  const halfx4 v = *reinterpret_cast<const halfx4 *>(p);
  return make_float4((float)v[0], (float)v[1], (float)v[2], (float)v[3]);
}
__device__ __forceinline__ void store4(float *p, float4 v) { *reinterpret_cast<float4 *>(p) = v; }
__device__ __forceinline__ void store4(_Float16 *p, float4 v) {
  halfx4 o;
  o[0] = (_Float16)v.x;
  o[1] = (_Float16)v.y;
  o[2] = (_Float16)v.z;
  o[3] = (_Float16)v.w;
  *reinterpret_cast<halfx4 *>(p) = o;
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
