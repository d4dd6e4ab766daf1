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

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float floatx2 __attribute__((ext_vector_type(2)));

// "f32x3": eight float32 values -> their three bfloat16 pieces, x = p1 + p2 + p3 EXACTLY (round to nearest each time:
// |x - p1| <= 2^-8 |x| is a float32 with <= 16 significant bits, |x - p1 - p2| <= 2^-16 |x| one with <= 8, which p3 holds;
// bfloat16 has float32's exponent range, so this holds for every finite float32 above 2^-110).  Packed conversions
// (v_cvt_pk_bf16_f32) and packed subtractions: ~40 VALU instructions per call.
__device__ __forceinline__ void split_bf16x3(const floatx4 lo, const floatx4 hi, bf16x8 &p1, bf16x8 &p2, bf16x8 &p3) {
  const float x[8] = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    const floatx2 v = {x[i], x[i + 1]};
    const bf16x2 a = __builtin_convertvector(v, bf16x2);
    const floatx2 r = v - __builtin_convertvector(a, floatx2);
    const bf16x2 b = __builtin_convertvector(r, bf16x2);
    const floatx2 r2 = r - __builtin_convertvector(b, floatx2);
    const bf16x2 c = __builtin_convertvector(r2, bf16x2);
    p1[i] = a[0]; p1[i + 1] = a[1];
    p2[i] = b[0]; p2[i + 1] = b[1];
    p3[i] = c[0]; p3[i + 1] = c[1];
  }
}

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

// "f32s" activation tensors are stored as their two float16 pieces (P format): flat element e of a dense
// NHWC tensor whose channel count is a multiple of 32 lives in 128-byte group e / 32 -- 32 hi halves, then 32
// lo halves -- exactly the layout of a weight row stage, so a conv GEMM's LDS-DMA moves such rows as it moves
// float32 ones and a fragment is two ds_read_b128 with no arithmetic.  hi = f16(v), lo = f16(v - hi): hi + lo
// is exact in float32 and carries 22 significant bits of v while lo is a normal float16 (|v| >= 2^-3); below that lo
// is a float16 subnormal with an absolute step of 2^-24 (about 20 bits at |v| = 0.03, 15 at 1e-3, float16's own 11
// at 1e-4).  The matrix cores keep subnormal inputs, which makes the small pieces usable, not finer.
__device__ __forceinline__ float4 load4_p(const void *base, size_t e) {
  const char *q = static_cast<const char *>(base) + (e >> 5) * 128 + (e & 31) * 2;
  const halfx4 hi = *reinterpret_cast<const halfx4 *>(q), lo = *reinterpret_cast<const halfx4 *>(q + 64);
  return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2],
                     (float)hi[3] + (float)lo[3]);
}
__device__ __forceinline__ void store4_p(void *base, size_t e, float4 v) {
  char *q = static_cast<char *>(base) + (e >> 5) * 128 + (e & 31) * 2;
  halfx4 hi, lo;
  hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
  lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
  lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
  *reinterpret_cast<halfx4 *>(q) = hi;
  *reinterpret_cast<halfx4 *>(q + 64) = lo;
}

// The same for a PAIR of adjacent lanes that hold channels [n, n+4) (even lane) and [n+4, n+8) (odd lane) of one
// pixel, n % 8 == 0 (every epilogue's thread map): the lanes swap one 8-byte half through a quad-permute DPP move
// so that the even lane moves the 16 bytes of hi pieces of all 8 channels and the odd lane the 16 bytes of lo
// pieces -- one 16-byte memory instruction per lane, like a float32 float4, instead of two 8-byte ones.
// `e` is the lane's OWN flat element index (channel n or n + 4).  All 64 lanes must be active.
__device__ __forceinline__ unsigned dpp_swap1(unsigned v) {  // value of lane ^ 1
  return (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /* quad_perm [1,0,3,2] */, 0xF, 0xF, true);
}
__device__ __forceinline__ void store4_p_pair(void *base, size_t e, float4 v, bool odd, bool guard) {
  halfx4 hi, lo;
  hi[0] = (_Float16)v.x; hi[1] = (_Float16)v.y; hi[2] = (_Float16)v.z; hi[3] = (_Float16)v.w;
  lo[0] = (_Float16)(v.x - (float)hi[0]); lo[1] = (_Float16)(v.y - (float)hi[1]);
  lo[2] = (_Float16)(v.z - (float)hi[2]); lo[3] = (_Float16)(v.w - (float)hi[3]);
  typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  const uintx2 h2 = __builtin_bit_cast(uintx2, hi), l2 = __builtin_bit_cast(uintx2, lo);
  // even lane gives its lo and gets the odd lane's hi; odd lane gives its hi and gets the even lane's lo
  const uintx2 give = odd ? h2 : l2;
  const uintx2 got = {dpp_swap1(give[0]), dpp_swap1(give[1])};
  const uintx4 out = odd ? uintx4{got[0], got[1], l2[0], l2[1]} : uintx4{h2[0], h2[1], got[0], got[1]};
  const size_t e8 = e & ~(size_t)7;  // the pair's first channel
  char *q = static_cast<char *>(base) + (e8 >> 5) * 128 + (e8 & 31) * 2 + (odd ? 64 : 0);
  if (guard) *reinterpret_cast<uintx4 *>(q) = out;
}
__device__ __forceinline__ float4 load4_p_pair(const void *base, size_t e, bool odd) {
  typedef unsigned uintx4 __attribute__((ext_vector_type(4)));
  typedef unsigned uintx2 __attribute__((ext_vector_type(2)));
  const size_t e8 = e & ~(size_t)7;
  const char *q = static_cast<const char *>(base) + (e8 >> 5) * 128 + (e8 & 31) * 2 + (odd ? 64 : 0);
  const uintx4 in = *reinterpret_cast<const uintx4 *>(q);  // even: hi of 8 channels; odd: lo of 8 channels
  // even keeps hi[0..3] and needs lo[0..3] (the odd lane's first half); odd keeps lo[4..7] and needs hi[4..7]
  const uintx2 give = odd ? uintx2{in[0], in[1]} : uintx2{in[2], in[3]};
  const uintx2 got = {dpp_swap1(give[0]), dpp_swap1(give[1])};
  const halfx4 hi = __builtin_bit_cast(halfx4, odd ? got : uintx2{in[0], in[1]});
  const halfx4 lo = __builtin_bit_cast(halfx4, odd ? uintx2{in[2], in[3]} : got);
  return make_float4((float)hi[0] + (float)lo[0], (float)hi[1] + (float)lo[1], (float)hi[2] + (float)lo[2],
                     (float)hi[3] + (float)lo[3]);
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
