// Host side of localizationNet (networks.py:30-46) and of the fused evaluation graph
// (model.py:98-123): checkpoint ingestion (BatchNorm folding, weight re-layout for the MFMA
// fragment scheme of conv_gemm.hip / conv1_pool.hip, float16 copies), workspace planning and the
// launch sequence for both storage precisions.
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "cnn_kernels.h"

namespace dvsg {
namespace {

constexpr float kBnEps = 1e-5f;  // slim resnet_arg_scope batch_norm_epsilon
const char *kPrefix = "stabNet/localizationNet/";  // model.py:114-117

struct HostArray {
  const float *data;
  int nd;
  int64_t dims[4];
  size_t size() const {
    size_t n = 1;
    for (int i = 0; i < nd; ++i) n *= (size_t)dims[i];
    return n;
  }
};

struct ConvLayer {
  int ksize, cin, cout, stride;
  bool relu;
  float *wt = nullptr;        // device, [cout][k*k*cin] float32 (conv1: [7][64][kConv1Ld])
  _Float16 *wt16 = nullptr;   // device, same layout in float16 (not for conv1)
  _Float16 *wt16s = nullptr;  // device, float16 hi / lo pairs [cout/64][128][k*k*cin] (conv_gemm.hip SPLIT; not for conv1)
  _Float16 *wt16p = nullptr;  // device, wt16s packed stage by stage for conv_gemm_wide16.hip (cin % 64 == 0 layers)
  _Float16 *wt16pa = nullptr; // device, the same in the order of its 128-byte-activation-row kernel
  _Float16 *wt16ph = nullptr; // device, the same in the order of its 3x3 stride-1 kernel (3x3 layers only)
  _Float16 *wt16q = nullptr, *wt16qa = nullptr, *wt16qh = nullptr;   // device, the PLAIN copy wt16 packed the same three ways
                              // (cin % 64 == 0, cout % 128 == 0 layers: what runs when a layer has no lo piece)
  _Float16 *wt32s = nullptr;  // device, "f32s" pieces [cout][k*k*cin/32][32 hi | 32 lo] (conv_gemm.hip SPLIT, T = float; not for conv1)
  unsigned short *wt3x = nullptr;  // device, "f32x3" bfloat16 pieces, packed stage by stage (launch_pack_x3; not for conv1)
  float *bias = nullptr;      // device, [cout] float32
  const void *weights(int prec, bool split) const {
    if (prec == kF32S) return wt32s;
    if (prec == kF32X) return wt3x;
    return prec == kF16 ? (split ? (const void *)wt16s : (const void *)wt16) : (const void *)wt;
  }
};

struct Unit {
  int block;   // 0..3
  int base, depth, stride;
  bool has_shortcut;
  ConvLayer shortcut, c1, c2, c3;
  // units that open blocks 2-4: `shortcut` and `conv1` read the same input -- their weight rows concatenated along N
  // ([shortcut | conv1], float32 and f32s pieces) run as ONE launch (forward(), "concat_sc")
  ConvLayer cat;
  bool has_cat = false;
};

struct BlockSpec {
  const char *name;
  int base, units, last_stride;
};
const BlockSpec kBlocks[4] = {{"block1", 64, 3, 2}, {"block2", 128, 4, 2}, {"block3", 256, 6, 2}, {"block4", 512, 3, 1}};
const int kDenseDims[4][2] = {{2048, 2048}, {2048, 1024}, {1024, 512}, {512, 50}};

}  // namespace
}  // namespace dvsg

using namespace dvsg;

struct dvsg_locnet {
  int c_in = 0;
  ConvLayer conv1;
  std::vector<Unit> units;
  float *dense_w[4] = {nullptr, nullptr, nullptr, nullptr};
  float *dense_b[4] = {nullptr, nullptr, nullptr, nullptr};
  float *v_src = nullptr;  // [25,2] model.py:105-110
  double *winv = nullptr;  // [25][28]: columns of the TPS system's inverse for v_src (see tps_apply_kernel)
  // float16 mode: which layers multiply by hi / lo weight PAIRS (bit 4 * kind + block, see g_f16_pair_mask).  All of them
  // until dvsg_locnet_calibrate_f16 has re-rounded the plain copies of blocks 2-4 with error feedback: then block 1 only.
  int f16_pair_mask = 0xFFFF;
  std::vector<void *> allocs;
};

namespace dvsg {
namespace {

typedef std::map<std::string, HostArray> ArrayMap;

int find(const ArrayMap &m, const std::string &name, std::initializer_list<int64_t> shape, const HostArray **out) {
  auto it = m.find(name);
  if (it == m.end()) return fail(DVSG_ERR_WEIGHTS, "checkpoint array missing: %s", name.c_str());
  const HostArray &a = it->second;
  bool ok = a.nd == (int)shape.size();
  int i = 0;
  for (int64_t s : shape) {
    if (ok && a.dims[i] != s) ok = false;
    ++i;
  }
  if (!ok) {
    std::string want, got;
    for (int64_t s : shape) want += std::to_string(s) + ",";
    for (int j = 0; j < a.nd; ++j) got += std::to_string(a.dims[j]) + ",";
    return fail(DVSG_ERR_WEIGHTS, "checkpoint array %s has shape [%s] expected [%s]", name.c_str(), got.c_str(),
                want.c_str());
  }
  *out = &a;
  return DVSG_OK;
}

template <typename T>
int upload(dvsg_locnet *net, const std::vector<T> &h, T **dev) {
  void *p = nullptr;
  DVSG_HIP(hipMalloc(&p, h.size() * sizeof(T)));
  net->allocs.push_back(p);
  DVSG_HIP(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  *dev = static_cast<T *>(p);
  return DVSG_OK;
}

// BatchNorm inference folded to y = conv(x, w * scale) + shift.
int bn_fold(const ArrayMap &m, const std::string &scope, int c, std::vector<float> *scale, std::vector<float> *shift) {
  const HostArray *g, *b, *mu, *var;
  if (int rc = find(m, scope + "/BatchNorm/gamma", {c}, &g)) return rc;
  if (int rc = find(m, scope + "/BatchNorm/beta", {c}, &b)) return rc;
  if (int rc = find(m, scope + "/BatchNorm/moving_mean", {c}, &mu)) return rc;
  if (int rc = find(m, scope + "/BatchNorm/moving_variance", {c}, &var)) return rc;
  scale->resize(c);
  shift->resize(c);
  for (int i = 0; i < c; ++i) {
    const float inv = g->data[i] / std::sqrt(var->data[i] + kBnEps);
    (*scale)[i] = inv;
    (*shift)[i] = b->data[i] - mu->data[i] * inv;
  }
  return DVSG_OK;
}

// (Re)make the stage-packed copies of a layer's PLAIN float16 weights (conv_gemm_wide16.hip; round 3 built the packing for
// the stacked hi / lo rows only, so a layer without the lo piece fetched half-line weight rows again)
int pack_plain(dvsg_locnet *net, ConvLayer *L) {
  if (L->cin % 64 != 0 || L->cout % 128 != 0 || !L->wt16) return DVSG_OK;
  auto pack = [&](int order, _Float16 **out) -> int {
    if (!*out) {
      void *pk = nullptr;
      DVSG_HIP(hipMalloc(&pk, wide16_packed_bytes(L->cout, L->cin, L->ksize)));
      net->allocs.push_back(pk);
      *out = static_cast<_Float16 *>(pk);
    }
    return launch_pack_wide16(L->wt16, *out, L->cout, L->cin, L->ksize, order, nullptr);
  };
  if (int rc = pack(0, &L->wt16q)) return rc;
  if (L->ksize == 1) {
    L->wt16qa = L->wt16q;
  } else {
    if (int rc = pack(1, &L->wt16qa)) return rc;
    if (int rc = pack(2, &L->wt16qh)) return rc;
  }
  DVSG_HIP(hipStreamSynchronize(nullptr));
  return DVSG_OK;
}

// Generic conv: HWIO -> [cout][kh][kw][cin] with the BN scale folded in; float32 and float16 copies.
int make_conv(dvsg_locnet *net, const ArrayMap &m, const std::string &scope, ConvLayer *L) {
  const HostArray *w;
  if (int rc = find(m, scope + "/weights", {L->ksize, L->ksize, L->cin, L->cout}, &w)) return rc;
  std::vector<float> scale, shift;
  if (int rc = bn_fold(m, scope, L->cout, &scale, &shift)) return rc;
  const int K = L->ksize * L->ksize * L->cin;
  std::vector<float> wt((size_t)L->cout * K);
  for (int k = 0; k < K; ++k)
    for (int n = 0; n < L->cout; ++n) wt[(size_t)n * K + k] = w->data[(size_t)k * L->cout + n] * scale[n];
  std::vector<_Float16> wt16(wt.size());
  for (size_t i = 0; i < wt.size(); ++i) wt16[i] = (_Float16)wt[i];
  // hi / lo pairs: w ~ hi + 2^-11 lo, both float16 (a float16 weight alone is off by up to 2^-12 relative,
  // the same way at every pixel); group g of 64 channels = rows [128 g, 128 g + 64) hi, then 64 rows lo
  std::vector<_Float16> wt16s(2 * wt.size());
  for (int n = 0; n < L->cout; ++n)
    for (int k = 0; k < K; ++k) {
      const float w32 = wt[(size_t)n * K + k];
      const _Float16 hi = (_Float16)w32;
      const _Float16 lo = (_Float16)((w32 - (float)hi) * 2048.0f);
      const size_t row = (size_t)(n / 64) * 128 + n % 64;
      wt16s[row * K + k] = hi;
      wt16s[(row + 64) * K + k] = lo;
    }
  // "f32s" pieces: w ~ hi + lo, hi = f16(w), lo = f16(w - hi) (unscaled: float16 subnormals are inputs the
  // matrix cores keep); a 32-k row stage is 128 bytes like a float32 one: 32 hi halves, then 32 lo halves
  std::vector<_Float16> wt32s(2 * wt.size());
  for (int n = 0; n < L->cout; ++n)
    for (int k = 0; k < K; ++k) {
      const float w32 = wt[(size_t)n * K + k];
      const _Float16 hi = (_Float16)w32;
      const size_t base = ((size_t)n * (K / 32) + k / 32) * 64;
      wt32s[base + k % 32] = hi;
      wt32s[base + 32 + k % 32] = (_Float16)(w32 - (float)hi);
    }
  if (int rc = upload(net, wt, &L->wt)) return rc;
  if (int rc = upload(net, wt16, &L->wt16)) return rc;
  if (int rc = upload(net, wt16s, &L->wt16s)) return rc;
  if (L->cin % 64 == 0 && L->cout % 64 == 0) {   // the layers conv_gemm_wide16.hip can take
    // packed copies: order 0 (64-byte activation rows), 1 (128-byte rows), 2 (3x3: a kernel row's taps from one run);
    // a 1x1 layer's K order is the same in 0 and 1: one copy
    auto pack = [&](int order, _Float16 **out) -> int {
      void *pk = nullptr;
      DVSG_HIP(hipMalloc(&pk, wide16_packed_bytes(2 * L->cout, L->cin, L->ksize)));
      net->allocs.push_back(pk);
      if (int rc = launch_pack_wide16(L->wt16s, pk, 2 * L->cout, L->cin, L->ksize, order, nullptr)) return rc;
      *out = static_cast<_Float16 *>(pk);
      return DVSG_OK;
    };
    if (int rc = pack(0, &L->wt16p)) return rc;
    if (L->ksize == 1) {
      L->wt16pa = L->wt16p;
    } else {
      if (int rc = pack(1, &L->wt16pa)) return rc;
      if (int rc = pack(2, &L->wt16ph)) return rc;
    }
    DVSG_HIP(hipStreamSynchronize(nullptr));
  }
  if (int rc = pack_plain(net, L)) return rc;
  if (int rc = upload(net, wt32s, &L->wt32s)) return rc;
  if (K % 32 == 0 && L->cout % 64 == 0) {   // "f32x3": three bfloat16 pieces per weight, packed stage by stage
    void *pk = nullptr;
    DVSG_HIP(hipMalloc(&pk, x3_packed_bytes(L->cout, K)));
    net->allocs.push_back(pk);
    if (int rc = launch_pack_x3(L->wt, pk, L->cout, K, nullptr)) return rc;
    DVSG_HIP(hipStreamSynchronize(nullptr));
    L->wt3x = static_cast<unsigned short *>(pk);
  }
  return upload(net, shift, &L->bias);
}

// conv1: [7][64][kConv1Ld]; within a kernel row a zero tap, then the taps (kw, raw channel c) in input
// memory order (cnn_kernels.h).  scale_RGB's group reversal (networks.py:10-14) is folded here: raw channel
// c multiplies the weight of scaled-tensor channel (2 - c/G) * G + c % G.
int make_conv1(dvsg_locnet *net, const ArrayMap &m, const std::string &scope, ConvLayer *L) {
  const HostArray *w;
  if (int rc = find(m, scope + "/weights", {7, 7, L->cin, 64}, &w)) return rc;
  std::vector<float> scale, shift;
  if (int rc = bn_fold(m, scope, 64, &scale, &shift)) return rc;
  const int cin = L->cin, G = cin / 3;
  std::vector<float> wt((size_t)7 * 64 * kConv1Ld, 0.f);
  for (int kh = 0; kh < 7; ++kh)
    for (int kw = 0; kw < 7; ++kw)
      for (int c = 0; c < cin; ++c) {
        const int cs = (2 - c / G) * G + c % G;
        for (int n = 0; n < 64; ++n)
          wt[((size_t)kh * 64 + n) * kConv1Ld + 1 + kw * cin + c] =
              w->data[(((size_t)kh * 7 + kw) * cin + cs) * 64 + n] * scale[n];
      }
  std::vector<_Float16> wth((size_t)7 * 64 * kConv1LdH, (_Float16)0.f);
  for (int kh = 0; kh < 7; ++kh)
    for (int n = 0; n < 64; ++n)
      for (int k = 0; k < kConv1K; ++k)
        wth[((size_t)kh * 64 + n) * kConv1LdH + k + 1] = (_Float16)wt[((size_t)kh * 64 + n) * kConv1Ld + k + 1];
  // "f32s" pieces, [7][2][64][kConv1LdH]: hi image then lo image per kernel row, lo scaled by 2^11; tap k sits at
  // k + 1 behind a zero tap (conv1_split_kernel stages the input row from one element before the window)
  std::vector<_Float16> wts((size_t)7 * 2 * 64 * kConv1LdH, (_Float16)0.f);
  for (int kh = 0; kh < 7; ++kh)
    for (int n = 0; n < 64; ++n)
      for (int k = 0; k < kConv1K; ++k) {
        const float w32 = wt[((size_t)kh * 64 + n) * kConv1Ld + k + 1];
        const _Float16 hi = (_Float16)w32;
        wts[(((size_t)kh * 2 + 0) * 64 + n) * kConv1LdH + k + 1] = hi;
        wts[(((size_t)kh * 2 + 1) * 64 + n) * kConv1LdH + k + 1] = (_Float16)((w32 - (float)hi) * 2048.0f);
      }
  // "f32x3" pieces [14][3][64][kConv1X3Ld] bfloat16 (cnn_kernels.h): stage 2 kh + half, position p = tap + 1
  auto bf16_rn = [](float f) -> unsigned short {   // round to nearest even on the upper 16 bits (finite inputs)
    uint32_t u;
    std::memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
  };
  auto bf16_val = [](unsigned short hbits) -> float {
    const uint32_t u = (uint32_t)hbits << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
  };
  std::vector<unsigned short> wtx((size_t)14 * kConv1X3StageElems, 0);
  for (int kh = 0; kh < 7; ++kh)
    for (int n = 0; n < 64; ++n)
      for (int pos = 1; pos <= kConv1K; ++pos) {
        const float w32 = wt[((size_t)kh * 64 + n) * kConv1Ld + pos];
        const unsigned short p1 = bf16_rn(w32);
        const float r1 = w32 - bf16_val(p1);
        const unsigned short p2 = bf16_rn(r1);
        const unsigned short p3 = bf16_rn(r1 - bf16_val(p2));
        const size_t base = (size_t)(2 * kh + pos / 80) * kConv1X3StageElems + (size_t)n * kConv1X3Ld + pos % 80;
        wtx[base] = p1;
        wtx[base + 64 * kConv1X3Ld] = p2;
        wtx[base + 2 * 64 * kConv1X3Ld] = p3;
      }
  if (int rc = upload(net, wt, &L->wt)) return rc;
  if (int rc = upload(net, wth, &L->wt16)) return rc;
  if (int rc = upload(net, wts, &L->wt32s)) return rc;
  if (int rc = upload(net, wtx, &L->wt3x)) return rc;
  return upload(net, shift, &L->bias);
}

struct Dims {
  int H1, W1, Hp, Wp, pad_top, pad_left;
};

Dims root_dims(int H, int W) {
  Dims d;
  d.H1 = (H - 1) / 2 + 1;  // conv2d_same(7, stride 2): pad 3/3 then VALID
  d.W1 = (W - 1) / 2 + 1;
  d.Hp = (d.H1 + 1) / 2;   // max_pool2d 3x3/2 SAME
  d.Wp = (d.W1 + 1) / 2;
  const int pth = std::max((d.Hp - 1) * 2 + 3 - d.H1, 0), ptw = std::max((d.Wp - 1) * 2 + 3 - d.W1, 0);
  d.pad_top = pth / 2;
  d.pad_left = ptw / 2;
  return d;
}

size_t align256(size_t n) { return (n + 255) & ~(size_t)255; }

// float16 precision: conv weights as hi / lo float16 pairs (default) or plain float16 (A/B only:
// dvsg_debug_set_option("f16_split", 0); 9/10 of the plain mode's F_t error is the weights' rounding)
int g_f16_split = 1;
// Which layers of the float16 mode carry the lo piece: bit 4 * kind + block (kind 0 = a unit's conv1, 1 = conv2, 2 = conv3,
// 3 = shortcut; block 0..3).  A plain float16 weight is off by up to 2^-12 relative at every pixel alike, an error the global
// average pool does not average away; how much of it reaches F_t depends on the layer (tools/f16_pair_sweep.py measures
// every block x kind).  A layer without the lo piece runs half the MFMAs and, in the big launches, 128 channels per tile.
// dvsg_debug_set_option("f16_pair_mask", m): A/B.
int g_f16_pair_mask = 0xFFFF;
enum LayerKind { kKindC1 = 0, kKindC2 = 1, kKindC3 = 2, kKindSc = 3 };
inline bool f16_pairs(const dvsg_locnet *net, int block, int kind) {
  return g_f16_split && ((g_f16_pair_mask & net->f16_pair_mask) >> (4 * kind + block)) & 1;
}
// block 1's shortcut conv inside the fused conv2 + conv3 kernel (dvsg_debug_set_option("fuse_shortcut", 0): A/B)
int g_fuse_shortcut = 1;
int g_x3_conv1 = 1;  // dvsg_debug_set_option("x3_conv1", 0): the f32x3 precision with the float32 conv1 kernel (A/B)
int g_x3_fuse = 3;   // dvsg_debug_set_option("x3_fuse", v): A/B of block 1's fusion in the f32x3 precision (forward())
// blocks 2-4's opening units: shortcut + conv1 as one launch (dvsg_debug_set_option("concat_sc", 0): A/B)
int g_concat_sc = 1;

struct Workspace {
  char *bufA, *bufB, *bufS, *r1, *r2;  // activations (element type = the run's precision)
  float *pool_part, *dpart0, *dpart1, *T, *Ft;
  char *splitk_slabs;  // kSplitKSlabBytes of partial tiles, reused by every split-K launch
  int *splitk_counters;  // kMaxConvLaunches x kSplitKMaxTiles tickets, zeroed once per forward
  size_t total;
};
constexpr int kMaxConvLaunches = 64;

// Sized for float32 activations; the float16 run uses the same plan with half the bytes.
Workspace plan(char *base, int B, int H, int W) {
  const Dims d = root_dims(H, W);
  // Walk the 16 units with forward()'s own (h, w) recurrence: with the ceil in (h - 1) / stride + 1
  // the per-sample element count does NOT shrink monotonically on tiny frames (8x8: block 4 needs
  // 1x1x2048 > 2x2x256), so the buffers are sized by the largest tenant, not by the first one.
  size_t big_el = std::max((size_t)d.H1 * d.W1 * 64, (size_t)d.Hp * d.Wp * 64);  // conv1 out (bufA), pool1 out (bufB)
  size_t small_el = 0;
  {
    int h = d.Hp, w = d.Wp;
    for (const BlockSpec &bs : kBlocks)
      for (int u = 1; u <= bs.units; ++u) {
        const int stride = u == bs.units ? bs.last_stride : 1;
        const int ho = (h - 1) / stride + 1, wo = (w - 1) / stride + 1;
        small_el = std::max(small_el, (size_t)h * w * bs.base);                  // r1 (conv1 out), r2 <= r1
        big_el = std::max(big_el, (size_t)ho * wo * bs.base * 4);                // unit out, shortcut
        if (u == 1) big_el = std::max(big_el, (size_t)h * w * bs.base * 5);     // [shortcut | conv1] of an opening unit (bufS)
        h = ho; w = wo;
      }
  }
  const size_t big = align256((size_t)B * big_el * sizeof(float));
  const size_t small = align256((size_t)B * small_el * sizeof(float));
  Workspace w;
  size_t off = 0;
  auto take = [&](size_t bytes) {
    char *p = base + off;
    off += align256(bytes);
    return p;
  };
  w.bufA = take(big);
  w.bufB = take(big);
  w.bufS = take(big);
  w.r1 = take(small);
  w.r2 = take(small);
  w.pool_part = reinterpret_cast<float *>(take((size_t)kPoolSplits * B * 2048 * sizeof(float)));
  w.dpart0 = reinterpret_cast<float *>(take((size_t)kDenseSplits * 16 * 2048 * sizeof(float)));
  w.dpart1 = reinterpret_cast<float *>(take((size_t)kDenseSplits * 16 * 2048 * sizeof(float)));
  w.T = reinterpret_cast<float *>(take((size_t)B * 2 * 28 * sizeof(float)));
  w.Ft = reinterpret_cast<float *>(take((size_t)B * 50 * sizeof(float)));
  w.splitk_slabs = take(kSplitKSlabBytes);
  w.splitk_counters = reinterpret_cast<int *>(take((size_t)kMaxConvLaunches * kSplitKMaxTiles * sizeof(int)));
  w.total = off;
  return w;
}

int run_conv(int prec, const ConvLayer &L, bool pairs, const void *x, int B, int H, int W, void *y, int Ho, int Wo,
             const void *res, int res_H, int res_W, int res_stride, bool relu, const Workspace &ws, int *launch_idx,
             hipStream_t s, int ldx = 0, int res_ld = 0, int relu_from = -1) {
  ConvGemm p;
  p.ldx = ldx; p.res_ld = res_ld; p.relu_from = relu_from;
  p.splitk_scratch = ws.splitk_slabs;
  p.splitk_scratch_bytes = kSplitKSlabBytes;
  p.splitk_counters = ws.splitk_counters + (size_t)((*launch_idx)++ % kMaxConvLaunches) * kSplitKMaxTiles;
  p.prec = prec == kF32S || prec == kF32X ? kF32 : prec;
  p.wsplit = prec == kF32S || (prec == kF16 && pairs);
  p.x3 = prec == kF32X;
  if (prec == kF32X && !L.wt3x)
    return fail(DVSG_ERR_UNSUPPORTED, "f32x3: layer %d -> %d (k = %d) has no packed bfloat16 pieces (K %% 32, Cout %% 64)", L.cin, L.cout, L.ksize);
  p.x = x; p.wt = L.weights(prec, p.wsplit != 0); p.bias = L.bias; p.res = res; p.y = y;
  if (prec == kF16 && pairs) {
    p.wt_packed = L.wt16p;
    p.wt_packed_a = L.wt16pa;
    p.wt_packed_h = L.wt16ph;
  } else if (prec == kF16) {
    p.wt_packed = L.wt16q;
    p.wt_packed_a = L.wt16qa;
    p.wt_packed_h = L.wt16qh;
  }
  p.B = B; p.H = H; p.W = W; p.Cin = L.cin; p.Ho = Ho; p.Wo = Wo; p.Cout = L.cout;
  p.ksize = L.ksize; p.stride = L.stride; p.pad = L.ksize == 3 ? 1 : 0;
  p.res_H = res_H; p.res_W = res_W; p.res_stride = res_stride;
  p.relu = relu;
  return launch_conv_gemm(p, s);
}

// ---- calibration of the float16 mode's PLAIN weights (dvsg_debug_calibrate_f16_weights) --------------------------------
// While armed, forward() adds up every bottleneck unit's three convolution inputs per channel: slot 3 u + {0: the unit's
// input (shortcut, conv1), 1: conv1's output (conv2's input), 2: conv2's output (conv3's input)}.
struct Calib {
  double *sums = nullptr;    // device [16 * 3][2048]
  float *part = nullptr;     // device [kCalibBlocks][2048]: per-block partial sums of the tensor being recorded
  double rows[48] = {0};     // pixels summed per slot
  bool on = false;
} g_calib;

// deterministic: block b adds up rows [b chunk, (b + 1) chunk) per channel into part[b][c]; channel_sum_final adds the
// blocks' partial sums in block order (double) onto out[c] -- the same bits from run to run, so the weights re-rounded from
// them are the same bits too
constexpr int kCalibBlocks = 512;
__global__ __launch_bounds__(256) void channel_sum_kernel(const float *__restrict__ x, long M, int C, long chunk,
                                                         float *__restrict__ part) {
  const long r0 = (long)blockIdx.x * chunk, r1 = r0 + chunk < M ? r0 + chunk : M;
  for (int c = threadIdx.x; c < C; c += 256) {
    float acc = 0.f;
    for (long r = r0; r < r1; ++r) acc += x[r * C + c];
    part[(size_t)blockIdx.x * 2048 + c] = acc;
  }
}
__global__ __launch_bounds__(256) void channel_sum_final(const float *__restrict__ part, int nblk, int C, double *__restrict__ out) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double acc = out[c];
  for (int b = 0; b < nblk; ++b) acc += (double)part[(size_t)b * 2048 + c];
  out[c] = acc;
}

int calib_record(int slot, const void *x, long M, int C, hipStream_t s) {
  if (!g_calib.on) return DVSG_OK;
  const long chunk = (M + kCalibBlocks - 1) / kCalibBlocks;
  const int nblk = (int)((M + chunk - 1) / chunk);
  hipLaunchKernelGGL(channel_sum_kernel, dim3(nblk), dim3(256), 0, s, static_cast<const float *>(x), M, C, chunk, g_calib.part);
  hipLaunchKernelGGL(channel_sum_final, dim3((C + 255) / 256), dim3(256), 0, s, g_calib.part, nblk, C,
                     g_calib.sums + (size_t)slot * 2048);
  g_calib.rows[slot] += (double)M;
  return check_launch("channel_sum_kernel");
}

// Runs the network in precision `prec`; stop_stage < 0 runs everything and writes F_t [B,50].
int forward(const dvsg_locnet *net, int prec, const Conv1Src &src, int src_kind, int B, int H, int W, float *F_t, int stop_stage,
            float *act_out, size_t act_out_bytes, int *act_dims, void *workspace, size_t workspace_bytes,
            hipStream_t s) {
  DVSG_REQUIRE(net && src.base && workspace, "locnet forward: NULL pointer");
  DVSG_REQUIRE(B > 0 && H >= 1 && W >= 1, "locnet forward: bad shape B=%d H=%d W=%d", B, H, W);
  DVSG_REQUIRE(((uintptr_t)workspace & 255) == 0, "locnet forward: workspace must be 256-byte aligned");
  const Workspace ws = plan(static_cast<char *>(workspace), B, H, W);
  if (ws.total > workspace_bytes)
    return fail(DVSG_ERR_WORKSPACE, "locnet forward: workspace %zu bytes < required %zu", workspace_bytes, ws.total);
  const Dims d = root_dims(H, W);
  // f32x3: float32 tensors throughout; conv1, the pools, block 1's fused kernel and the head are the float32 kernels
  // themselves, the conv GEMMs multiply bfloat16 pieces (run_conv)
  const int gprec = prec;
  if (prec == kF32X) prec = kF32;

  // parity tap: copy (float32 run) or convert (float16 run) the stage's activation to act_out
  auto tap = [&](int stage, const void *act, int h, int w, int c) -> int {
    if (stage != stop_stage) return 0;
    const size_t n = (size_t)B * h * w * c;
    if (n * sizeof(float) > act_out_bytes)
      return fail(DVSG_ERR_INVALID_ARG, "tap buffer %zu bytes < %zu", act_out_bytes, n * sizeof(float));
    if (prec == kF16) {
      if (int rc = launch_f16_to_f32(act, act_out, n, s)) return rc;
    } else if (prec == kF32S) {
      if (int rc = launch_p_to_f32(act, act_out, n, s)) return rc;
    } else {
      DVSG_HIP(hipMemcpyAsync(act_out, act, n * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    act_dims[0] = h; act_dims[1] = w; act_dims[2] = c;
    return 1;
  };
#define DVSG_TAP(stage, act, h, w, c)          \
  do {                                         \
    int t_ = tap(stage, act, h, w, c);         \
    if (t_ < 0) return t_;                     \
    if (t_ > 0) return DVSG_OK;                \
  } while (0)
#define DVSG_RUN(call)             \
  do {                             \
    if (int rc_ = (call)) return rc_; \
  } while (0)

  // split-K tickets of every conv launch of this pass (each launch owns its own segment)
  DVSG_RUN(launch_zero_tickets(ws.splitk_counters, (size_t)kMaxConvLaunches * kSplitKMaxTiles, s));
  int launch_idx = 0;
  // root: conv1 (+ fused scale_RGB; f32 multiply, output in `prec`) -> bufA, max pool -> bufB
  {
  MarkerRange mr("dvsg/conv1");
  DVSG_RUN(launch_conv1(gprec == kF32X && g_x3_conv1 ? kF32X : prec, src, src_kind, net->conv1.wt, net->conv1.wt16, net->conv1.wt32s,
                        net->conv1.wt3x, net->conv1.bias, ws.bufA, B, H, W, d.H1, d.W1, s));
  }
  DVSG_TAP(0, ws.bufA, d.H1, d.W1, 64);
  {
  MarkerRange mr("dvsg/pool1");
  DVSG_RUN(launch_maxpool(prec, ws.bufA, ws.bufB, B, d.H1, d.W1, 64, d.Hp, d.Wp, d.pad_top, d.pad_left, s));
  }
  DVSG_TAP(1, ws.bufB, d.Hp, d.Wp, 64);

  char *X = ws.bufB, *Y = ws.bufA;
  int h = d.Hp, w = d.Wp;
  int stage = 2;
  int unit_in_block = 0, last_block = -1;
  for (const Unit &u : net->units) {
    unit_in_block = u.block == last_block ? unit_in_block + 1 : 1;
    last_block = u.block;
    char range_name[40];
    std::snprintf(range_name, sizeof(range_name), "dvsg/block%d/unit_%d", u.block + 1, unit_in_block);
    MarkerRange mr(range_name);
    const int ho = (h - 1) / u.stride + 1, wo = (w - 1) / u.stride + 1;
    const void *res = X;
    int res_h = h, res_w = w, res_stride = u.stride;
    const int calib_slot = 3 * (stage - 2);
    DVSG_RUN(calib_record(calib_slot, X, (long)B * h * w, u.c1.cin, s));
    // (float16 mode: the fused kernel multiplies against the stacked hi / lo weights only)
    // (f32x3: block 1's units run conv2 + conv3 in the f32x3 fused kernel, conv_fused_x3.hip, the opening unit's shortcut as
    // its own f32x3 GEMM -- g_x3_fuse = 3; A/B: 0 never fused, 1 the opening unit only and in the exact float32 kernel with
    // its shortcut, 2 all three in the exact float32 kernel)
    const bool x3_fused = gprec == kF32X && g_x3_fuse == 3;
    const bool x3_unfused = gprec == kF32X && (g_x3_fuse == 0 || (g_x3_fuse == 1 && !u.has_shortcut));
    const bool fuse23 = !x3_unfused && conv_fusable(prec, u.c2.cin, u.c2.cout, u.c3.cout, u.c2.ksize) &&
                        (prec != kF16 || (f16_pairs(net, u.block, kKindC2) && f16_pairs(net, u.block, kKindC3) &&
                                          (!u.has_shortcut || f16_pairs(net, u.block, kKindSc))));
    // block 1's opening unit: its shortcut conv (64 -> 256) runs inside the fused conv2 + conv3 kernel
    const bool fuse_sc = fuse23 && !x3_fused && u.has_shortcut && u.stride == 1 && u.shortcut.cin == 64 && u.shortcut.cout == 256 &&
                         g_fuse_shortcut;
    // blocks 2-4's opening units, float32 / f32s: shortcut and conv1 as ONE launch over [shortcut | conv1] weight rows; its
    // output [M, depth + base] sits in bufS, the shortcut in columns [0, depth) (conv3's residual, row stride depth + base),
    // conv1's ReLU'd output behind it (conv2's input, same stride).  One launch and one read of the unit's input less.
    const bool cat = u.has_shortcut && !fuse_sc && u.has_cat && prec != kF16 && g_concat_sc && !g_calib.on && u.stride == 1;
    const void *c2_in = ws.r1;
    int c2_ldx = 0, res_ld = 0;
    if (cat) {
      DVSG_RUN(run_conv(gprec, u.cat, false, X, B, h, w, ws.bufS, h, w, nullptr, 0, 0, 1, false, ws, &launch_idx, s, 0, 0,
                        u.shortcut.cout));
      res = ws.bufS;
      res_h = ho; res_w = wo; res_stride = 1;
      res_ld = u.cat.cout;
      c2_in = ws.bufS + (size_t)u.shortcut.cout * sizeof(float);   // (f32s: 32 values are 128 bytes too)
      c2_ldx = u.cat.cout;
    } else {
    if (u.has_shortcut && !fuse_sc) {  // 1x1 conv + BN, no ReLU (stride is 1 wherever depth changes)
      DVSG_RUN(run_conv(gprec, u.shortcut, f16_pairs(net, u.block, kKindSc), X, B, h, w, ws.bufS, ho, wo, nullptr, 0, 0, 1, false, ws,
                        &launch_idx, s));
      res = ws.bufS;
      res_h = ho; res_w = wo; res_stride = 1;
    }
    DVSG_RUN(run_conv(gprec, u.c1, f16_pairs(net, u.block, kKindC1), X, B, h, w, ws.r1, h, w, nullptr, 0, 0, 1, true, ws, &launch_idx, s));
    }
    if (fuse23) {  // block 1: conv2 + conv3 in one kernel
      ConvFused f;
      const bool pcs = prec == kF32S;
      auto wts_of = [&](const ConvLayer &L) {
        if (x3_fused) return reinterpret_cast<const float *>(L.wt3x);
        return pcs ? reinterpret_cast<const float *>(L.wt32s) : prec == kF16 ? reinterpret_cast<const float *>(L.wt16s) : L.wt;
      };
      f.pieces = pcs;
      f.f16 = prec == kF16;
      f.x3 = x3_fused;
      f.x = reinterpret_cast<const float *>(ws.r1); f.wt2 = wts_of(u.c2); f.bias2 = u.c2.bias; f.wt3 = wts_of(u.c3); f.bias3 = u.c3.bias;
      f.res = static_cast<const float *>(res); f.y = reinterpret_cast<float *>(Y);
      f.B = B; f.H = h; f.W = w; f.Cin = u.c2.cin; f.Ho = ho; f.Wo = wo; f.Cout = u.c3.cout;
      f.stride = u.c2.stride; f.res_H = res_h; f.res_W = res_w; f.res_stride = res_stride;
      if (fuse_sc) {
        f.res = nullptr;
        f.sc_x = reinterpret_cast<const float *>(X); f.sc_wt = wts_of(u.shortcut); f.sc_bias = u.shortcut.bias;
        f.sc_cin = u.shortcut.cin;
      }
      DVSG_RUN(launch_conv3x3_1x1(f, s));
    } else {
      DVSG_RUN(calib_record(calib_slot + 1, ws.r1, (long)B * h * w, u.c2.cin, s));
      DVSG_RUN(run_conv(gprec, u.c2, f16_pairs(net, u.block, kKindC2), c2_in, B, h, w, ws.r2, ho, wo, nullptr, 0, 0, 1, true, ws,
                        &launch_idx, s, c2_ldx));
      DVSG_RUN(calib_record(calib_slot + 2, ws.r2, (long)B * ho * wo, u.c3.cin, s));
      DVSG_RUN(run_conv(gprec, u.c3, f16_pairs(net, u.block, kKindC3), ws.r2, B, ho, wo, Y, ho, wo, res, res_h, res_w, res_stride, true,
                        ws, &launch_idx, s, 0, res_ld));
    }
    h = ho; w = wo;
    DVSG_TAP(stage, Y, h, w, u.depth);
    ++stage;
    char *tmp = X; X = Y; Y = tmp;
  }

  // global average pool (float32 partial sums), then the float32 dense head in chunks of <= 16
  // samples; partial layouts are [b][split][k] so a batch chunk is a pointer offset.
  MarkerRange mr_head("dvsg/head");
  DVSG_RUN(launch_avgpool_partial(prec, X, ws.pool_part, B, h * w, 2048, s));
  const float inv_hw = 1.0f / (float)(h * w);
  if (stop_stage == 18) {
    const size_t bytes = (size_t)B * 2048 * sizeof(float);
    if (bytes > act_out_bytes) return fail(DVSG_ERR_INVALID_ARG, "tap buffer %zu bytes < %zu", act_out_bytes, bytes);
    DVSG_RUN(launch_dense_finalize(ws.pool_part, kPoolSplits, inv_hw, nullptr, act_out, B, 2048, s));
    act_dims[0] = 1; act_dims[1] = 1; act_dims[2] = 2048;
    return DVSG_OK;
  }
  for (int b0 = 0; b0 < B; b0 += 16) {
    const int bc = std::min(16, B - b0);
    float *pa = ws.dpart0, *pb = ws.dpart1;
    DVSG_RUN(launch_dense(ws.pool_part + (size_t)b0 * kPoolSplits * 2048, kPoolSplits, nullptr, inv_hw, 0,
                          net->dense_w[0], pa, bc, 2048, 2048, s));
    DVSG_RUN(launch_dense(pa, kDenseSplits, net->dense_b[0], 1.0f, 1, net->dense_w[1], pb, bc, 2048, 1024, s));
    DVSG_RUN(launch_dense(pb, kDenseSplits, net->dense_b[1], 1.0f, 1, net->dense_w[2], pa, bc, 1024, 512, s));
    DVSG_RUN(launch_dense(pa, kDenseSplits, net->dense_b[2], 1.0f, 1, net->dense_w[3], pb, bc, 512, 50, s));
    DVSG_RUN(launch_dense_finalize(pb, kDenseSplits, 1.0f, net->dense_b[3], F_t + (size_t)b0 * 50, bc, 50, s));
  }
  return DVSG_OK;
#undef DVSG_TAP
#undef DVSG_RUN
}

int forward(const dvsg_locnet *net, int prec, const float *patches, int B, int H, int W, float *F_t, int stop_stage,
            float *act_out, size_t act_out_bytes, int *act_dims, void *workspace, size_t workspace_bytes,
            hipStream_t s) {
  const Conv1Src src{patches, nullptr, 0};
  return forward(net, prec, src, kSrcWindow, B, H, W, F_t, stop_stage, act_out, act_out_bytes, act_dims, workspace,
                 workspace_bytes, s);
}

int stabilize(const dvsg_locnet *net, int prec, const float *patches_t, const float *u_t, const float *mask, int B, int H,
              int W, float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace, size_t workspace_bytes,
              void *stream) {
  DVSG_REQUIRE(net && patches_t && u_t && s_t_pred && workspace, "dvsg_stabilize: NULL pointer");
  DVSG_REQUIRE(B > 0 && B <= 65535, "dvsg_stabilize: B=%d out of range", B);
  DVSG_REQUIRE(prec == kF32 || prec == kF16 || prec == kF32S || prec == kF32X, "dvsg_stabilize: unknown precision %d", prec);
  DVSG_REQUIRE(!mask || net->c_in == 21, "dvsg_stabilize_masked: the mask covers the 18 history channels of a 7-frame window");
  const Workspace ws = plan(static_cast<char *>(workspace), B, H, W);
  if (ws.total > workspace_bytes)
    return fail(DVSG_ERR_WORKSPACE, "dvsg_stabilize: workspace %zu bytes < required %zu", workspace_bytes, ws.total);
  float *F = F_t ? F_t : ws.Ft;
  Conv1Src src{patches_t, nullptr, 0};
  src.mask = mask;   // eval_train.py:43-45: the CNN sees patches * mask, the warp below the unmasked u_t
  if (int rc = forward(net, prec, src, kSrcWindow, B, H, W, F, -1, nullptr, 0, nullptr, workspace, workspace_bytes,
                       as_stream(stream)))
    return rc;
  // model.py:120: stn(u_t, V_src, F_t, [h, w]) with V_src tiled over the batch (:111); float32
  MarkerRange mr("dvsg/tps");
  if (int rc = tps_apply_impl(net->winv, net->v_src, F, 1, B, 25, ws.T, stream)) return rc;
  return tps_warp_impl(u_t, net->v_src, 0, ws.T, B, H, W, 3, 25, H, W, s_t_pred, x_s, y_s, stream);
}

// The evaluation graph fed from a frame ring: window b = pool frames table[b][0..6], u_t = its newest
// frame table[b][6]; conv1 assembles the window in its load stage (eval.py:103-104) and, for a uint8 pool, applies the
// / 255. of eval.py:80 there; the warp reads u_t from the pool through the same table.
int stabilize_ring(const dvsg_locnet *net, int prec, const void *pool, int pool_is_u8, int n_pool, const int32_t *table,
                   const float *mask, int B, int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace,
                   size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(net && pool && table && s_t_pred && workspace, "dvsg_stabilize_ring: NULL pointer");
  DVSG_REQUIRE(B > 0 && B <= 65535 && n_pool > 0, "dvsg_stabilize_ring: B=%d n_pool=%d out of range", B, n_pool);
  DVSG_REQUIRE(prec == kF32 || prec == kF16 || prec == kF32S || prec == kF32X, "dvsg_stabilize_ring: unknown precision %d", prec);
  DVSG_REQUIRE(net->c_in == 21, "dvsg_stabilize_ring: the ring holds RGB frames, 7 per window");
  const Workspace ws = plan(static_cast<char *>(workspace), B, H, W);
  if (ws.total > workspace_bytes)
    return fail(DVSG_ERR_WORKSPACE, "dvsg_stabilize_ring: workspace %zu bytes < required %zu", workspace_bytes, ws.total);
  float *F = F_t ? F_t : ws.Ft;
  Conv1Src src{pool, table, n_pool};
  src.mask = mask;
  if (int rc = forward(net, prec, src, pool_is_u8 ? kSrcRingU8 : kSrcRingF32, B, H, W, F, -1, nullptr, 0, nullptr, workspace,
                       workspace_bytes, as_stream(stream)))
    return rc;
  MarkerRange mr("dvsg/tps");
  if (int rc = tps_apply_impl(net->winv, net->v_src, F, 1, B, 25, ws.T, stream)) return rc;
  return tps_warp_ring_impl(pool, pool_is_u8, n_pool, table + 6, 7, net->v_src, 0, ws.T, B, H, W, 25, s_t_pred, x_s, y_s,
                            stream);
}

int conv_gemm_op(int prec, int wsplit, const void *x, const void *wt, const float *bias, const void *res, void *y, int B,
                 int H, int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                 size_t scratch_bytes, void *stream) {
  DVSG_REQUIRE(x && wt && bias && y, "dvsg_conv_gemm: NULL pointer");
  DVSG_REQUIRE(!scratch || ((uintptr_t)scratch & 255) == 0, "dvsg_conv_gemm: scratch must be 256-byte aligned");
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0 && stride >= 1 && res_stride >= 1, "dvsg_conv_gemm: bad shape");
  ConvGemm p;
  p.prec = prec;
  p.wsplit = wsplit == 1;
  p.x3 = wsplit == 2;   // f32x3: packed bfloat16 piece stages
  p.x = x; p.wt = wt; p.bias = bias; p.res = res; p.y = y;
  p.B = B; p.H = H; p.W = W; p.Cin = Cin; p.Cout = Cout;
  p.Ho = (H - 1) / stride + 1; p.Wo = (W - 1) / stride + 1;
  p.ksize = ksize; p.stride = stride; p.pad = ksize == 3 ? 1 : 0;
  p.res_H = (p.Ho - 1) * res_stride + 1; p.res_W = (p.Wo - 1) * res_stride + 1;
  p.res_stride = res_stride;
  p.relu = relu;
  const size_t cbytes = align256((size_t)kSplitKMaxTiles * sizeof(int));
  if (scratch && scratch_bytes > cbytes) {  // [tickets | partial-tile slabs]
    if (int rc = launch_zero_tickets(static_cast<int *>(scratch), cbytes / sizeof(int), as_stream(stream))) return rc;
    p.splitk_counters = static_cast<int *>(scratch);
    p.splitk_scratch = static_cast<char *>(scratch) + cbytes;
    // the partial-tile slabs own at most kSplitKSlabBytes behind the tickets: whatever the caller's scratch holds beyond
    // that may carry the packed weight copies below, which a slab user sizing itself by this field must never reach
    p.splitk_scratch_bytes = std::min(scratch_bytes - cbytes, align256(kSplitKSlabBytes));
  }
  if (prec == kF16 && Cin % 64 == 0 && Cout % (wsplit ? 64 : 128) == 0 && (ksize == 1 || ksize == 3) && scratch) {
    // the packed weight copies conv_gemm_wide16.hip prefers (what dvsg_locnet_create makes once per layer), made here on
    // every call behind the tickets and the partial-tile slabs when the caller's scratch has room for them; without
    // them the layer runs from the [rows][K] layout
    const int rows = wsplit ? 2 * Cout : Cout;   // stacked hi / lo rows, or plain ones
    const size_t need = align256(wide16_packed_bytes(rows, Cin, ksize));
    const size_t base = cbytes + align256(kSplitKSlabBytes);
    if (scratch_bytes >= base + 3 * need) {
      char *pk = static_cast<char *>(scratch) + base;
      if (int rc = launch_pack_wide16(wt, pk, rows, Cin, ksize, 0, as_stream(stream))) return rc;
      p.wt_packed = pk;
      if (ksize == 1) {
        p.wt_packed_a = pk;
      } else {
        if (int rc = launch_pack_wide16(wt, pk + need, rows, Cin, ksize, 1, as_stream(stream))) return rc;
        if (int rc = launch_pack_wide16(wt, pk + 2 * need, rows, Cin, ksize, 2, as_stream(stream))) return rc;
        p.wt_packed_a = pk + need;
        p.wt_packed_h = pk + 2 * need;
      }
    }
  }
  return launch_conv_gemm(p, as_stream(stream));
}

}  // namespace
}  // namespace dvsg

extern "C" {

int dvsg_locnet_create(int n_arrays, const char *const *names, const float *const *host_data, const int *ndims,
                       const int64_t *dims, dvsg_locnet_t **out) {
  DVSG_REQUIRE(names && host_data && ndims && dims && out && n_arrays > 0, "dvsg_locnet_create: bad arguments");
  *out = nullptr;
  ArrayMap m;
  for (int i = 0; i < n_arrays; ++i) {
    DVSG_REQUIRE(names[i] && host_data[i] && ndims[i] >= 1 && ndims[i] <= 4, "dvsg_locnet_create: bad array %d", i);
    std::string name(names[i]);
    if (name.size() > 2 && name.compare(name.size() - 2, 2, ":0") == 0) name.resize(name.size() - 2);
    HostArray a;
    a.data = host_data[i];
    a.nd = ndims[i];
    for (int j = 0; j < 4; ++j) a.dims[j] = j < a.nd ? dims[(size_t)i * 4 + j] : 1;
    m[name] = a;
  }
  const std::string rn = std::string(kPrefix) + "resnet_v1_50";
  auto it = m.find(rn + "/conv1/weights");
  if (it == m.end()) return fail(DVSG_ERR_WEIGHTS, "checkpoint array missing: %s/conv1/weights", rn.c_str());
  if (it->second.nd != 4) return fail(DVSG_ERR_WEIGHTS, "conv1/weights must be 4-D");
  const int c_in = (int)it->second.dims[2];
  if (c_in != kConv1Cin)
    return fail(DVSG_ERR_UNSUPPORTED, "conv1 has %d input channels; this build supports the 7-frame window (21)", c_in);

  dvsg_locnet *net = new dvsg_locnet();
  net->c_in = c_in;
  int rc = DVSG_OK;
  auto bail = [&](int code) {
    dvsg_locnet_destroy(net);
    return code;
  };
  net->conv1 = ConvLayer{7, c_in, 64, 2, true};
  if ((rc = make_conv1(net, m, rn + "/conv1", &net->conv1))) return bail(rc);
  int depth_in = 64;
  for (const BlockSpec &bs : kBlocks) {
    for (int u = 1; u <= bs.units; ++u) {
      Unit unit;
      unit.block = (int)(&bs - kBlocks);
      unit.base = bs.base;
      unit.depth = bs.base * 4;
      unit.stride = u == bs.units ? bs.last_stride : 1;
      unit.has_shortcut = depth_in != unit.depth;
      const std::string sc = rn + "/" + bs.name + "/unit_" + std::to_string(u) + "/bottleneck_v1";
      if (unit.has_shortcut) {
        if (unit.stride != 1) return bail(fail(DVSG_ERR_UNSUPPORTED, "strided shortcut conv not expected in resnet_v1_50"));
        unit.shortcut = ConvLayer{1, depth_in, unit.depth, 1, false};
        if ((rc = make_conv(net, m, sc + "/shortcut", &unit.shortcut))) return bail(rc);
      }
      unit.c1 = ConvLayer{1, depth_in, bs.base, 1, true};
      unit.c2 = ConvLayer{3, bs.base, bs.base, unit.stride, true};
      unit.c3 = ConvLayer{1, bs.base, unit.depth, 1, false};
      if ((rc = make_conv(net, m, sc + "/conv1", &unit.c1))) return bail(rc);
      if ((rc = make_conv(net, m, sc + "/conv2", &unit.c2))) return bail(rc);
      if ((rc = make_conv(net, m, sc + "/conv3", &unit.c3))) return bail(rc);
      if (unit.has_shortcut && unit.shortcut.cin % 64 == 0 && unit.shortcut.cin > 64) {
        // [shortcut rows | conv1 rows]: device-to-device copies of the two layers' float32 rows, f32s pieces and biases
        const ConvLayer &a = unit.shortcut, &b = unit.c1;
        unit.cat = ConvLayer{1, a.cin, a.cout + b.cout, 1, false};
        const size_t ka = (size_t)a.cout * a.cin, kb = (size_t)b.cout * b.cin;
        void *w = nullptr, *ws2 = nullptr, *bi = nullptr, *w3 = nullptr;
        if (hipMalloc(&w, (ka + kb) * sizeof(float)) != hipSuccess || hipMalloc(&ws2, 2 * (ka + kb) * sizeof(_Float16)) != hipSuccess ||
            hipMalloc(&bi, (a.cout + b.cout) * sizeof(float)) != hipSuccess || hipMalloc(&w3, 6 * (ka + kb)) != hipSuccess)
          return bail(fail(DVSG_ERR_HIP, "hipMalloc failed"));
        net->allocs.push_back(w); net->allocs.push_back(ws2); net->allocs.push_back(bi); net->allocs.push_back(w3);
        hipError_t e = hipMemcpy(w, a.wt, ka * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(static_cast<float *>(w) + ka, b.wt, kb * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(ws2, a.wt32s, 2 * ka * sizeof(_Float16), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(static_cast<_Float16 *>(ws2) + 2 * ka, b.wt32s, 2 * kb * sizeof(_Float16), hipMemcpyDeviceToDevice);
        // (the f32x3 packing is by groups of 64 rows: the two layers' packed stages one behind the other)
        if (e == hipSuccess) e = hipMemcpy(w3, a.wt3x, 6 * ka, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(static_cast<char *>(w3) + 6 * ka, b.wt3x, 6 * kb, hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(bi, a.bias, a.cout * sizeof(float), hipMemcpyDeviceToDevice);
        if (e == hipSuccess) e = hipMemcpy(static_cast<float *>(bi) + a.cout, b.bias, b.cout * sizeof(float), hipMemcpyDeviceToDevice);
        if (e != hipSuccess) return bail(fail(DVSG_ERR_HIP, "concatenating shortcut | conv1: %s", hipGetErrorString(e)));
        unit.cat.wt = static_cast<float *>(w);
        unit.cat.wt32s = static_cast<_Float16 *>(ws2);
        unit.cat.wt3x = static_cast<unsigned short *>(w3);
        unit.cat.bias = static_cast<float *>(bi);
        unit.has_cat = true;
      }
      net->units.push_back(unit);
      depth_in = unit.depth;
    }
  }
  for (int i = 0; i < 4; ++i) {
    const std::string sc = std::string(kPrefix) + "df/dense" + std::to_string(i + 1);
    const HostArray *W, *b;
    if ((rc = find(m, sc + "/W", {kDenseDims[i][0], kDenseDims[i][1]}, &W))) return bail(rc);
    if ((rc = find(m, sc + "/b", {kDenseDims[i][1]}, &b))) return bail(rc);
    std::vector<float> hw(W->data, W->data + W->size()), hb(b->data, b->data + b->size());
    if ((rc = upload(net, hw, &net->dense_w[i]))) return bail(rc);
    if ((rc = upload(net, hb, &net->dense_b[i]))) return bail(rc);
  }
  std::vector<float> vs(50);  // model.py:105-110: 5x5 grid on [-1,1]^2, x fastest
  for (int i = 0; i < 5; ++i)
    for (int j = 0; j < 5; ++j) {
      vs[(i * 5 + j) * 2 + 0] = -1.0f + 0.5f * j;
      vs[(i * 5 + j) * 2 + 1] = -1.0f + 0.5f * i;
    }
  if ((rc = upload(net, vs, &net->v_src))) return bail(rc);
  {  // the TPS system of the constant V_src grid is inverted once (float64)
    void *w = nullptr, *scratch = nullptr;
    if (hipMalloc(&w, 25 * 28 * sizeof(double)) != hipSuccess) return bail(fail(DVSG_ERR_HIP, "hipMalloc failed"));
    net->allocs.push_back(w);
    if (hipMalloc(&scratch, 13 * 25 * 2 * sizeof(float)) != hipSuccess) return bail(fail(DVSG_ERR_HIP, "hipMalloc failed"));
    rc = tps_inverse_columns(net->v_src, 25, static_cast<double *>(w), static_cast<float *>(scratch), nullptr);
    (void)hipFree(scratch);
    if (rc) return bail(rc);
    net->winv = static_cast<double *>(w);
  }
  *out = net;
  return DVSG_OK;
}

int dvsg_locnet_destroy(dvsg_locnet_t *net) {
  if (!net) return DVSG_OK;
  for (void *p : net->allocs) (void)hipFree(p);
  delete net;
  return DVSG_OK;
}

int dvsg_locnet_in_channels(const dvsg_locnet_t *net) { return net ? net->c_in : 0; }

int dvsg_locnet_workspace_bytes(const dvsg_locnet_t *net, int B, int H, int W, size_t *bytes) {
  DVSG_REQUIRE(net && bytes, "dvsg_locnet_workspace_bytes: NULL pointer");
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0, "dvsg_locnet_workspace_bytes: bad shape B=%d H=%d W=%d", B, H, W);
  *bytes = plan(nullptr, B, H, W).total;
  return DVSG_OK;
}

int dvsg_locnet_forward_f32(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, float *F_t,
                            void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(F_t, "dvsg_locnet_forward_f32: NULL F_t");
  return forward(net, kF32, patches, B, H, W, F_t, -1, nullptr, 0, nullptr, workspace, workspace_bytes,
                 as_stream(stream));
}

int dvsg_locnet_forward_f16(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, float *F_t,
                            void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(F_t, "dvsg_locnet_forward_f16: NULL F_t");
  return forward(net, kF16, patches, B, H, W, F_t, -1, nullptr, 0, nullptr, workspace, workspace_bytes,
                 as_stream(stream));
}

int dvsg_locnet_forward_f32s(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, float *F_t,
                             void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(F_t, "dvsg_locnet_forward_f32s: NULL F_t");
  return forward(net, kF32S, patches, B, H, W, F_t, -1, nullptr, 0, nullptr, workspace, workspace_bytes,
                 as_stream(stream));
}

int dvsg_locnet_forward_f32x3(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, float *F_t,
                              void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(F_t, "dvsg_locnet_forward_f32x3: NULL F_t");
  return forward(net, kF32X, patches, B, H, W, F_t, -1, nullptr, 0, nullptr, workspace, workspace_bytes,
                 as_stream(stream));
}

int dvsg_locnet_forward_tap_f32x3(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, int stage,
                                  float *act_out, size_t act_out_bytes, int *act_dims_host, void *workspace,
                                  size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(act_out && act_dims_host, "dvsg_locnet_forward_tap_f32x3: NULL pointer");
  DVSG_REQUIRE(stage >= 0 && stage <= 18, "dvsg_locnet_forward_tap_f32x3: stage %d outside [0,18]", stage);
  return forward(net, kF32X, patches, B, H, W, nullptr, stage, act_out, act_out_bytes, act_dims_host, workspace,
                 workspace_bytes, as_stream(stream));
}

int dvsg_locnet_forward_tap_f32s(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, int stage,
                                 float *act_out, size_t act_out_bytes, int *act_dims_host, void *workspace,
                                 size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(act_out && act_dims_host, "dvsg_locnet_forward_tap_f32s: NULL pointer");
  DVSG_REQUIRE(stage >= 0 && stage <= 18, "dvsg_locnet_forward_tap_f32s: stage %d outside [0,18]", stage);
  return forward(net, kF32S, patches, B, H, W, nullptr, stage, act_out, act_out_bytes, act_dims_host, workspace,
                 workspace_bytes, as_stream(stream));
}

int dvsg_locnet_forward_tap_f32(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, int stage,
                                float *act_out, size_t act_out_bytes, int *act_dims_host, void *workspace,
                                size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(act_out && act_dims_host, "dvsg_locnet_forward_tap_f32: NULL pointer");
  DVSG_REQUIRE(stage >= 0 && stage <= 18, "dvsg_locnet_forward_tap_f32: stage %d outside [0,18]", stage);
  return forward(net, kF32, patches, B, H, W, nullptr, stage, act_out, act_out_bytes, act_dims_host, workspace,
                 workspace_bytes, as_stream(stream));
}

int dvsg_locnet_forward_tap_f16(const dvsg_locnet_t *net, const float *patches, int B, int H, int W, int stage,
                                float *act_out, size_t act_out_bytes, int *act_dims_host, void *workspace,
                                size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(act_out && act_dims_host, "dvsg_locnet_forward_tap_f16: NULL pointer");
  DVSG_REQUIRE(stage >= 0 && stage <= 18, "dvsg_locnet_forward_tap_f16: stage %d outside [0,18]", stage);
  return forward(net, kF16, patches, B, H, W, nullptr, stage, act_out, act_out_bytes, act_dims_host, workspace,
                 workspace_bytes, as_stream(stream));
}

int dvsg_conv_gemm_f32(const float *x, const float *wt, const float *bias, const float *res, float *y, int B, int H,
                       int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                       size_t scratch_bytes, void *stream) {
  return conv_gemm_op(kF32, 0, x, wt, bias, res, y, B, H, W, Cin, Cout, ksize, stride, relu, res_stride, scratch,
                      scratch_bytes, stream);
}

int dvsg_conv_gemm_f16(const void *x, const void *wt, const float *bias, const void *res, void *y, int B, int H,
                       int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                       size_t scratch_bytes, void *stream) {
  return conv_gemm_op(kF16, 0, x, wt, bias, res, y, B, H, W, Cin, Cout, ksize, stride, relu, res_stride, scratch,
                      scratch_bytes, stream);
}

int dvsg_conv_gemm_f32s(const void *x, const void *wt_pieces, const float *bias, const void *res, void *y, int B, int H,
                        int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                        size_t scratch_bytes, void *stream) {
  return conv_gemm_op(kF32, 1, x, wt_pieces, bias, res, y, B, H, W, Cin, Cout, ksize, stride, relu, res_stride, scratch,
                      scratch_bytes, stream);
}

int dvsg_conv_gemm_f32x3(const float *x, const void *wt_packed, const float *bias, const float *res, float *y, int B, int H,
                         int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                         size_t scratch_bytes, void *stream) {
  return conv_gemm_op(kF32, 2, x, wt_packed, bias, res, y, B, H, W, Cin, Cout, ksize, stride, relu, res_stride, scratch,
                      scratch_bytes, stream);
}

int dvsg_pack_weights_f32x3(const float *wt, void *wt_packed, int Cout, int K, void *stream) {
  return launch_pack_x3(wt, wt_packed, Cout, K, as_stream(stream));
}

int dvsg_f32_to_pieces(const float *x, void *y, size_t n, void *stream) {
  DVSG_REQUIRE(x && y, "dvsg_f32_to_pieces: NULL pointer");
  return launch_f32_to_p(x, y, n, as_stream(stream));
}

int dvsg_pieces_to_f32(const void *x, float *y, size_t n, void *stream) {
  DVSG_REQUIRE(x && y, "dvsg_pieces_to_f32: NULL pointer");
  return launch_p_to_f32(x, y, n, as_stream(stream));
}

int dvsg_conv_gemm_f16s(const void *x, const void *wt_split, const float *bias, const void *res, void *y, int B, int H,
                        int W, int Cin, int Cout, int ksize, int stride, int relu, int res_stride, void *scratch,
                        size_t scratch_bytes, void *stream) {
  return conv_gemm_op(kF16, 1, x, wt_split, bias, res, y, B, H, W, Cin, Cout, ksize, stride, relu, res_stride, scratch,
                      scratch_bytes, stream);
}

int dvsg_conv3x3_1x1_f32(const float *x, const float *wt2, const float *bias2, const float *wt3, const float *bias3,
                         const float *res, float *y, int B, int H, int W, int Cin, int Cout, int stride, int res_stride,
                         void *stream) {
  DVSG_REQUIRE(x && wt2 && bias2 && wt3 && bias3 && res && y, "dvsg_conv3x3_1x1_f32: NULL pointer");
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && res_stride >= 1,
               "dvsg_conv3x3_1x1_f32: bad shape B=%d H=%d W=%d stride=%d res_stride=%d", B, H, W, stride, res_stride);
  ConvFused f;
  f.x = x; f.wt2 = wt2; f.bias2 = bias2; f.wt3 = wt3; f.bias3 = bias3; f.res = res; f.y = y;
  f.B = B; f.H = H; f.W = W; f.Cin = Cin; f.Cout = Cout;
  f.Ho = (H - 1) / stride + 1; f.Wo = (W - 1) / stride + 1;
  f.stride = stride;
  f.res_H = (f.Ho - 1) * res_stride + 1; f.res_W = (f.Wo - 1) * res_stride + 1; f.res_stride = res_stride;
  return launch_conv3x3_1x1(f, as_stream(stream));
}

int dvsg_conv3x3_1x1_f32x3(const float *x, const void *wt2_packed, const float *bias2, const void *wt3_packed, const float *bias3,
                           const float *res, float *y, int B, int H, int W, int Cin, int Cout, int stride, int res_stride,
                           void *stream) {
  DVSG_REQUIRE(x && wt2_packed && bias2 && wt3_packed && bias3 && res && y, "dvsg_conv3x3_1x1_f32x3: NULL pointer");
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0 && (stride == 1 || stride == 2) && res_stride >= 1,
               "dvsg_conv3x3_1x1_f32x3: bad shape B=%d H=%d W=%d stride=%d res_stride=%d", B, H, W, stride, res_stride);
  ConvFused f;
  f.x = x; f.wt2 = static_cast<const float *>(wt2_packed); f.bias2 = bias2; f.wt3 = static_cast<const float *>(wt3_packed);
  f.bias3 = bias3; f.res = res; f.y = y;
  f.B = B; f.H = H; f.W = W; f.Cin = Cin; f.Cout = Cout;
  f.Ho = (H - 1) / stride + 1; f.Wo = (W - 1) / stride + 1;
  f.stride = stride;
  f.res_H = (f.Ho - 1) * res_stride + 1; f.res_W = (f.Wo - 1) * res_stride + 1; f.res_stride = res_stride;
  f.x3 = 1;
  return launch_conv3x3_1x1(f, as_stream(stream));
}

int dvsg_debug_set_option(const char *name, int value) {
  DVSG_REQUIRE(name, "dvsg_debug_set_option: NULL name");
  if (std::strcmp(name, "conv_variant") == 0) {
    set_conv_variant(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "wide16_min_tiles") == 0) {
    set_wide16_min_tiles(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "fused_hreuse") == 0) {
    set_fused_hreuse(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "wide16_hreuse") == 0) {
    set_wide16_hreuse(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "wide16_arows") == 0) {
    set_wide16_arows(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "wide16_packed") == 0) {
    set_wide16_packed(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "flow_tiled") == 0) {
    set_flow_tiled(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "flow_rounds") == 0) {
    set_flow_rounds(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "warp_xcd") == 0) {
    set_warp_xcd(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "conv1_variant") == 0) {
    set_conv1_variant(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "fuse_conv") == 0) {
    set_fuse_conv(value);
    return DVSG_OK;
  }
  if (std::strcmp(name, "concat_sc") == 0) {
    g_concat_sc = value != 0;
    return DVSG_OK;
  }
  if (std::strcmp(name, "x3_conv1") == 0) {
    g_x3_conv1 = value;
    return DVSG_OK;
  }
  if (std::strcmp(name, "x3_fuse") == 0) {
    g_x3_fuse = value;
    return DVSG_OK;
  }
  if (std::strcmp(name, "fuse_shortcut") == 0) {
    g_fuse_shortcut = value != 0;
    return DVSG_OK;
  }
  if (std::strcmp(name, "f16_split") == 0) {
    g_f16_split = value != 0;
    return DVSG_OK;
  }
  if (std::strcmp(name, "f16_pair_mask") == 0) {
    g_f16_pair_mask = value & 0xFFFF;
    return DVSG_OK;
  }
  return fail(DVSG_ERR_INVALID_ARG, "dvsg_debug_set_option: unknown option %s", name);
}

// Error-feedback rounding of the float16 mode's PLAIN weights (the copies a layer without the lo piece multiplies by).
// A float16 weight is off by up to 2^-12 relative, the same way at every pixel: through the mean activation mu_k of its
// input channel that is a BIAS of its output channel, sum_k (q_k - w_k) mu_k, which the global average pool does not
// average away -- 9/10 of the plain mode's F_t error, and what the hi / lo pairs exist to remove (at twice the MFMAs).
// Here every weight is rounded to one of its two float16 neighbours so that the running sum of (q_k - w_k) mu_k along K
// stays within half a step: the bias goes, the zero-mean part -- which the pool does average -- stays.  mu comes from ONE
// float32 pass over `patches` (calibration windows) with block 1's fusion off, recorded per unit and convolution input.
// mode 0: round to nearest again (undo); 1: mu = 1 (A/B: does nothing for F_t -- channel means are far from uniform);
// 2: calibrated means.  Measured (tools/f16_ef_sweep.py, two 720p windows, synthetic checkpoint): plain weights 1.9e-5 ->
// 2.7e-6 (hi / lo pairs everywhere: 1.6e-6), 22 % less time per step.
static int calibrate_f16(dvsg_locnet *net, const float *patches, int B, int H, int W, int mode, void *workspace,
                         size_t workspace_bytes, hipStream_t s) {
  std::vector<double> mu((size_t)48 * 2048, 1.0);
  if (mode == 2) {   // calibrated channel means
    void *sums = nullptr, *part = nullptr, *F = nullptr;
    hipError_t e = hipMalloc(&sums, 48 * 2048 * sizeof(double));
    if (e == hipSuccess) e = hipMalloc(&part, (size_t)kCalibBlocks * 2048 * sizeof(float));
    if (e == hipSuccess) e = hipMalloc(&F, (size_t)B * 50 * sizeof(float));
    if (e == hipSuccess) e = hipMemsetAsync(sums, 0, 48 * 2048 * sizeof(double), s);
    int rc = e == hipSuccess ? DVSG_OK : fail(DVSG_ERR_HIP, "calibration: %s", hipGetErrorString(e));
    if (rc == DVSG_OK) {
      g_calib.sums = static_cast<double *>(sums);
      g_calib.part = static_cast<float *>(part);
      for (double &r : g_calib.rows) r = 0;
      const int fuse_was = get_fuse_conv();
      set_fuse_conv(0);       // block 1's conv2 output only exists in LDS when fused
      g_calib.on = true;
      rc = forward(net, kF32, patches, B, H, W, static_cast<float *>(F), -1, nullptr, 0, nullptr, workspace, workspace_bytes, s);
      g_calib.on = false;
      set_fuse_conv(fuse_was);
      if (rc == DVSG_OK && hipMemcpyAsync(mu.data(), sums, mu.size() * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess)
        rc = fail(DVSG_ERR_HIP, "copy of the calibration sums failed");
      if (rc == DVSG_OK && hipStreamSynchronize(s) != hipSuccess) rc = fail(DVSG_ERR_HIP, "calibration pass failed");
    }
    (void)hipFree(sums);
    (void)hipFree(part);
    (void)hipFree(F);
    if (rc) return rc;
    for (int slot = 0; slot < 48; ++slot)
      for (int c = 0; c < 2048; ++c) mu[(size_t)slot * 2048 + c] = g_calib.rows[slot] > 0 ? mu[(size_t)slot * 2048 + c] / g_calib.rows[slot] : 1.0;
  }
  auto f16_bits = [](_Float16 v) { unsigned short b; std::memcpy(&b, &v, 2); return b; };
  auto f16_from = [](unsigned short b) { _Float16 v; std::memcpy(&v, &b, 2); return v; };
  auto neighbour = [&](_Float16 q, bool up) -> _Float16 {   // next float16 above / below q
    unsigned short b = f16_bits(q);
    if ((b & 0x7fff) == 0) return f16_from(up ? 0x0001 : 0x8001);
    const bool neg = (b & 0x8000) != 0;
    b = (unsigned short)((up != neg) ? b + 1 : b - 1);
    return f16_from(b);
  };
  auto redo = [&](ConvLayer &L, const double *m) -> int {
    const int K = L.ksize * L.ksize * L.cin;
    std::vector<float> wt((size_t)L.cout * K);
    DVSG_HIP(hipMemcpy(wt.data(), L.wt, wt.size() * sizeof(float), hipMemcpyDeviceToHost));
    std::vector<_Float16> q16(wt.size());
    for (int n = 0; n < L.cout; ++n) {
      double E = 0.0;
      for (int k = 0; k < K; ++k) {
        const float w32 = wt[(size_t)n * K + k];
        const _Float16 q0 = (_Float16)w32;
        _Float16 q = q0;
        if (mode != 0 && (float)q0 != w32) {
          const _Float16 other = neighbour(q0, (float)q0 < w32);
          const double mk = m[k % L.cin];                  // k = (kh, kw, c): the channel's mean for every tap
          const double e0 = E + ((double)(float)q0 - w32) * mk, e1 = E + ((double)(float)other - w32) * mk;
          const bool take = std::fabs(e1) < std::fabs(e0) && std::isfinite((float)other);
          q = take ? other : q0;
          E = take ? e1 : e0;
        }
        q16[(size_t)n * K + k] = q;
      }
    }
    DVSG_HIP(hipMemcpy(L.wt16, q16.data(), q16.size() * sizeof(_Float16), hipMemcpyHostToDevice));
    return pack_plain(net, &L);
  };
  int ui = 0;
  for (Unit &u : net->units) {
    const double *mx = mu.data() + (size_t)(3 * ui) * 2048;
    if (u.has_shortcut) if (int rc = redo(u.shortcut, mx)) return rc;
    if (int rc = redo(u.c1, mx)) return rc;
    if (int rc = redo(u.c2, mx + 2048)) return rc;
    if (int rc = redo(u.c3, mx + 2 * 2048)) return rc;
    ++ui;
  }
  return DVSG_OK;
}

int dvsg_locnet_calibrate_f16(dvsg_locnet_t *net, const float *patches, int B, int H, int W, void *workspace,
                              size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(net && workspace, "dvsg_locnet_calibrate_f16: NULL pointer");
  if (!patches) {   // undo: round-to-nearest plain copies, pairs everywhere
    net->f16_pair_mask = 0xFFFF;
    return calibrate_f16(net, nullptr, 0, 0, 0, 0, workspace, workspace_bytes, as_stream(stream));
  }
  DVSG_REQUIRE(B > 0 && H > 0 && W > 0, "dvsg_locnet_calibrate_f16: bad shape B=%d H=%d W=%d", B, H, W);
  if (int rc = calibrate_f16(net, patches, B, H, W, 2, workspace, workspace_bytes, as_stream(stream))) return rc;
  net->f16_pair_mask = 0x1111;   // block 1 (whose fused kernels multiply by the stacked pairs) keeps them; blocks 2-4 run plain
  return DVSG_OK;
}

int dvsg_debug_calibrate_f16_weights(dvsg_locnet_t *net, const float *patches, int B, int H, int W, int mode, void *workspace,
                                     size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(net && patches && workspace && mode >= 0 && mode <= 2, "dvsg_debug_calibrate_f16_weights: bad arguments");
  return calibrate_f16(net, patches, B, H, W, mode, workspace, workspace_bytes, as_stream(stream));
}

int dvsg_stabilize_f32(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B, int H, int W,
                       float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace, size_t workspace_bytes,
                       void *stream) {
  return stabilize(net, kF32, patches_t, u_t, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s, workspace, workspace_bytes, stream);
}

int dvsg_stabilize_f32s(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B, int H, int W,
                        float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace, size_t workspace_bytes,
                        void *stream) {
  return stabilize(net, kF32S, patches_t, u_t, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s, workspace, workspace_bytes, stream);
}

int dvsg_stabilize_f16(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B, int H, int W,
                       float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace, size_t workspace_bytes,
                       void *stream) {
  return stabilize(net, kF16, patches_t, u_t, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s, workspace, workspace_bytes, stream);
}

int dvsg_stabilize_f32x3(const dvsg_locnet_t *net, const float *patches_t, const float *u_t, int B, int H, int W,
                         float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace, size_t workspace_bytes,
                         void *stream) {
  return stabilize(net, kF32X, patches_t, u_t, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s, workspace, workspace_bytes, stream);
}

static int ring_precision(int precision) {
  return precision == DVSG_PRECISION_F32 ? kF32 : precision == DVSG_PRECISION_F16 ? kF16 : precision == DVSG_PRECISION_F32S ? kF32S
         : precision == DVSG_PRECISION_F32X3 ? kF32X : -1;
}

int dvsg_stabilize_ring_f32(const dvsg_locnet_t *net, int precision, const float *pool, int n_pool, const int32_t *table,
                            int B, int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace,
                            size_t workspace_bytes, void *stream) {
  return stabilize_ring(net, ring_precision(precision), pool, 0, n_pool, table, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s,
                        workspace, workspace_bytes, stream);
}

int dvsg_stabilize_ring_u8(const dvsg_locnet_t *net, int precision, const uint8_t *pool, int n_pool, const int32_t *table,
                           int B, int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s, void *workspace,
                           size_t workspace_bytes, void *stream) {
  return stabilize_ring(net, ring_precision(precision), pool, 1, n_pool, table, nullptr, B, H, W, s_t_pred, F_t, x_s, y_s,
                        workspace, workspace_bytes, stream);
}

// eval_train.py's evaluation graph (:25-51): F_t = localizationNet(patches_t * mask), the warp on the unmasked u_t.
int dvsg_stabilize_masked_f32(const dvsg_locnet_t *net, int precision, const float *patches_t, const float *u_t,
                              const float *mask, int B, int H, int W, float *s_t_pred, float *F_t, float *x_s, float *y_s,
                              void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(mask, "dvsg_stabilize_masked_f32: NULL mask");
  return stabilize(net, ring_precision(precision), patches_t, u_t, mask, B, H, W, s_t_pred, F_t, x_s, y_s, workspace,
                   workspace_bytes, stream);
}

int dvsg_stabilize_ring_masked_f32(const dvsg_locnet_t *net, int precision, const float *pool, int n_pool,
                                   const int32_t *table, const float *mask, int B, int H, int W, float *s_t_pred, float *F_t,
                                   float *x_s, float *y_s, void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(mask, "dvsg_stabilize_ring_masked_f32: NULL mask");
  return stabilize_ring(net, ring_precision(precision), pool, 0, n_pool, table, mask, B, H, W, s_t_pred, F_t, x_s, y_s,
                        workspace, workspace_bytes, stream);
}

int dvsg_stabilize_ring_masked_u8(const dvsg_locnet_t *net, int precision, const uint8_t *pool, int n_pool,
                                  const int32_t *table, const float *mask, int B, int H, int W, float *s_t_pred, float *F_t,
                                  float *x_s, float *y_s, void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(mask, "dvsg_stabilize_ring_masked_u8: NULL mask");
  return stabilize_ring(net, ring_precision(precision), pool, 1, n_pool, table, mask, B, H, W, s_t_pred, F_t, x_s, y_s,
                        workspace, workspace_bytes, stream);
}

// The CNN alone on a masked source -- F_t [B,25,2] into `out` (stage = -1) or the parity tap of `stage` (0..18).
// src_kind 0: `src` is a window tensor [B,H,W,21] (n_pool, table unused); 1 / 2: a float32 / uint8 frame pool + table.
int dvsg_locnet_forward_masked(const dvsg_locnet_t *net, int precision, const void *src_ptr, int src_kind, int n_pool,
                               const int32_t *table, const float *mask, int B, int H, int W, int stage, float *out,
                               size_t out_bytes, int *act_dims_host, void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(net && src_ptr && mask && out, "dvsg_locnet_forward_masked: NULL pointer");
  DVSG_REQUIRE(src_kind >= 0 && src_kind <= 2 && (src_kind == 0 || (table && n_pool > 0)),
               "dvsg_locnet_forward_masked: bad source (kind %d)", src_kind);
  DVSG_REQUIRE(net->c_in == 21, "dvsg_locnet_forward_masked: the mask covers the 18 history channels of a 7-frame window");
  DVSG_REQUIRE(stage >= -1 && stage <= 18 && (stage < 0 || act_dims_host), "dvsg_locnet_forward_masked: stage %d outside [-1,18]", stage);
  DVSG_REQUIRE(stage >= 0 || out_bytes >= (size_t)B * 50 * sizeof(float), "dvsg_locnet_forward_masked: F_t needs B*50 floats");
  const int prec = ring_precision(precision);
  DVSG_REQUIRE(prec >= 0, "dvsg_locnet_forward_masked: unknown precision %d", precision);
  Conv1Src src{src_ptr, src_kind ? table : nullptr, src_kind ? n_pool : 0};
  src.mask = mask;
  return forward(net, prec, src, src_kind, B, H, W, stage < 0 ? out : nullptr, stage, stage < 0 ? nullptr : out, out_bytes,
                 act_dims_host, workspace, workspace_bytes, as_stream(stream));
}

int dvsg_locnet_forward_ring(const dvsg_locnet_t *net, int precision, const void *pool, int pool_is_u8, int n_pool,
                             const int32_t *table, int B, int H, int W, int stage, float *out, size_t out_bytes,
                             int *act_dims_host, void *workspace, size_t workspace_bytes, void *stream) {
  DVSG_REQUIRE(pool && table && out && n_pool > 0, "dvsg_locnet_forward_ring: bad arguments");
  DVSG_REQUIRE(stage >= -1 && stage <= 18 && (stage < 0 || act_dims_host), "dvsg_locnet_forward_ring: stage %d outside [-1,18]", stage);
  DVSG_REQUIRE(stage >= 0 || out_bytes >= (size_t)B * 50 * sizeof(float), "dvsg_locnet_forward_ring: F_t needs B*50 floats");
  const int prec = ring_precision(precision);
  DVSG_REQUIRE(prec >= 0, "dvsg_locnet_forward_ring: unknown precision %d", precision);
  const Conv1Src src{pool, table, n_pool};
  return forward(net, prec, src, pool_is_u8 ? kSrcRingU8 : kSrcRingF32, B, H, W, stage < 0 ? out : nullptr, stage,
                 stage < 0 ? nullptr : out, out_bytes, act_dims_host, workspace, workspace_bytes, as_stream(stream));
}

}  // extern "C"
