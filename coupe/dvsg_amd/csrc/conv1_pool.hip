// Root of resnet_v1_50 (slim `conv2d_same(64, 7, stride=2)` + BatchNorm + ReLU, then
// `max_pool2d(3x3, stride 2, SAME)`) with scale_RGB (networks.py:6-16) fused into conv1's
// load stage.
#include <cstdint>
#include <type_traits>

#include "cnn_device.h"
#include "cnn_kernels.h"

namespace dvsg {
namespace {

// ----------------------------------------------------------------------------------------
// conv1: 7x7 / stride 2 / pad 3, 21 -> 64.
//
// A workgroup owns 128 consecutive output pixels of one output row and all 64 channels.
// For kernel row kh the 7*21 = 147 taps of an output pixel are ONE contiguous run of the
// input row, and neighbouring output pixels start 42 floats apart: the raw input row
// segment (261 px * 21 ch = 5481 floats, 22 KB) is staged once in LDS and every A fragment
// is read from it in place as lds[42 * pixel + k] -- overlapping windows, no im2col, each
// input byte fetched once per kernel row.  42 r mod 64 visits every even bank once over
// r = 0..31, so the ds_read_b64 fragment reads are conflict-free as they stand; the weight
// rows use a stride of 150 floats (2 x odd) for the same reason.
//
// scale_RGB: y = 255 x - mean, channel groups reversed.  The zero padding of conv2d_same
// happens AFTER scale_RGB, so pad taps must be 0 in the scaled domain: the scale is applied
// per element while staging (valid elements only), and the group reversal is a permutation
// of conv1's input channels that is folded into the weights at load time (locnet.hip).
//
// NW waves per workgroup: NW/2 along the pixels x 2 along the channels.
// ----------------------------------------------------------------------------------------
constexpr int C1_TILE = 128;
constexpr int C1_SEG_PAD = 5488;                              // staged elements: 1 + (2 * 127 + 7) * 21 = 5482, in whole groups of four
constexpr int C1_WELEMS = 64 * kConv1Ld;                      // 9600 floats per kernel row
constexpr int C1_LDC = 68;                                    // epilogue tile row stride

// ----------------------------------------------------------------------------------------
// The staged segment of an input row, shared by the three conv1 kernels.  Element e of the segment is
// input-row float (2 wo0 - 3) * 21 - 1 + e: it starts one element before the first window (the weight rows
// carry a zero tap in front), which is a multiple of four elements of the row, and is fetched as GROUPS
// 16-byte groups per thread (thread tid: groups tid, tid + NT, ...) -- 6 loads per thread and kernel row
// where element-wise fetching issued 22.  When the rows themselves are 16-byte aligned (ALIGNED: W a
// multiple of 4) every group is inside the row or outside it as a whole and the loads are unconditional and
// aligned; otherwise a group inside the row is one 4-byte-aligned dwordx4 load and a group that straddles
// an end of the row goes element by element.  Loads are issued from clamped indices whatever the validity
// (a predicated load whose value feeds arithmetic makes hipcc wait for each load in turn: serialised L2
// round trips); validity is a per-thread bit mask applied when the value is scaled.
// scale_RGB happens here, in float32: x * 255 - mean, two roundings like the TF ops (this file is compiled
// with -ffp-contract=off); element e is raw channel (e - 1) mod 21 of its pixel, group g = c / 7 lands in
// output group 2 - g and gets that group's mean; elements outside the image are 0 in the SCALED domain.
// ----------------------------------------------------------------------------------------
// MASKED (eval_train.py:43-45, 53-64: `patches * mask` in front of the CNN): src.mask is ONE plane [B,H,W] -- the projective
// warp of an all-ones image is the same in each of the 18 history channels -- and the raw value of a staged element of
// channel c < 18 is multiplied by its pixel's mask value before the scale: (x * m) * 255 - mean, three roundings like the
// TF ops.  The four elements of a group belong to at most two neighbouring pixels: two mask loads per group and row.
template <int NT, int GROUPS, bool ALIGNED, bool MASKED = false>
struct Conv1Row {
  static constexpr bool kRing = false;
  typedef float floatx4_u __attribute__((ext_vector_type(4), aligned(4)));
  floatx4 nmean[GROUPS];   // minus the channel-group mean of each element: x * 255 + (-mean) == x * 255 - mean exactly
  long idx[GROUPS];
  unsigned col_ok, full;
  bool fast;               // wave-uniform: every element this wave stages lies inside the image row (or behind the segment's
                           // end, where the weights are zero): no per-element masks -- packed multiply / add / f16 convert,
                           // 1.5 VALU per element where the masked path spends ~7 (SQ_INSTS_VALU: 10 per MFMA in the f16 kernel)
  struct Data {            // one input row in flight (prefetched under the MFMAs of the rows before it)
    floatx4 reg[GROUPS];
    float m0[MASKED ? GROUPS : 1], m1[MASKED ? GROUPS : 1];   // mask values of the group's first / second pixel
    bool row_ok;
  };
  const float *img;      // window b
  long row_elems;
  int H;
  const float *mplane;   // MASKED: mask plane of window b
  int mW;
  int mcol0[MASKED ? GROUPS : 1], mcol1[MASKED ? GROUPS : 1];   // columns (clamped into the row) of the group's two pixels
  unsigned m_first, m_second;   // per element: multiplied by m0 / by m1 (neither: channel >= 18, or not an element of the row)

  __device__ __forceinline__ void init(const Conv1Src &src, int tid, int b, int wo0, int H_, int W, int seg_elems) {
    H = H_;
    row_elems = (long)W * kConv1Cin;
    img = static_cast<const float *>(src.base) + (long)b * H * row_elems;
    const long seg0 = (long)(2 * wo0 - 3) * kConv1Cin - 1;
    col_ok = full = 0;
    if constexpr (MASKED) {
      mplane = src.mask + (long)b * H * W;
      mW = W;
      m_first = m_second = 0;
    }
    bool mine = true;
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
      const int e0 = 4 * (tid + NT * i);
      unsigned m = 0;
      if constexpr (MASKED) {
        const long ge0 = seg0 + e0;
        const int c0 = (int)(ge0 >= 0 ? ge0 / kConv1Cin : -((-ge0 + kConv1Cin - 1) / kConv1Cin));   // floor
        mcol0[i] = c0 < 0 ? 0 : (c0 >= W ? W - 1 : c0);
        mcol1[i] = c0 + 1 < 0 ? 0 : (c0 + 1 >= W ? W - 1 : c0 + 1);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const long ge = ge0 + j;
          if (ge < 0 || ge >= row_elems) continue;
          const int col = (int)(ge / kConv1Cin), ch = (int)(ge - (long)col * kConv1Cin);
          if (ch >= kConv1Cin - 3) continue;                     // the newest frame is never masked (eval_train.py:62)
          if (col == c0) m_first |= 1u << (4 * i + j); else m_second |= 1u << (4 * i + j);
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int e = e0 + j;
        const long ge = seg0 + e;
        const int g = ((e + kConv1Cin - 1) % kConv1Cin) / (kConv1Cin / 3);
        nmean[i][j] = g == 0 ? -123.68f : (g == 1 ? -116.779f : -103.939f);
        if (e < seg_elems && ge >= 0 && ge < row_elems) m |= 1u << j;
        if (e < seg_elems && !(ge >= 0 && ge < row_elems)) mine = false;   // a pad column this thread must zero
      }
      col_ok |= m << (4 * i);
      if (m == 15u) full |= 1u << i;
      idx[i] = (ALIGNED && m != 15u) ? 0 : seg0 + e0;   // ALIGNED: a group outside the row reads the row's first
    }
    // (elements behind the segment's end are staged as whatever finite value the clamped load returned: they meet zero weights)
    fast = ALIGNED && __builtin_amdgcn_ballot_w64(!mine) == 0;
  }
  // input row `hi` (clamped into the image; `row_ok` remembers whether it was inside)
  __device__ __forceinline__ void load(Data &d, int hi) const {
    d.row_ok = hi >= 0 && hi < H;
    const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
    const float *xrow = img + (long)hc * row_elems;
    if constexpr (MASKED) {
      const float *mrow = mplane + (long)hc * mW;
#pragma unroll
      for (int i = 0; i < GROUPS; ++i) {
        d.m0[i] = mrow[mcol0[i]];
        d.m1[i] = mrow[mcol1[i]];
      }
    }
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
      if constexpr (ALIGNED) {
        d.reg[i] = *reinterpret_cast<const floatx4 *>(xrow + idx[i]);
      } else if ((full >> i) & 1u) {
        d.reg[i] = *reinterpret_cast<const floatx4_u *>(xrow + idx[i]);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) d.reg[i][j] = ((col_ok >> (4 * i + j)) & 1u) ? xrow[idx[i] + j] : 0.f;
      }
    }
  }
  // group i of a row in flight, scaled
  __device__ __forceinline__ floatx4 scaled(const Data &d, int i) const {
    floatx4 x = d.reg[i];
    if constexpr (MASKED) {   // patches * mask (eval_train.py:64), then scale_RGB; a factor of 1.0f is exact
      floatx4 m4;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        m4[j] = ((m_first >> (4 * i + j)) & 1u) ? d.m0[i] : (((m_second >> (4 * i + j)) & 1u) ? d.m1[i] : 1.0f);
      x = x * m4;
    }
    if (fast) {   // (both conditions are uniform over the wave / the workgroup)
      if (!d.row_ok) return floatx4{0.f, 0.f, 0.f, 0.f};
      return x * 255.0f + nmean[i];
    }
    const unsigned ok = d.row_ok ? col_ok : 0u;
    floatx4 v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = ((ok >> (4 * i + j)) & 1u) ? x[j] * 255.0f + nmean[i][j] : 0.f;
    return v;
  }
};

// ----------------------------------------------------------------------------------------
// The same staged segment assembled from a FRAME RING instead of a window tensor (SURVEY.md 8f-1/-2,
// eval.py:103-104: `np.concatenate` of the 7 window frames fused into this load stage).  The source is a pool
// of RGB frames [n_pool][H][W][3] -- float32 in [0,1], or the raw uint8 frames, whose `/ 255.` (eval.py:80) is
// fused here too -- and window b consists of the pool frames table[b][0..6], oldest to newest.  Element e of the
// staged segment (pixel p = (e - 1) / 21 of the segment, channel c = 3 f + rgb) is float 3 p' + rgb of row
// `hi` of frame f: per frame the segment is ONE contiguous run of 3 * 261 row elements, fetched in groups of
// four from one pixel before the run (a multiple of four elements of the row, like the window path's "one
// element before"), and each element is scattered to its place in the pixel-major LDS image.  So the LDS
// image, the weights and the MFMA loop are the window kernels' own; only the staging differs: the same number
// of global loads (4 elements each), ds_write_b32 / b16 per element instead of b128 per group.
// uint8: f32(double(v) / 255.) * 255.f == (float)v EXACTLY for all 256 byte values (exhaustive:
// tests/test_frames_cpu.py), so the scaled value float(v) - mean is bit-identical to the float path's
// x * 255 - mean on the converted frame, with one conversion and one subtraction per element.
// An index outside [0, n_pool) stages zeros (raw domain), like dvsg_window_gather_f32.
// ----------------------------------------------------------------------------------------
constexpr int C1_RING_PX = 2 * (C1_TILE - 1) + 7;            // 261 pixels of the input row feed a tile
constexpr int C1_RING_RG = (3 * (C1_RING_PX + 1) + 3) / 4;   // 197 groups of four per frame: one pixel of lead-in + the run
constexpr int C1_RING_GROUPS = 7 * C1_RING_RG;               // 1379

// MASKED: as for Conv1Row -- the groups of the six history frames (window slots 0..5) multiply their raw values by the mask
// plane's value at their pixel.  uint8 frames then need x = float32(v / 255.) itself (eval_train.py:119: the frame
// reader's float64 quotient, rounded once by the feed): q = v * r, q' = fma(fma(-q, 255, v), r, q) with r = f32(1 / 255) is
// that value for all 256 bytes (exhaustive: tests/test_frames_cpu.py), and (x * 1.0f) * 255.f == (float)v exactly.
template <int NT, int GROUPS, typename TS, bool ALIGNED, bool MASKED = false>
struct Conv1RingRow {
  static constexpr bool kRing = true;
  static constexpr bool kU8 = sizeof(TS) == 1;
  static_assert(NT * GROUPS >= C1_RING_GROUPS, "not enough threads x groups for 7 frame runs");
  static_assert(GROUPS <= 8, "per-element masks are 32 bits wide");
  typedef float floatx4_u __attribute__((ext_vector_type(4), aligned(4)));
  floatx4 nmean[GROUPS];   // minus the channel-group mean of each element
  bool fast;               // wave-uniform: no pad column and no missing frame among this wave's elements (see Conv1Row)
  long off[GROUPS];        // element offset of the group in the pool, row 0 of its frame
  int l0[GROUPS];          // LDS element of the first value's PIXEL (channel 3 f of it)
  unsigned phase;          // 2 bits per group: rgb of the group's first value
  unsigned col_ok, st_ok;  // per element: inside the image row / inside the staged segment
  unsigned no_frame;       // per group: its window slot names a frame outside the pool (reads as a frame of zeros)
  struct Data {            // one input row in flight
    floatx4 regf[kU8 ? 1 : GROUPS];
    unsigned regu[kU8 ? GROUPS : 1];
    float m0[MASKED ? GROUPS : 1], m1[MASKED ? GROUPS : 1];
    bool row_ok;
  };
  const TS *pool;
  long row_elems;
  int H;
  const float *mplane;   // MASKED: mask plane of window b
  int mW;
  int mcol0[MASKED ? GROUPS : 1], mcol1[MASKED ? GROUPS : 1];
  unsigned m_first, m_second;   // per element: multiplied by m0 / m1 (a group of the newest frame: neither)

  __device__ __forceinline__ void init(const Conv1Src &src, int tid, int b, int wo0, int H_, int W, int /*seg_elems*/) {
    H = H_;
    row_elems = (long)W * 3;
    pool = static_cast<const TS *>(src.base);
    const long frame_elems = (long)H * row_elems;
    const long r_first = (long)(2 * wo0 - 4) * 3;    // one pixel before the segment: a multiple of 4
    col_ok = st_ok = phase = no_frame = 0;
    if constexpr (MASKED) {
      mplane = src.mask + (long)b * H * W;
      mW = W;
      m_first = m_second = 0;
    }
    bool mine = true;
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
      const int q = tid + NT * i;
      const int f = q / C1_RING_RG, g = q - f * C1_RING_RG;
      const bool in_range = q < C1_RING_GROUPS;
      const int fi = in_range ? src.table[b * 7 + f] : -1;
      const bool frame_ok = fi >= 0 && fi < src.n_pool;
      const long r0 = r_first + 4 * g;
      unsigned mc = 0, ms = 0;
      if constexpr (MASKED) {
        const int c0 = (int)(r0 >= 0 ? r0 / 3 : -((-r0 + 2) / 3));   // floor: the group's first pixel
        mcol0[i] = c0 < 0 ? 0 : (c0 >= W ? W - 1 : c0);
        mcol1[i] = c0 + 1 < 0 ? 0 : (c0 + 1 >= W ? W - 1 : c0 + 1);
        if (in_range && f < 6) {                                      // eval_train.py:58,62: the history frames only
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const long r = r0 + j;
            if (r < 0 || r >= row_elems) continue;
            if ((int)(r / 3) == c0) m_first |= 1u << (4 * i + j); else m_second |= 1u << (4 * i + j);
          }
        }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int t = 4 * g + j;                      // element of the run, lead-in pixel included
        const int c = 3 * f + t % 3;
        const int grp = c / (kConv1Cin / 3);
        nmean[i][j] = grp == 0 ? -123.68f : (grp == 1 ? -116.779f : -103.939f);
        if (in_range && t >= 3 && t < 3 * (C1_RING_PX + 1)) ms |= 1u << j;
        if (in_range && r0 + j >= 0 && r0 + j < row_elems) mc |= 1u << j;
      }
      if (in_range && !frame_ok) no_frame |= 1u << i;   // raw value 0 (scaled: -mean), read from the pool's frame 0 and dropped
      if ((ms & ~mc) != 0 || (in_range && !frame_ok)) mine = false;
      // ALIGNED: a group lies inside the row or outside it as a whole (both ends of the row and the run's start are
      // multiples of four elements); outside it reads row `hc` of the pool's first frame and its values are dropped.
      // Element-wise otherwise.
      const bool whole = mc == 15u;
      if (ALIGNED && !whole) mc = 0;
      col_ok |= mc << (4 * i);
      st_ok |= ms << (4 * i);
      off[i] = (ALIGNED && !whole) ? 0 : (frame_ok ? (long)fi * frame_elems : 0) + r0;
      // value j of the group: t = 4 g + j, pixel t / 3 of the run (the lead-in pixel is -1 of the segment), rgb t % 3,
      // LDS element 1 + 21 (t / 3 - 1) + 3 f + t % 3
      l0[i] = 1 + kConv1Cin * ((4 * g) / 3 - 1) + 3 * f;
      phase |= (unsigned)((4 * g) % 3) << (2 * i);
    }
    fast = __builtin_amdgcn_ballot_w64(!mine) == 0;
  }
  __device__ __forceinline__ void load(Data &d, int hi) const {
    d.row_ok = hi >= 0 && hi < H;
    const int hc = hi < 0 ? 0 : (hi >= H ? H - 1 : hi);
    const long roff = (long)hc * row_elems;
    if constexpr (MASKED) {
      const float *mrow = mplane + (long)hc * mW;
#pragma unroll
      for (int i = 0; i < GROUPS; ++i) {
        d.m0[i] = mrow[mcol0[i]];
        d.m1[i] = mrow[mcol1[i]];
      }
    }
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
      const TS *p = pool + off[i] + roff;
      if constexpr (kU8) {
        if constexpr (ALIGNED) {
          d.regu[i] = *reinterpret_cast<const unsigned *>(p);
        } else {
          unsigned w = 0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if ((col_ok >> (4 * i + j)) & 1u) w |= (unsigned)p[j] << (8 * j);
          d.regu[i] = w;
        }
      } else {
        if constexpr (ALIGNED) {
          d.regf[i] = *reinterpret_cast<const floatx4 *>(p);
        } else if (((col_ok >> (4 * i)) & 15u) == 15u) {
          d.regf[i] = *reinterpret_cast<const floatx4_u *>(p);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) d.regf[i][j] = ((col_ok >> (4 * i + j)) & 1u) ? (float)p[j] : 0.f;
        }
      }
    }
  }
  // every staged element of a row in flight, scaled: put(LDS element index, value)
  template <typename PUT>
  __device__ __forceinline__ void scatter(const Data &d, PUT put) const {
    const unsigned ok = d.row_ok ? col_ok : 0u;
    const bool plain = fast && d.row_ok;   // uniform: no masks to apply
#pragma unroll
    for (int i = 0; i < GROUPS; ++i) {
      const int ph = (phase >> (2 * i)) & 3u;
      floatx4 raw4;
      if constexpr (MASKED) {
        floatx4 m4, x;
#pragma unroll
        for (int j = 0; j < 4; ++j)
          m4[j] = ((m_first >> (4 * i + j)) & 1u) ? d.m0[i] : (((m_second >> (4 * i + j)) & 1u) ? d.m1[i] : 1.0f);
        if constexpr (kU8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float v = (float)((d.regu[i] >> (8 * j)) & 255u);
            const float q = v * (1.0f / 255.0f);
            x[j] = __builtin_fmaf(__builtin_fmaf(-q, 255.0f, v), 1.0f / 255.0f, q);   // f32(v / 255.), correctly rounded
          }
        } else {
          x = d.regf[i];
        }
        raw4 = (x * m4) * 255.0f;
      } else if constexpr (kU8) {
#pragma unroll
        for (int j = 0; j < 4; ++j) raw4[j] = (float)((d.regu[i] >> (8 * j)) & 255u);   // == f32(v / 255.) * 255.f, exactly
      } else {
        raw4 = d.regf[i] * 255.0f;
      }
      const floatx4 sc4 = raw4 + nmean[i];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v;
        if (plain) {
          v = sc4[j];
        } else {
          const float raw = ((no_frame >> i) & 1u) ? 0.f : raw4[j];
          v = ((ok >> (4 * i + j)) & 1u) ? raw + nmean[i][j] : 0.f;
        }
        const int t = ph + j, step = t / 3;                              // t in 0..5
        if ((st_ok >> (4 * i + j)) & 1u) put(l0[i] + kConv1Cin * step + (t - 3 * step), v);
      }
    }
  }
};

// SRC of the conv1 kernels: bit 0 = rows on the 16-byte (4-byte for uint8) grid; 0/1 window tensor, 2/3 float32 frame
// ring, 4/5 uint8 frame ring; + 8: the history channels are multiplied by src.mask (eval_train.py's graph).
template <int NT, int GROUPS, int SRC>
struct Conv1RowSel {
  static_assert((SRC & 6) == 0, "ring sources are specialised below");
  typedef Conv1Row<NT, GROUPS, (SRC & 1) != 0, (SRC & 8) != 0> type;
};
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 10> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, float, false, true> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 11> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, float, true, true> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 12> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, uint8_t, false, true> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 13> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, uint8_t, true, true> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 2> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, float, false> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 3> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, float, true> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 4> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, uint8_t, false> type; };
template <int NT, int GROUPS>
struct Conv1RowSel<NT, GROUPS, 5> { typedef Conv1RingRow<NT, (C1_RING_GROUPS + NT - 1) / NT, uint8_t, true> type; };

#ifdef DVSG_STAMPS  // diagnostic build (tools/stamp_probe_conv1.py): per-workgroup phase times of conv1_kernel
__device__ unsigned long long g_c1_stamps[8 * 65536];
#define C1_STAMP() __builtin_amdgcn_s_memtime()
#endif

template <int NW, typename TO, int SRC>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW / 2, NW / 2)))
void conv1_kernel(const Conv1Src src, const float *__restrict__ wt1, const float *__restrict__ bias,
                  TO *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles) {
  constexpr int NT = 64 * NW;
  constexpr int MI = 4 / (NW / 2);                       // 32-pixel MFMA blocks per wave: 2 or 1
  constexpr int IN4 = (C1_SEG_PAD / 4 + NT - 1) / NT;    // 6 or 3 groups of four elements per thread per kernel row
  constexpr int WLOADS = (C1_WELEMS / 4 + NT - 1) / NT;  // 10 or 5 float4
  static_assert(C1_TILE * C1_LDC <= C1_WELEMS, "epilogue tile must fit in the weight stage");
  __shared__ __attribute__((aligned(16))) float w_s[C1_WELEMS];
  __shared__ __attribute__((aligned(16))) float in_s[C1_SEG_PAD];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;

  // each XCD takes a contiguous range of tiles: the 3-4 output rows that share an input row run on
  // the same L2 (consecutive block ids are dealt round-robin over the 8 XCDs)
  int blk = xcd_remap(blockIdx.x, gridDim.x);
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho = blk % Ho;
  const int b = blk / Ho;
  const int wo0 = wt_i * C1_TILE;

  typedef typename Conv1RowSel<NT, IN4, SRC>::type Row;
  Row row;
  row.init(src, tid, b, wo0, H, W, C1_SEG_PAD);
  if constexpr (Row::kRing) {   // elements the scatter never writes (the zero tap's slot, the tail) must be finite
    for (int e = tid; e < C1_SEG_PAD; e += NT) in_s[e] = 0.f;
  }
  typename Row::Data rd;
  floatx4 w_reg[WLOADS];
  auto load_stage = [&](int kh) __attribute__((always_inline)) {
    row.load(rd, 2 * ho + kh - 3);
    const floatx4 *wsrc = reinterpret_cast<const floatx4 *>(wt1 + (size_t)kh * C1_WELEMS);
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      w_reg[i] = wsrc[q < C1_WELEMS / 4 ? q : 0];
    }
  };
  auto store_stage = [&]() __attribute__((always_inline)) {
    if constexpr (Row::kRing) {
      row.scatter(rd, [&](int e, float v) __attribute__((always_inline)) { in_s[e] = v; });
    } else {
#pragma unroll
      for (int i = 0; i < IN4; ++i) {
        const int q = tid + NT * i;
        const floatx4 v = row.scaled(rd, i);
        if (q < C1_SEG_PAD / 4) reinterpret_cast<floatx4 *>(in_s)[q] = v;
      }
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      if (q < C1_WELEMS / 4) reinterpret_cast<floatx4 *>(w_s)[q] = w_reg[i];
    }
  };

  floatx16 acc[MI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[mi][q] = 0.f;

#ifdef DVSG_STAMPS
  unsigned long long st[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  const unsigned long long t_begin = C1_STAMP(), r_begin = __builtin_amdgcn_s_memrealtime();
#endif
  load_stage(0);
#ifdef DVSG_STAMPS
  st[0] = C1_STAMP() - t_begin;  // prologue (index setup) + first load issue
#endif
  for (int kh = 0; kh < 7; ++kh) {
#ifdef DVSG_STAMPS  // (each stamp drains lgkmcnt: the phase times below are upper bounds)
    const unsigned long long s0 = C1_STAMP();
    __syncthreads();
    const unsigned long long s1 = C1_STAMP();
    store_stage();
    const unsigned long long s2 = C1_STAMP();
    __syncthreads();
    const unsigned long long s3 = C1_STAMP();
    if (kh + 1 < 7) load_stage(kh + 1);
    const unsigned long long s4 = C1_STAMP();
    st[1] += s1 - s0;  // barrier 1 (+ vmcnt(0): the prefetched row and weights)
    st[2] += s2 - s1;  // scale + LDS stores
    st[3] += s3 - s2;  // barrier 2
    st[4] += s4 - s3;  // issue of the next row's loads
#else
    __syncthreads();  // everyone is done reading the previous kernel row
    store_stage();
    __syncthreads();
    if (kh + 1 < 7) load_stage(kh + 1);
#endif
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA loop
    const float *a0 = in_s + 2 * kConv1Cin * (wm * 32 * MI + r) + 2 * h;
    const float *bp = w_s + (wn * 32 + r) * kConv1Ld + 2 * h;
    // software-pipelined: the operands of step u + 1 are requested before the MFMAs of step u are issued (the
    // compiler's own schedule waits for each step's reads with nothing else in flight)
    constexpr int STEPS = kConv1Kpad / 4, PAIRS = (STEPS + 1) / 2;   // two steps per request: one ds_read2_b64 per operand
    float2 va[2][2][MI], vb[2][2];
    auto read_pair = [&](int slot, int g) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (2 * g + j >= STEPS) break;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
          va[slot][j][mi] = *reinterpret_cast<const float2 *>(a0 + mi * (2 * kConv1Cin * 32) + 4 * (2 * g + j));
        vb[slot][j] = *reinterpret_cast<const float2 *>(bp + 4 * (2 * g + j));
      }
    };
    read_pair(0, 0);
#pragma unroll
    for (int g = 0; g < PAIRS; ++g) {
      if (g + 1 < PAIRS) read_pair((g + 1) & 1, g + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        if (2 * g + j >= STEPS) break;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma32(va[g & 1][j][mi].x, vb[g & 1][j].x, acc[mi]);
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) acc[mi] = mfma32(va[g & 1][j][mi].y, vb[g & 1][j].y, acc[mi]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

#ifdef DVSG_STAMPS
  const unsigned long long t_loop_end = C1_STAMP();
#endif
  // ---- epilogue: transpose through the (idle) weight stage so stores are float4s along channels
  float *Cs = w_s;
  __syncthreads();
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wm * 32 * MI + mi * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + wn * 32 + r] = acc[mi][q];
  __syncthreads();
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
  TO *yrow = y + (((size_t)b * Ho + ho) * Wo + wo0) * 64 + 4 * col4;
#pragma unroll 4
  for (int row = row0; row < C1_TILE; row += NT / 16) {
    if (wo0 + row >= Wo) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * C1_LDC + 4 * col4);
    v.x = fmaxf(v.x + b4.x, 0.f);
    v.y = fmaxf(v.y + b4.y, 0.f);
    v.z = fmaxf(v.z + b4.z, 0.f);
    v.w = fmaxf(v.w + b4.w, 0.f);
    store4(yrow + (size_t)row * 64, v);
  }
#ifdef DVSG_STAMPS
  if (tid == 0 && blockIdx.x < 65536) {
    unsigned long long *o = g_c1_stamps + (size_t)blockIdx.x * 8;
    const unsigned long long t_end = C1_STAMP();
    o[0] = st[0]; o[1] = st[1]; o[2] = st[2]; o[3] = st[3]; o[4] = st[4];
    o[5] = t_loop_end - t_begin;            // up to the end of the last MFMA loop
    o[6] = t_end - t_begin;                 // lifetime (stores issued, not necessarily landed)
    o[7] = __builtin_amdgcn_s_memrealtime() - r_begin;
  }
#endif
}

// ----------------------------------------------------------------------------------------
// 3x3 / stride 2 max pool with TF 'SAME' padding (pad_before = pad_total / 2: nothing on
// the top/left for even sizes).  One thread = one output pixel x 4 channels.
// ----------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void maxpool_kernel(const T *__restrict__ x, T *__restrict__ y, int H, int W,
                                                     int C4, int Ho, int Wo, int pad_top, int pad_left,
                                                     size_t total) {
  // one element per thread; each XCD takes a contiguous range of blocks so that the output rows
  // sharing an input row (3x3 window, stride 2) read it out of the same L2
  {
    const size_t e = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (e >= total) return;
    const int c4 = (int)(e % C4);
    size_t t = e / C4;
    const int wo = (int)(t % Wo);
    t /= Wo;
    const int ho = (int)(t % Ho);
    const size_t b = t / Ho;
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int hi = 2 * ho - pad_top + i;
      if (hi < 0 || hi >= H) continue;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const int wi = 2 * wo - pad_left + j;
        if (wi < 0 || wi >= W) continue;
        const float4 v = load4(x + (((b * H + hi) * W + wi) * C4 + c4) * 4);
        m.x = fmaxf(m.x, v.x);
        m.y = fmaxf(m.y, v.y);
        m.z = fmaxf(m.z, v.z);
        m.w = fmaxf(m.w, v.w);
      }
    }
    store4(y + e * 4, m);
  }
}

// float16 tensors, 8 channels = 16 bytes per thread: with 4 channels a thread moves 8-byte pieces and the kernel ran
// at 3.2 TB/s of algorithmic traffic where the float32 one reaches 5.1 (comparisons in float16 are exact: max of
// representable values)
__global__ __launch_bounds__(256) void maxpool_h8_kernel(const _Float16 *__restrict__ x, _Float16 *__restrict__ y, int H,
                                                         int W, int C8, int Ho, int Wo, int pad_top, int pad_left,
                                                         size_t total) {
  const size_t e = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int c8 = (int)(e % C8);
  size_t t = e / C8;
  const int wo = (int)(t % Wo);
  t /= Wo;
  const int ho = (int)(t % Ho);
  const size_t b = t / Ho;
  halfx8 m;
#pragma unroll
  for (int q = 0; q < 8; ++q) m[q] = (_Float16)(-INFINITY);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hi = 2 * ho - pad_top + i;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int wi = 2 * wo - pad_left + j;
      if (wi < 0 || wi >= W) continue;
      const halfx8 v = *reinterpret_cast<const halfx8 *>(x + (((b * H + hi) * W + wi) * C8 + c8) * 8);
#pragma unroll
      for (int q = 0; q < 8; ++q) m[q] = v[q] > m[q] ? v[q] : m[q];
    }
  }
  *reinterpret_cast<halfx8 *>(y + e * 8) = m;
}

// ----------------------------------------------------------------------------------------
// conv1 for the float16 precision: same decomposition (one output row x 128 pixels x 64 channels
// per workgroup, the input row segment staged once per kernel row and read in place as
// overlapping windows), but the scaled input is stored in LDS as float16 and multiplied on
// v_mfma_f32_32x32x16_f16.  A lane's operand is 8 consecutive taps = 16 bytes at byte offset
// 84 pixel + 32 step + 16 h: only 4-byte aligned, so it is fetched as four ds_read_b32
// (21 r mod 32 is a bijection: conflict-free); the weight rows are laid out with a 336-byte
// stride (21 x 16 B) for conflict-free ds_read_b128 and arrive by LDS-DMA.  Both LDS images are
// double-buffered: kernel row kh+1 is fetched while kh is multiplied, one barrier per row.
// The 147 taps of a row are padded to 160 (ten 16-tap MFMA steps) with zero weights.
// ----------------------------------------------------------------------------------------
constexpr int C1H_STEPS = 10;
constexpr int C1H_LD = 168;                        // halfs per weight row in LDS / global
constexpr int C1H_WBYTES = 64 * C1H_LD * 2;        // 21504 bytes per kernel row = 21 LDS-DMA pieces
constexpr int C1H_SEG = 5504;                      // halfs per staged input row (5481 + zero tail)
constexpr int C1H_IN4 = (C1H_SEG / 4 + 255) / 256;     // 6 groups of four elements per thread

template <typename TO, int SRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_f16_kernel(const Conv1Src src, const _Float16 *__restrict__ wt1h, const float *__restrict__ bias,
                      TO *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles) {
  static_assert(C1_TILE * C1_LDC * 4 <= 2 * C1H_WBYTES, "epilogue tile must fit in the weight stages");
  __shared__ __attribute__((aligned(16))) char lds[2 * C1H_WBYTES + 2 * C1H_SEG * 2];
  char *w_s = lds;
  char *in_s = lds + 2 * C1H_WBYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;   // wave w: pixels [32 w, 32 w + 32) x all 64 channels (see conv1_split_kernel)

  int blk = xcd_remap(blockIdx.x, gridDim.x);  // see conv1_kernel
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho = blk % Ho;
  const int b = blk / Ho;
  const int wo0 = wt_i * C1_TILE;

  typedef typename Conv1RowSel<256, C1H_IN4, SRC>::type Row;
  Row row;
  row.init(src, tid, b, wo0, H, W, C1H_SEG);
  if constexpr (Row::kRing) {   // both images: the zero tap's slot and the tail are read against zero weights
    for (int e = tid; e < C1H_SEG; e += 256) reinterpret_cast<unsigned *>(in_s)[e] = 0u;
    __syncthreads();   // the first scatter writes halves of words other threads zero: the zeros must land first
  }

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  typename Row::Data rd;
  auto load_stage = [&](int kh, int buf) __attribute__((always_inline)) {
    // weights of kernel row kh: 21 pieces of 1 KiB, piece j by wave j % 4, straight into LDS
    const char *wsrc = reinterpret_cast<const char *>(wt1h) + (size_t)kh * C1H_WBYTES;
    for (int j = wave; j < C1H_WBYTES / 1024; j += 4)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + j * 1024 + lane * 16), (lptr_t)(w_s + buf * C1H_WBYTES + j * 1024),
                                       16, 0, 0);
    row.load(rd, 2 * ho + kh - 3);
  };
  auto store_stage = [&](int buf) __attribute__((always_inline)) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    if constexpr (Row::kRing) {
      _Float16 *dsth = reinterpret_cast<_Float16 *>(in_s + buf * C1H_SEG * 2);
      row.scatter(rd, [&](int e, float v) __attribute__((always_inline)) { dsth[e] = (_Float16)v; });
    } else {
      half4_t *dst = reinterpret_cast<half4_t *>(in_s + buf * C1H_SEG * 2);
#pragma unroll
      for (int i = 0; i < C1H_IN4; ++i) {
        const int q = tid + 256 * i;
        const floatx4 v = row.scaled(rd, i);   // then one rounding to f16
        half4_t hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[j] = (_Float16)v[j];
        if (q < C1H_SEG / 4) dst[q] = hv;
      }
    }
  };

  floatx16 acc[2];   // [channel block]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[ni][q] = 0.f;

  load_stage(0, 0);
  store_stage(0);
  __syncthreads();
  for (int kh = 0; kh < 7; ++kh) {
    const int buf = kh & 1;
    if (kh + 1 < 7) load_stage(kh + 1, buf ^ 1);
    __builtin_amdgcn_sched_barrier(0);
    const unsigned *a0 = reinterpret_cast<const unsigned *>(in_s + buf * C1H_SEG * 2) + kConv1Cin * (wave * 32 + r) + 4 * h;
    const char *bp = w_s + buf * C1H_WBYTES + 2 * (r * C1H_LD + 8 * h);
    // software-pipelined one step ahead, like conv1_split_kernel
    struct Frag {
      union {
        unsigned u[4];
        halfx8 v;
      } a;
      halfx8 b[2];
    };
    auto read_frag = [&](Frag &f, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) f.a.u[j] = a0[8 * t + j];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.b[ni] = *reinterpret_cast<const halfx8 *>(bp + 2 * ni * 32 * C1H_LD + 32 * t);
    };
    Frag fr[2];
    read_frag(fr[0], 0);
#pragma unroll
    for (int t = 0; t < C1H_STEPS; ++t) {
      if (t + 1 < C1H_STEPS) read_frag(fr[(t + 1) & 1], t + 1);
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t & 1].a.v, fr[t & 1].b[0], acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fr[t & 1].a.v, fr[t & 1].b[1], acc[1], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (kh + 1 < 7) store_stage(buf ^ 1);
    __syncthreads();
  }

  float *Cs = reinterpret_cast<float *>(lds);
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + ni * 32 + r] = acc[ni][q];
  __syncthreads();
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
  TO *yrow = y + (((size_t)b * Ho + ho) * Wo + wo0) * 64 + 4 * col4;
#pragma unroll 4
  for (int row = row0; row < C1_TILE; row += 16) {
    if (wo0 + row >= Wo) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * C1_LDC + 4 * col4);
    v.x = fmaxf(v.x + b4.x, 0.f);
    v.y = fmaxf(v.y + b4.y, 0.f);
    v.z = fmaxf(v.z + b4.z, 0.f);
    v.w = fmaxf(v.w + b4.w, 0.f);
    store4(yrow + (size_t)row * 64, v);
  }
}

// ----------------------------------------------------------------------------------------
// conv1, float16 precision, TWO output rows per workgroup (rows 2 p and 2 p + 1 of one 128-pixel tile).
//
// Why.  rocprofv3 counters of conv1_f16_kernel at 3840x2160, batch 32 (profiles/r02_f16_pmc_sq.csv): matrix pipe busy
// 27 %, SQ_ACTIVE_INST_ANY / SQ_WAVE_CYCLES = 0.44 with two waves per SIMD -- the SIMDs issue instructions 88 % of the time:
// the kernel is bound by INSTRUCTION ISSUE (and, at 304 KB of L2 -> CU traffic per tile, close to the L2's rate), not by
// the matrix cores or HBM (FETCH_SIZE = the input once).  Per kernel row and wave it issues ~110 VALU (scale_RGB, the
// float16 rounding), 60 LDS fragment reads, 11 memory instructions and 20 MFMAs: 11 other instructions per MFMA.
// Two rows per workgroup attack every term:
//   * the weights of a kernel row (21.5 KB by LDS-DMA) and their fragment reads serve both rows: per MFMA half the weight
//     traffic and half the B reads;
//   * the kernel rows are visited as two chains, kh = 0, 2, 4, 6 and kh = 1, 3, 5.  Within a chain the input row that
//     output row 2 p + 1 needs at step kh (row a + kh + 2) is the one output row 2 p needs at step kh + 2: it stays where
//     it is, and only ONE new input row is staged per step -- 9 stagings per pair of output rows instead of 14;
//   * a step is 40 MFMAs, so the two barriers and the address arithmetic of a step cost half as much per MFMA.
// Staging as in conv1_kernel: the next step's new row(s) are fetched into registers under this step's MFMAs and, after
// the barrier that ends the step, scaled, rounded and stored over the row nobody needs any more; the weights of the next
// step arrive by LDS-DMA in the other weight buffer meanwhile.  LDS: 2 x 21.5 KB of weights + 2 x 11 KB of rows, two
// workgroups per CU, as before.
// ----------------------------------------------------------------------------------------
template <typename TO, int SRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_f16_pair_kernel(const Conv1Src src, const _Float16 *__restrict__ wt1h, const float *__restrict__ bias,
                           TO *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles, int hpairs) {
  static_assert(C1_TILE * C1_LDC * 4 <= 2 * C1H_WBYTES, "epilogue tile must fit in the weight stages");
  __shared__ __attribute__((aligned(16))) char lds[2 * C1H_WBYTES + 2 * C1H_SEG * 2];
  char *w_s = lds;
  char *in_s = lds + 2 * C1H_WBYTES;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;   // wave w: pixels [32 w, 32 w + 32) x all 64 channels x both rows

  int blk = xcd_remap(blockIdx.x, gridDim.x);  // see conv1_kernel
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho0 = 2 * (blk % hpairs);
  const int b = blk / hpairs;
  const int wo0 = wt_i * C1_TILE;

  typedef typename Conv1RowSel<256, C1H_IN4, SRC>::type Row;
  Row row;
  row.init(src, tid, b, wo0, H, W, C1H_SEG);
  if constexpr (Row::kRing) {   // both images: the zero tap's slot and the tail are read against zero weights
    for (int e = tid; e < C1H_SEG; e += 256) reinterpret_cast<unsigned *>(in_s)[e] = 0u;
    __syncthreads();   // the first scatter writes halves of words other threads zero: the zeros must land first
  }
  typename Row::Data d0, d1;

  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;
  auto weights_dma = [&](int kh, int wbuf) __attribute__((always_inline)) {
    const char *wsrc = reinterpret_cast<const char *>(wt1h) + (size_t)kh * C1H_WBYTES;
    for (int j = wave; j < C1H_WBYTES / 1024; j += 4)
      __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + j * 1024 + lane * 16), (lptr_t)(w_s + wbuf * C1H_WBYTES + j * 1024),
                                       16, 0, 0);
  };
  auto store_row = [&](const typename Row::Data &d, int buf) __attribute__((always_inline)) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    if constexpr (Row::kRing) {
      _Float16 *dsth = reinterpret_cast<_Float16 *>(in_s + buf * C1H_SEG * 2);
      row.scatter(d, [&](int e, float v) __attribute__((always_inline)) { dsth[e] = (_Float16)v; });
    } else {
      half4_t *dst = reinterpret_cast<half4_t *>(in_s + buf * C1H_SEG * 2);
#pragma unroll
      for (int i = 0; i < C1H_IN4; ++i) {
        const int q = tid + 256 * i;
        const floatx4 v = row.scaled(d, i);   // then one rounding to f16
        half4_t hv;
#pragma unroll
        for (int j = 0; j < 4; ++j) hv[j] = (_Float16)v[j];
        if (q < C1H_SEG / 4) dst[q] = hv;
      }
    }
  };

  floatx16 acc[2][2];   // [output row][channel block]
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][ni][q] = 0.f;

  // input row of (output row ho0 + j, kernel row kh)
  const int a = 2 * ho0 - 3;
  // prologue: rows a, a + 2 and the weights of kernel row 0
  row.load(d0, a);
  row.load(d1, a + 2);
  weights_dma(0, 0);
  store_row(d0, 0);
  store_row(d1, 1);
  __syncthreads();
  int lo = 0;   // buffer of output row ho0's input row; the other one holds output row ho0 + 1's
  // (seven explicit instances: left to `#pragma unroll` the optimizer gives up on this body and the kernel-row order
  // becomes a table lookup)
  auto step = [&](auto S) __attribute__((always_inline)) {
    constexpr int s = decltype(S)::value;
    constexpr int kOrder[8] = {0, 2, 4, 6, 1, 3, 5, -1};
    constexpr int kh = kOrder[s], khn = kOrder[s + 1];
    constexpr bool chain = khn == kh + 2;        // the next step continues this chain: one new row
    if (khn >= 0) {
      row.load(d0, chain ? a + khn + 2 : a + khn);
      if (!chain) row.load(d1, a + khn + 2);
      weights_dma(khn, (s + 1) & 1);
    }
    __builtin_amdgcn_sched_barrier(0);
    const unsigned *a0 = reinterpret_cast<const unsigned *>(in_s + lo * C1H_SEG * 2) + kConv1Cin * (wave * 32 + r) + 4 * h;
    const unsigned *a1 = reinterpret_cast<const unsigned *>(in_s + (lo ^ 1) * C1H_SEG * 2) + kConv1Cin * (wave * 32 + r) + 4 * h;
    const char *bp = w_s + (s & 1) * C1H_WBYTES + 2 * (r * C1H_LD + 8 * h);
    struct Frag {
      union {
        unsigned u[4];
        halfx8 v;
      } a[2];
      halfx8 b[2];
    };
    auto read_frag = [&](Frag &f, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f.a[0].u[j] = a0[8 * t + j];
        f.a[1].u[j] = a1[8 * t + j];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.b[ni] = *reinterpret_cast<const halfx8 *>(bp + 2 * ni * 32 * C1H_LD + 32 * t);
    };
    Frag fr[2];
    read_frag(fr[0], 0);
#pragma unroll
    for (int t = 0; t < C1H_STEPS; ++t) {
      if (t + 1 < C1H_STEPS) read_frag(fr[(t + 1) & 1], t + 1);
      __builtin_amdgcn_sched_barrier(0);
      const Frag &f = fr[t & 1];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.a[j].v, f.b[ni], acc[j][ni], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();   // everyone has read this step's rows; the prefetched row(s) and the next weights have landed
    if (khn >= 0) {
      if (chain) {
        store_row(d0, lo);   // over the row of output row ho0: the other one becomes its row for the next step
        lo ^= 1;
      } else {
        store_row(d0, lo);
        store_row(d1, lo ^ 1);
      }
      __syncthreads();
    }
  };
  step(std::integral_constant<int, 0>{});
  step(std::integral_constant<int, 1>{});
  step(std::integral_constant<int, 2>{});
  step(std::integral_constant<int, 3>{});
  step(std::integral_constant<int, 4>{});
  step(std::integral_constant<int, 5>{});
  step(std::integral_constant<int, 6>{});

  float *Cs = reinterpret_cast<float *>(lds);
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if (j) __syncthreads();   // the first row's tile has been read
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q)
        Cs[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + ni * 32 + r] = acc[j][ni][q];
    __syncthreads();
    if (ho0 + j < Ho) {
      TO *yrow = y + (((size_t)b * Ho + ho0 + j) * Wo + wo0) * 64 + 4 * col4;
#pragma unroll 4
      for (int rw = row0; rw < C1_TILE; rw += 16) {
        if (wo0 + rw >= Wo) break;
        float4 v = *reinterpret_cast<const float4 *>(Cs + rw * C1_LDC + 4 * col4);
        v.x = fmaxf(v.x + b4.x, 0.f);
        v.y = fmaxf(v.y + b4.y, 0.f);
        v.z = fmaxf(v.z + b4.z, 0.f);
        v.w = fmaxf(v.w + b4.w, 0.f);
        store4(yrow + (size_t)rw * 64, v);
      }
    }
  }
}

// ----------------------------------------------------------------------------------------
// conv1, float16 precision, for the big launches: a workgroup MARCHES down a band of output rows of one 128-pixel
// column tile, four output rows ("quad") at a time, and its eight waves are SPECIALISED --
//
//   waves 0-3  multiply: pixels [32 w, 32 w + 32) x 64 channels x the quad's 4 rows (128 accumulator registers);
//              their instruction stream is fragment reads + MFMAs only (80 MFMAs per step);
//   waves 4-7  stage: fetch the input rows the NEXT step needs, scale_RGB them, round to float16, store them in LDS,
//              and start the LDS-DMA of the next step's weight row.
//
// A wave of each kind shares a SIMD, so the staging arithmetic issues in the shadow of the matrix pipe instead of
// between its MFMAs: conv1_f16_pair_kernel was bound by instruction ISSUE (SQ_ACTIVE_INST_ANY = 0.34 of the wave cycles
// with two waves per SIMD, 6.8 VALU + 1.9 LDS + 3 scalar instructions per MFMA, matrix pipe busy 35 %).
//
// Marching makes the staged rows a SLIDING WINDOW.  A step is one kernel row kh for the quad's four output rows
// r0 .. r0 + 3, which need the input rows a + kh + {0, 2, 4, 6} (a = 2 r0 - 3).  The steps of a quad run as two chains,
// kh = 0, 2, 4, 6 and kh = 1, 3, 5; a chain's window moves up by one input row of its parity per step and -- because the
// next quad starts 8 input rows further down -- continues seamlessly from quad to quad.  So every input row is staged
// ONCE per column tile (8 stagings per quad = 2 per output row; the one-row kernel staged 7, the pair kernel 4.5), into
// one of two rings of five row buffers (odd and even input rows): four rows of the current window + the one being
// filled.  The weights of a kernel row serve four output rows (a quarter of the weight traffic and B-fragment reads per
// MFMA of the one-row kernel).  One barrier per step is the only synchronisation: what step s + 1 needs is staged during
// step s.
//
// The MFMA operands are swapped against the other conv1 kernels (A = weights, B = pixels), so a lane ends up with 16
// CHANNELS of one pixel: bias, ReLU, the float16 rounding and 8-byte stores happen from the accumulators, with no LDS
// transpose (there is no LDS left for one: 10 x 11 KB of rows + 2 x 21.5 KB of weights = 153 KB, one workgroup per CU).
// Used when the launch gives every CU several bands (launch_conv1); small launches keep conv1_f16_pair_kernel.
// ----------------------------------------------------------------------------------------
constexpr int C1M_QUAD = 4;                         // output rows per quad
constexpr int C1M_RING = 5;                         // row buffers per parity
constexpr int C1M_LDS = 2 * C1M_RING * C1H_SEG * 2 + 2 * C1H_WBYTES;   // 153 088 bytes

template <int SRC>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_f16_march_kernel(const Conv1Src src, const _Float16 *__restrict__ wt1h, const float *__restrict__ bias,
                            _Float16 *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles, int bands, int quads_per_band) {
  __shared__ __attribute__((aligned(16))) char lds[C1M_LDS];
  char *w_s = lds;                                   // [2][C1H_WBYTES]
  char *in_s = lds + 2 * C1H_WBYTES;                 // [2 parities][C1M_RING][C1H_SEG halves]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const bool stager = wave >= 4;
  const int r = lane & 31, h = lane >> 5;

  int blk = xcd_remap(blockIdx.x, gridDim.x);        // an XCD takes neighbouring column tiles / bands of one image
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int band = blk % bands;
  const int b = blk / bands;
  const int wo0 = wt_i * C1_TILE;
  const int rb0 = band * quads_per_band * C1M_QUAD;                       // first output row of the band
  const int nq = min(quads_per_band, (Ho - rb0 + C1M_QUAD - 1) / C1M_QUAD);
  const int nsteps = 7 * nq;
  // step s: quad s / 7, kernel row kOrder[s % 7]; its window of input rows starts at 2 (rb0 + 4 quad) - 3 + kh
  auto window_of = [&](int s) __attribute__((always_inline)) -> int {
    const int quad = s / 7, ks = s - 7 * quad;
    const int kh = ks < 4 ? 2 * ks : 2 * ks - 7;
    return 2 * (rb0 + C1M_QUAD * quad) - 3 + kh;
  };
  // ring slot (in halves from in_s) of input row `row`: odd rows (the even-kh chain) in the first ring
  auto slot_of = [&](int row) __attribute__((always_inline)) -> int {
    const int u = (row + 64) >> 1;                   // rows >= -3: non-negative
    return (((row & 1) ? 0 : C1M_RING) + u % C1M_RING) * C1H_SEG;
  };
  typedef const __attribute__((address_space(1))) void *gptr_t;
  typedef __attribute__((address_space(3))) void *lptr_t;

  if (stager) {
    // ------------------------------------------------------------------------------ staging waves
    // Their few instructions go first when both waves of a SIMD are ready (s_setprio): with the default oldest-first
    // arbitration the staging wave -- the younger one -- only issues while the multiplying wave is stalled, and a step takes
    // 5550 cycles instead of 4950 (stamps, tools/stamp_probe_march.py).  The multiplying waves raise their own priority for
    // the flush, where they have no MFMAs to hide behind.
    __builtin_amdgcn_s_setprio(2);
    const int st = tid - 256;
    typedef typename Conv1RowSel<256, C1H_IN4, SRC>::type Row;
    Row row;
    row.init(src, st, b, wo0, H, W, C1H_SEG);
    if constexpr (Row::kRing) {   // the zero tap's slot and the tails of all ten row buffers meet zero weights: finite
      for (int e = st; e < C1M_RING * C1H_SEG; e += 256) reinterpret_cast<unsigned *>(in_s)[e] = 0u;
    }
    __syncthreads();              // (every wave of the workgroup: nobody's first row store may be overtaken by the zeroing)
    auto weights_dma = [&](int kh, int wbuf) __attribute__((always_inline)) {
      const char *wsrc = reinterpret_cast<const char *>(wt1h) + (size_t)kh * C1H_WBYTES;
#pragma unroll
      for (int i = 0; i < (C1H_WBYTES / 1024 + 3) / 4; ++i) {   // 21 pieces of 1 KiB over the 4 staging waves
        const int j = wave - 4 + 4 * i;
        if (j < C1H_WBYTES / 1024)
          __builtin_amdgcn_global_load_lds((gptr_t)(wsrc + j * 1024 + lane * 16), (lptr_t)(w_s + wbuf * C1H_WBYTES + j * 1024),
                                           16, 0, 0);
      }
    };
    auto store_row = [&](const typename Row::Data &dd, int input_row) __attribute__((always_inline)) {
      typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
      _Float16 *base = reinterpret_cast<_Float16 *>(in_s) + slot_of(input_row);
      if constexpr (Row::kRing) {
        // element 0 is the zero tap's slot: the scatter never writes it, but `flush_quad` below transposes conv1 OUTPUTS
        // through halves [0, 5120) of the even ring slots -- a stale output there meets a zero weight and must be finite;
        // a float16 output that overflowed to inf would turn the tile's first pixel of a later quad into NaN.  One lane
        // rewrites it with every staged row (the window path below rewrites the whole segment anyway).
        if (st == 0) base[0] = (_Float16)0.f;
        row.scatter(dd, [&](int e, float v) __attribute__((always_inline)) { base[e] = (_Float16)v; });
      } else {
        half4_t *dst = reinterpret_cast<half4_t *>(base);
#pragma unroll
        for (int i = 0; i < C1H_IN4; ++i) {
          const int q = st + 256 * i;
          const floatx4 v = row.scaled(dd, i);
          if (q < C1H_SEG / 4) dst[q] = __builtin_convertvector(v, half4_t);
        }
      }
    };
    // What a step needs that the steps before it have not staged: the top n rows of its window.  The first step of a chain
    // in the band's FIRST quad: the whole window (4 rows); kh = 1 of a later quad: two rows (its chain has three steps for
    // four output rows); every other step: one row.  ks is a compile-time constant wherever it matters (14 unrolled steps
    // per loop iteration): the per-step code has no data-dependent branches left but "first quad of the band?".
    auto row_base = [&](int quad) __attribute__((always_inline)) -> int { return 2 * (rb0 + C1M_QUAD * quad) - 3; };
    // The staging is pipelined TWO steps deep: during step s the rows of step s + 1 -- fetched during step s - 1 -- are
    // scaled and stored, and the global loads of step s + 2's rows are issued (into the other of two register sets), so a
    // row has a whole step (~1.3 us) to arrive.  One step deep, every step waited for the memory latency of its own rows.
    typename Row::Data setA[2], setB[2];
    // fetch (into a register set) the top rows of the window of step (quad, KS); returns how many (<= 2)
    auto issue = [&](auto KS, int quad, typename Row::Data *set) __attribute__((always_inline)) -> int {
      constexpr int ks = decltype(KS)::value;
      constexpr int kh = ks < 4 ? 2 * ks : 2 * ks - 7;
      const int top = row_base(quad) + kh + 6;
      row.load(set[0], top);
      if (ks == 4 || (ks == 0 && quad == 0)) {
        row.load(set[1], top - 2);
        return 2;
      }
      return 1;
    };
    // scale, round and store the rows of step (quad, KS) out of their register set, then start the LDS-DMA of its weight row
    auto commit = [&](auto KS, int quad, const typename Row::Data *set) __attribute__((always_inline)) {
      constexpr int ks = decltype(KS)::value;
      constexpr int kh = ks < 4 ? 2 * ks : 2 * ks - 7;
      const int top = row_base(quad) + kh + 6;
      store_row(set[0], top);
      if (ks == 4 || (ks == 0 && quad == 0)) store_row(set[1], top - 2);
      if ((ks == 0 || ks == 4) && quad == 0) {   // the band's first window of a chain: its two older rows, on the spot
        typename Row::Data extra[2];
        row.load(extra[0], top - 4);
        row.load(extra[1], top - 6);
        store_row(extra[0], top - 4);
        store_row(extra[1], top - 6);
      }
      weights_dma(kh, (7 * quad + ks) & 1);
    };
    // End of a step for a staging wave: its LDS stores and everything OLDER than the `newer` row loads it has just issued
    // (this step's weight DMA in particular) must have landed; the row loads themselves stay in flight across the barrier
    // -- __syncthreads() would wait for them too (vmcnt(0)) and put their latency back into every step.  A row is
    // kLoadsPerRow load instructions when the rows are on the aligned grid; otherwise the count is data dependent and the
    // wave waits for everything.
    constexpr int kLoadsPerRow = (SRC & 1) ? C1H_IN4 : 0;
    auto step_barrier = [&](int newer_rows) __attribute__((always_inline)) {
      if (kLoadsPerRow > 0 && newer_rows == 2) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * kLoadsPerRow) : "memory");
      } else if (kLoadsPerRow > 0 && newer_rows == 1) {
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(kLoadsPerRow) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    };
    // (the fences keep the program order the vmcnt arithmetic of step_barrier counts on: weight DMA, THEN the new row loads)
    auto order_fence = [&]() __attribute__((always_inline)) {
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
    };
#ifdef DVSG_STAMPS
    unsigned long long st_work = 0, st_wait = 0, st_load = 0, st_t = C1_STAMP();
#endif
    typedef std::integral_constant<int, 0> K0;
    typedef std::integral_constant<int, 1> K1;
    issue(K0{}, 0, setA);
    commit(K0{}, 0, setA);
    order_fence();
    int inflight = nsteps > 1 ? issue(K1{}, 0, setB) : 0;
    order_fence();
    step_barrier(inflight);
#ifdef DVSG_STAMPS
    st_t = C1_STAMP();
#endif
    // One step of the staging waves, K = 0..13 within a pair of quads: during step s = (quad, ks) store step s + 1's rows
    // (in the set of the other parity, fetched a step ago) and fetch step s + 2's into this parity's set.
    auto stager_step = [&](auto KK, int quad_pair) __attribute__((always_inline)) -> bool {
      constexpr int K = decltype(KK)::value;
      constexpr int ks = K % 7, ks1 = (K + 1) % 7, ks2 = (K + 2) % 7;
      const int quad = quad_pair + K / 7, quad1 = quad_pair + (K + 1) / 7, quad2 = quad_pair + (K + 2) / 7;
      const int sidx = 7 * quad + ks;
      if (sidx >= nsteps) return false;
      typename Row::Data *cur = (K & 1) ? setA : setB;    // step s + 1's rows
      typename Row::Data *nxt = (K & 1) ? setB : setA;    // step s + 2's rows go here
#ifdef DVSG_STAMPS
      { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); const unsigned long long n_ = C1_STAMP(); st_load += n_ - st_t; st_t = n_; }
#endif
      if (sidx + 1 < nsteps) commit(std::integral_constant<int, ks1>{}, quad1, cur);
      order_fence();
      const int fl = sidx + 2 < nsteps ? issue(std::integral_constant<int, ks2>{}, quad2, nxt) : 0;
      order_fence();
#ifdef DVSG_STAMPS
      { const unsigned long long n_ = C1_STAMP(); st_work += n_ - st_t; st_t = n_; }
#endif
      step_barrier(fl);
#ifdef DVSG_STAMPS
      { const unsigned long long n_ = C1_STAMP(); st_wait += n_ - st_t; st_t = n_; }
#endif
      return true;
    };
    for (int qp = 0; qp < nq; qp += 2) {
#define C1M_STEP(K) if (!stager_step(std::integral_constant<int, K>{}, qp)) break;
      C1M_STEP(0) C1M_STEP(1) C1M_STEP(2) C1M_STEP(3) C1M_STEP(4) C1M_STEP(5) C1M_STEP(6)
      C1M_STEP(7) C1M_STEP(8) C1M_STEP(9) C1M_STEP(10) C1M_STEP(11) C1M_STEP(12) C1M_STEP(13)
#undef C1M_STEP
    }
#ifdef DVSG_STAMPS
    if (tid == 256 && blockIdx.x < 65536) {
      g_c1_stamps[(size_t)blockIdx.x * 8 + 4] = st_work;   // staging wave: loads issued + rows stored + DMA issued, all steps
      g_c1_stamps[(size_t)blockIdx.x * 8 + 5] = st_wait;   // staging wave: waiting at the step barriers
      g_c1_stamps[(size_t)blockIdx.x * 8 + 7] = st_load;   // staging wave: waiting for the rows fetched a step earlier
    }
#endif
    return;
  }

  // -------------------------------------------------------------------------------- multiplying waves
  floatx16 acc[C1M_QUAD][2];   // [output row of the quad][channel block]; rows = channels (A = weights), columns = pixels
#pragma unroll
  for (int j = 0; j < C1M_QUAD; ++j)
#pragma unroll
    for (int ni = 0; ni < 2; ++ni)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[j][ni][q] = 0.f;
  // bias of the 32 channels a lane ends up with: 32 ni + 8 g + 4 h + {0..3}
  float4 bias4[2][4];
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int g = 0; g < 4; ++g) bias4[ni][g] = *reinterpret_cast<const float4 *>(bias + 32 * ni + 8 * g + 4 * h);
  __syncthreads();                                   // pairs with the staging waves' barrier after the LDS zeroing
  // A quad's results: bias, ReLU, float16 -- and a TRANSPOSE through LDS, because a lane holds 16 channels of ONE pixel
  // and storing them as they are (8 bytes per lane, 32 different cache lines per instruction) took 13 500 cycles per quad,
  // a quarter of the kernel.  Space: two row buffers of the even-row ring that are dead between the quad's last step
  // (kh = 5: window a + 5 .. a + 11) and the staging for the next quad's kh = 1 (window a + 9 .. a + 15), i.e. the
  // slots of rows a + 5 and a + 7; each wave transposes its own 32 pixels there, half a row's channels at a time, and
  // writes 16 bytes per lane (64 contiguous bytes per pixel).  No barrier: LDS operations of one wave complete in order.
  auto flush_quad = [&](int quad) __attribute__((always_inline)) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    // unit = (output row j, channel block ni): 32 pixels x 32 channels = 2 KB, 80-byte pixel stride (16-byte aligned reads,
    // 20 r mod 64 banks: conflict-free 8-byte writes); two buffers per wave, so that unit u + 1 is written while unit u's
    // reads are in flight -- the LDS round trips of the 8 units of a quad overlap instead of adding up
    constexpr int PXS = 40;                           // halves per pixel
    constexpr int UNIT = 32 * PXS;                    // halves per buffer
    _Float16 *tb = reinterpret_cast<_Float16 *>(in_s) + slot_of(2 * (rb0 + C1M_QUAD * quad) - 3 + (wave < 2 ? 5 : 7)) +
                   (wave & 1) * (2 * UNIT);
    const int cpx = lane >> 2, chunk = lane & 3;      // read side: 16 pixels per pass, 4 chunks of 16 bytes per pixel
    auto write_unit = [&](int u) __attribute__((always_inline)) {
      const int j = u >> 1, ni = u & 1;
      _Float16 *dst = tb + (u & 1) * UNIT + r * PXS + 4 * h;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 bb = bias4[ni][g];
        floatx4 v = {acc[j][ni][4 * g], acc[j][ni][4 * g + 1], acc[j][ni][4 * g + 2], acc[j][ni][4 * g + 3]};
        v = v + floatx4{bb.x, bb.y, bb.z, bb.w};
        v = __builtin_elementwise_max(v, floatx4{0.f, 0.f, 0.f, 0.f});
        *reinterpret_cast<half4_t *>(dst + 8 * g) = __builtin_convertvector(v, half4_t);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[j][ni][4 * g + q] = 0.f;
      }
    };
    halfx8 rd[2][2];
    auto read_unit = [&](int u) __attribute__((always_inline)) {
#pragma unroll
      for (int pass = 0; pass < 2; ++pass)
        rd[u & 1][pass] = *reinterpret_cast<const halfx8 *>(tb + (u & 1) * UNIT + (16 * pass + cpx) * PXS + 8 * chunk);
    };
    auto store_unit = [&](int u) __attribute__((always_inline)) {
      const int j = u >> 1, ni = u & 1;
      const int ho = rb0 + C1M_QUAD * quad + j;
#pragma unroll
      for (int pass = 0; pass < 2; ++pass) {
        const int pxo = wo0 + 32 * wave + 16 * pass + cpx;
#ifdef DVSG_FLUSH_NOSTORE   // (diagnostic: what do the global stores of the flush cost?)
        if (ho < -1 && pxo < Wo)
#else
        if (ho < Ho && pxo < Wo)
#endif
          *reinterpret_cast<halfx8 *>(y + (((size_t)b * Ho + ho) * Wo + pxo) * 64 + 32 * ni + 8 * chunk) = rd[u & 1][pass];
      }
    };
    // software pipeline: a unit's reads are issued a whole unit of arithmetic before the stores that need them (the LDS is
    // ~70 % busy with the other SIMDs' fragment reads: a round trip is several hundred cycles).  LDS operations of one wave
    // complete in order, so read_unit(u) sees write_unit(u) and write_unit(u + 2) cannot overtake read_unit(u).
    __builtin_amdgcn_s_setprio(3);
    write_unit(0);
    write_unit(1);
    read_unit(0);
#pragma unroll
    for (int u = 0; u < 2 * C1M_QUAD; ++u) {
      if (u + 1 < 2 * C1M_QUAD) read_unit(u + 1);
      if (u + 2 < 2 * C1M_QUAD) write_unit(u + 2);
      store_unit(u);
    }
    __builtin_amdgcn_s_setprio(0);
  };
  __syncthreads();                                   // the first window and weight row are in place
#ifdef DVSG_STAMPS
  unsigned long long m_work = 0, m_wait = 0, m_flush = 0, m_t = C1_STAMP();
  const unsigned long long m_begin = m_t;
#endif
  for (int s = 0; s < nsteps; ++s) {
    const int quad = s / 7, ks = s - 7 * quad;
    if (ks == 0 && quad > 0) flush_quad(quad - 1);
#ifdef DVSG_STAMPS
    { const unsigned long long n_ = C1_STAMP(); m_flush += n_ - m_t; m_t = n_; }
#endif
    const int w0 = window_of(s);
    // LDS byte offsets of this lane's fragments, each in ONE register (laundered through an empty asm), so that the ten
    // t-steps differ by the read instruction's own offset field (<= 1020 bytes); left alone the compiler keeps "lane part +
    // (LDS base + t offset)" and spends a v_add_u32 per read -- 80 VALU slots per step on the port the MFMAs issue through
    typedef const __attribute__((address_space(3))) unsigned *lds_u32_t;
    typedef const __attribute__((address_space(3))) halfx8 *lds_h8_t;
    const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) char *)lds;
    unsigned pix_off[C1M_QUAD];
#pragma unroll
    for (int j = 0; j < C1M_QUAD; ++j) {
      pix_off[j] = lds0 + 2 * C1H_WBYTES + 2 * slot_of(w0 + 2 * j) + 4 * (kConv1Cin * (wave * 32 + r) + 4 * h);
      asm volatile("" : "+v"(pix_off[j]));
    }
    unsigned w_off = lds0 + (s & 1) * C1H_WBYTES + 2 * (r * C1H_LD + 8 * h);
    asm volatile("" : "+v"(w_off));
    struct Frag {
      union {
        unsigned u[4];
        halfx8 v;
      } p[C1M_QUAD];
      halfx8 w[2];
    };
    auto read_frag = [&](Frag &f, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) f.w[ni] = *(lds_h8_t)(size_t)(w_off + 2 * ni * 32 * C1H_LD + 32 * t);
#pragma unroll
      for (int j = 0; j < C1M_QUAD; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) f.p[j].u[q] = *(lds_u32_t)(size_t)(pix_off[j] + 4 * (8 * t + q));
    };
    Frag fr[2];
    read_frag(fr[0], 0);
#pragma unroll
    for (int t = 0; t < C1H_STEPS; ++t) {
      if (t + 1 < C1H_STEPS) read_frag(fr[(t + 1) & 1], t + 1);
      const Frag &f = fr[t & 1];
#pragma unroll
      for (int j = 0; j < C1M_QUAD; ++j)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          acc[j][ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.w[ni], f.p[j].v, acc[j][ni], 0, 0, 0);
      // This wave is the only one that multiplies on its SIMD, so the next fragments' 10 LDS reads must issue BETWEEN this
      // step's MFMAs -- each of which keeps the matrix pipe busy for 32 cycles -- not in front of them: grouped
      // [reads][8 MFMAs] a t-step took 350 cycles instead of 256.
#define C1M_GROUP(NREADS)                                                      \
  __builtin_amdgcn_sched_group_barrier(0x008, 1, 0); /* one MFMA */            \
  __builtin_amdgcn_sched_group_barrier(0x100, NREADS, 0); /* LDS reads */
      C1M_GROUP(2) C1M_GROUP(2) C1M_GROUP(1) C1M_GROUP(1) C1M_GROUP(1) C1M_GROUP(1) C1M_GROUP(1) C1M_GROUP(1)
#undef C1M_GROUP
    }
#ifdef DVSG_STAMPS
    { const unsigned long long n_ = C1_STAMP(); m_work += n_ - m_t; m_t = n_; }
#endif
    __syncthreads();
#ifdef DVSG_STAMPS
    { const unsigned long long n_ = C1_STAMP(); m_wait += n_ - m_t; m_t = n_; }
#endif
  }
  flush_quad(nq - 1);
#ifdef DVSG_STAMPS
  if (tid == 0 && blockIdx.x < 65536) {
    unsigned long long *o = g_c1_stamps + (size_t)blockIdx.x * 8;
    o[0] = m_work;    // multiplying wave: fragment reads + MFMAs, all steps
    o[1] = m_wait;    // multiplying wave: waiting at the step barriers
    o[2] = m_flush;   // multiplying wave: bias / ReLU / stores between quads
    o[3] = (unsigned long long)nsteps;
    o[6] = C1_STAMP() - m_begin;
  }
#endif
}

// ----------------------------------------------------------------------------------------
// conv1 for the "f32s" precision: conv1_kernel's decomposition and staging (one output row x 128 pixels x
// 64 channels per workgroup, the input row segment prefetched into registers under the previous kernel
// row's MFMAs, scale_RGB applied while staging, overlapping windows read in place), with the products
// formed on the float16 matrix cores from two float16 pieces per operand (cnn_kernels.h, conv_gemm.hip).
// The scaled input is split ONCE, when it is staged -- hi = f16(v), lo = f16(v - hi), two float16 images
// of the row in LDS, together as large as the float32 row was -- so the 7 overlapping windows and the two
// channel halves that read an element do not split it again: a lane's operand is 8 consecutive taps =
// 16 bytes at byte offset 84 pixel + 32 step + 16 h of either image, four ds_read_b32 (21 r mod 32 is a
// bijection: conflict-free).  The weights' pieces sit in LDS as two [64][168]-half images per kernel row
// (336-byte rows: conflict-free ds_read_b128), prefetched through registers like the input.  The scaled
// input reaches +-150, which would amplify the absolute error of an unscaled lo weight piece, so here
// lo_w = f16((w - hi_w) x 2^11) and its products accumulate apart, folded in with 2^-11 at the end.
// Ten 16-tap steps per kernel row (147 taps padded to 160 with zero weights): 60 MFMAs of 32 cycles per
// wave and kernel row where the exact kernel issues 150 of 64.
// ----------------------------------------------------------------------------------------
constexpr int C1S_SEG = 5504;                        // halves per staged image of the row (1 + 5481 + tail; reads reach 5495)
constexpr int C1S_IN4 = (C1S_SEG / 4 + 255) / 256;     // 6 groups of four elements per thread
constexpr int C1S_WHALFS = 2 * 64 * kConv1LdH;       // halves per kernel row: hi image, then lo image
constexpr int C1S_WBYTES = C1S_WHALFS * 2;           // 43008

template <typename TO, int SRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_split_kernel(const Conv1Src src, const _Float16 *__restrict__ wt1s, const float *__restrict__ bias,
                        TO *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles) {
  constexpr int NT = 256;
  constexpr int WLOADS = (C1S_WBYTES / 16 + NT - 1) / NT;    // 11 float4
  static_assert(C1_TILE * C1_LDC * 4 <= C1S_WBYTES, "epilogue tile must fit in the weight stage");
  __shared__ __attribute__((aligned(16))) char w_s[C1S_WBYTES];
  __shared__ __attribute__((aligned(16))) unsigned in_hi[C1S_SEG / 2], in_lo[C1S_SEG / 2];  // half2 per word
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;   // wave w: pixels [32 w, 32 w + 32) x all 64 channels (two 32-channel blocks):
                                            // an activation fragment is read once for both blocks (the LDS is this kernel's bound)

  int blk = xcd_remap(blockIdx.x, gridDim.x);  // see conv1_kernel
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho = blk % Ho;
  const int b = blk / Ho;
  const int wo0 = wt_i * C1_TILE;

  typedef typename Conv1RowSel<NT, C1S_IN4, SRC>::type Row;
  Row row;
  row.init(src, tid, b, wo0, H, W, C1S_SEG);
  if constexpr (Row::kRing) {
    for (int e = tid; e < C1S_SEG / 2; e += NT) in_hi[e] = in_lo[e] = 0u;
  }
  typename Row::Data rd;
  floatx4 w_reg[WLOADS];
  auto load_stage = [&](int kh) __attribute__((always_inline)) {
    row.load(rd, 2 * ho + kh - 3);
    const floatx4 *wsrc = reinterpret_cast<const floatx4 *>(wt1s + (size_t)kh * C1S_WHALFS);
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      w_reg[i] = wsrc[q < C1S_WBYTES / 16 ? q : 0];
    }
  };
  auto store_stage = [&]() __attribute__((always_inline)) {
    typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
    if constexpr (Row::kRing) {
      row.scatter(rd, [&](int e, float v) __attribute__((always_inline)) {
        const _Float16 hv = (_Float16)v;
        reinterpret_cast<_Float16 *>(in_hi)[e] = hv;
        reinterpret_cast<_Float16 *>(in_lo)[e] = (_Float16)(v - (float)hv);
      });
    } else {
#pragma unroll
      for (int i = 0; i < C1S_IN4; ++i) {
        const int q = tid + NT * i;
        const floatx4 v = row.scaled(rd, i);   // then the two pieces
        half4_t hv, lv;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          hv[j] = (_Float16)v[j];
          lv[j] = (_Float16)(v[j] - (float)hv[j]);
        }
        if (q < C1S_SEG / 4) {
          reinterpret_cast<half4_t *>(in_hi)[q] = hv;
          reinterpret_cast<half4_t *>(in_lo)[q] = lv;
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      if (q < C1S_WBYTES / 16) reinterpret_cast<floatx4 *>(w_s)[q] = w_reg[i];
    }
  };

  floatx16 acc[2], accl[2];   // [channel block]
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[ni][q] = accl[ni][q] = 0.f;

  load_stage(0);
  for (int kh = 0; kh < 7; ++kh) {
    __syncthreads();  // everyone is done reading the previous kernel row
    store_stage();
    __syncthreads();
    if (kh + 1 < 7) load_stage(kh + 1);
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA loop
    // word index of the lane's first tap pair: (42 pixel + 8 h) / 2
    const int a0 = kConv1Cin * (wave * 32 + r) + 4 * h;
    const _Float16 *bhi0 = reinterpret_cast<const _Float16 *>(w_s) + r * kConv1LdH + 8 * h;
    const _Float16 *blo0 = bhi0 + 64 * kConv1LdH;
    // software-pipelined: the fragments of step t + 1 are requested before the MFMAs of step t are issued (left to
    // itself the compiler waits for each pair of steps' reads with nothing in flight)
    struct Frag {
      union {
        unsigned u[4];
        halfx8 v;
      } ahi, alo;
      halfx8 bhi[2], blo[2];
    };
    auto read_frag = [&](Frag &f, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        f.ahi.u[j] = in_hi[a0 + 8 * t + j];
        f.alo.u[j] = in_lo[a0 + 8 * t + j];
      }
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        f.bhi[ni] = *reinterpret_cast<const halfx8 *>(bhi0 + ni * 32 * kConv1LdH + 16 * t);
        f.blo[ni] = *reinterpret_cast<const halfx8 *>(blo0 + ni * 32 * kConv1LdH + 16 * t);
      }
    };
    Frag fr[2];
    read_frag(fr[0], 0);
#pragma unroll
    for (int t = 0; t < 10; ++t) {
      if (t + 1 < 10) read_frag(fr[(t + 1) & 1], t + 1);
      __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs that do not need them
      const Frag &f = fr[t & 1];
#pragma unroll
      for (int ni = 0; ni < 2; ++ni) {
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi.v, f.bhi[ni], acc[ni], 0, 0, 0);
        acc[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.alo.v, f.bhi[ni], acc[ni], 0, 0, 0);
        accl[ni] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f.ahi.v, f.blo[ni], accl[ni], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: fold the scaled lo products in, transpose through the (idle) weight stage
  float *Cs = reinterpret_cast<float *>(w_s);
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + ni * 32 + r] = acc[ni][q] + accl[ni][q] * (1.0f / 2048.0f);
  __syncthreads();
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
#pragma unroll 4
  for (int row = row0; row < C1_TILE; row += 16) {
    if (wo0 + row >= Wo) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * C1_LDC + 4 * col4);
    v.x = fmaxf(v.x + b4.x, 0.f);
    v.y = fmaxf(v.y + b4.y, 0.f);
    v.z = fmaxf(v.z + b4.z, 0.f);
    v.w = fmaxf(v.w + b4.w, 0.f);
    store4_p(y, (((size_t)b * Ho + ho) * Wo + wo0 + row) * 64 + 4 * col4, v);   // float16 pieces (cnn_device.h)
  }
}

// ----------------------------------------------------------------------------------------
// "f32x3" conv1: float32 in, float32 out, every product from THREE bfloat16 pieces per operand (conv_gemm_tile.h, X3: all 24
// bits of both operands, six of the nine cross terms, the five small ones in their own accumulators).  Built like
// conv1_split_kernel: the scaled input row is split ONCE while it is staged -- three bfloat16 images of the row in LDS, 33 KB
// -- so the loop holds LDS reads and MFMAs only (a split in registers would be ~40 VALU per fragment that the matrix pipe
// does not hide, tools/x3_overlap_probe.hip).  The weight pieces of a kernel row are 3 x 64 x 160 bfloat16 = 61 KB, too many
// for two workgroups per CU next to the input images: a kernel row is staged as two HALVES of 80 taps (5 MFMA steps each,
// [piece][64][88] bfloat16 = 33 KB; 176-byte rows: 11 sixteen-byte slots, odd, so a ds_read_b128 lane group is conflict-free),
// 14 stages per tile, the input images restaged at every other one.
// ----------------------------------------------------------------------------------------
constexpr int C1X_SEG = 5504;                        // bfloat16 per staged image of the row (1 + 5481 + tail; reads reach 5495)
constexpr int C1X_IN4 = (C1X_SEG / 4 + 255) / 256;     // 6 groups of four elements per thread
constexpr int C1X_WBYTES = kConv1X3StageElems * 2;   // 33 792 bytes per half kernel row
constexpr int C1X_WS = C1_TILE * C1_LDC * 4;         // the weight stage also holds the epilogue's transpose tile (34 816)
static_assert(C1X_WS >= C1X_WBYTES, "weight stage smaller than a half kernel row");

template <int SRC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void conv1_x3_kernel(const Conv1Src src, const unsigned short *__restrict__ wt1x, const float *__restrict__ bias,
                     float *__restrict__ y, int H, int W, int Ho, int Wo, int wtiles) {
  constexpr int NT = 256;
  constexpr int WLOADS = (C1X_WBYTES / 16 + NT - 1) / NT;    // 9 float4
  __shared__ __attribute__((aligned(16))) char w_s[C1X_WS];
  __shared__ __attribute__((aligned(16))) unsigned in_p[3][C1X_SEG / 2];  // two bfloat16 per word
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int r = lane & 31, h = lane >> 5;   // wave w: pixels [32 w, 32 w + 32) x all 64 channels (two 32-channel blocks)

  int blk = xcd_remap(blockIdx.x, gridDim.x);  // see conv1_kernel
  const int wt_i = blk % wtiles;
  blk /= wtiles;
  const int ho = blk % Ho;
  const int b = blk / Ho;
  const int wo0 = wt_i * C1_TILE;

  typedef typename Conv1RowSel<NT, C1X_IN4, SRC>::type Row;
  Row row;
  row.init(src, tid, b, wo0, H, W, C1X_SEG);
  if constexpr (Row::kRing) {
    for (int e = tid; e < C1X_SEG / 2; e += NT) in_p[0][e] = in_p[1][e] = in_p[2][e] = 0u;
  }
  typename Row::Data rd;
  floatx4 w_reg[WLOADS];
  auto load_stage = [&](int st) __attribute__((always_inline)) {   // stage st: kernel row st / 2, taps [80 (st & 1), + 80)
    if ((st & 1) == 0) row.load(rd, 2 * ho + (st >> 1) - 3);
    const floatx4 *wsrc = reinterpret_cast<const floatx4 *>(wt1x + (size_t)st * kConv1X3StageElems);
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      w_reg[i] = wsrc[q < C1X_WBYTES / 16 ? q : 0];
    }
  };
  auto pieces = [](float v, __bf16 &p1, __bf16 &p2, __bf16 &p3) __attribute__((always_inline)) {
    p1 = (__bf16)v;
    const float r1 = v - (float)p1;
    p2 = (__bf16)r1;
    p3 = (__bf16)(r1 - (float)p2);
  };
  auto store_stage = [&](int st) __attribute__((always_inline)) {
    typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
    if ((st & 1) == 0) {
      if constexpr (Row::kRing) {
        row.scatter(rd, [&](int e, float v) __attribute__((always_inline)) {
          __bf16 p1, p2, p3;
          pieces(v, p1, p2, p3);
          reinterpret_cast<__bf16 *>(in_p[0])[e] = p1;
          reinterpret_cast<__bf16 *>(in_p[1])[e] = p2;
          reinterpret_cast<__bf16 *>(in_p[2])[e] = p3;
        });
      } else {
#pragma unroll
        for (int i = 0; i < C1X_IN4; ++i) {
          const int q = tid + NT * i;
          const floatx4 v = row.scaled(rd, i);   // then the three pieces
          bf16x4 v1, v2, v3;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            __bf16 p1, p2, p3;
            pieces(v[j], p1, p2, p3);
            v1[j] = p1; v2[j] = p2; v3[j] = p3;
          }
          if (q < C1X_SEG / 4) {
            reinterpret_cast<bf16x4 *>(in_p[0])[q] = v1;
            reinterpret_cast<bf16x4 *>(in_p[1])[q] = v2;
            reinterpret_cast<bf16x4 *>(in_p[2])[q] = v3;
          }
        }
      }
    }
#pragma unroll
    for (int i = 0; i < WLOADS; ++i) {
      const int q = tid + NT * i;
      if (q < C1X_WBYTES / 16) reinterpret_cast<floatx4 *>(w_s)[q] = w_reg[i];
    }
  };

  floatx16 acc[2], accs[2];   // [channel block]: the large cross term, the five small ones
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[ni][q] = accs[ni][q] = 0.f;

  load_stage(0);
  for (int st = 0; st < 14; ++st) {
    __syncthreads();  // everyone is done reading the previous stage
    store_stage(st);
    __syncthreads();
    if (st + 1 < 14) load_stage(st + 1);
    __builtin_amdgcn_sched_barrier(0);  // prefetch stays ahead of the MFMA loop
    // word index of the lane's first tap pair: (42 pixel + 80 (st & 1) + 8 h) / 2
    const int a0 = kConv1Cin * (wave * 32 + r) + 40 * (st & 1) + 4 * h;
    const __bf16 *b0 = reinterpret_cast<const __bf16 *>(w_s) + r * kConv1X3Ld + 8 * h;
    // software-pipelined: the activation fragments of step t + 1 (4-byte-aligned reads, the slow ones) are requested before the
    // MFMAs of step t are issued; the weight fragments of a step are read when it starts (holding two steps of them as well
    // spills the ring and masked sources' staging registers)
    struct Frag {
      union {
        unsigned u[4];
        bf16x8 v;
      } a[3];
    };
    auto read_a = [&](Frag &f, int t) __attribute__((always_inline)) {
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int j = 0; j < 4; ++j) f.a[p].u[j] = in_p[p][a0 + 8 * t + j];
    };
    // (the ring and masked sources hold more staging registers: one set of fragments there, or the kernel spills)
    constexpr bool PIPE = SRC < 2;
    Frag fr[PIPE ? 2 : 1];
    if (PIPE) read_a(fr[0], 0);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      bf16x8 bw[3][2];
#pragma unroll
      for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ni = 0; ni < 2; ++ni)
          bw[p][ni] = *reinterpret_cast<const bf16x8 *>(b0 + (p * 64 + ni * 32) * kConv1X3Ld + 16 * t);
      if (PIPE) {
        if (t + 1 < 5) read_a(fr[(t + 1) & 1], t + 1);
      } else {
        read_a(fr[0], t);
      }
      __builtin_amdgcn_sched_barrier(0);   // keep the reads ahead of the MFMAs that do not need them
      const Frag &f = fr[PIPE ? (t & 1) : 0];
#define DVSG_C1X_TERM(C, PA, PB) \
  _Pragma("unroll") for (int ni = 0; ni < 2; ++ni) C[ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.a[PA].v, bw[PB][ni], C[ni], 0, 0, 0)
      DVSG_C1X_TERM(accs, 2, 0);
      DVSG_C1X_TERM(accs, 0, 2);
      DVSG_C1X_TERM(accs, 1, 1);
      DVSG_C1X_TERM(accs, 1, 0);
      DVSG_C1X_TERM(accs, 0, 1);
      DVSG_C1X_TERM(acc, 0, 0);
#undef DVSG_C1X_TERM
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- epilogue: join the two accumulators, transpose through the (idle) weight stage
  float *Cs = reinterpret_cast<float *>(w_s);
  __syncthreads();
#pragma unroll
  for (int ni = 0; ni < 2; ++ni)
#pragma unroll
    for (int q = 0; q < 16; ++q)
      Cs[(wave * 32 + (q & 3) + 8 * (q >> 2) + 4 * h) * C1_LDC + ni * 32 + r] = acc[ni][q] + accs[ni][q];
  __syncthreads();
  const int col4 = tid & 15, row0 = tid >> 4;
  const float4 b4 = *reinterpret_cast<const float4 *>(bias + 4 * col4);
#pragma unroll 4
  for (int row = row0; row < C1_TILE; row += 16) {
    if (wo0 + row >= Wo) break;
    float4 v = *reinterpret_cast<const float4 *>(Cs + row * C1_LDC + 4 * col4);
    v.x = fmaxf(v.x + b4.x, 0.f);
    v.y = fmaxf(v.y + b4.y, 0.f);
    v.z = fmaxf(v.z + b4.z, 0.f);
    v.w = fmaxf(v.w + b4.w, 0.f);
    store4(y + (((size_t)b * Ho + ho) * Wo + wo0 + row) * 64 + 4 * col4, v);
  }
}

// max pool on a P-format tensor ("f32s"): the pieces are joined (exact), compared, and the maximum's pieces stored
__global__ __launch_bounds__(256) void maxpool_p_kernel(const void *__restrict__ x, void *__restrict__ y, int H, int W,
                                                       int C4, int Ho, int Wo, int pad_top, int pad_left, size_t total) {
  const size_t e = (size_t)xcd_remap(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
  if (e >= total) return;
  const int c4 = (int)(e % C4);
  size_t t = e / C4;
  const int wo = (int)(t % Wo);
  t /= Wo;
  const int ho = (int)(t % Ho);
  const size_t b = t / Ho;
  float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int hi = 2 * ho - pad_top + i;
    if (hi < 0 || hi >= H) continue;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int wi = 2 * wo - pad_left + j;
      if (wi < 0 || wi >= W) continue;
      const float4 v = load4_p(x, (((b * H + hi) * W + wi) * C4 + c4) * 4);
      m.x = fmaxf(m.x, v.x);
      m.y = fmaxf(m.y, v.y);
      m.z = fmaxf(m.z, v.z);
      m.w = fmaxf(m.w, v.w);
    }
  }
  store4_p(y, e * 4, m);
}

// Bands of the marching kernel: the launch must give each of the 256 CUs (one workgroup each) several bands' worth of
// work, and a band should be long enough to amortise its first window (4 + 4 rows staged with nothing to multiply): at
// least 3 workgroups per CU with >= 4 quads (16 output rows) per band, else 0 = do not use the kernel.  Returns the number
// of bands per column tile and image; *quads_per_band the quads of each.
int march_bands(int B, int Ho, int wtiles, int *quads_per_band) {
  const int quads = (Ho + C1M_QUAD - 1) / C1M_QUAD;
  const long cols = (long)B * wtiles;
  for (int qpb = 8; qpb >= 4; --qpb) {               // 32 .. 16 output rows per band
    const int bands = (quads + qpb - 1) / qpb;
    if (cols * bands >= 3 * 256) {
      if (quads_per_band) *quads_per_band = qpb;
      return bands;
    }
  }
  return 0;
}

int g_conv1_variant = 0;  // dvsg_debug_set_option("conv1_variant", v): 0 = auto; 1 = float32 kernel with 8 waves; 2 = float16 output from
                          // the float32 multiply; 3 = float16, one output row per workgroup; 4 = never the marching kernel; 5 = always

}  // namespace

void set_conv1_variant(int v) { g_conv1_variant = v; }

int launch_conv1(int out_prec, const Conv1Src &src, int src_kind, const float *wt1, const void *wt1h, const void *wt1s,
                 const void *wt1x, const float *bias, void *y, int B, int H, int W, int Ho, int Wo, hipStream_t s) {
  const int wtiles = ceil_div(Wo, C1_TILE);
  const long blocks = (long)wtiles * Ho * B;
  DVSG_REQUIRE(blocks > 0 && blocks < (1L << 31), "conv1: grid of %ld workgroups out of range", blocks);
  DVSG_REQUIRE((long)W * kConv1Cin < (1L << 31), "conv1: input row too long");
  DVSG_REQUIRE(src_kind == kSrcWindow || (src.table && src.n_pool > 0), "conv1: a frame ring needs its index table");
  const double in_bytes = src_kind == kSrcRingU8 ? 1.0 : 4.0;   // each window element once (ring frames are shared by windows:
  ProfScope prof(kClsConv1, s, 2.0 * (double)B * Ho * Wo * 64 * 49 * kConv1Cin,   // an upper bound of the algorithmic read)
                 in_bytes * (double)B * H * W * kConv1Cin + (double)elem_size(out_prec) * B * Ho * Wo * 64);
  const dim3 grid((unsigned)blocks);
  // rows that start on 16-byte boundaries (4-byte for uint8 frames): the staged segment then consists of whole aligned groups
  const bool aligned = W % 4 == 0 && reinterpret_cast<uintptr_t>(src.base) % (src_kind == kSrcRingU8 ? 4 : 16) == 0;
  const int SRC = 2 * src_kind + (aligned ? 1 : 0) + (src.mask ? 8 : 0);
  // the f32s activations are float16 pieces (P format): only conv1_split_kernel writes them, whatever the A/B switch says
  // (the float32 kernel's output would be read back as pieces: silent garbage)
  if (out_prec == kF32S && !wt1s) return fail(DVSG_ERR_UNSUPPORTED, "conv1: the f32s precision needs the piece weights");
#define DVSG_K_F32(SRC) conv1_kernel<4, float, SRC>
#define DVSG_K_F16(SRC) conv1_f16_kernel<_Float16, (SRC) & 7>   /* A/B kernel: never launched with a mask */
#define DVSG_K_SPLIT(SRC) conv1_split_kernel<float, SRC>
#define DVSG_K_X3(SRC) conv1_x3_kernel<SRC>
#define DVSG_C1(KERNEL, NTHREADS, WPTR, YPTR)                                                                            \
  do {                                                                                                                   \
    switch (SRC) {                                                                                                       \
      case 0: hipLaunchKernelGGL((KERNEL(0)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 1: hipLaunchKernelGGL((KERNEL(1)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 2: hipLaunchKernelGGL((KERNEL(2)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 3: hipLaunchKernelGGL((KERNEL(3)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 4: hipLaunchKernelGGL((KERNEL(4)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 5: hipLaunchKernelGGL((KERNEL(5)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 8: hipLaunchKernelGGL((KERNEL(8)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 9: hipLaunchKernelGGL((KERNEL(9)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 10: hipLaunchKernelGGL((KERNEL(10)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 11: hipLaunchKernelGGL((KERNEL(11)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      case 12: hipLaunchKernelGGL((KERNEL(12)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
      default: hipLaunchKernelGGL((KERNEL(13)), grid, dim3(NTHREADS), 0, s, src, WPTR, bias, YPTR, H, W, Ho, Wo, wtiles); break; \
    }                                                                                                                    \
  } while (0)
  if (out_prec == kF32X && !wt1x) return fail(DVSG_ERR_UNSUPPORTED, "conv1: the f32x3 precision needs the bfloat16 piece weights");
  if (out_prec == kF32X) {
    DVSG_C1(DVSG_K_X3, 256, static_cast<const unsigned short *>(wt1x), static_cast<float *>(y));
  } else if (out_prec == kF32S) {
    DVSG_C1(DVSG_K_SPLIT, 256, static_cast<const _Float16 *>(wt1s), static_cast<float *>(y));
  } else if (out_prec == kF16 && wt1h && g_conv1_variant == 3 && !src.mask) {   // A/B: one output row per workgroup
    DVSG_C1(DVSG_K_F16, 256, static_cast<const _Float16 *>(wt1h), static_cast<_Float16 *>(y));
  } else if (out_prec == kF16 && wt1h && g_conv1_variant != 2 && g_conv1_variant != 4 && !src.mask &&
             (g_conv1_variant == 5 || march_bands(B, Ho, wtiles, nullptr) > 0)) {   // (a masked window -- eval_train.py's
    // graph -- takes the two-row kernel below: same products in the same order)
    // big launches: marching, wave-specialised workgroups (conv1_f16_march_kernel), one per CU
    // (conv1_variant 5 forces it at any size, bands of <= 3 quads: the parity tests run it on small frames)
    int qpb = 0;
    int bands = march_bands(B, Ho, wtiles, &qpb);
    if (g_conv1_variant == 5) {
      qpb = 3;
      bands = ((Ho + C1M_QUAD - 1) / C1M_QUAD + qpb - 1) / qpb;
    }
    const dim3 mgrid((unsigned)((long)wtiles * bands * B));
    const _Float16 *wp = static_cast<const _Float16 *>(wt1h);
    _Float16 *yp = static_cast<_Float16 *>(y);
    switch (SRC) {
      case 0: hipLaunchKernelGGL((conv1_f16_march_kernel<0>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
      case 1: hipLaunchKernelGGL((conv1_f16_march_kernel<1>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
      case 2: hipLaunchKernelGGL((conv1_f16_march_kernel<2>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
      case 3: hipLaunchKernelGGL((conv1_f16_march_kernel<3>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
      case 4: hipLaunchKernelGGL((conv1_f16_march_kernel<4>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
      default: hipLaunchKernelGGL((conv1_f16_march_kernel<5>), mgrid, dim3(512), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, bands, qpb); break;
    }
  } else if (out_prec == kF16 && wt1h && g_conv1_variant != 2) {   // two output rows per workgroup (conv1_f16_pair_kernel)
    const int hpairs = (Ho + 1) / 2;
    const long pblocks = (long)wtiles * hpairs * B;
    const dim3 pgrid((unsigned)pblocks);
    const _Float16 *wp = static_cast<const _Float16 *>(wt1h);
    _Float16 *yp = static_cast<_Float16 *>(y);
    switch (SRC) {
      case 0: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 0>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 1: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 1>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 2: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 2>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 3: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 3>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 4: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 4>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 5: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 5>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 8: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 8>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 9: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 9>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 10: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 10>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 11: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 11>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      case 12: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 12>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
      default: hipLaunchKernelGGL((conv1_f16_pair_kernel<_Float16, 13>), pgrid, dim3(256), 0, s, src, wp, bias, yp, H, W, Ho, Wo, wtiles, hpairs); break;
    }
  } else if (out_prec == kF16) {  // conv1_variant 2: f32 multiply, f16 output (window tensors only)
    DVSG_REQUIRE(src_kind == kSrcWindow && !src.mask, "conv1: conv1_variant 2 takes an unmasked window tensor");
    hipLaunchKernelGGL((conv1_kernel<4, _Float16, 0>), grid, dim3(256), 0, s, src, wt1, bias, static_cast<_Float16 *>(y), H, W,
                       Ho, Wo, wtiles);
  } else if (g_conv1_variant != 0 && src_kind == kSrcWindow && !src.mask) {
    hipLaunchKernelGGL((conv1_kernel<8, float, 0>), grid, dim3(512), 0, s, src, wt1, bias, static_cast<float *>(y), H, W, Ho,
                       Wo, wtiles);
  } else {
    DVSG_C1(DVSG_K_F32, 256, wt1, static_cast<float *>(y));
  }
#undef DVSG_C1
#undef DVSG_K_F32
#undef DVSG_K_F16
#undef DVSG_K_SPLIT
#undef DVSG_K_X3
  return check_launch("conv1_kernel");
}

#ifdef DVSG_STAMPS
extern "C" int dvsg_debug_read_conv1_stamps(void *host, size_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_c1_stamps), bytes) == hipSuccess ? 0 : -3;
}
#endif

int launch_maxpool(int prec, const void *x, void *y, int B, int H, int W, int C, int Ho, int Wo, int pad_top,
                   int pad_left, hipStream_t s) {
  DVSG_REQUIRE(C % 4 == 0 && (prec != kF32S || C % 32 == 0), "maxpool: C=%d must be a multiple of 4 (32 in the f32s format)", C);
  const size_t total = (size_t)B * Ho * Wo * (C / 4);
  const size_t want = (total + 255) / 256;
  DVSG_REQUIRE(want < (1u << 31), "maxpool: %zu blocks do not fit a grid", want);
  const int blocks = (int)want;
  ProfScope prof(kClsMaxpool, s, 0.0, (double)elem_size(prec) * C * ((double)B * H * W + (double)B * Ho * Wo));
  if (prec == kF32S)
    hipLaunchKernelGGL(maxpool_p_kernel, dim3(blocks), dim3(256), 0, s, x, y, H, W, C / 4, Ho, Wo, pad_top, pad_left, total);
  else if (prec == kF16 && C % 8 == 0) {
    const size_t total8 = total / 2;
    hipLaunchKernelGGL(maxpool_h8_kernel, dim3((unsigned)((total8 + 255) / 256)), dim3(256), 0, s, static_cast<const _Float16 *>(x),
                       static_cast<_Float16 *>(y), H, W, C / 8, Ho, Wo, pad_top, pad_left, total8);
  } else if (prec == kF16)
    hipLaunchKernelGGL(maxpool_kernel<_Float16>, dim3(blocks), dim3(256), 0, s, static_cast<const _Float16 *>(x),
                       static_cast<_Float16 *>(y), H, W, C / 4, Ho, Wo, pad_top, pad_left, total);
  else
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(blocks), dim3(256), 0, s, static_cast<const float *>(x),
                       static_cast<float *>(y), H, W, C / 4, Ho, Wo, pad_top, pad_left, total);
  return check_launch("maxpool_kernel");
}

}  // namespace dvsg
