// Launch policy of the implicit-GEMM convolution kernel (conv_gemm_tile.h: tile geometry, staging, work decomposition):
// which tile width, split-K / stream-K, and the float32 / float16 / f32s instantiations.  The f32x3 instantiations
// (bfloat16 x 3 products) live in conv_gemm_x3.hip.
#include "conv_gemm_tile.h"

namespace dvsg {

int g_conv_variant = 0;  // dvsg_debug_set_option("conv_variant", v): 0 = auto, 1 = 4 waves, 2 = 8 waves,
                         // 3 = no split-K, 4 = 64-wide tiles only, 5 = no 256 x 128 float16 tiles, 6 = no stream-K tail
long g_wide16_min_tiles = 128;   // float16 mode: 256 x 128 tiles from this many of them (a quarter of a round of 512 workgroups).
                                 // Round 2 measured 256 (128 and below lost at batch 1-2 with the kernels of then); with packed
                                 // weight stages, 128-byte activation rows and the 3x3 row reuse 128 is -0.5 % at batch 16, -4.5 %
                                 // at batch 4 (720p), equal at batch 1; 64 and 32 lose 4-27 % at batch 1-4

void set_conv_variant(int v) { g_conv_variant = v; }
void set_wide16_min_tiles(int v) { g_wide16_min_tiles = v; }

namespace {
__global__ __launch_bounds__(256) void pack_x3_kernel(const float *__restrict__ wt, unsigned short *__restrict__ out, int rows, int K) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (size_t)rows * K) return;
  const int n = (int)(i / K), k = (int)(i - (size_t)n * K);
  const float w = wt[i];
  const __bf16 p1 = (__bf16)w;
  const float r1 = w - (float)p1;
  const __bf16 p2 = (__bf16)r1;
  const __bf16 p3 = (__bf16)(r1 - (float)p2);
  const int R = n & 63, kk = k & 31;
  const size_t base = (((size_t)(n >> 6) * (K >> 5) + (k >> 5)) * 3) * 2048 + R * 32 + (((kk >> 3) ^ ((R >> 2) & 3)) << 3) + (kk & 7);
  out[base] = __builtin_bit_cast(unsigned short, p1);
  out[base + 2048] = __builtin_bit_cast(unsigned short, p2);
  out[base + 4096] = __builtin_bit_cast(unsigned short, p3);
}
__global__ __launch_bounds__(256) void zero_tickets_kernel(int *__restrict__ t, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) t[i] = 0;
}
}  // namespace

int launch_pack_x3(const float *wt, void *out, int rows, int K, hipStream_t s) {
  DVSG_REQUIRE(wt && out && rows > 0 && rows % 64 == 0 && K > 0 && K % 32 == 0, "pack_x3: rows=%d must be a multiple of 64, K=%d of 32", rows, K);
  const size_t n = (size_t)rows * K;
  hipLaunchKernelGGL(pack_x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, wt, static_cast<unsigned short *>(out), rows, K);
  return check_launch("pack_x3_kernel");
}

int launch_zero_tickets(int *tickets, size_t n, hipStream_t s) {
  if (n == 0) return DVSG_OK;
  hipLaunchKernelGGL(zero_tickets_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, tickets, n);
  return check_launch("zero_tickets_kernel");
}

int launch_conv_gemm(const ConvGemm &p, hipStream_t s) {
  DVSG_REQUIRE(p.prec == kF32 || p.prec == kF16, "conv_gemm: unknown precision %d", p.prec);
  DVSG_REQUIRE(p.ksize == 1 || p.ksize == 3, "conv_gemm: kernel size %d unsupported", p.ksize);
  const int bke = ROWB / (int)elem_size(p.prec);
  DVSG_REQUIRE(p.Cin % bke == 0 && p.Cout % 64 == 0, "conv_gemm: Cin=%d must be a multiple of %d and Cout=%d of 64",
               p.Cin, bke, p.Cout);
  const long M = (long)p.B * p.Ho * p.Wo;
  DVSG_REQUIRE(M > 0 && M < (1L << 31) - BM, "conv_gemm: M=%ld out of range", M);
  ConvGemmDev d;
  d.x = p.x; d.wt = p.wt; d.bias = p.bias; d.res = p.res; d.y = p.y;
  d.H = p.H; d.W = p.W; d.Cin = p.Cin; d.Ho = p.Ho; d.Wo = p.Wo; d.Cout = p.Cout;
  d.stride = p.stride; d.pad = p.pad;
  d.res_H = p.res_H; d.res_W = p.res_W; d.res_stride = p.res_stride;
  d.ldx = p.ldx > 0 ? p.ldx : p.Cin;
  d.res_ld = p.res_ld > 0 ? p.res_ld : p.Cout;
  d.relu_from = p.relu_from >= 0 ? p.relu_from : 0x7fffffff;
  DVSG_REQUIRE(p.prec != kF16 || (d.ldx == p.Cin && d.res_ld == p.Cout && p.relu_from < 0),
               "conv_gemm: row strides / a partial ReLU are the float32 kernels' only");
  DVSG_REQUIRE(d.ldx >= p.Cin && d.ldx % bke == 0 && d.res_ld >= p.Cout && d.res_ld % 4 == 0 && (p.relu_from < 0 || (p.relu_from % 64 == 0 && !p.relu)),
               "conv_gemm: bad strides ldx=%d res_ld=%d relu_from=%d", d.ldx, d.res_ld, p.relu_from);
  d.M = (int)M;
  d.K = p.ksize * p.ksize * p.Cin;
  d.mtiles = (int)((M + BM - 1) / BM);
  const int res = !p.res ? 0 : (p.res_stride == 1 && p.res_H == p.Ho && p.res_W == p.Wo ? 1 : 2);
  // algorithmic work: 2 M N K flops; bytes = input + weights + output (+ residual) once each
  const double es = (double)elem_size(p.prec);
  ProfScope prof(p.ksize == 3 ? kClsConv3x3 : kClsConv1x1, s, 2.0 * (double)M * p.Cout * d.K,
                 es * ((double)p.B * p.H * p.W * p.Cin + (double)p.Cout * d.K +
                       (double)M * p.Cout * (p.res ? 2.0 : 1.0)));
  // 128-wide n tiles when there are enough of them to fill the chip, else 64-wide -- except that a
  // launch of 256..511 wide tiles (230 measured: no gain) with a long K loop runs wide as ONE stream-K round (block 4 at
  // batch 16, 720p: 460 wide tiles, or 920 narrow ones = 1.8 rounds, both 90 % full otherwise).
  const bool split = p.wsplit != 0 && p.prec == kF16;    // f16: hi / lo weight rows stacked along N
  const bool x3 = p.x3 != 0;                             // f32x3: float32 tensors, products from three bfloat16 pieces per operand
  DVSG_REQUIRE(!x3 || (p.prec == kF32 && p.wsplit == 0), "conv_gemm: f32x3 runs on float32 tensors");
  const bool psplit = (p.wsplit != 0 && p.prec == kF32) || x3;   // f32s: float32 storage, products from float16 pieces
                                                         // (f32x3 shares its launch policy: tiles several times cheaper than exact ones)
  // split weights: 128 physical weight rows per 64 output channels, always one 128-wide tile each
  const long tiles128 = split ? (long)d.mtiles * (p.Cout / 64) : p.Cout % 128 == 0 ? (long)d.mtiles * (p.Cout / 128) : 0;
  const int kt_all = d.K / bke;
  const size_t streamk_need = (size_t)kResident * 2 * BM * 128 * sizeof(float);
  const bool streamk_ok = g_conv_variant == 0 && p.splitk_scratch && streamk_need <= p.splitk_scratch_bytes && res == 0;
  // (f32s: only for the very long K loops -- block 4's 3x3 layers; 1024- and 2048-deep 1x1 layers of 460 tiles run
  // 52 vs 87 us and 96 vs 157 us without it)
  const bool streamk_all = streamk_ok && tiles128 >= kResident / 2 && tiles128 < kResident && kt_all >= (psplit ? 128 : 32);
  const bool wide = split || (g_conv_variant != 4 && (tiles128 >= kResident || streamk_all));
  d.ntiles = split ? p.Cout / 64 : p.Cout / (wide ? 128 : 64);
  // Tile order.  Each XCD runs a contiguous range of tiles.  With nt fastest that range covers every
  // weight panel, which is right while a panel (BN x K) is small; block 4's 3x3 conv has 2.4 MB
  // panels, and its stream-K workgroups stream them at 512 different K phases, so each tile fetched
  // its whole panel past the 4 MB L2 (1.2 GB per launch).  With mt fastest an XCD stays on one or
  // two panels and re-reads the (much smaller) activations instead.
  d.mt_fast = d.ntiles > 1 && (size_t)(wide ? 128 : 64) * d.K * elem_size(p.prec) >= ((size_t)2 << 20);
  // float16 mode: a launch of several rounds of tiles runs in the 256 x 128 / 64-byte-stage geometry (conv_gemm_wide16.hip)
  if (split && g_conv_variant != 5 && (long)((M + 255) / 256) * (p.Cout / 64) >= g_wide16_min_tiles) return launch_conv_wide16(p, s);
  // ... and so does a plain-float16 layer (no lo piece: locnet.hip's pair policy) whose 128-channel tiles fill the chip
  if (p.prec == kF16 && !split && g_conv_variant != 5 && p.Cout % 128 == 0 && p.Cin % 64 == 0 &&
      (long)((M + 255) / 256) * (p.Cout / 128) >= g_wide16_min_tiles)
    return launch_conv_wide16(p, s);
  d.ksplit = 1;
  d.slabs = static_cast<float *>(p.splitk_scratch);
  d.counters = p.splitk_counters;
  const long tiles = (long)d.mtiles * d.ntiles;
  // Split-K when the launch has too few tiles for the 512 resident workgroups (batch 1-2): slices
  // of >= 2 stages, at most 8 per tile, partial tiles + tickets in the caller's scratch.
  // (measured at batch 1, 720p: pays for <= 128 tiles and K rows of >= 4 KiB, i.e. the 3x3 convs of
  // blocks 3-4 and block 4's 1x1 convs; shorter K loops lose more to the reduction than they gain;
  // round 2, up to 16 / 32 slices per tile: a 512x288 frame takes 1.21 / 1.40 ms instead of 1.09, 720p unchanged)
  if (g_conv_variant != 3 && p.splitk_scratch && (!wide || split) && tiles <= 128 && kt_all >= 32) {
    const int ks = (int)std::min<long>(8, std::min<long>(kResident / tiles, kt_all / 2));
    const size_t need = (size_t)tiles * ks * BM * (split ? 128 : 64) * sizeof(float);
    if (ks > 1 && need <= p.splitk_scratch_bytes && tiles <= kSplitKMaxTiles) d.ksplit = ks;
  }
  // Stream-K tail for the big 128-wide launches: only when the last round is clearly under-filled,
  // every workgroup's share is at least two K stages, and K rows are >= 4 KiB (batch 16, 720p: the
  // 3x3 convs of blocks 2-3 -7 %, block 4's shortcut -5 %, block 3's 1024->256 convs -1.6 %).
  // With K rows of 2 KiB the partial tiles, 64 KiB each way, cost more than the tail saves
  // (measured +3..7 % on those layers).
  // (Measured and rejected: the same tail treatment for the 64-wide tiles of block 1 -- K rows of
  // 2.3 KiB, shares of ~9 stages -- lost 2-6 % per launch.)
  int streamk_tail = 0;
  if (streamk_all) {
    streamk_tail = (int)tiles;
  } else if (streamk_ok && wide && d.ksplit == 1 && tiles > kResident && !psplit && !split) {
    // (f32s, and the float16 mode with its hi / lo weights: with tiles several times cheaper in matrix-core time
    // the partial-tile traffic of a tail costs more than the tail it removes -- f32s: 103 vs 142 us on block 3's
    // 1024 -> 256 layers, f16: +2.4 % end to end at 720p without it; only a launch that is less than one round,
    // above, still pays)
    const int r = (int)(tiles % kResident);
    if (r > 0 && r <= kResident * 4 / 5 && kt_all >= 32 && (long)r * kt_all >= 2L * kResident) streamk_tail = r;
  }
  // (Measured and rejected: running a sliver of a last round -- 16-64 tiles after 7-28 full rounds,
  // blocks 1-3 -- as a separate split-K launch.  The hardware does not run tiles in lock-step rounds,
  // and the drain between the two launches costs more than the sliver: every such layer got 1-5 %
  // slower.)
  if (x3) return launch_conv_gemm_x3(d, p.ksize, wide, streamk_tail, p.relu != 0, res, s);
  if (psplit)
    return p.ksize == 1 ? launch_ks<float, 1, true>(d, wide, streamk_tail, p.relu != 0, res, s)
                        : launch_ks<float, 3, true>(d, wide, streamk_tail, p.relu != 0, res, s);
  if (p.prec == kF32)
    return p.ksize == 1 ? launch_ks<float, 1>(d, wide, streamk_tail, p.relu != 0, res, s)
                        : launch_ks<float, 3>(d, wide, streamk_tail, p.relu != 0, res, s);
  if (split)
    return p.ksize == 1 ? launch_ks<_Float16, 1, true>(d, wide, streamk_tail, p.relu != 0, res, s)
                        : launch_ks<_Float16, 3, true>(d, wide, streamk_tail, p.relu != 0, res, s);
  return p.ksize == 1 ? launch_ks<_Float16, 1>(d, wide, streamk_tail, p.relu != 0, res, s)
                      : launch_ks<_Float16, 3>(d, wide, streamk_tail, p.relu != 0, res, s);
}

}  // namespace dvsg
