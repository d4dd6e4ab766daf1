// "f32x3" instantiations of the implicit-GEMM convolution kernel (conv_gemm_tile.h, X3): float32 tensors, every product
// from three bfloat16 pieces per operand on the bfloat16 matrix cores.  Always four waves per workgroup whose tiles are
// 64 wide along N -- 128 x 128 tiles as 2 x 2 waves of 64 x 64, 128 x 64 tiles as 4 x 1 waves of 32 x 64 -- so that one
// in-register split of an activation fragment serves two weight blocks.
#include "conv_gemm_tile.h"

namespace dvsg {
namespace {

template <int KS>
int launch_x3_ks(ConvGemmDev d, bool wide, int streamk_tail, bool relu, int res, hipStream_t s) {
  const int tiles = d.mtiles * d.ntiles;
  d.tile_begin = 0;
  d.tile_count = tiles;
  if (d.ksplit > 1) return launch_cfg<float, 64, 4, 1, KS, 1, false, true>(d, tiles * d.ksplit, relu, res, s);
  if (wide && streamk_tail > 0) {
    d.tile_begin = tiles - streamk_tail;
    d.tile_count = streamk_tail;
    return launch_cfg<float, 128, 2, 2, KS, 2, false, true>(d, d.tile_begin + 512, relu, res, s);
  }
  return wide ? launch_cfg<float, 128, 2, 2, KS, 0, false, true>(d, tiles, relu, res, s)
              : launch_cfg<float, 64, 4, 1, KS, 0, false, true>(d, tiles, relu, res, s);
}

}  // namespace

int launch_conv_gemm_x3(const ConvGemmDev &d, int ksize, bool wide, int streamk_tail, bool relu, int res, hipStream_t s) {
  return ksize == 1 ? launch_x3_ks<1>(d, wide, streamk_tail, relu, res, s) : launch_x3_ks<3>(d, wide, streamk_tail, relu, res, s);
}

}  // namespace dvsg
