"""Oracle (test infrastructure): warp_with_optical_flow.py `tf_warp` restated in NumPy
float32.  parity unpinned (see oracle/__init__.py).

Reference: warp_with_optical_flow.py:91-176.
"""
import numpy as np

from .tfops import F32, add_n4


def padded_bilinear(im, x, y):
    """The shared tail of sampler B and sampler C: 1-px zero ring, clamp of the pixel
    coordinates to [-1, W] / [-1, H], floor, min() on the upper index, weights from the
    UNclamped x0+1 (warp_with_optical_flow.py:103-104,128-173; the same text at
    spatial_transformer.py:504-505,517-562).

    im [B,H,W,C]; x, y [B,N] in (unpadded) pixel units.  Returns [B,N,C].
    """
    im = np.asarray(im, dtype=F32)
    B, H, W, C = im.shape
    edge = 1
    imp = np.pad(im, [[0, 0], [edge, edge], [edge, edge], [0, 0]], mode="constant")  # :104
    width_f = F32(W)
    height_f = F32(H)
    x = np.clip(np.asarray(x, dtype=F32), F32(-edge), F32(width_f - F32(1) + F32(edge)))   # :128
    y = np.clip(np.asarray(y, dtype=F32), F32(-edge), F32(height_f - F32(1) + F32(edge)))  # :129
    x = (x + F32(edge)).astype(F32)                                                        # :131
    y = (y + F32(edge)).astype(F32)
    x0_f = np.floor(x).astype(F32)                                                         # :135
    y0_f = np.floor(y).astype(F32)
    x1_f = (x0_f + F32(1)).astype(F32)
    y1_f = (y0_f + F32(1)).astype(F32)
    x0 = x0_f.astype(np.int32)
    y0 = y0_f.astype(np.int32)
    x1 = np.minimum(x1_f, F32(width_f - F32(1) + F32(2 * edge))).astype(np.int32)          # :142
    y1 = np.minimum(y1_f, F32(height_f - F32(1) + F32(2 * edge))).astype(np.int32)         # :143
    bidx = np.arange(B)[:, None]
    I00 = imp[bidx, y0, x0]                                                                # :162
    I01 = imp[bidx, y0, x1]
    I10 = imp[bidx, y1, x0]
    I11 = imp[bidx, y1, x1]
    w00 = ((x1_f - x) * (y1_f - y)).astype(F32)[..., None]                                 # :168
    w01 = ((x - x0_f) * (y1_f - y)).astype(F32)[..., None]
    w10 = ((x1_f - x) * (y - y0_f)).astype(F32)[..., None]
    w11 = ((x - x0_f) * (y - y0_f)).astype(F32)[..., None]
    return add_n4(w00 * I00, w01 * I01, w10 * I10, w11 * I11)                              # :173


def tf_warp(im, flow, out_height, out_width):
    """warp_with_optical_flow.py:96: backward warp by a dense flow in pixel units
    (flow[...,0] = dx, flow[...,1] = dy).  out size must equal the input size (:148)."""
    im = np.asarray(im, dtype=F32)
    flow = np.asarray(flow, dtype=F32)
    B, H, W, C = im.shape
    assert (out_height, out_width) == (H, W)
    gx, gy = np.meshgrid(np.arange(W), np.arange(H))                                       # :107
    x = (gx.astype(F32)[None] + flow[..., 0]).astype(F32).reshape(B, -1)                   # :117-119
    y = (gy.astype(F32)[None] + flow[..., 1]).astype(F32).reshape(B, -1)
    out = padded_bilinear(im, x, y)
    return out.reshape(B, out_height, out_width, C)                                        # :175
