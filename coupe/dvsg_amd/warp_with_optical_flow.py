"""Drop-in for the reference's warp_with_optical_flow.py `tf_warp` on MI355X.

`tf_warp(im, flow, out_height, out_width) -> out` (warp_with_optical_flow.py:96-176):
im [B,H,W,C], flow [B,H,W,2] in pixels (dx, dy).  Compute: `dvsg_flow_warp_f32`.
"""
from . import _lib
from ._tensor import as_dev, empty, like_input, ptr, stream


def tf_warp(im, flow, out_height, out_width):
    it = as_dev(im, "im")
    ft = as_dev(flow, "flow")
    if it.dim() != 4 or ft.dim() != 4 or ft.shape[3] != 2 or ft.shape[:3] != it.shape[:3]:
        raise ValueError("im must be [B,H,W,C] and flow [B,H,W,2]")
    B, H, W, C = it.shape
    if (int(out_height), int(out_width)) != (H, W):
        # the reference's flat index base (warp_with_optical_flow.py:148) is only consistent then
        raise ValueError("tf_warp requires out_height/out_width == the input size")
    out = empty((B, H, W, C), it)
    _lib.call("dvsg_flow_warp_f32", ptr(it), ptr(ft), B, H, W, C, ptr(out), stream())
    return like_input(out, im)
