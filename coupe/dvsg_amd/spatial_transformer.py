"""Drop-in for the reference's spatial_transformer.py on MI355X.

Same classes and call surface: `AffineTransformer(out_size).transform(inp, theta) -> out`
(spatial_transformer.py:13,32), `ProjectiveTransformer(out_size).transform(inp, theta) -> out`
(:365,384), `ElasticTransformer(out_size, param_dim=32, param_dim_per_side=4).transform(inp,
theta, forward=True) -> (out, x_s, y_s)` (:101,137), plus `bilinear_interp` (:496, sampler B).
Compute: `dvsg_grid_{affine,projective,elastic}_f32` (grid generation fused with the
sampler-B gather) and `dvsg_stn_sample_f32`.  `bicubic_interp` is broken upstream (line 633)
and is not offered.
"""
import math

import numpy as np
import torch

from . import _lib
from ._tensor import as_dev, device, empty, like_input, ptr, stream


def bilinear_interp(im, x, y, out_size):
    """spatial_transformer.py:496-563: returns [B*out_h*out_w, C]."""
    it = as_dev(im)
    xt = as_dev(x).reshape(-1)
    yt = as_dev(y).reshape(-1)
    B, H, W, C = it.shape
    out_h, out_w = int(out_size[0]), int(out_size[1])
    if xt.numel() != B * out_h * out_w or yt.numel() != xt.numel():
        raise ValueError("x / y must hold B*out_h*out_w coordinates")
    out = empty((B, out_h, out_w, C), it)
    _lib.call("dvsg_stn_sample_f32", ptr(it), ptr(xt), ptr(yt), B, H, W, C, out_h, out_w, ptr(out), stream())
    return like_input(out.reshape(-1, C), im)


def _interpolate(im, x, y, out_size, method="bilinear"):
    """spatial_transformer.py:489-494."""
    if method == "bilinear":
        return bilinear_interp(im, x, y, out_size)
    raise NotImplementedError("interp_method %r: bicubic_interp is broken in the reference (line 633)" % method)


class _GridTransformer(object):
    _fn = None
    param_dim = 0

    def __init__(self, out_size, name, interp_method="bilinear", **kwargs):
        if interp_method != "bilinear":
            raise NotImplementedError("only interp_method='bilinear' works in the reference")
        self.name = name
        self.out_size = out_size
        self.interp_method = interp_method

    def _run(self, inp, theta, want_coords):
        it = as_dev(inp)
        B, H, W, C = it.shape
        tt = as_dev(theta).reshape(B, self.param_dim)
        out_h, out_w = int(self.out_size[0]), int(self.out_size[1])
        out = empty((B, out_h, out_w, C), it)
        xs = empty((B * out_h * out_w,), it) if want_coords else None
        ys = empty((B * out_h * out_w,), it) if want_coords else None
        _lib.call(self._fn, ptr(tt), ptr(it), B, H, W, C, out_h, out_w, ptr(out), ptr(xs), ptr(ys), stream())
        return out, xs, ys

    def transform(self, inp, theta):
        out, _, _ = self._run(inp, theta, False)
        return like_input(out, inp)

    def _transform(self, inp, theta):
        """Source coordinates only (x_s_flat, y_s_flat)."""
        it = as_dev(inp)
        B = it.shape[0]
        tt = as_dev(theta).reshape(B, self.param_dim)
        out_h, out_w = int(self.out_size[0]), int(self.out_size[1])
        xs = empty((B * out_h * out_w,), it)
        ys = empty((B * out_h * out_w,), it)
        _lib.call(self._fn, ptr(tt), 0, B, 1, 1, 1, out_h, out_w, 0, ptr(xs), ptr(ys), stream())
        return like_input(xs, inp), like_input(ys, inp)


class AffineTransformer(_GridTransformer):
    """spatial_transformer.py:5-91.  theta [B,6] (row-major 2x3)."""
    _fn = "dvsg_grid_affine_f32"
    param_dim = 6

    def __init__(self, out_size, name="SpatialAffineTransformer", interp_method="bilinear", **kwargs):
        super().__init__(out_size, name, interp_method, **kwargs)


class ProjectiveTransformer(_GridTransformer):
    """spatial_transformer.py:364-452.  theta [B,8]; the ninth entry of the homography is 1."""
    _fn = "dvsg_grid_projective_f32"
    param_dim = 8

    def __init__(self, out_size, name="SpatialProjectiveTransformer", interp_method="bilinear", **kwargs):
        super().__init__(out_size, name, interp_method, **kwargs)


class ElasticTransformer(object):
    """spatial_transformer.py:93-362."""

    def __init__(self, out_size, param_dim=2 * 16, param_dim_per_side=4,
                 name="SpatialElasticTransformer", interp_method="bilinear", **kwargs):
        if interp_method != "bilinear":
            raise NotImplementedError("only interp_method='bilinear' works in the reference")
        num_control_points = int(param_dim / 2)
        assert param_dim == 2 * num_control_points, "param_dim must be 2 times a square of an integer."
        self.name = name
        self.param_dim = param_dim
        self.interp_method = interp_method
        self.num_control_points = num_control_points
        self.num_control_points_per_side = param_dim_per_side
        self.out_size = out_size
        self.grid_size = math.floor(math.sqrt(self.num_control_points))
        assert self.grid_size * self.grid_size == self.num_control_points, \
            "num_control_points must be a square of an int"
        n = self.num_control_points
        src = np.empty((2, n), dtype=np.float32)
        linv = np.empty((n, n + 3), dtype=np.float32)
        _lib.call("dvsg_elastic_constants_f32", self.grid_size, src.ctypes.data, linv.ctypes.data)
        dev = device()
        self.source_points = torch.from_numpy(src).to(dev)   # [2,n]  (:131)
        self.L_inv = torch.from_numpy(linv).to(dev)          # [n,n+3] (:358)
        self.num_pixels = self.out_size[0] * self.out_size[1]

    def transform(self, inp, theta, forward=True, **kwargs):
        """Returns (output, x_s, y_s) (spatial_transformer.py:137-193)."""
        it = as_dev(inp)
        B, H, W, C = it.shape
        n = self.num_control_points
        # theta are offsets of the control points; absolute = source_points + theta (:161)
        th = (self.source_points.unsqueeze(0) + as_dev(theta).reshape(-1, 2, n)).contiguous()
        out_h, out_w = int(self.out_size[0]), int(self.out_size[1])
        out = empty((B, out_h, out_w, C), it)
        xs = empty((B * out_h * out_w,), it)
        ys = empty((B * out_h * out_w,), it)
        _lib.call("dvsg_grid_elastic_f32", ptr(th), ptr(self.L_inv), ptr(self.source_points), n, ptr(it),
                  B, H, W, C, out_h, out_w, ptr(out), ptr(xs), ptr(ys), stream())
        return like_input(out, inp), like_input(xs, inp), like_input(ys, inp)

    def get_abs_theta(self, theta):
        """spatial_transformer.py:195-217."""
        n = self.num_control_points
        th = self.source_points.unsqueeze(0) + as_dev(theta).reshape(-1, 2, n)
        th = th.transpose(1, 2)
        xy = torch.clamp((th + 1.0) / 2.0, 0, 1)
        s = self.num_control_points_per_side
        return like_input(xy.reshape(-1, s, s, 2), theta)

    def get_abs_src_points(self, batch_size):
        """spatial_transformer.py:260-274."""
        sp = self.source_points.unsqueeze(0).transpose(1, 2)
        xy = torch.clamp((sp + 1.0) / 2.0, 0, 1)
        s = self.num_control_points_per_side
        return xy.reshape(-1, s, s, 2)
