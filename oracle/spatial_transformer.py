"""Oracle (test infrastructure): spatial_transformer.py restated in NumPy float32.
parity unpinned (see oracle/__init__.py).

Reference: spatial_transformer.py:5-91 (Affine), 93-362 (Elastic), 364-452 (Projective),
460-482 (_meshgrid), 496-563 (bilinear_interp = sampler B).  bicubic_interp (565-673) is
broken upstream (`channels` undefined at 633) and is not restated.
"""
import math

import numpy as np

from .tfops import F32, seq_matmul_small, tf_linspace
from .warp_with_optical_flow import padded_bilinear


def _meshgrid(out_size):
    """spatial_transformer.py:460-482: rows [x_t; y_t; 1], each [H*W], x fastest."""
    out_h, out_w = int(out_size[0]), int(out_size[1])
    x_t, y_t = np.meshgrid(tf_linspace(-1.0, 1.0, out_w), tf_linspace(-1.0, 1.0, out_h))  # :474
    x_t_flat = x_t.reshape(1, -1)
    y_t_flat = y_t.reshape(1, -1)
    return np.concatenate([x_t_flat, y_t_flat, np.ones_like(x_t_flat)], 0).astype(F32)     # :480


def bilinear_interp(im, x, y, out_size):
    """Sampler B, spatial_transformer.py:496-563.  x, y [B*N] or [B,N] normalised."""
    im = np.asarray(im, dtype=F32)
    B, H, W, C = im.shape
    x = np.asarray(x, dtype=F32).reshape(B, -1)
    y = np.asarray(y, dtype=F32).reshape(B, -1)
    x = ((x + F32(1.0)).astype(F32) / F32(2.0) * F32(F32(W) - F32(1.0))).astype(F32)       # :515
    y = ((y + F32(1.0)).astype(F32) / F32(2.0) * F32(F32(H) - F32(1.0))).astype(F32)       # :516
    return padded_bilinear(im, x, y).reshape(-1, C)                                        # :517-562


def _interpolate(im, x, y, out_size, method="bilinear"):
    """spatial_transformer.py:489-494."""
    if method == "bilinear":
        return bilinear_interp(im, x, y, out_size)
    raise NotImplementedError("bicubic_interp is broken in the reference (line 633)")


class AffineTransformer(object):
    """spatial_transformer.py:5-91."""

    def __init__(self, out_size, name="SpatialAffineTransformer", interp_method="bilinear", **kwargs):
        self.name = name
        self.out_size = out_size
        self.param_dim = 6
        self.interp_method = interp_method
        self.pixel_grid = _meshgrid(out_size)                                              # :29

    def _transform(self, inp, theta):
        B = inp.shape[0]
        theta = np.asarray(theta, dtype=F32).reshape(-1, 2, 3)                             # :80
        T_g = seq_matmul_small(theta, np.broadcast_to(self.pixel_grid, (B,) + self.pixel_grid.shape))  # :85
        return T_g[:, 0].reshape(-1), T_g[:, 1].reshape(-1)

    def transform(self, inp, theta):
        inp = np.asarray(inp, dtype=F32)
        x_s, y_s = self._transform(inp, theta)
        out = _interpolate(inp, x_s, y_s, self.out_size, self.interp_method)
        return out.reshape(-1, self.out_size[0], self.out_size[1], inp.shape[3])           # :67


class ProjectiveTransformer(object):
    """spatial_transformer.py:364-452."""

    def __init__(self, out_size, name="SpatialProjectiveTransformer", interp_method="bilinear", **kwargs):
        self.name = name
        self.out_size = out_size
        self.param_dim = 8
        self.interp_method = interp_method
        self.pixel_grid = _meshgrid(out_size)                                              # :381

    def _transform(self, inp, theta):
        B = inp.shape[0]
        theta = np.asarray(theta, dtype=F32).reshape(B, 8)                                 # :429
        theta = np.concatenate([theta, np.ones((B, 1), dtype=F32)], 1).reshape(B, 3, 3)    # :430-431
        T_g = seq_matmul_small(theta, np.broadcast_to(self.pixel_grid, (B,) + self.pixel_grid.shape))  # :437
        x_s, y_s, z_s = T_g[:, 0], T_g[:, 1], T_g[:, 2]
        nz = z_s != 0
        safe = np.where(nz, z_s, F32(1))
        x_s = np.where(nz, (x_s / safe).astype(F32), F32(0))                               # :446 div_no_nan
        y_s = np.where(nz, (y_s / safe).astype(F32), F32(0))                               # :447
        return x_s.reshape(-1).astype(F32), y_s.reshape(-1).astype(F32)

    def transform(self, inp, theta):
        inp = np.asarray(inp, dtype=F32)
        x_s, y_s = self._transform(inp, theta)
        out = _interpolate(inp, x_s, y_s, self.out_size, self.interp_method)
        return out.reshape(-1, self.out_size[0], self.out_size[1], inp.shape[3])           # :420


class ElasticTransformer(object):
    """spatial_transformer.py:93-362."""

    def __init__(self, out_size, param_dim=2 * 16, param_dim_per_side=4,
                 name="SpatialElasticTransformer", interp_method="bilinear", **kwargs):
        num_control_points = int(param_dim / 2)
        assert param_dim == 2 * num_control_points
        self.name = name
        self.param_dim = param_dim
        self.interp_method = interp_method
        self.num_control_points = num_control_points
        self.num_control_points_per_side = param_dim_per_side
        self.out_size = out_size
        self.grid_size = math.floor(math.sqrt(self.num_control_points))
        assert self.grid_size * self.grid_size == self.num_control_points
        self.source_points = self.get_meshgrid(self.grid_size, self.grid_size)             # :131
        self.pixel_grid = self.get_meshgrid(self.out_size[1], self.out_size[0])            # :133
        self.num_pixels = self.out_size[0] * self.out_size[1]
        self.pixel_distances, self.L_inv = self._initialize_tps(self.source_points, self.pixel_grid)

    @staticmethod
    def U_func(points1, points2):
        """:298-310: U = r^2 log r^2 with log(0) -> 0 through the is_inf mask."""
        r_sq = np.sum(np.square((points1 - points2).astype(F32)), axis=0, dtype=F32).T
        with np.errstate(divide="ignore"):
            log_r = np.log(r_sq).astype(F32)
        log_r = np.where(np.isinf(log_r), F32(0), log_r)
        return (r_sq * log_r).astype(F32)

    @staticmethod
    def get_meshgrid(grid_size_x, grid_size_y):
        """:313-322: [2, n] rows (x, y), x fastest."""
        x_points, y_points = np.meshgrid(tf_linspace(-1.0, 1.0, int(grid_size_x)),
                                         tf_linspace(-1.0, 1.0, int(grid_size_y)))
        return np.concatenate([x_points.reshape(1, -1), y_points.reshape(1, -1)], 0).astype(F32)

    def _initialize_tps(self, source_points, pixel_grid):
        """:324-362."""
        n = self.num_control_points
        tL = self.U_func(source_points[:, :, None], source_points[:, None, :])             # :339
        L_top = np.concatenate([np.zeros((2, 3), F32), source_points], 1)                  # :342
        L_mid = np.concatenate([np.zeros((1, 2), F32), np.ones((1, n + 1), F32)], 1)       # :343
        L_bot = np.concatenate([source_points.T, np.ones((n, 1), F32), tL], 1)             # :344
        L = np.concatenate([L_top, L_mid, L_bot], 0).astype(F32)
        L_inv = np.linalg.inv(L).astype(F32)                                               # :347
        distances = self.U_func(pixel_grid[:, :, None], source_points[:, None, :])         # :352
        ones = np.ones((1, self.num_pixels), F32)
        pixel_distances = np.concatenate([ones, distances], 0).astype(F32)                 # :357
        L_inv = np.ascontiguousarray(L_inv[:, 3:].T)                                       # :358
        return pixel_distances, L_inv

    def _transform(self, inp, theta):
        """:276-296.  theta [B,2,n] absolute control point positions."""
        B = inp.shape[0]
        n = self.num_control_points
        theta = theta.reshape(-1, n)                                                       # :283
        coefficients = np.matmul(theta, self.L_inv).astype(F32).reshape(-1, 2, n + 3)      # :284-285
        right_mat = np.concatenate([self.pixel_grid, self.pixel_distances], 0)             # :288
        tp = seq_matmul_small(coefficients, np.broadcast_to(right_mat, (B,) + right_mat.shape))  # :290
        return tp[:, 0].reshape(-1), tp[:, 1].reshape(-1), coefficients

    def transform(self, inp, theta, forward=True, **kwargs):
        """:137-193.  Returns (output, x_s, y_s)."""
        inp = np.asarray(inp, dtype=F32)
        theta = np.asarray(theta, dtype=F32)
        theta = (self.source_points[None] + theta.reshape(-1, 2, self.num_control_points)).astype(F32)  # :161
        x_s, y_s, _ = self._transform(inp, theta)
        out = _interpolate(inp, x_s, y_s, self.out_size, self.interp_method)
        return out.reshape(-1, self.out_size[0], self.out_size[1], inp.shape[3]), x_s, y_s

    def get_abs_theta(self, theta):
        """:195-217."""
        theta = np.asarray(theta, dtype=F32)
        theta = (self.source_points[None] + theta.reshape(-1, 2, self.num_control_points)).astype(F32)
        theta = theta.transpose(0, 2, 1)
        x = np.clip((theta[:, :, 0:1] + F32(1)) / F32(2), 0, 1)
        y = np.clip((theta[:, :, 1:2] + F32(1)) / F32(2), 0, 1)
        s = self.num_control_points_per_side
        return np.concatenate([x, y], 2).astype(F32).reshape(-1, s, s, 2)

    def get_abs_src_points(self, batch_size):
        """:260-274."""
        sp = self.source_points[None].transpose(0, 2, 1)
        x = np.clip((sp[:, :, 0:1] + F32(1)) / F32(2), 0, 1)
        y = np.clip((sp[:, :, 1:2] + F32(1)) / F32(2), 0, 1)
        s = self.num_control_points_per_side
        return np.concatenate([x, y], 2).astype(F32).reshape(-1, s, s, 2)
