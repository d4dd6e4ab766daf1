"""Oracle (test infrastructure): ThinPlateSpline.py / ThinPlateSpline2.py restated in NumPy
float32, op for op.  parity unpinned (see oracle/__init__.py).

Reference: ThinPlateSpline.py:4-170 (vector mode), ThinPlateSpline2.py:4-170 (target mode,
differs only at line 160).
"""
import numpy as np

from .tfops import F32, add_n4, seq_matmul_small, tf_linspace

EPS = F32(1e-6)  # ThinPlateSpline.py:106,153


def solve_system(coord, rhs_points, dtype=np.float32):
    """ThinPlateSpline.py:143-166 `_solve_system`.

    coord [B,P,2]; rhs_points [B,P,2] is `coord+vector` (ThinPlateSpline.py:161) or
    `target` (ThinPlateSpline2.py:160).  Returns T [B,2,P+3].  ``dtype=float64`` gives the
    exactly-rounded answer used to bound the float32 LU noise (cond(W) ~ 4e2).
    """
    coord = np.asarray(coord, dtype=dtype)
    rhs_points = np.asarray(rhs_points, dtype=dtype)
    B, P, _ = coord.shape
    ones = np.ones((B, P, 1), dtype=dtype)
    p = np.concatenate([ones, coord], 2)                      # :148
    p_1 = p.reshape(B, P, 1, 3)                               # :150
    p_2 = p.reshape(B, 1, P, 3)                               # :151
    d2 = np.sum(np.square(p_1 - p_2), 3, dtype=dtype)         # :152
    r = (d2 * np.log(d2 + dtype(1e-6))).astype(dtype)         # :153
    zeros = np.zeros((B, 3, 3), dtype=dtype)
    W_0 = np.concatenate([p, r], 2)                           # :156
    W_1 = np.concatenate([zeros, p.transpose(0, 2, 1)], 2)    # :157
    W = np.concatenate([W_0, W_1], 1)                         # :158
    W_inv = np.linalg.inv(W).astype(dtype)                    # :159 tf.matrix_inverse
    tp = np.concatenate([rhs_points, np.zeros((B, 3, 2), dtype=dtype)], 1)  # :161-162
    T = np.matmul(W_inv, tp).astype(dtype)                    # :163
    return np.ascontiguousarray(T.transpose(0, 2, 1))         # :164


def meshgrid_xy(height, width):
    """x_t per column, y_t per row (ThinPlateSpline.py:93-96)."""
    return tf_linspace(-1.0, 1.0, width), tf_linspace(-1.0, 1.0, height)


def source_coords(T, coord, out_h, out_w):
    """ThinPlateSpline.py:92-134: grid = [1; x_t; y_t; r_0..r_{P-1}], (x_s, y_s) = T @ grid.

    Returns x_s, y_s as [B, out_h*out_w] float32 (the reference returns them flattened
    over the batch too: reshape(-1) gives exactly that).
    """
    T = np.asarray(T, dtype=F32)
    coord = np.asarray(coord, dtype=F32)
    B, P, _ = coord.shape
    xl, yl = meshgrid_xy(out_h, out_w)
    x_t = np.tile(xl.reshape(1, out_w), (out_h, 1)).reshape(1, -1)   # :93-94,98
    y_t = np.tile(yl.reshape(out_h, 1), (1, out_w)).reshape(1, -1)   # :95-96,99
    xs = np.empty((B, out_h * out_w), dtype=F32)
    ys = np.empty((B, out_h * out_w), dtype=F32)
    for b in range(B):
        px = coord[b, :, 0:1]                                         # :102
        py = coord[b, :, 1:2]                                         # :103
        d2 = (np.square(x_t - px) + np.square(y_t - py)).astype(F32)  # :104
        r = (d2 * np.log((d2 + EPS).astype(F32)).astype(F32)).astype(F32)  # :105
        grid = np.concatenate([np.ones_like(x_t), x_t, y_t, r], 0)   # :110
        Tg = seq_matmul_small(T[b:b + 1], grid[None])[0]             # :129
        xs[b] = Tg[0]
        ys[b] = Tg[1]
    return xs, ys


def interpolate_a(im, x, y):
    """Sampler A, ThinPlateSpline.py:30-90 `_interpolate`.

    im [B,H,W,C]; x, y [B, N] normalised source coords (N == H*W: the base offset at
    ThinPlateSpline.py:63 needs out size == input size).  Returns [B, N, C].
    """
    im = np.asarray(im, dtype=F32)
    B, H, W, C = im.shape
    x = np.asarray(x, dtype=F32)
    y = np.asarray(y, dtype=F32)
    width_f = F32(W)
    height_f = F32(H)
    x = ((x + F32(1.0)).astype(F32) * width_f).astype(F32) / F32(2.0)    # :48
    y = ((y + F32(1.0)).astype(F32) * height_f).astype(F32) / F32(2.0)   # :49
    x0 = np.floor(x).astype(np.int32)                                    # :52
    x1 = x0 + 1
    y0 = np.floor(y).astype(np.int32)
    y1 = y0 + 1
    x0 = np.clip(x0, 0, W - 1)                                           # :57-60
    x1 = np.clip(x1, 0, W - 1)
    y0 = np.clip(y0, 0, H - 1)
    y1 = np.clip(y1, 0, H - 1)
    bidx = np.arange(B)[:, None]
    Ia = im[bidx, y0, x0]                                                # :75 (x0,y0)
    Ib = im[bidx, y1, x0]                                                # :76 (x0,y1)
    Ic = im[bidx, y0, x1]                                                # :77 (x1,y0)
    Id = im[bidx, y1, x1]                                                # :78 (x1,y1)
    x0_f = x0.astype(F32)
    x1_f = x1.astype(F32)
    y0_f = y0.astype(F32)
    y1_f = y1.astype(F32)
    wa = ((x1_f - x) * (y1_f - y)).astype(F32)[..., None]                # :85
    wb = ((x1_f - x) * (y - y0_f)).astype(F32)[..., None]                # :86
    wc = ((x - x0_f) * (y1_f - y)).astype(F32)[..., None]                # :87
    wd = ((x - x0_f) * (y - y0_f)).astype(F32)[..., None]                # :88
    return add_n4(wa * Ia, wb * Ib, wc * Ic, wd * Id)                    # :89


def _tps(U, coord, rhs_points, out_size):
    U = np.asarray(U, dtype=F32)
    B, H, W, C = U.shape
    out_h, out_w = int(out_size[0]), int(out_size[1])
    T = solve_system(coord, rhs_points)
    xs, ys = source_coords(T, coord, out_h, out_w)
    out = interpolate_a(U, xs, ys).reshape(B, out_h, out_w, C)           # :138-140
    return out, xs.reshape(-1), ys.reshape(-1)                           # :141,170


def ThinPlateSpline(U, coord, vector, out_size):
    """ThinPlateSpline.py:4: returns (output, x_s_flat, y_s_flat)."""
    coord = np.asarray(coord, dtype=F32)
    vector = np.asarray(vector, dtype=F32)
    return _tps(U, coord, (coord + vector).astype(F32), out_size)


def ThinPlateSpline2(U, source, target, out_size):
    """ThinPlateSpline2.py:4: as above with RHS = target (line 160)."""
    return _tps(U, np.asarray(source, dtype=F32), np.asarray(target, dtype=F32), out_size)


def border_discontinuity_mask(xs, ys, H, W, delta=1e-2):
    """Pixels whose oracle source coordinate lies within `delta` px of a sampler-A jump
    ({0, W-1} in x, {0, H-1} in y; SURVEY.md section 7 hard part 3).  xs, ys normalised."""
    x = (np.asarray(xs, dtype=np.float64) + 1.0) * W / 2.0
    y = (np.asarray(ys, dtype=np.float64) + 1.0) * H / 2.0
    m = np.zeros(x.shape, dtype=bool)
    for v, edges in ((x, (0.0, W - 1.0)), (y, (0.0, H - 1.0))):
        for e in edges:
            m |= np.abs(v - e) < delta
    return m


# ------------------------------------------------------------------------------------------------------------
# A float64 ARBITER of the same map (not a restatement of reference arithmetic: the reference computes in float32).
# Two correct float32 evaluations of the 28-term map differ by their rounding noise (SURVEY.md section 7 hard part
# 2: ~3e-3 px at W = 3840), so "GPU against the float32 oracle" mixes the GPU's error with the oracle's own.  The
# arbiter evaluates the map the reference DEFINES -- the same float32 inputs (control points, right-hand side, the
# float32 grid values of tf.linspace, the float32 constant 1e-6), every operation in float64 -- and lets a test ask
# the two float32 evaluations separately how far each is from it.
# ------------------------------------------------------------------------------------------------------------
def source_coords_f64(coord, rhs_points, out_h, out_w, rows=None):
    """(x_s, y_s) of ThinPlateSpline.py:92-134,143-166 in float64, [B, n_rows*out_w]; `rows` (default all) selects
    output rows, so a 3840x2160 grid can be checked a band at a time."""
    coord64 = np.asarray(np.asarray(coord, dtype=F32), dtype=np.float64)
    T = solve_system(coord64, np.asarray(np.asarray(rhs_points, dtype=F32), dtype=np.float64), dtype=np.float64)
    xl, yl = meshgrid_xy(out_h, out_w)
    xl = xl.astype(np.float64)
    yl = yl.astype(np.float64)
    if rows is not None:
        yl = yl[np.asarray(rows)]
    B, P, _ = coord64.shape
    eps = np.float64(EPS)
    xs = np.empty((B, yl.size * out_w), dtype=np.float64)
    ys = np.empty_like(xs)
    for b in range(B):
        dx2 = np.square(xl[None, :] - coord64[b, :, 0:1])          # [P, W]
        dy2 = np.square(yl[None, :] - coord64[b, :, 1:2])          # [P, rows]
        d2 = dy2[:, :, None] + dx2[:, None, :]                      # [P, rows, W]
        r = (d2 * np.log(d2 + eps)).reshape(P, -1)
        x_t = np.tile(xl[None, :], (yl.size, 1)).reshape(-1)
        y_t = np.tile(yl[:, None], (1, out_w)).reshape(-1)
        for k, dst in ((0, xs), (1, ys)):
            dst[b] = T[b, k, 0] + T[b, k, 1] * x_t + T[b, k, 2] * y_t + T[b, k, 3:] @ r
    return xs, ys


def interpolate_a_f64(im, x, y):
    """Sampler A (ThinPlateSpline.py:30-90) evaluated in float64 at float64 coordinates x, y [B,N]: the frame the
    reference's definition yields when the map is evaluated without float32 noise.  Returns [B,N,C] float64."""
    im = np.asarray(im, dtype=np.float64)
    B, H, W, C = im.shape
    x = (np.asarray(x, dtype=np.float64) + 1.0) * W / 2.0
    y = (np.asarray(y, dtype=np.float64) + 1.0) * H / 2.0
    x0 = np.floor(x).astype(np.int64)
    y0 = np.floor(y).astype(np.int64)
    x1, y1 = x0 + 1, y0 + 1
    x0, x1 = np.clip(x0, 0, W - 1), np.clip(x1, 0, W - 1)
    y0, y1 = np.clip(y0, 0, H - 1), np.clip(y1, 0, H - 1)
    bidx = np.arange(B)[:, None]
    wa = ((x1 - x) * (y1 - y))[..., None]
    wb = ((x1 - x) * (y - y0))[..., None]
    wc = ((x - x0) * (y1 - y))[..., None]
    wd = ((x - x0) * (y - y0))[..., None]
    return wa * im[bidx, y0, x0] + wb * im[bidx, y1, x0] + wc * im[bidx, y0, x1] + wd * im[bidx, y1, x1]
