"""Known-answer tests for the CPU oracle, derived from the reference source text (SURVEY.md
section 8c items 1-12).  The reference ships no tests or fixtures of its own, so these are the
only independent pins the oracle has ("parity unpinned" otherwise, oracle/__init__.py)."""
import numpy as np
import pytest

import inputs
from oracle import networks as onet
from oracle import spatial_transformer as ost
from oracle import thin_plate_spline as otps
from oracle import warp_with_optical_flow as oflow
from oracle.tfops import tf_linspace


def test_linspace_is_start_plus_step_times_i():
    x = tf_linspace(-1.0, 1.0, 1280)
    step = np.float32(np.float32(2.0) / np.float32(1279))
    assert x.dtype == np.float32 and x[0] == -1.0
    assert np.array_equal(x, (np.float32(-1.0) + step * np.arange(1280, dtype=np.float32)).astype(np.float32))
    assert abs(float(x[-1]) - 1.0) < 2e-7
    assert np.array_equal(tf_linspace(-1.0, 1.0, 1), np.array([-1.0], np.float32))


def test_tf_warp_zero_flow_is_identity():                      # (1)
    im = inputs.smooth_frames(1, 2, 20, 30)
    assert np.array_equal(oflow.tf_warp(im, np.zeros((2, 20, 30, 2), np.float32), 20, 30), im)


def test_tf_warp_integer_shift_and_zero_ring():                # (2)
    im = inputs.smooth_frames(2, 1, 20, 30)
    f = np.zeros((1, 20, 30, 2), np.float32)
    f[..., 0], f[..., 1] = 4.0, 1.0
    exp = np.zeros_like(im)
    exp[:, :19, :26] = im[:, 1:, 4:]
    assert np.array_equal(oflow.tf_warp(im, f, 20, 30), exp)


def test_tf_warp_far_out_of_bounds_is_zero():                  # (3)
    im = inputs.smooth_frames(3, 1, 12, 16) + 0.5
    f = np.full((1, 12, 16, 2), 100.0, np.float32)
    assert np.array_equal(oflow.tf_warp(im, f, 12, 16), np.zeros_like(im))


def test_tf_warp_half_pixel_is_average():
    im = inputs.smooth_frames(4, 1, 8, 8, 1)
    f = np.zeros((1, 8, 8, 2), np.float32)
    f[..., 0] = 0.5
    out = oflow.tf_warp(im, f, 8, 8)
    assert np.allclose(out[0, :, :7, 0], 0.5 * (im[0, :, :7, 0] + im[0, :, 1:, 0]), atol=1e-7)
    assert np.allclose(out[0, :, 7, 0], 0.5 * im[0, :, 7, 0], atol=1e-7)   # fades into the zero ring


def test_tf_warp_is_scipys_bilinear_with_a_zero_ring():
    """Independent of the restatement: warp_with_optical_flow.py:96-176 samples the frame bilinearly at (x + u, y + v)
    with zeros outside -- SciPy's map_coordinates(order=1, mode="grid-constant", cval=0): it blends into the zero ring too."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(5)
    H, W = 17, 23
    im = rng.uniform(0, 1, (1, H, W, 2)).astype(np.float32)
    flow = rng.uniform(-3.0, 3.0, (1, H, W, 2)).astype(np.float32)
    got = oflow.tf_warp(im, flow, H, W)[0]
    gy, gx = np.meshgrid(np.arange(H, dtype=np.float64), np.arange(W, dtype=np.float64), indexing="ij")
    xq = (gx.astype(np.float32) + flow[0, ..., 0]).astype(np.float64)
    yq = (gy.astype(np.float32) + flow[0, ..., 1]).astype(np.float64)
    for c in range(2):
        ref = map_coordinates(im[0, :, :, c].astype(np.float64), [yq, xq], order=1, mode="grid-constant", cval=0.0)
        assert np.abs(got[..., c] - ref).max() < 2e-6


def test_tps_zero_vector_gives_identity_map():                 # (4)
    B, H, W = 1, 24, 40
    coord = inputs.v_src(B)
    T = otps.solve_system(coord, coord, dtype=np.float64)
    assert np.abs(T[0, :, :3] - np.array([[0, 1, 0], [0, 0, 1]])).max() < 1e-12
    assert np.abs(T[0, :, 3:]).max() < 1e-12
    U = inputs.smooth_frames(5, B, H, W)
    out, xs, ys = otps.ThinPlateSpline(U, coord, np.zeros_like(coord), (H, W))
    xt = np.tile(tf_linspace(-1, 1, W)[None], (H, 1)).reshape(-1)
    assert np.abs(xs - xt).max() < 1e-5
    # sampler A: x = (x_s + 1) * W / 2 -> the last column / row samples x = W, y = H -> 0
    assert np.abs(out[0, :, -1]).max() < 1e-3 and np.abs(out[0, -1, :]).max() < 1e-3


def test_tps_uniform_vector_is_translation():                  # (5)
    B, H, W = 1, 16, 24
    coord = inputs.v_src(B)
    vec = np.zeros_like(coord)
    vec[..., 0], vec[..., 1] = 0.25, -0.125
    _, xs, ys = otps.ThinPlateSpline(inputs.smooth_frames(6, B, H, W), coord, vec, (H, W))
    xt = np.tile(tf_linspace(-1, 1, W)[None], (H, 1)).reshape(-1)
    yt = np.tile(tf_linspace(-1, 1, H)[:, None], (1, W)).reshape(-1)
    assert np.abs(xs - (xt + 0.25)).max() < 1e-5 and np.abs(ys - (yt - 0.125)).max() < 1e-5


def test_tps_interpolates_control_points():                    # (6)
    B = 3
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(7, B, scale=0.1)
    T = otps.solve_system(coord, (coord + vec).astype(np.float32), dtype=np.float64)
    for b in range(B):
        for k in range(25):
            d2 = ((coord[b, k].astype(np.float64) - coord[b].astype(np.float64)) ** 2).sum(1)
            basis = np.concatenate([[1.0], coord[b, k], d2 * np.log(d2 + 1e-6)])
            assert np.abs(T[b] @ basis - (coord[b, k] + vec[b, k])).max() < 1e-5


def test_tps_map_is_scipys_thin_plate_spline():
    """Independent of the restatement: the map of ThinPlateSpline.py:92-166 is THE thin-plate spline through the control
    points (kernel d^2 log(d^2 + 1e-6) = 2 r^2 log r up to the epsilon, affine part, side conditions in W's last three
    rows), so SciPy's RBFInterpolator (kernel r^2 log r, degree 1, no smoothing) through the same points must give the
    same source coordinates everywhere on the grid -- a scaled kernel changes the weights, not the interpolant."""
    from scipy.interpolate import RBFInterpolator
    H, W, B = 24, 40, 2
    coord = inputs.v_src(B).astype(np.float64)
    vec = inputs.control_vectors(3, B, scale=0.2).astype(np.float64)
    _, xs, ys = otps.ThinPlateSpline(np.zeros((B, H, W, 1), np.float32), coord, vec, (H, W))
    xs, ys = xs.reshape(B, -1), ys.reshape(B, -1)
    gx, gy = np.meshgrid(np.linspace(-1, 1, W), np.linspace(-1, 1, H))
    pts = np.stack([gx.ravel(), gy.ravel()], 1)
    for b in range(B):
        ref = RBFInterpolator(coord[b], coord[b] + vec[b], kernel="thin_plate_spline", degree=1)(pts)
        assert np.abs(xs[b] - ref[:, 0]).max() < 2e-5 and np.abs(ys[b] - ref[:, 1]).max() < 2e-5


def test_sampler_a_is_plain_bilinear_inside_the_frame():
    """Independent of the restatement: away from the border ThinPlateSpline.py:30-90 is ordinary bilinear interpolation
    at pixel coordinates ((x + 1) W / 2, (y + 1) H / 2) -- SciPy's map_coordinates(order=1) on the same points."""
    from scipy.ndimage import map_coordinates
    rng = np.random.default_rng(11)
    H, W = 19, 31
    im = rng.uniform(0, 1, (1, H, W, 2)).astype(np.float32)
    n = 400
    xp = rng.uniform(0.0, W - 1.001, n)
    yp = rng.uniform(0.0, H - 1.001, n)
    x = (xp * 2.0 / W - 1.0).astype(np.float32)
    y = (yp * 2.0 / H - 1.0).astype(np.float32)
    got = otps.interpolate_a(np.tile(im, (1, 1, 1, 1)), x[None, :H * W], y[None, :H * W])[0]
    xq = (x.astype(np.float64) + 1.0) * W / 2.0          # the coordinates the float32 inputs really name
    yq = (y.astype(np.float64) + 1.0) * H / 2.0
    for c in range(2):
        ref = map_coordinates(im[0, :, :, c].astype(np.float64), [yq, xq], order=1, mode="nearest")
        assert np.abs(got[:, c] - ref).max() < 2e-5


def test_tps2_is_tps_with_target():
    B, H, W = 2, 16, 24
    U = inputs.smooth_frames(8, B, H, W)
    coord, vec = inputs.v_src(B), inputs.control_vectors(9, B)
    a = otps.ThinPlateSpline(U, coord, vec, (H, W))
    b = otps.ThinPlateSpline2(U, coord, (coord + vec).astype(np.float32), (H, W))
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_sampler_a_out_of_range_cancels_and_jumps_at_border():
    im = np.ones((1, 4, 6, 1), np.float32)
    W, H = 6, 4
    px = np.array([[-0.5, 0.25, 4.75, 5.25, 7.0]], np.float32)           # pixel units
    xs = (px * 2 / W - 1).astype(np.float32)
    ys = np.full_like(xs, (1.5 * 2 / H - 1))
    out = otps.interpolate_a(im, xs, ys)[0, :, 0]
    assert np.allclose(out, [0, 1, 1, 0, 0], atol=1e-6)


def test_projective_identity_and_div_no_nan():                 # (7), (8)
    B, H, W = 1, 10, 14
    im = inputs.smooth_frames(10, B, H, W)
    t = ost.ProjectiveTransformer((H, W))
    ident = np.array([[1, 0, 0, 0, 1, 0, 0, 0]], np.float32)
    assert np.abs(t.transform(im, ident) - im).max() < 1e-5
    xs, ys = t._transform(im, np.array([[1, 0, 0, 0, 1, 0, 1, 0]], np.float32))   # z = x_t + 1
    assert np.all(xs.reshape(H, W)[:, 0] == 0) and np.all(ys.reshape(H, W)[:, 0] == 0)
    assert np.all(np.isfinite(xs))


def test_affine_identity_and_shift():
    B, H, W = 1, 9, 13
    im = inputs.smooth_frames(11, B, H, W)
    t = ost.AffineTransformer((H, W))
    assert np.abs(t.transform(im, np.array([[1, 0, 0, 0, 1, 0]], np.float32)) - im).max() < 1e-5
    # a shift of exactly one pixel in x: 2/(W-1) in normalised units
    out = t.transform(im, np.array([[1, 0, 2.0 / (W - 1), 0, 1, 0]], np.float32))
    assert np.abs(out[0, :, :-1] - im[0, :, 1:]).max() < 1e-5
    assert np.abs(out[0, :, -1]).max() < 1e-5


def test_elastic_zero_theta_is_identity():                     # (9)
    B, H, W = 1, 12, 18
    im = inputs.smooth_frames(12, B, H, W)
    t = ost.ElasticTransformer((H, W))
    out, xs, ys = t.transform(im, np.zeros((B, 32), np.float32))
    assert np.abs(out - im).max() < 1e-4
    assert t.L_inv.shape == (16, 19) and t.pixel_distances.shape == (17, H * W)


def test_scale_rgb_group_reversal():                           # (10)
    x = np.zeros((1, 1, 1, 21), np.float32)
    x[..., :] = np.arange(21, dtype=np.float32) / 255.0
    y = onet.scale_RGB(x)[0, 0, 0]
    exp = np.concatenate([np.arange(14, 21) - 103.939, np.arange(7, 14) - 116.779, np.arange(0, 7) - 123.68])
    assert np.allclose(y, exp, atol=1e-4)
    y3 = onet.scale_RGB(np.array([[[[1.0, 0.5, 0.0]]]], np.float32))[0, 0, 0]   # RGB -> BGR
    assert np.allclose(y3, [0 - 103.939, 127.5 - 116.779, 255 - 123.68], atol=1e-4)


def test_max_pool_same_alignment():                            # (11)
    x = np.arange(6 * 8, dtype=np.float32).reshape(1, 6, 8, 1)
    y = onet.max_pool_3x3_s2_same(x)
    assert y.shape == (1, 3, 4, 1)
    # even size: no pad on the top/left; window of out (0,0) = rows 0..2, cols 0..2
    assert y[0, 0, 0, 0] == x[0, :3, :3, 0].max() and y[0, 2, 3, 0] == x[0, 4:, 6:, 0].max()
    xo = np.arange(5 * 7, dtype=np.float32).reshape(1, 5, 7, 1)
    yo = onet.max_pool_3x3_s2_same(xo)          # odd size: one pad row/col on each side
    assert yo.shape == (1, 3, 4, 1) and yo[0, 0, 0, 0] == xo[0, :2, :2, 0].max()


def test_conv2d_same_stride2_tap_alignment():                  # (12)
    x = np.zeros((1, 8, 8, 1), np.float32)
    x[0, 3, 5, 0] = 1.0
    w = np.arange(9, dtype=np.float32).reshape(3, 3, 1, 1)
    y = onet.conv2d_same_slim(x, w, 2)[0, :, :, 0]
    assert y.shape == (4, 4)
    # explicit pad 1/1 then VALID: out(i,j) covers input rows 2i-1..2i+1
    exp = np.zeros((4, 4), np.float32)
    for i in range(4):
        for j in range(4):
            kh, kw = 3 - (2 * i - 1), 5 - (2 * j - 1)
            if 0 <= kh < 3 and 0 <= kw < 3:
                exp[i, j] = w[kh, kw, 0, 0]
    assert np.array_equal(y, exp)
    w7 = np.random.default_rng(0).standard_normal((7, 7, 1, 1)).astype(np.float32)
    y7 = onet.conv2d_same_slim(x, w7, 2)[0, :, :, 0]     # root conv: pad 3/3, out = ceil(n/2)
    assert y7.shape == (4, 4) and y7[0, 1] == w7[3 - (0 - 3), 5 - (2 - 3), 0, 0]


def test_numpy_and_torch_cnn_agree(synthetic_weights):
    from oracle.cnn_torch import TorchLocNet
    x = inputs.window_frames(13, 2, 40, 56)
    a = onet.localizationNet(x, 25, synthetic_weights)
    b = TorchLocNet(synthetic_weights).forward(x)
    assert a.shape == (2, 25, 2) and np.abs(a - b).max() < 2e-6
    assert 0.002 < np.abs(a).mean() < 0.2      # realistic control-point displacements


def test_resnet_shapes_match_survey_appendix(synthetic_weights):
    taps = {}
    x = inputs.window_frames(14, 1, 45, 80)            # 720p / 16
    onet.localizationNet(x, 25, synthetic_weights, taps=taps)
    assert taps["conv1"].shape == (1, 23, 40, 64) and taps["pool1"].shape == (1, 12, 20, 64)
    assert taps["block1/unit_3"].shape == (1, 6, 10, 256) and taps["block2/unit_4"].shape == (1, 3, 5, 512)
    assert taps["block3/unit_6"].shape == (1, 2, 3, 1024) and taps["block4/unit_3"].shape == (1, 2, 3, 2048)


def test_float64_arbiter_is_the_same_map():
    """oracle/thin_plate_spline.py `source_coords_f64` / `interpolate_a_f64` (the float64 arbiter of the 4K tests) against
    the float32 restatement at a size where float32 noise is small, against SciPy's thin-plate RBF (an implementation
    that shares nothing with either), and on the known answers: zero vectors give the unit grid, a uniform vector a shift."""
    from scipy.interpolate import RBFInterpolator
    B, H, W = 2, 40, 56
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(901, B, scale=0.1)
    rhs = (coord + vec).astype(np.float32)
    xo, yo = otps.source_coords(otps.solve_system(coord, rhs), coord, H, W)
    x64, y64 = otps.source_coords_f64(coord, rhs, H, W)
    assert np.abs(x64 - xo).max() < 2e-5 and np.abs(y64 - yo).max() < 2e-5
    xr, yr = otps.source_coords_f64(coord, rhs, H, W, rows=[3, 17])
    assert np.array_equal(xr[:, :W], x64[:, 3 * W:4 * W]) and np.array_equal(yr[:, W:], y64[:, 17 * W:18 * W])
    xl, yl = otps.meshgrid_xy(H, W)
    pts = np.stack([np.tile(xl, H), np.repeat(yl, W)], 1).astype(np.float64)
    rbf = RBFInterpolator(coord[0].astype(np.float64), rhs[0].astype(np.float64), kernel="thin_plate_spline", degree=1)(pts)
    assert np.abs(rbf[:, 0] - x64[0]).max() < 2e-5 and np.abs(rbf[:, 1] - y64[0]).max() < 2e-5   # (r^2 log r vs d2 log(d2 + 1e-6))
    x0, y0 = otps.source_coords_f64(coord, coord, H, W)
    assert np.abs(x0[0] - pts[:, 0]).max() < 1e-12 and np.abs(y0[0] - pts[:, 1]).max() < 1e-12
    xs, _ = otps.source_coords_f64(coord, (coord + np.float32([0.25, 0.0])).astype(np.float32), H, W)
    assert np.abs(xs[0] - (pts[:, 0] + 0.25)).max() < 1e-12
    U = inputs.smooth_frames(902, B, H, W)
    a32 = otps.interpolate_a(U, xo, yo)
    a64 = otps.interpolate_a_f64(U, x64, y64)
    border = otps.border_discontinuity_mask(xo, yo, H, W, delta=1e-2)
    assert np.abs(a64 - a32).max(axis=2)[~border].max() < 1e-4


def test_torch_tps_port_is_the_numpy_restatement():
    """oracle/tps_torch.py (the multi-threaded warp leg of bench.py's cpu_baseline) computes what the NumPy
    restatement of ThinPlateSpline.py computes: grid within float32 evaluation noise, pixels outside the counted
    border-discontinuity pixels."""
    from oracle import tps_torch
    B, H, W = 2, 40, 56
    U = inputs.smooth_frames(911, B, H, W)
    coord, vec = inputs.v_src(B), inputs.control_vectors(912, B, scale=0.1)
    out, xs, ys = otps.ThinPlateSpline(U, coord, vec, (H, W))
    tout, txs, tys = tps_torch.ThinPlateSpline(U, coord, vec, (H, W))
    assert tout.shape == out.shape and tout.dtype == np.float32
    assert np.abs(txs - xs).max() < 2e-5 and np.abs(tys - ys).max() < 2e-5
    border = otps.border_discontinuity_mask(xs, ys, H, W, delta=1e-2).reshape(B, H, W)
    assert np.abs(tout - out).max(axis=3)[~border].max() < 1e-4
