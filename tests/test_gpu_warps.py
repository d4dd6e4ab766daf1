"""GPU parity: the HIP warps (through the C ABI, via the reference-shaped facade) against the
CPU oracle on the same seeded inputs.  Tolerances are stated per test; the reference graph
is float32 and so are both sides."""
import numpy as np
import pytest

import inputs
from oracle import spatial_transformer as ost
from oracle import thin_plate_spline as otps
from oracle import warp_with_optical_flow as oflow
from oracle.networks import scale_RGB as o_scale_RGB

pytestmark = pytest.mark.gpu

# warped-pixel tolerance of BASELINE.json (max abs err vs reference semantics)
PIX_TOL = 1e-3


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    from coupe.dvsg_amd import _lib
    _lib.load()
    return torch.device("cuda:0")


def _grid_err_px(xs, ys, xo, yo, H, W):
    ex = np.abs(np.asarray(xs, np.float64) - np.asarray(xo, np.float64)).max() * W / 2.0
    ey = np.abs(np.asarray(ys, np.float64) - np.asarray(yo, np.float64)).max() * H / 2.0
    return max(ex, ey)


# ------------------------------------------------------------------------------- TPS solve
@pytest.mark.parametrize("P_side", [5, 4, 7])
def test_tps_solve_matches_oracle(dev, P_side):
    from coupe.dvsg_amd.ThinPlateSpline import solve_system
    B = 6
    lin = np.linspace(-1, 1, P_side)
    coord = np.tile(np.array([[x, y] for y in lin for x in lin], np.float32)[None], (B, 1, 1))
    rng = np.random.default_rng(3)
    coord = coord + (0.02 * rng.standard_normal(coord.shape)).astype(np.float32)  # batch-varying
    vec = inputs.control_vectors(4, B, P_side * P_side)
    T = solve_system(coord, vec, True)
    T64 = otps.solve_system(coord, (coord + vec).astype(np.float32), dtype=np.float64)
    T32 = otps.solve_system(coord, (coord + vec).astype(np.float32))
    # float64 solve of the float32-built system: agrees with the exactly rounded answer to
    # ~1e-5 (the matrix entries themselves carry float32 log rounding, amplified by cond ~4e2)
    assert np.abs(T - T64).max() < 2e-5
    # and sits inside the float32-LU noise band of the reference's own tf.matrix_inverse
    assert np.abs(T - T32).max() < 5e-4
    T2 = solve_system(coord, (coord + vec).astype(np.float32), False)  # ThinPlateSpline2 mode
    assert np.abs(T2 - T).max() < 1e-6


# ---------------------------------------------------------------- TPS grid + sampler A
@pytest.mark.parametrize("B,H,W,C", [(2, 72, 128, 3), (1, 37, 53, 3), (2, 40, 300, 1), (1, 33, 65, 5)])
def test_tps_warp_given_T(dev, B, H, W, C):
    """Isolates grid generation + sampler A: feed the ORACLE's T through the C ABI."""
    import torch
    from coupe.dvsg_amd import _lib
    U = inputs.smooth_frames(11, B, H, W, C)
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(12, B)
    T = otps.solve_system(coord, (coord + vec).astype(np.float32))
    xo, yo = otps.source_coords(T, coord, H, W)
    ref = otps.interpolate_a(U, xo, yo).reshape(B, H, W, C)
    tU, tc, tT = (torch.from_numpy(a).to(dev) for a in (U, coord, T))
    out = torch.empty((B, H, W, C), device=dev)
    xs = torch.empty((B * H * W,), device=dev)
    ys = torch.empty((B * H * W,), device=dev)
    _lib.call("dvsg_tps_warp_f32", tU.data_ptr(), tc.data_ptr(), tT.data_ptr(), B, H, W, C, 25, H, W,
              out.data_ptr(), xs.data_ptr(), ys.data_ptr(), 0)
    torch.cuda.synchronize()
    xs, ys, out = xs.cpu().numpy().reshape(B, -1), ys.cpu().numpy().reshape(B, -1), out.cpu().numpy()
    gerr = _grid_err_px(xs, ys, xo, yo, H, W)
    assert gerr < 2e-3, "grid error %.3g px" % gerr
    mask = otps.border_discontinuity_mask(xo, yo, H, W).reshape(B, H, W)
    err = np.abs(out - ref).max(axis=3)
    assert err[~mask].max() < PIX_TOL
    assert mask.mean() < 0.05


def test_tps_warp_white_noise_stress(dev):
    """SURVEY.md section 7, hard part 2: the stress case reported separately.  On white-noise frames the
    image gradient is up to 1 per pixel, so a warped pixel may differ by the full source-coordinate
    error (in pixels, both axes) -- and by no more than that."""
    import torch
    from coupe.dvsg_amd import _lib
    B, H, W, C = 2, 72, 128, 3
    U = np.random.default_rng(31).uniform(0.0, 1.0, (B, H, W, C)).astype(np.float32)
    coord = inputs.v_src(B)
    T = otps.solve_system(coord, (coord + inputs.control_vectors(32, B)).astype(np.float32))
    xo, yo = otps.source_coords(T, coord, H, W)
    ref = otps.interpolate_a(U, xo, yo).reshape(B, H, W, C)
    tU, tc, tT = (torch.from_numpy(a).to(dev) for a in (U, coord, T))
    out = torch.empty((B, H, W, C), device=dev)
    xs = torch.empty((B * H * W,), device=dev)
    ys = torch.empty((B * H * W,), device=dev)
    _lib.call("dvsg_tps_warp_f32", tU.data_ptr(), tc.data_ptr(), tT.data_ptr(), B, H, W, C, 25, H, W,
              out.data_ptr(), xs.data_ptr(), ys.data_ptr(), 0)
    torch.cuda.synchronize()
    xs, ys = xs.cpu().numpy().reshape(B, -1), ys.cpu().numpy().reshape(B, -1)
    ex = np.abs(xs - xo).reshape(B, H, W) * W / 2
    ey = np.abs(ys - yo).reshape(B, H, W) * H / 2
    mask = otps.border_discontinuity_mask(xo, yo, H, W).reshape(B, H, W)
    err = np.abs(out.cpu().numpy() - ref).max(axis=3)
    # slack: the pixel coordinate (x + 1) * W / 2 itself carries half an ulp of W (4e-6 px at W = 128)
    slack = err - 2.0 * (ex + ey)
    assert slack[~mask].max() <= 2e-5, "worst excess %.3g (grid error there %.3g px)" % (slack[~mask].max(), (ex + ey)[~mask][np.argmax(slack[~mask])])
    assert err[~mask].max() < 5e-3


@pytest.mark.parametrize("out_size", [(72, 128), (36, 64), (50, 90)])
def test_thin_plate_spline_facade(dev, out_size):
    """Whole operator (solve + grid + sampler A), reference signature and return order."""
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline, ThinPlateSpline2
    B, H, W = 2, 72, 128
    U = inputs.smooth_frames(21, B, H, W)
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(22, B)
    out, x, y = ThinPlateSpline(U, coord, vec, out_size)
    ro, rx, ry = otps.ThinPlateSpline(U, coord, vec, out_size)
    assert out.shape == ro.shape and x.shape == rx.shape and y.shape == ry.shape
    oh, ow = out_size
    gerr = _grid_err_px(x, y, rx, ry, H, W)
    assert gerr < 2e-2, "grid error %.3g px" % gerr   # includes the float32 LU noise of the oracle's T
    mask = otps.border_discontinuity_mask(rx, ry, H, W, delta=3e-2).reshape(B, oh, ow)
    err = np.abs(out - ro).max(axis=3)
    assert err[~mask].max() < 3e-3
    out2, x2, y2 = ThinPlateSpline2(U, coord, (coord + vec).astype(np.float32), out_size)
    assert np.abs(x2 - x).max() < 1e-6 and np.abs(out2 - out).max() < 1e-5


def test_tps_known_answers(dev):
    """zero vector -> x_s = x_t; uniform vector -> translation."""
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline
    B, H, W = 1, 48, 80
    U = inputs.smooth_frames(31, B, H, W)
    coord = inputs.v_src(B)
    _, x, y = ThinPlateSpline(U, coord, np.zeros_like(coord), (H, W))
    xt = np.tile(np.linspace(-1, 1, W)[None], (H, 1)).reshape(-1)
    yt = np.tile(np.linspace(-1, 1, H)[:, None], (1, W)).reshape(-1)
    assert np.abs(x - xt).max() < 5e-6 and np.abs(y - yt).max() < 5e-6
    shift = np.zeros_like(coord)
    shift[..., 0] = 0.125
    shift[..., 1] = -0.25
    _, x, y = ThinPlateSpline(U, coord, shift, (H, W))
    assert np.abs(x - (xt + 0.125)).max() < 5e-6 and np.abs(y - (yt - 0.25)).max() < 5e-6


def test_singular_control_points_raise(dev):
    """tf.matrix_inverse (ThinPlateSpline.py:159) raises InvalidArgument on a non-invertible system:
    repeated or collinear control points.  The facade raises too; the raw stream-ordered entry
    point marks such a sample's T as NaN (never a finite garbage map) and leaves the others alone."""
    import torch
    from coupe.dvsg_amd import DvsgError, _lib
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline, solve_system
    B, H, W = 3, 24, 32
    U = inputs.smooth_frames(35, B, H, W)
    coord = inputs.v_src(B).copy()
    vec = inputs.control_vectors(36, B)
    good = solve_system(coord, vec)
    coord[1, 7] = coord[1, 3]                      # sample 1: a repeated control point
    with pytest.raises(DvsgError, match="1 of 3 samples"):
        ThinPlateSpline(U, coord, vec, (H, W))
    line = inputs.v_src(1).copy()
    line[0, :, 1] = 0.5 * line[0, :, 0]            # all 25 points on one line
    with pytest.raises(DvsgError, match="not invertible"):
        solve_system(line, vec[:1])
    ct, vt = torch.from_numpy(coord).to(dev), torch.from_numpy(vec).to(dev)
    T = torch.zeros((B, 2, 28), device=dev)
    _lib.call("dvsg_tps_solve_f32", ct.data_ptr(), vt.data_ptr(), 1, B, 25, T.data_ptr(),
              torch.cuda.current_stream().cuda_stream)
    T = T.cpu().numpy()
    assert np.isnan(T[1]).all() and np.array_equal(T[0], good[0]) and np.array_equal(T[2], good[2])


def test_tps_torch_tensors_stay_on_device(dev):
    import torch
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline
    U = torch.from_numpy(inputs.smooth_frames(41, 1, 32, 48)).to(dev)
    coord = torch.from_numpy(inputs.v_src(1)).to(dev)
    out, x, y = ThinPlateSpline(U, coord, torch.zeros_like(coord), (32, 48))
    assert out.is_cuda and x.is_cuda and y.is_cuda and out.shape == (1, 32, 48, 3)


# ----------------------------------------------------------------------- tf_warp (sampler C)
@pytest.mark.parametrize("B,H,W,C", [(2, 72, 128, 3), (1, 31, 47, 3), (1, 40, 64, 2)])
def test_tf_warp_matches_oracle(dev, B, H, W, C):
    from coupe.dvsg_amd.warp_with_optical_flow import tf_warp
    im = inputs.smooth_frames(51, B, H, W, C)
    flow = inputs.smooth_flow(52, B, H, W)
    out = tf_warp(im, flow, H, W)
    ref = oflow.tf_warp(im, flow, H, W)
    # every op is a separately rounded float32 op on both sides: expect (near) bit equality
    assert np.abs(out - ref).max() <= 1e-6


@pytest.mark.parametrize("B,H,W", [(2, 37, 53), (1, 8, 128), (1, 9, 129), (3, 64, 300), (1, 1, 1), (2, 5, 4), (1, 200, 131),
                                   (2, 180, 320), (1, 720, 1280)])
def test_tf_warp_strip_kernel_gives_the_gather_kernel_bits(dev, B, H, W):
    """`dvsg_flow_warp_f32` on RGB frames streams the source rows of a 128-column strip through an LDS ring (window: the step
    +- 12 px) and takes the taps from there; a pixel whose flow leaves the window gathers from global memory.  Same geometry
    and blend functions on the same values as the gather kernel (`flow_tiled` = 0): the SAME BITS -- for the smooth flow
    of BASELINE configs[2], white noise, a constant flow, a flow that leaves the margins everywhere, flows that point at
    and beyond every image border, and NaN / Inf free results on all of them; also against the oracle (<= 1e-6)."""
    import torch
    from coupe.dvsg_amd import _lib
    rng = np.random.default_rng(1000 * H + W)
    im = torch.from_numpy(inputs.smooth_frames(53, B, H, W, 3)).to(dev)
    st = torch.cuda.current_stream().cuda_stream
    flows = {"cfg3": inputs.smooth_flow(54, B, H, W),
             "noise": (4.0 * rng.standard_normal((B, H, W, 2))).astype(np.float32),
             "const": np.broadcast_to(np.float32([3.3, -1.7]), (B, H, W, 2)).copy(),
             "wide": (40.0 * rng.standard_normal((B, H, W, 2))).astype(np.float32),            # beyond the +- 12 px window
             "edges": np.stack([rng.uniform(-W - 3, W + 3, (B, H, W)), rng.uniform(-H - 3, H + 3, (B, H, W))], -1).astype(np.float32),
             "integer": np.rint(6.0 * rng.standard_normal((B, H, W, 2))).astype(np.float32)}
    for name, f in flows.items():
        fl = torch.from_numpy(f).to(dev)
        outs = {}
        try:
            for v in (0, 1, 2):
                _lib.call("dvsg_debug_set_option", b"flow_tiled", v)
                out = torch.full((B, H, W, 3), float("nan"), device=dev)
                _lib.call("dvsg_flow_warp_f32", im.data_ptr(), fl.data_ptr(), B, H, W, 3, out.data_ptr(), st)
                outs[v] = out
        finally:
            _lib.call("dvsg_debug_set_option", b"flow_tiled", 1)
        assert bool(torch.isfinite(outs[1]).all()), name
        assert torch.equal(outs[1], outs[0]) and torch.equal(outs[2], outs[0]), (name, float((outs[1] - outs[0]).abs().max()))
        if H * W <= 64 * 300:
            assert np.abs(outs[1].cpu().numpy() - oflow.tf_warp(im.cpu().numpy(), f, H, W)).max() <= 1e-6, name
    # a frame tensor off the 16-byte grid takes the gather kernel (the strip kernel's chunks are aligned loads): same result
    if H * W >= 4:
        buf = torch.empty(B * H * W * 3 + 1, device=dev)
        view = buf[1:].reshape(B, H, W, 3)
        view.copy_(im)
        fl = torch.from_numpy(flows["cfg3"]).to(dev)
        out = torch.empty((B, H, W, 3), device=dev)
        ref = torch.empty((B, H, W, 3), device=dev)
        _lib.call("dvsg_flow_warp_f32", view.data_ptr(), fl.data_ptr(), B, H, W, 3, out.data_ptr(), st)
        _lib.call("dvsg_flow_warp_f32", im.data_ptr(), fl.data_ptr(), B, H, W, 3, ref.data_ptr(), st)
        assert torch.equal(out, ref)


def test_tf_warp_known_answers(dev):
    """SURVEY.md 8c (1)-(3)."""
    from coupe.dvsg_amd.warp_with_optical_flow import tf_warp
    B, H, W = 1, 24, 40
    im = inputs.smooth_frames(61, B, H, W)
    zero = np.zeros((B, H, W, 2), np.float32)
    assert np.array_equal(tf_warp(im, zero, H, W), im)
    f = zero.copy()
    f[..., 0] = 3.0
    f[..., 1] = -2.0
    out = tf_warp(im, f, H, W)
    exp = np.zeros_like(im)
    exp[:, 2:, :W - 3] = im[:, :H - 2, 3:]
    assert np.array_equal(out, exp)
    f[..., 0] = W + 1.0
    assert np.array_equal(tf_warp(im, f, H, W), np.zeros_like(im))
    with pytest.raises(ValueError):
        tf_warp(im, zero, H - 1, W)


# ------------------------------------------------------------- spatial transformers (sampler B)
def test_bilinear_interp_matches_oracle(dev):
    from coupe.dvsg_amd.spatial_transformer import bilinear_interp
    B, H, W, C = 2, 40, 72, 3
    im = inputs.smooth_frames(71, B, H, W, C)
    rng = np.random.default_rng(72)
    oh, ow = 33, 50
    x = rng.uniform(-1.3, 1.3, B * oh * ow).astype(np.float32)
    y = rng.uniform(-1.3, 1.3, B * oh * ow).astype(np.float32)
    out = bilinear_interp(im, x, y, (oh, ow))
    ref = ost.bilinear_interp(im, x, y, (oh, ow))
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 1e-6


def test_affine_transformer(dev):
    from coupe.dvsg_amd.spatial_transformer import AffineTransformer
    B, H, W = 3, 40, 72
    im = inputs.smooth_frames(81, B, H, W)
    rng = np.random.default_rng(82)
    theta = (np.array([1, 0, 0, 0, 1, 0], np.float32)[None] + 0.1 * rng.standard_normal((B, 6))).astype(np.float32)
    out = AffineTransformer((H, W)).transform(im, theta)
    ref = ost.AffineTransformer((H, W)).transform(im, theta)
    assert np.abs(out - ref).max() <= 2e-6
    ident = np.tile(np.array([1, 0, 0, 0, 1, 0], np.float32)[None], (B, 1))
    out = AffineTransformer((H, W)).transform(im, ident)
    assert np.abs(out - im).max() < 2e-5   # identity up to 1 ulp of coordinate (W-1 scaling)


def test_projective_transformer(dev):
    from coupe.dvsg_amd.spatial_transformer import ProjectiveTransformer
    B, H, W, C = 2, 40, 72, 18   # random_mask warps an 18-channel mask (model.py:160-164)
    im = inputs.smooth_frames(91, B, H, W, C)
    rng = np.random.default_rng(92)
    Hm = rng.uniform(-1, 1, (B, 8)) * np.array([0.1, 0.1, 0.5, 0.1, 0.1, 0.5, 0.1, 0.1])
    Hm = (Hm + np.array([1, 0, 0, 0, 1, 0, 0, 0])).astype(np.float32)
    t = ProjectiveTransformer((H, W))
    out = t.transform(im, Hm)
    o = ost.ProjectiveTransformer((H, W))
    ref = o.transform(im, Hm)
    assert out.shape == ref.shape
    assert np.abs(out - ref).max() <= 5e-5
    xs, ys = t._transform(im, Hm)
    rx, ry = o._transform(im, Hm)
    assert np.abs(xs - rx).max() < 1e-6 and np.abs(ys - ry).max() < 1e-6
    # tf.div_no_nan: theta = [1,0,0, 0,1,0, 1,0] gives z = x_t + 1, exactly 0 on
    # the first output column (x_t = -1) -> coordinates 0 there
    theta = np.tile(np.array([1, 0, 0, 0, 1, 0, 1, 0], np.float32)[None], (B, 1))
    xs, ys = t._transform(im, theta)
    rx, ry = o._transform(im, theta)
    assert np.abs(xs - rx).max() < 1e-6 and np.abs(ys - ry).max() < 1e-6
    assert np.all(xs.reshape(B, H, W)[:, :, 0] == 0.0) and np.all(ys.reshape(B, H, W)[:, :, 0] == 0.0)


@pytest.mark.parametrize("side", [4, 3])
def test_elastic_transformer(dev, side):
    from coupe.dvsg_amd.spatial_transformer import ElasticTransformer
    B, H, W = 2, 40, 72
    n = side * side
    im = inputs.smooth_frames(101, B, H, W)
    rng = np.random.default_rng(102)
    theta = (0.05 * rng.standard_normal((B, 2 * n))).astype(np.float32)
    t = ElasticTransformer((H, W), param_dim=2 * n, param_dim_per_side=side)
    o = ost.ElasticTransformer((H, W), param_dim=2 * n, param_dim_per_side=side)
    assert np.abs(t.L_inv.cpu().numpy() - o.L_inv).max() < 2e-4      # float64 vs float32 inverse
    out, xs, ys = t.transform(im, theta)
    ref, rx, ry = o.transform(im, theta)
    assert out.shape == ref.shape
    assert np.abs(xs - rx).max() * W / 2 < 2e-2 and np.abs(ys - ry).max() * H / 2 < 2e-2
    assert np.abs(out - ref).max() < 3e-3
    out0, xs0, _ = t.transform(im, np.zeros_like(theta))
    assert np.abs(out0 - im).max() < 1e-4
    assert np.abs(t.get_abs_theta(theta) - o.get_abs_theta(theta)).max() < 1e-6


def test_scale_rgb(dev):
    from coupe.dvsg_amd.networks import scale_RGB
    x = inputs.window_frames(111, 2, 16, 24)
    assert np.array_equal(scale_RGB(x), o_scale_RGB(x))
    x3 = inputs.smooth_frames(112, 1, 8, 8, 3)
    assert np.array_equal(scale_RGB(x3), o_scale_RGB(x3))


def test_argument_errors(dev):
    from coupe.dvsg_amd import DvsgError, _lib
    with pytest.raises(DvsgError, match="P=2"):
        _lib.call("dvsg_tps_solve_f32", 8, 8, 1, 1, 2, 8, 0)
    with pytest.raises(DvsgError, match="NULL"):
        _lib.call("dvsg_flow_warp_f32", 0, 0, 1, 4, 4, 3, 0, 0)


# ------------------------------------------------- sampler A far outside the frame, at scale
def _sampler_a_noise_bound(xo, yo, H, W):
    """Rounding noise of sampler A's four-term sum (ThinPlateSpline.py:81-89) per unit of intensity.  The
    weights come from the CLIPPED tap indices (:57-60), so a source coordinate d px outside the frame gives
    coincident taps with weights of size ~d that cancel only up to rounding: |noise| <~ ulp(wx wy).  Inside
    the frame the weights are <= 1 and this is ~1e-7."""
    x = (np.asarray(xo, np.float64) + 1.0) * W / 2.0
    y = (np.asarray(yo, np.float64) + 1.0) * H / 2.0
    wx = np.maximum(np.abs(np.clip(np.floor(x) + 1, 0, W - 1) - x), np.abs(x - np.clip(np.floor(x), 0, W - 1)))
    wy = np.maximum(np.abs(np.clip(np.floor(y) + 1, 0, H - 1) - y), np.abs(y - np.clip(np.floor(y), 0, H - 1)))
    return 8.0 * 2.0 ** -23 * np.maximum(wx, 1.0) * np.maximum(wy, 1.0)


@pytest.mark.parametrize("frames", ["smooth", "white_noise"])
def test_tps_large_displacement_out_of_frame(dev, frames):
    """Control vectors of scale 0.5 plus a uniform shift of 1.5 (normalised units: three quarters of the frame):
    a large part of every output frame samples OUTSIDE the image, where sampler A clips the tap indices first and
    takes the weights from the clipped integers (ThinPlateSpline.py:57-60, 81-88) -- the taps coincide and the
    weights cancel to ~0 instead of fading out.  Until now that cancellation was only exercised on the last
    row / column.  Checked with the oracle's T fed through the C ABI (isolates grid + sampler) and end to end
    through the facade against the float64-solved system (the oracle's float32 LU noise grows with |rhs|)."""
    import torch
    from coupe.dvsg_amd import _lib
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline
    B, H, W, C = 2, 72, 128, 3
    if frames == "smooth":
        U = inputs.smooth_frames(71, B, H, W, C)
    else:
        U = np.random.default_rng(72).uniform(0.0, 1.0, (B, H, W, C)).astype(np.float32)
    coord = inputs.v_src(B)
    vec = inputs.control_vectors(73, B, scale=0.5)
    vec[0, :, 0] += 1.5            # sample 0 shifted in x, sample 1 in y (and the other way in x)
    vec[1, :, 1] += 1.5
    vec[1, :, 0] -= 0.75
    T = otps.solve_system(coord, (coord + vec).astype(np.float32), dtype=np.float64).astype(np.float32)
    xo, yo = otps.source_coords(T, coord, H, W)
    ref = otps.interpolate_a(U, xo, yo).reshape(B, H, W, C)
    xp, yp = (xo.astype(np.float64) + 1) * W / 2, (yo.astype(np.float64) + 1) * H / 2
    outside = ((xp < -1) | (xp > W) | (yp < -1) | (yp > H)).reshape(B, H, W)
    assert 0.3 < outside.mean() < 0.95, "the case must put a large part of the frame outside: %.2f" % outside.mean()
    noise = _sampler_a_noise_bound(xo, yo, H, W).reshape(B, H, W)
    # the oracle's own output out there is the cancellation residue, not an image value
    assert (np.abs(ref).max(axis=3)[outside] <= noise[outside]).all()
    tU, tc, tT = (torch.from_numpy(a).to(dev) for a in (U, coord, T))
    out = torch.empty((B, H, W, C), device=dev)
    xs = torch.empty((B * H * W,), device=dev)
    ys = torch.empty((B * H * W,), device=dev)
    _lib.call("dvsg_tps_warp_f32", tU.data_ptr(), tc.data_ptr(), tT.data_ptr(), B, H, W, C, 25, H, W,
              out.data_ptr(), xs.data_ptr(), ys.data_ptr(), 0)
    torch.cuda.synchronize()
    xs, ys, out = xs.cpu().numpy().reshape(B, -1), ys.cpu().numpy().reshape(B, -1), out.cpu().numpy()
    ex = np.abs(xs - xo).reshape(B, H, W) * W / 2
    ey = np.abs(ys - yo).reshape(B, H, W) * H / 2
    print("large displacement (%s): outside %.0f%%, grid error %.3g px, |T| max %.3g" % (frames, 100 * outside.mean(), max(ex.max(), ey.max()), np.abs(T).max()))
    assert max(ex.max(), ey.max()) < 5e-3           # measured 7.5e-4 px at 128 x 72 (|T| up to 4: ~10x the small-vector cases')
    mask = otps.border_discontinuity_mask(xo, yo, H, W, delta=3e-2).reshape(B, H, W)
    err = np.abs(out - ref).max(axis=3)
    assert (np.abs(out).max(axis=3)[outside] <= noise[outside]).all(), "outside the frame the taps must cancel"
    # inside: a warped value moves by at most the coordinate error times the image gradient (<= 1 per px)
    slack = err - 2.0 * (ex + ey) - 2.0 * noise
    assert slack[~mask].max() <= 2e-5, "worst excess %.3g" % slack[~mask].max()
    if frames == "smooth":
        assert err[~mask & ~outside].max() < 3e-3
    # end to end (GPU float64 solve -> T -> grid -> sampler): same frames, same bounds
    o2, x2, y2 = ThinPlateSpline(U, coord, vec, (H, W))
    ex2 = np.abs(x2.reshape(B, H, W) - xo.reshape(B, H, W)) * W / 2
    ey2 = np.abs(y2.reshape(B, H, W) - yo.reshape(B, H, W)) * H / 2
    assert max(ex2.max(), ey2.max()) < 3e-2
    err2 = np.abs(o2 - ref).max(axis=3)
    assert (err2 - 2.0 * (ex2 + ey2) - 2.0 * noise)[~mask].max() <= 2e-5
    assert (np.abs(o2).max(axis=3)[outside & ~mask] <= noise[outside & ~mask]).all()
