"""GPU parity at BASELINE.json's full sizes (1280x720, B=16 / B=64; one 3840x2160 frame), where
the NumPy oracle is too slow to run on every pixel of every sample: size-independent properties
(identity, integer shifts, linearity, batch invariance, run-to-run determinism) plus spot checks
of single samples against the oracle."""
import numpy as np
import pytest

import inputs

pytestmark = pytest.mark.gpu
H, W = 720, 1280


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _frames(dev, B, h=H, w=W, c=3, seed=0):
    import torch
    g = torch.Generator(device=dev).manual_seed(seed)
    return torch.rand((B, h, w, c), generator=g, device=dev)


def test_tf_warp_b64_720p_properties(dev):
    """configs[2]: identity, integer shift, linearity in the image, one sample vs the oracle."""
    import torch
    from coupe.dvsg_amd.warp_with_optical_flow import tf_warp
    from oracle.warp_with_optical_flow import tf_warp as o_warp
    B = 64
    im = _frames(dev, B)
    zero = torch.zeros((B, H, W, 2), device=dev)
    assert torch.equal(tf_warp(im, zero, H, W), im)
    sh = zero.clone()
    sh[..., 0] = 5.0
    sh[..., 1] = -3.0
    out = tf_warp(im, sh, H, W)
    assert torch.equal(out[:, 3:, :W - 5], im[:, :H - 3, 5:])
    assert float(out[:, :3].abs().max()) == 0.0 and float(out[:, :, W - 5:].abs().max()) == 0.0
    flow = torch.from_numpy(inputs.smooth_flow(7, 1, H, W)).to(dev).repeat(B, 1, 1, 1)
    a = tf_warp(im, flow, H, W)
    im2 = _frames(dev, B, seed=1)
    b = tf_warp(im2, flow, H, W)
    ab = tf_warp(0.25 * im + 0.5 * im2, flow, H, W)
    assert float((ab - (0.25 * a + 0.5 * b)).abs().max()) < 1e-6          # linear in the image
    ref = o_warp(im[5:6].cpu().numpy(), flow[5:6].cpu().numpy(), H, W)
    assert np.abs(a[5:6].cpu().numpy() - ref).max() <= 1e-6


def test_tps_b16_720p_properties(dev):
    """configs[1] warp stage: zero / uniform control vectors, per-sample independence, and one
    sample against the oracle at full resolution."""
    import torch
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline
    from oracle import thin_plate_spline as otps
    B = 16
    U = torch.from_numpy(inputs.smooth_frames(3, 2, H, W)).to(dev).repeat(8, 1, 1, 1)
    coord = torch.from_numpy(inputs.v_src(B)).to(dev)
    _, x, y = ThinPlateSpline(U, coord, torch.zeros_like(coord), (H, W))
    xt = torch.linspace(-1, 1, W, device=dev).repeat(H)
    yt = torch.linspace(-1, 1, H, device=dev).repeat_interleave(W)
    assert float((x.reshape(B, -1) - xt).abs().max()) < 5e-6 and float((y.reshape(B, -1) - yt).abs().max()) < 5e-6
    vec = torch.from_numpy(inputs.control_vectors(4, B)).to(dev)
    out, x, y = ThinPlateSpline(U, coord, vec, (H, W))
    out1, x1, y1 = ThinPlateSpline(U[9:10], coord[9:10], vec[9:10], (H, W))
    assert torch.equal(out[9:10], out1) and torch.equal(x.reshape(B, -1)[9], x1)   # batch invariance
    ro, rx, ry = otps.ThinPlateSpline(U[9:10].cpu().numpy(), coord[9:10].cpu().numpy(), vec[9:10].cpu().numpy(), (H, W))
    gerr = max(np.abs(x1.cpu().numpy() - rx).max() * W / 2, np.abs(y1.cpu().numpy() - ry).max() * H / 2)
    assert gerr < 2e-2, "grid error %.3g px" % gerr
    mask = otps.border_discontinuity_mask(rx, ry, H, W, delta=3e-2).reshape(1, H, W)
    err = np.abs(out1.cpu().numpy() - ro).max(axis=3)
    assert err[~mask].max() < 1e-3          # BASELINE.json: warped-frame max abs error < 1e-3
    assert mask.mean() < 0.01


def test_locnet_b16_720p(dev, synthetic_weights):
    """configs[1] CNN stage: bitwise run-to-run determinism, batch invariance up to float32
    re-association (where a tile's K loop is cut by split-K / stream-K depends on the launch's
    tile count and the tile's place in it), and two of the 16 windows against the (torch-CPU)
    oracle at 720p."""
    import torch
    from coupe.dvsg_amd.networks import LocNet
    from oracle.cnn_torch import TorchLocNet
    B = 16
    net = LocNet(synthetic_weights)
    x = torch.from_numpy(inputs.window_frames(5, 2, H, W)).to(dev)
    x = torch.cat([x, x.flip(0)] * 4, 0)                  # 16 windows, two distinct ones
    F = net.forward(x)
    assert torch.equal(F, net.forward(x))                 # deterministic (no atomics anywhere)
    for a, b in ((0, 3), (1, 2), (0, 15)):
        assert float((F[a] - F[b]).abs().max()) <= 1e-6
    F1 = net.forward(x[1:2])
    assert float((F1 - F[1:2]).abs().max()) <= 1e-6       # independent of the batch around it
    ref = TorchLocNet(synthetic_weights).forward(x[:2].cpu().numpy())
    assert np.abs(F[:2].cpu().numpy() - ref).max() <= 1e-5


def test_stabilize_single_4k_frame(dev, synthetic_weights):
    """One 3840x2160 window (configs[4] resolution, float32): F_t against the torch-CPU oracle and
    the identity-warp property; exercises every kernel at sizes 9x the 720p case."""
    import torch
    from coupe.dvsg_amd.model import Session, StabNet
    from oracle.cnn_torch import TorchLocNet
    h, w = 2160, 3840
    x = torch.from_numpy(inputs.window_frames(9, 1, h // 8, w // 8)).to(dev)
    x = torch.nn.functional.interpolate(x.permute(0, 3, 1, 2), size=(h, w), mode="bilinear",
                                        align_corners=False).permute(0, 2, 3, 1).contiguous()
    model = StabNet(h, w).load_weights(synthetic_weights)
    ins, outs = model.get_evaluation_model(7)
    s_t_pred, F = Session().run([outs["s_t_pred"], outs["F_t"]], {ins["patches_t"]: x, ins["u_t"]: x[..., 18:].contiguous()})
    ref = TorchLocNet(synthetic_weights).forward(x.cpu().numpy())
    assert np.abs(F.cpu().numpy() - ref).max() <= 2e-5
    assert s_t_pred.shape == (1, h, w, 3) and bool(torch.isfinite(s_t_pred).all())
    # sampler A's out-of-range taps cancel only up to rounding (weights of coincident taps, x ~ 4e3)
    assert -1e-4 <= float(s_t_pred.min()) and float(s_t_pred.max()) <= 1.0 + 1e-4


def test_stream_split_is_invisible(dev, synthetic_weights):
    """LocNet.stabilize with the batch split over two or three HIP streams returns the same results
    (up to float32 re-association in the convolutions, whose K-split policy follows the tile count
    of each launch; a coordinate that moves by 1e-6 px can still flip a pixel sitting exactly on
    sampler A's border discontinuity)."""
    import torch
    from coupe.dvsg_amd.networks import LocNet
    B, h, w = 6, 96, 160
    net = LocNet(synthetic_weights)
    x = torch.from_numpy(inputs.window_frames(11, B, h, w)).to(dev)
    u = x[..., 18:].contiguous()
    res = []
    for n in (1, 2, 3):
        out = torch.empty((B, h, w, 3), device=dev)
        F = torch.empty((B, 25, 2), device=dev)
        xs = torch.empty((B * h * w,), device=dev)
        ys = torch.empty_like(xs)
        net.stabilize(x, u, out, F, xs, ys, n_streams=n)
        torch.cuda.synchronize()
        res.append((out, F, xs, ys))
    for other in res[1:]:
        for a, b in zip(res[0][1:], other[1:]):
            assert float((a - b).abs().max()) <= 1e-6
        assert float(((res[0][0] - other[0]).abs() > 1e-4).float().mean()) < 1e-3


def test_tps_warp_4k_against_oracle(dev):
    """One 3840x2160 frame through `dvsg_tps_warp_f32` with NON-zero control vectors against the NumPy oracle
    (ThinPlateSpline.py:92-141; the 4K tests so far checked the identity map only).  Given the oracle's own T, so
    that what is measured is the float32 evaluation of the 28-term map and sampler A at W = 3840: two correct
    float32 evaluations differ by a few 1e-6 in normalised coordinates, i.e. by ~1e-2 px here (SURVEY.md section 7,
    hard part 2, predicts ~7e-3), and a warped value by that times the local image gradient.  The frame is
    band-limited noise at 1/16 resolution (|dI/dx| <= 0.1 per pixel)."""
    import torch
    from coupe.dvsg_amd import _lib
    from oracle import thin_plate_spline as otps
    h, w, C = 2160, 3840, 3
    U = inputs.smooth_frames(61, 1, h, w, C, factor=16)
    coord = inputs.v_src(1)
    vec = inputs.control_vectors(62, 1)
    T = otps.solve_system(coord, (coord + vec).astype(np.float32))
    xo, yo = otps.source_coords(T, coord, h, w)
    ref = otps.interpolate_a(U, xo, yo).reshape(1, h, w, C)
    tU, tc, tT = (torch.from_numpy(a).to(dev) for a in (U, coord, T))
    out = torch.empty((1, h, w, C), device=dev)
    xs = torch.empty((h * w,), device=dev)
    ys = torch.empty((h * w,), device=dev)
    _lib.call("dvsg_tps_warp_f32", tU.data_ptr(), tc.data_ptr(), tT.data_ptr(), 1, h, w, C, 25, h, w,
              out.data_ptr(), xs.data_ptr(), ys.data_ptr(), 0)
    torch.cuda.synchronize()
    ex = np.abs(xs.cpu().numpy() - xo[0]) * w / 2
    ey = np.abs(ys.cpu().numpy() - yo[0]) * h / 2
    mask = otps.border_discontinuity_mask(xo, yo, h, w, delta=5e-2).reshape(h, w)
    err = np.abs(out.cpu().numpy()[0] - ref[0]).max(axis=2)
    print("4K TPS warp vs oracle: grid error %.3g px (x) %.3g px (y); pixels max %.3g, median %.3g outside %d border pixels "
          "(%.4f%% of the frame)" % (ex.max(), ey.max(), err[~mask].max(), np.median(err), int(mask.sum()), 100 * mask.mean()))
    # measured (round 3, MI355X): grid 3.1e-3 px (x) / 9.7e-4 px (y), pixels 2.0e-4 max, 5.5e-6 median, 732 border pixels
    assert max(ex.max(), ey.max()) < 1e-2, "grid error %.3g px" % max(ex.max(), ey.max())
    assert err[~mask].max() < 1e-3, "pixel error %.3g" % err[~mask].max()      # BASELINE.json's bound, at 4K
    # a warped value may move by the coordinate error times the gradient and no more
    assert (err - 0.25 * (ex + ey).reshape(h, w))[~mask].max() < 2e-5
    assert mask.mean() < 0.005


def test_tps_warp_4k_against_the_float64_arbiter(dev):
    """Whose error is the 3e-3 px?  `test_tps_warp_4k_against_oracle` measures the distance between two float32
    evaluations of the 28-term map at W = 3840; here both are compared with a float64 evaluation of the map the
    reference DEFINES (oracle/thin_plate_spline.py `source_coords_f64`: the same float32 inputs, every operation in
    float64).  End to end from (coord, vector): the GPU solves the system itself (`dvsg_tps_solve_f32`), the oracle
    inverts in float32 like tf.matrix_inverse.  The GPU's grid must not be further from the arbiter than 1.25 x the
    float32 oracle's own distance, and its pixels -- against sampler A evaluated in float64 on the arbiter's grid --
    stay inside BASELINE.json's 1e-3."""
    import torch
    from coupe.dvsg_amd import _lib
    from oracle import thin_plate_spline as otps
    h, w, C = 2160, 3840, 3
    U = inputs.smooth_frames(61, 1, h, w, C, factor=16)
    coord = inputs.v_src(1)
    worst = {}
    for seed, scale in ((62, 0.05), (63, 0.2)):
        vec = inputs.control_vectors(seed, 1, scale=scale)
        rhs = (coord + vec).astype(np.float32)
        ref32, xo, yo = otps.ThinPlateSpline(U, coord, vec, (h, w))
        x64 = np.empty(h * w)
        y64 = np.empty(h * w)
        ref64 = np.empty((h * w, C))
        for r0 in range(0, h, 240):                                   # a band of rows at a time (25 x rows x W float64 terms)
            rows = np.arange(r0, min(h, r0 + 240))
            xb, yb = otps.source_coords_f64(coord, rhs, h, w, rows=rows)
            x64[r0 * w:(rows[-1] + 1) * w], y64[r0 * w:(rows[-1] + 1) * w] = xb[0], yb[0]
            ref64[r0 * w:(rows[-1] + 1) * w] = otps.interpolate_a_f64(U, xb, yb)[0]
        tU, tc, tv = (torch.from_numpy(a).to(dev) for a in (U, coord, vec))
        T = torch.empty((1, 2, 28), device=dev)
        out = torch.empty((1, h, w, C), device=dev)
        xs = torch.empty((h * w,), device=dev)
        ys = torch.empty((h * w,), device=dev)
        _lib.call("dvsg_tps_solve_f32", tc.data_ptr(), tv.data_ptr(), 1, 1, 25, T.data_ptr(), 0)
        _lib.call("dvsg_tps_warp_f32", tU.data_ptr(), tc.data_ptr(), T.data_ptr(), 1, h, w, C, 25, h, w,
                  out.data_ptr(), xs.data_ptr(), ys.data_ptr(), 0)
        torch.cuda.synchronize()
        xg, yg = xs.cpu().numpy().astype(np.float64), ys.cpu().numpy().astype(np.float64)
        g_px = max(np.abs(xg - x64).max() * w / 2, np.abs(yg - y64).max() * h / 2)
        o_px = max(np.abs(xo - x64).max() * w / 2, np.abs(yo - y64).max() * h / 2)
        mask = otps.border_discontinuity_mask(x64, y64, h, w, delta=5e-2).reshape(h, w)
        g_pix = np.abs(out.cpu().numpy()[0].reshape(-1, C) - ref64).max(axis=1).reshape(h, w)[~mask].max()
        o_pix = np.abs(ref32.reshape(-1, C) - ref64).max(axis=1).reshape(h, w)[~mask].max()
        print("4K TPS vs float64 arbiter (vectors x %.2f): grid GPU %.3g px, float32 oracle %.3g px; pixels GPU %.3g, oracle %.3g "
              "(outside %d border pixels)" % (scale, g_px, o_px, g_pix, o_pix, int(mask.sum())))
        # measured (MI355X, round 4): vectors x 0.05 -- grid GPU 3.7e-3 px, float32 oracle 8.2e-3 px; pixels 2.4e-4 / 5.8e-4;
        # x 0.2 -- 9.3e-3 / 1.5e-2 px; 4.9e-4 / 8.5e-4: the GPU (fused multiply-adds, the system solved in float64) is the
        # closer of the two float32 evaluations
        worst[scale] = (g_px, o_px, g_pix, o_pix)
        assert g_px <= 1.25 * o_px + 1e-4, "GPU grid %.3g px from the arbiter, the float32 oracle %.3g px" % (g_px, o_px)
        assert g_pix < 1e-3, "GPU pixels %.3g from the float64 frame" % g_pix
        assert mask.mean() < 0.005
