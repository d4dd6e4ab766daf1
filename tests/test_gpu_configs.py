"""GPU parity at the five workloads BASELINE.json names, each at its full size through the C ABI.

configs[0]  one 256x256 window through the evaluation graph, against the committed golden
configs[1]  B=16 1280x720: `dvsg_stabilize_f32` end to end, two windows against the CPU oracle
configs[2]  B=64 tf_warp: tests/test_gpu_fullsize.py::test_tf_warp_b64_720p_properties
configs[3]  one rank's 64-window 720p shard of the 512-window job (the 8-GPU run itself is the
            driver's: bench.py --gpus 8), in batches of 16 through `stabilize_windows_sharded`
configs[4]  B=32 3840x2160 through `dvsg_stabilize_f16`: determinism, batch invariance, the
            identity-warp property of the 4K TPS stage, F_t of two windows against the CPU oracle
Where the NumPy / torch-CPU oracle is too slow for every sample, size-independent properties cover
the whole batch and the oracle spot-checks single windows.
"""
import os
import sys

import numpy as np
import pytest

import inputs
from oracle import thin_plate_spline as otps
from oracle.cnn_torch import TorchLocNet

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
H720, W720 = 720, 1280


@pytest.fixture(scope="module")
def dev():
    import torch
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _gpu_windows(B, H, W, seed, dev):
    """bench.py's device-side synthetic windows (band-limited noise, 7 shifted views)."""
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    import bench
    return bench.gpu_windows(B, H, W, seed, dev)


def _oracle_window(weights, x1, H, W):
    """CPU oracle of one window [1,H,W,21] (torch-CPU CNN + NumPy TPS): F_t, s_t_pred, x_s, y_s."""
    F = TorchLocNet(weights).forward(x1)
    pred, xs, ys = otps.ThinPlateSpline(x1[..., 18:], inputs.v_src(1), F, (H, W))
    return F, pred, xs, ys


def test_cfg0_single_256_window(synthetic_weights):
    """configs[0]: `sess.run` on one 256x256 window, NumPy in / NumPy out as eval.py:106-110 does."""
    from coupe.dvsg_amd.model import Session, StabNet
    with np.load(os.path.join(GOLD, "cfg0_256.npz"), allow_pickle=False) as z:
        g = {k: z[k] for k in z.files}
    H = W = 256
    x = inputs.window_frames(5001, 1, H, W)
    model = StabNet(H, W).load_weights(synthetic_weights)
    ins, outs = model.get_evaluation_model(7)
    pred, F, xs, ys = Session().run([outs["s_t_pred"], outs["F_t"], outs["x_offset_t"], outs["y_offset_t"]],
                                    {ins["patches_t"]: x, ins["u_t"]: x[..., 18:]})
    assert isinstance(pred, np.ndarray) and pred.shape == (1, H, W, 3) and F.shape == (1, 25, 2)
    assert np.abs(F - g["F_t"]).max() <= 1e-5
    gerr = max(np.abs(xs[::16] - g["xs_sub"]).max() * W / 2, np.abs(ys[::16] - g["ys_sub"]).max() * H / 2)
    assert gerr < 2e-2, "grid error %.3g px" % gerr
    mask = np.unpackbits(g["border_mask_bits"])[:H * W].astype(bool).reshape(1, H, W)
    err = np.abs(pred - g["s_t_pred"]).max(axis=3)
    assert err[~mask].max() < 1e-3, "pixel error %.3g" % err[~mask].max()
    assert mask.mean() < 0.02


@pytest.mark.parametrize("wseed,fseed", [(0, 77), (5, 1077)])
def test_cfg1_stabilize_b16_720p_end_to_end(dev, wseed, fseed):
    """configs[1]: the benchmarked call itself -- ONE `dvsg_stabilize_f32` over B=16 720p windows --
    with two of the windows checked end to end against the CPU oracle: F_t <= 1e-5, source grid
    < 2e-2 px, warped pixels < 1e-3 outside the counted sampler-A border-discontinuity pixels.  Two seeds of
    checkpoint and frames (round 4: every 720p end-to-end figure used to ride on seed 0)."""
    import torch
    from coupe.dvsg_amd.networks import LocNet
    from coupe.dvsg_amd.weights import make_synthetic_weights
    synthetic_weights = make_synthetic_weights(seed=wseed)
    B, H, W = 16, H720, W720
    x = _gpu_windows(B, H, W, fseed, dev)
    u = x[..., 18:].contiguous()
    net = LocNet(synthetic_weights)
    out = torch.empty((B, H, W, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    xs = torch.empty((B * H * W,), device=dev)
    ys = torch.empty_like(xs)
    net.stabilize(x, u, out, F, xs, ys)
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all())
    xs, ys = xs.reshape(B, -1), ys.reshape(B, -1)
    for b in (3, 12):
        rF, rpred, rxs, rys = _oracle_window(synthetic_weights, x[b:b + 1].cpu().numpy(), H, W)
        assert np.abs(F[b:b + 1].cpu().numpy() - rF).max() <= 1e-5
        gerr = max(np.abs(xs[b].cpu().numpy() - rxs).max() * W / 2, np.abs(ys[b].cpu().numpy() - rys).max() * H / 2)
        assert gerr < 2e-2, "window %d: grid error %.3g px" % (b, gerr)
        mask = otps.border_discontinuity_mask(rxs, rys, H, W, delta=3e-2).reshape(H, W)
        err = np.abs(out[b].cpu().numpy() - rpred[0]).max(axis=2)
        print("cfg1 seeds (%d, %d) window %d: F_t %.3g, grid %.3g px, pixels %.3g outside %d border pixels"
              % (wseed, fseed, b, np.abs(F[b:b + 1].cpu().numpy() - rF).max(), gerr, err[~mask].max(), int(mask.sum())))
        assert err[~mask].max() < 1e-3, "window %d: pixel error %.3g" % (b, err[~mask].max())
        assert mask.mean() < 0.01


def test_cfg3_one_rank_shard_of_64_windows(dev, synthetic_weights):
    """configs[3] rehearsal on one card: rank 3's 64-window shard of the 512-window job, in batches
    of 16 through `stabilize_windows_sharded` (a one-rank process group, so the sharding and gather
    code runs), spot-checked against the CPU oracle and against a direct call."""
    import torch
    import torch.distributed as dist
    from coupe.dvsg_amd.clip import shard_range, stabilize_windows_sharded
    from coupe.dvsg_amd.model import Session, StabNet
    H, W = H720, W720
    lo, hi = shard_range(512, 8, 3)
    assert (lo, hi) == (192, 256)
    n = hi - lo
    x = torch.cat([_gpu_windows(16, H, W, 1000 + lo + i, dev) for i in range(0, n, 16)], 0)
    u = x[..., 18:].contiguous()
    model = StabNet(H, W).load_weights(synthetic_weights)
    ins, outs = model.get_evaluation_model(7)
    sess = Session()
    calls = []

    def run_fn(p, uu):
        calls.append(int(p.shape[0]))
        return sess.run(outs["s_t_pred"], {ins["patches_t"]: p, ins["u_t"]: uu})

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        res = stabilize_windows_sharded(run_fn, x, u, batch=16)
    finally:
        dist.destroy_process_group()
    assert calls == [16, 16, 16, 16] and res.shape == (n, H, W, 3) and res.is_cuda
    direct = run_fn(x[32:48], u[32:48])
    assert torch.equal(direct, res[32:48])
    for b in (5, 58):
        _, rpred, rxs, rys = _oracle_window(synthetic_weights, x[b:b + 1].cpu().numpy(), H, W)
        mask = otps.border_discontinuity_mask(rxs, rys, H, W, delta=3e-2).reshape(H, W)
        err = np.abs(res[b].cpu().numpy() - rpred[0]).max(axis=2)
        assert err[~mask].max() < 1e-3, "window %d: pixel error %.3g" % (b, err[~mask].max())


@pytest.mark.parametrize("calibrated", [False, True])
def test_cfg4_b32_4k_f16(dev, synthetic_weights, calibrated):
    """(calibrated: after `dvsg_locnet_calibrate_f16` on ONE other 3840x2160 window -- blocks 2-4 on error-feedback-rounded
    plain float16 weights, 22 % less time -- the same bounds hold.)
    configs[4]: B=32 3840x2160 windows in ONE `dvsg_stabilize_f16` call (22 GB of windows and a
    ~60 GB workspace fit the 288 GB of HBM: no sub-batching).  float16 storage cannot be bit-compatible
    with the float32 reference; stated bound against the float32 CPU oracle at this size:
    F_t < 1e-5 (|F_t| ~ 0.1; float16 activations, hi / lo float16 weight pairs -- tests/test_gpu_f16.py).  Exact properties: bitwise run-to-run determinism, batch invariance up to
    accumulation order, and the identity map of the (float32) 4K TPS stage."""
    import torch
    from coupe.dvsg_amd.ThinPlateSpline import ThinPlateSpline
    from coupe.dvsg_amd.networks import LocNet
    B, H, W = 32, 2160, 3840
    net = LocNet(synthetic_weights)
    if calibrated:
        net.calibrate_f16(_gpu_windows(1, H, W, 499, dev))
    x = torch.cat([_gpu_windows(8, H, W, 400 + i, dev) for i in range(4)], 0)
    u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    net.stabilize(x, u, out, F, precision="f16")
    torch.cuda.synchronize()
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(F).all())
    assert -1e-4 <= float(out.min()) and float(out.max()) <= 1.0 + 1e-4
    out2 = torch.empty_like(out)
    F2 = torch.empty_like(F)
    net.stabilize(x, u, out2, F2, precision="f16")
    torch.cuda.synchronize()
    assert torch.equal(F, F2) and torch.equal(out, out2)                # deterministic
    del out2
    o1 = torch.empty((1, H, W, 3), device=dev)
    F1 = torch.empty((1, 25, 2), device=dev)
    net.stabilize(x[21:22], u[21:22], o1, F1, precision="f16")          # window 21 alone
    assert float((F1 - F[21:22]).abs().max()) <= 2e-5
    assert float(((o1 - out[21:22]).abs() > 1e-3).float().mean()) < 1e-3
    ref = TorchLocNet(synthetic_weights)
    for b in (0, 21):
        rF = ref.forward(x[b:b + 1].cpu().numpy())
        err = np.abs(F[b:b + 1].cpu().numpy() - rF).max()
        assert err < 1e-5, "window %d: f16 F_t error %.3g" % (b, err)
    # window 21 in PIXELS, end to end against the float32 oracle (torch-CPU CNN -> NumPy TPS + sampler A at 4K):
    # float16 CNN error (F_t above, x W/2 = 1920 px per unit) + the float32 evaluation noise of the map at W = 3840
    # (tests/test_gpu_fullsize.py::test_tps_warp_4k_against_oracle) times the frames' gradient (<= 0.2 per pixel)
    rpred, rxs, rys = otps.ThinPlateSpline(u[21:22].cpu().numpy(), inputs.v_src(1), rF, (H, W))
    mask = otps.border_discontinuity_mask(rxs, rys, H, W, delta=5e-2).reshape(H, W)
    got21 = out[21].cpu().numpy()
    perr = np.abs(got21 - rpred[0]).max(axis=2)
    # ... and against the float64 ARBITER of the same definition (oracle/thin_plate_spline.py `source_coords_f64` +
    # `interpolate_a_f64`: the oracle's F_t, the map and sampler A evaluated in float64).  At W = 3840 the float32 oracle
    # is itself ~3e-3 px x gradient away from that frame (tests/test_gpu_fullsize.py::..._float64_arbiter), so the
    # distance between two float32 evaluations is not the GPU's error: the bound of record is the one against the arbiter.
    rhs = (inputs.v_src(1) + rF).astype(np.float32)
    u21 = u[21:22].cpu().numpy()
    e64 = np.zeros((H, W))
    o64 = np.zeros((H, W))
    for r0 in range(0, H, 240):
        rows = np.arange(r0, min(H, r0 + 240))
        xb, yb = otps.source_coords_f64(inputs.v_src(1), rhs, H, W, rows=rows)
        f64 = otps.interpolate_a_f64(u21, xb, yb)[0].reshape(len(rows), W, 3)
        e64[rows] = np.abs(got21[rows] - f64).max(axis=2)
        o64[rows] = np.abs(rpred[0][rows] - f64).max(axis=2)
    print("cfg4 (calibrated: %s) window 21: F_t error %.3g; pixels vs float32 oracle max %.3g median %.3g; vs float64 arbiter: GPU %.3g, "
          "float32 oracle %.3g (outside %d border pixels)"
          % (calibrated, err, perr[~mask].max(), np.median(perr), e64[~mask].max(), o64[~mask].max(), int(mask.sum())))
    # measured: F_t 9.6e-7; pixels 9.0e-4 max / 9.0e-5 median against the float32 oracle (round 3: 9.6e-4) -- but 3.9e-4
    # against the float64 arbiter, from which the float32 ORACLE is 9.0e-4 away: the zero margin of round 3 was the
    # oracle's own float32 noise at W = 3840, not the float16 mode's error (476 border pixels)
    assert e64[~mask].max() < 1e-3, "window 21: pixel error %.3g against the float64 arbiter" % e64[~mask].max()
    assert perr[~mask].max() < 1e-3 + o64[~mask].max(), "window 21: pixel error %.3g against the float32 oracle" % perr[~mask].max()
    assert mask.mean() < 0.005
    del out, o1, rpred, perr
    # the warp stage of configs[4] (float32 in both modes) at 4K: zero control vectors -> identity grid
    coord = torch.from_numpy(inputs.v_src(2)).to(dev)
    o, xg, yg = ThinPlateSpline(u[:2], coord, torch.zeros_like(coord), (H, W))
    xt = torch.linspace(-1, 1, W, device=dev).repeat(H)
    yt = torch.linspace(-1, 1, H, device=dev).repeat_interleave(W)
    assert float((xg.reshape(2, -1) - xt).abs().max()) < 5e-6 and float((yg.reshape(2, -1) - yt).abs().max()) < 5e-6
    # identity grid + sampler A = bilinear resampling at (j W / (W - 1), i H / (H - 1)) (ThinPlateSpline.py:48-49
    # scale by W, not W - 1); last row / column excluded (they map to x = W: the clipped taps cancel)
    xsrc = torch.arange(W - 1, device=dev, dtype=torch.float64) * (W / (W - 1.0))
    ysrc = torch.arange(H - 1, device=dev, dtype=torch.float64) * (H / (H - 1.0))
    x0, y0 = xsrc.floor().long(), ysrc.floor().long()
    fx, fy = (xsrc - x0).float()[None, :, None], (ysrc - y0).float()[:, None, None]
    u0 = u[0]
    want = ((1 - fy) * ((1 - fx) * u0[y0][:, x0] + fx * u0[y0][:, x0 + 1])
            + fy * ((1 - fx) * u0[y0 + 1][:, x0] + fx * u0[y0 + 1][:, x0 + 1]))
    assert float((o[0, :H - 1, :W - 1] - want).abs().max()) < 1e-3
