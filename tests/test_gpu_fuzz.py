"""GPU: randomly drawn shapes through the whole path, against the CPU oracle and against itself.

The fixed-shape tests pin the cases someone thought of; this one draws (B, H, W) from a seeded generator -- biased towards
the edges of the kernels' tilings (128-pixel conv1 column tiles, 4-row quads, the 16-byte row grid, one-pixel frames) --
and checks, per shape and precision: F_t against the torch-CPU oracle, the source grid and the warped frame against the
NumPy oracle (models/thin_plate_spline.py restated), the frame-ring entry points (float32 and uint8 pools) against
the gathered window BIT for bit, eval_train.py's MASKED graph (the mask plane fused into conv1's staging: window, float ring
and uint8 ring agree bit for bit, F_t against the oracle on `patches * mask`), and tf_warp's strip kernel against its gather
kernel bit for bit.  DVSG_FUZZ_N raises the number of shapes (default 24, ~1 min); DVSG_FUZZ_SEED moves
the draw.  A failure prints the shape: add it to the fixed lists of test_gpu_cnn.py / test_gpu_ring.py with the fix."""
import os

import numpy as np
import pytest

import inputs
from oracle import thin_plate_spline as otps
from oracle.cnn_torch import TorchLocNet

pytestmark = pytest.mark.gpu

N = int(os.environ.get("DVSG_FUZZ_N", "24"))
SEED = int(os.environ.get("DVSG_FUZZ_SEED", "20261004"))
F_TOL = {"f32": 1e-5, "f32s": 1e-5, "f32x3": 1e-5, "f16": 5e-5}


def _shapes(n, seed):
    rng = np.random.default_rng(seed)
    edge_w = [1, 2, 3, 4, 5, 7, 8, 253, 254, 255, 256, 257, 258, 259, 509, 510, 511, 512, 513, 514, 515, 640, 766, 767, 769]
    edge_h = [1, 2, 3, 4, 5, 6, 7, 8, 9, 15, 16, 17, 31, 33, 63, 64, 65]
    out = []
    for _ in range(n):
        W = int(rng.choice(edge_w)) if rng.random() < 0.45 else int(rng.integers(1, 700))
        H = int(rng.choice(edge_h)) if rng.random() < 0.35 else int(rng.integers(1, 300))
        B = int(rng.integers(1, 7))
        if B * H * W > 600_000:          # keep the CPU oracle in seconds
            B = max(1, 600_000 // (H * W))
        out.append((B, H, W))
    return out


@pytest.fixture(scope="module")
def net(synthetic_weights):
    import torch
    assert torch.cuda.is_available()
    from coupe.dvsg_amd.networks import LocNet
    return LocNet(synthetic_weights)


@pytest.mark.parametrize("B,H,W", _shapes(N, SEED))
def test_random_shape_against_the_oracle_and_the_ring(net, synthetic_weights, B, H, W):
    import torch
    x_np = inputs.window_frames(31 * H + W, B, H, W)
    x = torch.from_numpy(x_np).cuda()
    u = x[..., 18:].contiguous()
    F_ref = TorchLocNet(synthetic_weights).forward(x_np)
    pred_ref, xs_ref, ys_ref = otps.ThinPlateSpline(x_np[..., 18:], inputs.v_src(B), F_ref, (H, W))
    # the window as a frame ring: frame f of window b = pool frame 7 b + f (+ one unused frame in front)
    pool = torch.cat([torch.zeros((1, H, W, 3), device="cuda"),
                      x.reshape(B, H, W, 7, 3).permute(0, 3, 1, 2, 4).reshape(7 * B, H, W, 3)]).contiguous()
    table = (1 + torch.arange(7 * B, device="cuda", dtype=torch.int32)).reshape(B, 7).contiguous()
    pool8 = torch.randint(0, 256, (7 * B + 1, H, W, 3), device="cuda", dtype=torch.uint8,
                          generator=torch.Generator(device="cuda").manual_seed(H * 1000 + W))
    x8 = (pool8.double() / 255.0).float()[table.long()].permute(0, 2, 3, 1, 4).reshape(B, H, W, 21).contiguous()
    for prec in ("f32", "f32s", "f32x3", "f16"):
        out = torch.empty((B, H, W, 3), device="cuda")
        F = torch.empty((B, 25, 2), device="cuda")
        xs = torch.empty((B, H * W), device="cuda")
        ys = torch.empty((B, H * W), device="cuda")
        net.stabilize(x, u, out, F, xs, ys, precision=prec)
        assert torch.isfinite(out).all() and torch.isfinite(F).all()
        ferr = np.abs(F.cpu().numpy() - F_ref).max()
        assert ferr <= F_TOL[prec], "%s F_t error %.3g at B=%d H=%d W=%d" % (prec, ferr, B, H, W)
        if prec == "f32":
            # the grid follows F_t: its own oracle value at the oracle's F_t, in pixels (W / 2, H / 2 per grid unit)
            gx = np.abs(xs.cpu().numpy().reshape(-1) - np.asarray(xs_ref).reshape(-1)).max() * W / 2
            gy = np.abs(ys.cpu().numpy().reshape(-1) - np.asarray(ys_ref).reshape(-1)).max() * H / 2
            assert max(gx, gy) < 2e-2, "grid error %.3g px at B=%d H=%d W=%d" % (max(gx, gy), B, H, W)
            # warped pixels, away from sampler A's jumps at the frame border (hard part 3)
            mask = otps.border_discontinuity_mask(xs_ref, ys_ref, H, W, delta=2e-2).reshape(B, H, W)
            err = np.abs(out.cpu().numpy() - pred_ref).max(axis=3)
            if (~mask).any():
                assert err[~mask].max() < 1e-3, "pixel error %.3g at B=%d H=%d W=%d" % (err[~mask].max(), B, H, W)
        # ring entry points: index and byte work -- the gathered window's bits
        out_r = torch.empty_like(out)
        F_r = torch.empty_like(F)
        net.stabilize_ring(pool, table, out_r, F_r, precision=prec)
        assert torch.equal(F_r, F) and torch.equal(out_r, out), "%s float ring differs at B=%d H=%d W=%d" % (prec, B, H, W)
        net.stabilize(x8, x8[..., 18:].contiguous(), out, F, precision=prec)
        net.stabilize_ring(pool8, table, out_r, F_r, precision=prec)
        assert torch.equal(F_r, F) and torch.equal(out_r, out), "%s uint8 ring differs at B=%d H=%d W=%d" % (prec, B, H, W)
        # eval_train.py's masked graph: the three sources agree bit for bit; F_t against the oracle on patches * mask
        from coupe.dvsg_amd.networks import random_mask_plane
        plane = random_mask_plane(inputs.mask_homographies(17 * H + W, B) * np.float32(1.0), H, W)
        net.stabilize(x8, x8[..., 18:].contiguous(), out, F, precision=prec, mask=plane)
        net.stabilize_ring(pool8, table, out_r, F_r, precision=prec, mask=plane)
        assert torch.equal(F_r, F) and torch.equal(out_r, out), "%s masked uint8 ring differs at B=%d H=%d W=%d" % (prec, B, H, W)
        net.stabilize(x, u, out, F, precision=prec, mask=plane)
        net.stabilize_ring(pool, table, out_r, F_r, precision=prec, mask=plane)
        assert torch.equal(F_r, F) and torch.equal(out_r, out), "%s masked float ring differs at B=%d H=%d W=%d" % (prec, B, H, W)
        if prec == "f32":
            xm = x_np.copy()
            xm[..., :18] *= plane.cpu().numpy()[..., None]
            ferr = np.abs(F.cpu().numpy() - TorchLocNet(synthetic_weights).forward(xm)).max()
            assert ferr <= 1e-5, "masked F_t error %.3g at B=%d H=%d W=%d" % (ferr, B, H, W)
    # tf_warp: the strip kernel (LDS-staged source rows) gives the gather kernel's bits on a flow that leaves the window in places
    from coupe.dvsg_amd import _lib
    rng = np.random.default_rng(H * 7 + W)
    flow = torch.from_numpy((8.0 * rng.standard_normal((B, H, W, 2))).astype(np.float32)).cuda()
    got = {}
    try:
        for v in (0, 1):
            _lib.call("dvsg_debug_set_option", b"flow_tiled", v)
            o = torch.full((B, H, W, 3), float("nan"), device="cuda")
            _lib.call("dvsg_flow_warp_f32", u.data_ptr(), flow.data_ptr(), B, H, W, 3, o.data_ptr(), torch.cuda.current_stream().cuda_stream)
            got[v] = o
    finally:
        _lib.call("dvsg_debug_set_option", b"flow_tiled", 1)
    assert torch.equal(got[0], got[1]), "tf_warp strip kernel differs at B=%d H=%d W=%d" % (B, H, W)
