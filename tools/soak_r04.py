#!/usr/bin/env python3
"""Round-4 soak: the new paths must reproduce their first result bit for bit over many launches with fresh inputs each time --
(1) tf_warp: strip kernel against the gather kernel on random flows of random scale (incl. NaN-free extremes) and shapes;
(2) the masked graph (window / float ring / uint8 ring) at 720p; (3) the calibrated float16 mode: two calibrations give the
same weights, repeated steps the same frames.  python tools/soak_r04.py [rounds]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402
from coupe.dvsg_amd import _lib  # noqa: E402
from coupe.dvsg_amd.networks import LocNet, random_mask_plane  # noqa: E402
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(20261005)
rng = np.random.default_rng(20261005)
bad = 0
for it in range(rounds):
    B = int(rng.integers(1, 9))
    H = int(rng.choice([1, 7, 8, 9, 15, 16, 17, 33, 180, 360, 719, 720, 1080]))
    W = int(rng.choice([1, 3, 127, 128, 129, 255, 256, 257, 640, 1279, 1280, 1920]))
    if B * H * W > 8_000_000:
        B = max(1, 8_000_000 // (H * W))
    im = torch.rand((B, H, W, 3), generator=g, device=dev)
    scale = float(rng.choice([0.0, 0.5, 4.0, 11.9, 12.1, 40.0, 2000.0]))
    flow = scale * torch.randn((B, H, W, 2), generator=g, device=dev)
    outs = []
    for v in (0, 1, 1):
        _lib.call("dvsg_debug_set_option", b"flow_tiled", v)
        o = torch.full((B, H, W, 3), float("nan"), device=dev)
        _lib.call("dvsg_flow_warp_f32", im.data_ptr(), flow.data_ptr(), B, H, W, 3, o.data_ptr(), st)
        outs.append(o)
    if not (torch.equal(outs[0], outs[1]) and torch.equal(outs[1], outs[2]) and bool(torch.isfinite(outs[1]).all())):
        bad += 1
        print("tf_warp MISMATCH at B=%d H=%d W=%d scale=%g" % (B, H, W, scale), flush=True)
_lib.call("dvsg_debug_set_option", b"flow_tiled", 1)
print("tf_warp strip vs gather kernel: %d rounds, %d mismatches" % (rounds, bad), flush=True)

net = LocNet(make_synthetic_weights(0))
H, W, B = 720, 1280, 3
bad_m = 0
for it in range(max(4, rounds // 4)):
    pool8 = torch.randint(0, 256, (9, H, W, 3), device=dev, dtype=torch.uint8, generator=g)
    table = torch.randint(0, 9, (B, 7), device=dev, dtype=torch.int32, generator=g)
    poolf = torch.empty(pool8.shape, dtype=torch.float32, device=dev)
    _lib.call("dvsg_frames_u8_to_f32", pool8.data_ptr(), 9 * H * W, 0, poolf.data_ptr(), st)
    x = poolf[table.long()].permute(0, 2, 3, 1, 4).reshape(B, H, W, 21).contiguous()
    u = torch.stack([poolf[int(table[b, 6])] for b in range(B)]).contiguous()
    Hm = inputs.mask_homographies(int(rng.integers(1 << 30)), B)
    plane = random_mask_plane(Hm, H, W)
    for prec in ("f32", "f16", "f32s"):
        res = []
        for src in ("window", "ringf", "ring8", "window"):
            o = torch.empty((B, H, W, 3), device=dev)
            F = torch.empty((B, 25, 2), device=dev)
            if src == "window":
                net.stabilize(x, u, o, F, precision=prec, mask=plane)
            else:
                net.stabilize_ring(poolf if src == "ringf" else pool8, table, o, F, precision=prec, mask=plane)
            res.append((o, F))
        if not all(torch.equal(a[0], res[0][0]) and torch.equal(a[1], res[0][1]) for a in res[1:]):
            bad_m += 1
            print("masked MISMATCH round %d %s" % (it, prec), flush=True)
print("masked graph, three sources x three precisions at 720p: %d rounds, %d mismatches" % (max(4, rounds // 4), bad_m), flush=True)

calib = torch.from_numpy(inputs.window_frames(991, 2, 288, 512)).to(dev)
x = torch.from_numpy(inputs.window_frames(7, 2, H, W)).to(dev)
u = x[..., 18:].contiguous()
net.calibrate_f16(calib)
ref_o = torch.empty((2, H, W, 3), device=dev)
ref_F = torch.empty((2, 25, 2), device=dev)
net.stabilize(x, u, ref_o, ref_F, precision="f16")
bad_c = 0
for it in range(max(4, rounds // 2)):
    if it % 4 == 0:
        net.calibrate_f16(None)
        net.calibrate_f16(calib)
    o = torch.empty_like(ref_o)
    F = torch.empty_like(ref_F)
    net.stabilize(x, u, o, F, precision="f16")
    if not (torch.equal(o, ref_o) and torch.equal(F, ref_F)):
        bad_c += 1
print("calibrated float16 mode (re-calibrated every 4th step): %d steps, %d mismatches" % (max(4, rounds // 2), bad_c), flush=True)
sys.exit(1 if bad or bad_m or bad_c else 0)
