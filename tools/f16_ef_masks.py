#!/usr/bin/env python3
"""Calibrated error-feedback plain float16 weights (dvsg_debug_calibrate_f16_weights mode 2): F_t error at 720p against the
float64 CPU arbiter for pair masks between "all pairs" and "none", calibrated on 288x512 or on 720p windows of other
seeds, on test windows of three seeds; step time at 4K B=8.  Run on the GPU box."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import inputs  # noqa: E402
from coupe.dvsg_amd import _lib  # noqa: E402
from coupe.dvsg_amd.networks import LocNet  # noqa: E402
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402
from oracle.cnn_torch import TorchLocNet  # noqa: E402

dev = torch.device("cuda")
torch.set_num_threads(16)
H, W = 720, 1280
w = make_synthetic_weights(0)
net = LocNet(w)
tests = []
oracle = TorchLocNet(w, dtype=torch.float64)
for seed in (7, 77, 777):
    x = inputs.window_frames(seed, 1, H, W)
    tests.append((torch.from_numpy(x).cuda(), oracle.forward(x)))


def bit(kind, block):
    return 1 << (4 * kind + block)


ALL = 0xFFFF
B1 = 0x1111
masks = [("all pairs", ALL), ("none", 0), ("block 1 only", B1),
         ("block 1 + every conv2 (3x3)", B1 | bit(1, 1) | bit(1, 2) | bit(1, 3)),
         ("block 1 + every conv1", B1 | bit(0, 1) | bit(0, 2) | bit(0, 3)),
         ("block 1 + every conv3", B1 | bit(2, 1) | bit(2, 2) | bit(2, 3)),
         ("block 1 + shortcuts", B1 | bit(3, 1) | bit(3, 2) | bit(3, 3)),
         ("blocks 1, 2", B1 | 0x2222), ("blocks 1, 4", B1 | 0x8888), ("blocks 1, 3", B1 | 0x4444),
         ("blocks 1, 2, 3", B1 | 0x2222 | 0x4444), ("blocks 1, 3, 4", B1 | 0x4444 | 0x8888)]
big4k = bench.gpu_windows(8, 2160, 3840, 2, dev)


def timeit(p, n=3):
    B, h, ww, _ = p.shape
    u = p[..., 18:].contiguous()
    o = torch.empty((B, h, ww, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for cname, (cb, ch, cw, cseed) in (("calibrated on 2 x 288x512", (2, 288, 512, 991)), ("calibrated on 2 x 720p", (2, 720, 1280, 992)),
                                    ("calibrated on 1 x 720p", (1, 720, 1280, 993))):
    calib = torch.from_numpy(inputs.window_frames(cseed, cb, ch, cw)).cuda()
    ws, nbytes = net.workspace(cb, ch, cw)
    _lib.call("dvsg_debug_calibrate_f16_weights", net.handle, calib.data_ptr(), cb, ch, cw, 2, ws.data_ptr(), nbytes,
              torch.cuda.current_stream().cuda_stream)
    for mname, m in masks:
        _lib.call("dvsg_debug_set_option", b"f16_pair_mask", m)
        errs = [np.abs(net.forward(xd, precision="f16").cpu().numpy() - ref).max() for xd, ref in tests]
        t = timeit(big4k) if cname.endswith("288x512") else float("nan")
        print("%-26s %-30s mask 0x%04x  F_t err %s  4K B=8 %.2f ms" % (cname, mname, m, " ".join("%.2e" % e for e in errs), t), flush=True)
_lib.call("dvsg_debug_set_option", b"f16_pair_mask", ALL)
