#!/usr/bin/env python3
"""float16 mode without the lo weight pieces: does error-feedback rounding of the plain float16 weights
(dvsg_debug_calibrate_f16_weights: mode 0 round-to-nearest, 1 running error sum bounded, 2 the same weighted by calibrated
channel means) bring F_t back to the pairs' level?  F_t error of two 720p windows against the float32 CPU oracle for a
few pair masks and both checkpoints (synthetic, stress), and the step times.  Run on the GPU box."""
import ctypes
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
from oracle.stress_weights import make_stress_weights  # noqa: E402

dev = torch.device("cuda")
torch.set_num_threads(16)
H, W = 720, 1280
x = inputs.window_frames(7, 2, H, W)
xd = torch.from_numpy(x).cuda()
calib = torch.from_numpy(inputs.window_frames(991, 2, 288, 512)).cuda()      # other frames, another size
ALL = 0xFFFF
B1 = 0x1111


def timeit(net, p, n=5):
    B, h, ww, _ = p.shape
    u = p[..., 18:].contiguous()
    o = torch.empty((B, h, ww, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    for _ in range(2):
        net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


for wname, w in (("synthetic", make_synthetic_weights(0)), ("stress", make_stress_weights(0))):
    net = LocNet(w)
    ref = TorchLocNet(w, dtype=torch.float64).forward(x)
    scale = np.abs(ref).max()
    ws, nbytes = net.workspace(2, 288, 512)
    for mode in (0, 1, 2):
        _lib.call("dvsg_debug_calibrate_f16_weights", net.handle, calib.data_ptr(), 2, 288, 512, mode, ws.data_ptr(), nbytes,
                  torch.cuda.current_stream().cuda_stream)
        for mname, m in (("all pairs", ALL), ("pairs in block 1 only", B1), ("no pairs", 0)):
            _lib.call("dvsg_debug_set_option", b"f16_pair_mask", m)
            F = net.forward(xd, precision="f16").cpu().numpy()
            err = np.abs(F - ref).max()
            print("%-9s rounding mode %d  %-22s F_t err %.2e (rel %.2e; %.3f px at 720p)" % (wname, mode, mname, err, err / scale, err * W / 2), flush=True)
    _lib.call("dvsg_debug_set_option", b"f16_pair_mask", ALL)
    if wname == "synthetic":
        big = bench.gpu_windows(16, H, W, 1, dev)
        big4k = bench.gpu_windows(8, 2160, 3840, 2, dev)
        for mname, m in (("all pairs", ALL), ("pairs in block 1 only", B1)):
            _lib.call("dvsg_debug_set_option", b"f16_pair_mask", m)
            print("timing %-22s 720p B=16 %.2f ms   4K B=8 %.2f ms" % (mname, timeit(net, big), timeit(net, big4k, 3)), flush=True)
        _lib.call("dvsg_debug_set_option", b"f16_pair_mask", ALL)
        del big, big4k
    del net
