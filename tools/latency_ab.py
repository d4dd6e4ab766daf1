#!/usr/bin/env python3
"""Same-process A/B of the batch-1 path under a debug option: back-to-back dvsg_stabilize_f32 calls at 512x288 and 1280x720
(and batch 2), F_t compared between the settings.  python tools/latency_ab.py [option [values...]]   (default conv_variant 7 0;
AB_PRECISION=f32x3: in that precision; the option is left at its LAST value)"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs  # noqa: E402
from coupe.dvsg_amd import _lib  # noqa: E402
from coupe.dvsg_amd.networks import LocNet  # noqa: E402
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else "conv_variant"
values = [int(v) for v in sys.argv[2:]] or [7, 0]
PREC = os.environ.get("AB_PRECISION", "f32")
net = LocNet(make_synthetic_weights(0))
for B, H, W in ((1, 288, 512), (1, 720, 1280), (2, 288, 512), (2, 720, 1280)):
    x = torch.from_numpy(inputs.window_frames(5, B, H, W)).cuda()
    u = x[..., 18:].contiguous()
    o = torch.empty((B, H, W, 3), device="cuda")
    F = torch.empty((B, 25, 2), device="cuda")
    ref = None
    for rep in range(2):
        for v in values:
            _lib.call("dvsg_debug_set_option", opt.encode(), v)
            for _ in range(5):
                net.stabilize(x, u, o, F, precision=PREC)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(100):
                net.stabilize(x, u, o, F, precision=PREC)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / 100
            Fh = F.cpu().numpy().copy()
            if ref is None:
                ref = Fh
            print("B=%d %4dx%-4d %s=%d  %.3f ms per call   F_t vs first setting: %.2e" % (B, W, H, opt, v, 1e3 * dt, np.abs(Fh - ref).max()), flush=True)
