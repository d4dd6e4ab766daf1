#!/usr/bin/env python3
"""The CPU baseline of SURVEY.md 8d in full (bench.py's default run only takes the bounded B=1 sample):
the torch-CPU CNN + torch-CPU TPS warp oracle, both on every usable core ("port": the literal TF-CPU reference cannot run,
BASELINE.md; the one-thread NumPy warp of rounds 1-3 is timed beside it) at
1280x720 for B=1 and B=16, median of >= 5 runs after 2 warm-ups, the CNN and the warp timed apart, CPU
model and thread count recorded.  Writes one JSON object to stdout.
    python tools/cpu_baseline_full.py > profiles/rNN_cpu_baseline.json        (about 3 minutes on 16 cores)"""
import json
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
from coupe.dvsg_amd.weights import make_synthetic_weights  # noqa: E402
from oracle.cnn_torch import TorchLocNet  # noqa: E402
from oracle.thin_plate_spline import ThinPlateSpline as o_tps_numpy  # noqa: E402
from oracle.tps_torch import ThinPlateSpline as o_tps  # noqa: E402

H, W = 720, 1280
cores = bench.usable_cores()
torch.set_num_threads(cores)
net = TorchLocNet(make_synthetic_weights(seed=0))
out = {"cpu_model": bench.cpu_model(), "threads": cores, "height": H, "width": W, "kind": "port",
       "what": "torch-CPU float32 CNN (F.conv2d, TF padding, BN folded) + torch-CPU TPS solve / grid / sampler A, all threads; "
               "numpy_warp_s = the one-thread NumPy restatement of the warp (rounds 1-3's warp leg)"}
for B in (1, 16):
    x = inputs.window_frames(1234, B, H, W)
    vsrc = inputs.v_src(B)
    cnn, warp = [], []
    runs = 7 if B == 1 else 5
    for r in range(runs + 2):
        t0 = time.perf_counter()
        Ft = net.forward(x)
        t1 = time.perf_counter()
        o_tps(x[..., 18:], vsrc, Ft, (H, W))
        t2 = time.perf_counter()
        if r >= 2:
            cnn.append(t1 - t0)
            warp.append(t2 - t1)
    c, w = float(np.median(cnn)), float(np.median(warp))
    t0 = time.perf_counter()
    o_tps_numpy(x[:1, ..., 18:], vsrc[:1], Ft[:1], (H, W))
    out["B%d" % B] = {"runs": runs, "cnn_s": c, "warp_s": w, "frames_per_s": B / (c + w), "cnn_frames_per_s": B / c,
                      "warp_frames_per_s": B / w, "numpy_warp_s_per_frame": time.perf_counter() - t0}
print(json.dumps(out, indent=1))
