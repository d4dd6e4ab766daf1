#!/usr/bin/env python3
"""Steady-state latency of the eval.py clip loop (batch 1, autoregressive) at 720p."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
from coupe.dvsg_amd.clip import stabilize_clip
from coupe.dvsg_amd.model import Session, StabNet
from coupe.dvsg_amd.weights import make_synthetic_weights
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (720, 1280)
net = StabNet(H, W).load_weights(make_synthetic_weights(0)); net.get_evaluation_model(7)
sess = Session()
frames = torch.from_numpy(inputs.smooth_frames(1, 4, H, W)).cuda().repeat(16, 1, 1, 1)   # 64 frames on the device
stabilize_clip(net, sess, frames[:4]); torch.cuda.synchronize()
for prec in ("f32", "f32x3", "f32s", "f16"):
    net.precision = prec
    t0 = time.perf_counter(); out = stabilize_clip(net, sess, frames); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("%s clip loop: %d frames, %.2f ms/frame, %.1f frames/s" % (prec, out.shape[0], 1e3 * dt / out.shape[0], out.shape[0] / dt))
x = torch.rand((1, H, W, 21), device="cuda"); u = x[..., 18:].contiguous()
o = torch.empty((1, H, W, 3), device="cuda"); F = torch.empty((1, 25, 2), device="cuda")
for prec in ("f32", "f32x3", "f32s", "f16"):
    for _ in range(3): net.locnet.stabilize(x, u, o, F, precision=prec)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): net.locnet.stabilize(x, u, o, F, precision=prec)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 50
    print("%s dvsg_stabilize B=1 back to back: %.2f ms" % (prec, 1e3 * dt))
