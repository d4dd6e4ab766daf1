#!/usr/bin/env python3
"""Same-process A/B of a dvsg_debug_set_option value on the float16 step: tools/ab_option.py <option> <v0> <v1> [B H W] ...
Alternates the two values, 5 rounds of 6 steps each, prints the median ms/step of each and the F_t difference."""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
opt, v0, v1 = sys.argv[1].encode(), int(sys.argv[2]), int(sys.argv[3])
shapes = [tuple(int(t) for t in a.split("x")) for a in sys.argv[4:]] or [(16, 720, 1280), (16, 2160, 3840)]
prec = os.environ.get("AB_PRECISION", "f16")
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(0))
for B, H, W in shapes:
    x = torch.cat([bench.gpu_windows(min(8, B), H, W, 50 + i, dev) for i in range((B + 7) // 8)], 0)[:B].contiguous()
    u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev)
    Fs = {}
    times = {v0: [], v1: []}
    for rnd in range(5):
        for v in (v0, v1):
            _lib.call("dvsg_debug_set_option", opt, v)
            F = torch.empty((B, 25, 2), device=dev)
            net.stabilize(x, u, out, F, precision=prec)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                net.stabilize(x, u, out, F, precision=prec)
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 6)
            Fs[v] = F
    print("B=%d %dx%d %s  %s=%d: %.3f ms/step   %s=%d: %.3f ms/step   (%+.1f %%)   F_t diff %.3g" % (
        B, W, H, prec, opt.decode(), v0, statistics.median(times[v0]), opt.decode(), v1, statistics.median(times[v1]),
        100 * (statistics.median(times[v1]) / statistics.median(times[v0]) - 1), float((Fs[v0] - Fs[v1]).abs().max())), flush=True)
    del x, u, out
    torch.cuda.empty_cache()
