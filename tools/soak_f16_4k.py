#!/usr/bin/env python3
"""Run-to-run determinism of the float16 mode at 3840x2160 (the sizes where a wait that is one stage short shows: tens of
thousands of tiles per launch): `runs` calls of dvsg_stabilize_f16 on 4 windows, and of the ring entry point, against the
first result, bit for bit.  Usage: tools/soak_f16_4k.py [runs]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(0))
B, H, W = 4, 2160, 3840
x = bench.gpu_windows(B, H, W, 7, dev)
u = x[..., 18:].contiguous()
pool = x.reshape(B, H, W, 7, 3).permute(0, 3, 1, 2, 4).reshape(7 * B, H, W, 3).contiguous()
table = torch.arange(7 * B, device=dev, dtype=torch.int32).reshape(B, 7).contiguous()
ref_out = torch.empty((B, H, W, 3), device=dev); ref_F = torch.empty((B, 25, 2), device=dev)
net.stabilize(x, u, ref_out, ref_F, precision="f16")
out = torch.empty_like(ref_out); F = torch.empty_like(ref_F)
bad = bad_ring = 0
for i in range(runs):
    net.stabilize(x, u, out, F, precision="f16")
    bad += 0 if (torch.equal(F, ref_F) and torch.equal(out, ref_out)) else 1
    net.stabilize_ring(pool, table, out, F, precision="f16")
    bad_ring += 0 if (torch.equal(F, ref_F) and torch.equal(out, ref_out)) else 1
print("float16, %d windows of %dx%d: %d runs, %d differ from the first; ring %d differ" % (B, W, H, runs, bad, bad_ring))
sys.exit(1 if bad or bad_ring else 0)
