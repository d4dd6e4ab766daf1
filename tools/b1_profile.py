#!/usr/bin/env python3
"""33 back-to-back batch-1 dvsg_stabilize calls at a given size, for rocprofv3 --kernel-trace --stats
(which kernels make up a frame of the latency path).  Usage: tools/b1_profile.py H W [precision]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from coupe.dvsg_amd.model import StabNet
from coupe.dvsg_amd.weights import make_synthetic_weights
H, W = int(sys.argv[1]), int(sys.argv[2])
PREC = sys.argv[3] if len(sys.argv) > 3 else "f32"
net = StabNet(H, W).load_weights(make_synthetic_weights(0))
x = torch.rand((1, H, W, 21), device="cuda"); u = x[..., 18:].contiguous()
o = torch.empty((1, H, W, 3), device="cuda"); F = torch.empty((1, 25, 2), device="cuda")
for _ in range(3): net.locnet.stabilize(x, u, o, F, precision=PREC)
torch.cuda.synchronize()
for _ in range(30): net.locnet.stabilize(x, u, o, F, precision=PREC)
torch.cuda.synchronize()
