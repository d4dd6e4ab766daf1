#!/usr/bin/env python3
"""Soak test of run-to-run determinism: thousands of dvsg_stabilize calls at batch 16 / 3 / 1 (stream-K
tails, split-K slices, the fused block-1 kernel, alternating 1 and 2 HIP streams) must reproduce the
first result bit for bit -- the last-arriver ticket reductions sum in K order whoever arrives last."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import inputs
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
net = LocNet(make_synthetic_weights(0))
dev = torch.device("cuda:0")
for (B, H, W, n) in ((16, 720, 1280, 300), (1, 720, 1280, 600), (1, 288, 512, 1000), (3, 200, 320, 600)):
    x = torch.from_numpy(inputs.window_frames(5, min(B, 2), H, W)).to(dev)
    if B > x.shape[0]:
        x = torch.cat([x] * (B // x.shape[0] + 1), 0)[:B].contiguous()
    u = x[..., 18:].contiguous()
    ref_out = torch.empty((B, H, W, 3), device=dev); ref_F = torch.empty((B, 25, 2), device=dev)
    net.stabilize(x, u, ref_out, ref_F)
    torch.cuda.synchronize()
    out = torch.empty_like(ref_out); F = torch.empty_like(ref_F)
    bad = 0
    for i in range(n):
        for prec in ("f32",):
            net.stabilize(x, u, out, F, n_streams=1 + (i % 2) if B >= 4 else 1, precision=prec)
        if i % 2 == 0 or B < 4:   # single-stream runs must be bitwise identical to the first one
            if not (torch.equal(F, ref_F) and torch.equal(out, ref_out)):
                bad += 1
    torch.cuda.synchronize()
    print("B=%d %dx%d: %d runs, %d differ" % (B, W, H, n, bad), flush=True)
    assert bad == 0
print("soak ok")
