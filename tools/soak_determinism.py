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
PRECS = tuple(sys.argv[1].split(",")) if len(sys.argv) > 1 else ("f32",)   # e.g. f32,f16,f32s
SCALE = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0                    # fraction of the run counts
for (B, H, W, n) in ((16, 720, 1280, 300), (1, 720, 1280, 600), (1, 288, 512, 1000), (3, 200, 320, 600)):
    x = torch.from_numpy(inputs.window_frames(5, min(B, 2), H, W)).to(dev)
    if B > x.shape[0]:
        x = torch.cat([x] * (B // x.shape[0] + 1), 0)[:B].contiguous()
    u = x[..., 18:].contiguous()
    # the same windows as a frame ring: frame f of window b is pool frame 7 b + f; the ring step must give the same bits
    pool = x.reshape(B, H, W, 7, 3).permute(0, 3, 1, 2, 4).reshape(7 * B, H, W, 3).contiguous()
    pool8 = (pool * 255.0).round().clamp(0, 255).to(torch.uint8)
    table = torch.arange(7 * B, device=dev, dtype=torch.int32).reshape(B, 7).contiguous()
    n = max(4, int(n * SCALE))
    for prec in PRECS:
        ref_out = torch.empty((B, H, W, 3), device=dev); ref_F = torch.empty((B, 25, 2), device=dev)
        net.stabilize(x, u, ref_out, ref_F, precision=prec)
        ref8_out = torch.empty_like(ref_out); ref8_F = torch.empty_like(ref_F)
        x8 = (pool8.double() / 255.0).float().reshape(B, 7, H, W, 3).permute(0, 2, 3, 1, 4).reshape(B, H, W, 21).contiguous()
        net.stabilize(x8, x8[..., 18:].contiguous(), ref8_out, ref8_F, precision=prec)   # eval.py:80 then the window path
        del x8
        torch.cuda.synchronize()
        out = torch.empty_like(ref_out); F = torch.empty_like(ref_F)
        bad = bad_ring = bad_ring8 = 0
        for i in range(n):
            net.stabilize(x, u, out, F, n_streams=1 + (i % 2) if B >= 4 else 1, precision=prec)
            if i % 2 == 0 or B < 4:   # single-stream runs must be bitwise identical to the first one
                if not (torch.equal(F, ref_F) and torch.equal(out, ref_out)):
                    bad += 1
            if i % 4 == 0:            # the ring step on the same frames: the gathered window's bits
                net.stabilize_ring(pool, table, out, F, precision=prec)
                if not (torch.equal(F, ref_F) and torch.equal(out, ref_out)):
                    bad_ring += 1
                net.stabilize_ring(pool8, table, out, F, precision=prec)
                if not (torch.equal(F, ref8_F) and torch.equal(out, ref8_out)):
                    bad_ring8 += 1
        torch.cuda.synchronize()
        print("B=%d %dx%d %s: %d runs, %d differ; float ring %d, uint8 ring %d of %d differ" %
              (B, W, H, prec, n, bad, bad_ring, bad_ring8, (n + 3) // 4), flush=True)
        assert bad == 0 and bad_ring == 0 and bad_ring8 == 0
print("soak ok")
