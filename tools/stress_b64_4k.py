#!/usr/bin/env python3
"""Stress: 64 windows of 3840x2160 in ONE dvsg_stabilize_f16 call -- twice configs[4].  Block 1's activation tensors
are 64 x 540 x 960 x 256 = 8.5e9 elements (> 2^32: every element offset has to be 64-bit), the windows 11e9, the
workspace ~120 GB.  The second half of the batch must give what a 32-window call on those windows gives (the tile
decomposition of the big launches does not depend on the batch: same products, same order), and a uint8 frame ring
(448 frames would be 11 GB; a sliding table over 70 frames here) must run at the same size.
Usage: tools/stress_b64_4k.py        (needs ~200 GB of free HBM)"""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(0))
B, H, W = 64, 2160, 3840
x = torch.cat([bench.gpu_windows(8, H, W, 900 + i, dev) for i in range(B // 8)], 0)
u = x[..., 18:].contiguous()
out = torch.empty((B, H, W, 3), device=dev)
F = torch.empty((B, 25, 2), device=dev)
t0 = time.time()
net.stabilize(x, u, out, F, precision="f16")
torch.cuda.synchronize()
print("B=64 4K f16: %.1f ms (first call), peak HBM %.1f GB" % (1e3 * (time.time() - t0), torch.cuda.max_memory_allocated() / 1e9), flush=True)
assert bool(torch.isfinite(F).all()) and bool(torch.isfinite(out).all())
o2 = torch.empty((32, H, W, 3), device=dev)
F2 = torch.empty((32, 25, 2), device=dev)
net.stabilize(x[32:], u[32:], o2, F2, precision="f16")
torch.cuda.synchronize()
dF = float((F2 - F[32:]).abs().max())
dO = float((o2 - out[32:]).abs().max())
print("windows 32..63 against a 32-window call on them: F_t max diff %.3g, pixels max diff %.3g" % (dF, dO), flush=True)
assert dF <= 2e-5 and dO <= 2e-3
del x, o2
torch.cuda.empty_cache()
# uint8 frame ring at the same batch: 70 frames, window b = frames b .. b + 6
pool = torch.empty((B + 6, H, W, 3), device=dev, dtype=torch.uint8)   # band-limited frames: the batch's newest frames as bytes
pool[:B] = (u * 255.0).round().clamp(0, 255).to(torch.uint8)
pool[B:] = pool[:6]
del u
table = (torch.arange(B, device=dev)[:, None] + torch.arange(7, device=dev)[None, :]).to(torch.int32).contiguous()
net.stabilize_ring(pool, table, out, F, precision="f16")
torch.cuda.synchronize()
assert bool(torch.isfinite(F).all()) and bool(torch.isfinite(out).all())
# window 63 of the ring against the same window through the float window path
fr = torch.empty((7, H, W, 3), device=dev)
fr.copy_((pool[63:70].double() / 255.0).float())
x1 = fr.permute(1, 2, 0, 3).reshape(1, H, W, 21).contiguous()
o1 = torch.empty((1, H, W, 3), device=dev)
F1 = torch.empty((1, 25, 2), device=dev)
net.stabilize(x1, x1[..., 18:].contiguous(), o1, F1, precision="f16")
torch.cuda.synchronize()
dF = float((F1 - F[63:]).abs().max())
bad = float(((o1 - out[63:]).abs() > 1e-3).float().mean())
print("uint8 ring, window 63 against the window alone: F_t max diff %.3g, pixels off by > 1e-3: %.3g of the frame" % (dF, bad), flush=True)
assert dF <= 2e-5 and bad < 1e-3
print("stress ok")
