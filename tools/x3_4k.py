#!/usr/bin/env python3
"""f32x3 at 3840x2160 (tensors of > 2^32 elements at B = 20): F_t and the warped frames against the exact path, step times."""
import sys, torch
sys.path.insert(0, "/root/repo")
import bench
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(0))
for B, H, W in [(2, 2160, 3840), (20, 2160, 3840)]:
    x = torch.cat([bench.gpu_windows(min(4, B), H, W, 60 + i, dev) for i in range((B + 3) // 4)], 0)[:B].contiguous()
    u = x[..., 18:].contiguous()
    outs = {}
    for prec in ("f32", "f32x3"):
        out = torch.empty((B, H, W, 3), device=dev); F = torch.empty((B, 25, 2), device=dev)
        net.stabilize(x, u, out, F, precision=prec); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); net.stabilize(x, u, out, F, precision=prec); e1.record(); torch.cuda.synchronize()
        outs[prec] = (F.clone(), out[:, ::7, ::7].clone(), e0.elapsed_time(e1))
    d = (outs["f32"][1] - outs["f32x3"][1]).abs().amax(dim=3)
    inner = d[:, 1:-1, 1:-1]
    print("B=%d %dx%d: |F_t x3 - f32| %.3g; sampled pixels: max diff %.3g, %d of %d above 1e-3 (sampler A's border jumps), interior max %.3g; ms f32 %.1f x3 %.1f"
          % (B, W, H, float((outs["f32"][0] - outs["f32x3"][0]).abs().max()), float(d.max()), int((d > 1e-3).sum()), d.numel(),
             float(inner.max()), outs["f32"][2], outs["f32x3"][2]), flush=True)
    del x, u, out; net._ws = None; torch.cuda.empty_cache()
