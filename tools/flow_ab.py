#!/usr/bin/env python3
"""A/B of the sampler kernels under a debug option (default: warp_xcd 0 / 1) on BASELINE configs[2] and its neighbours:
tf_warp B=64 720p with the cfg-3 flow, a constant flow, per-pixel white noise; the projective STN and the TPS warp.
Prints GB/s of ALGORITHMIC bytes and checks that both settings give the same bits.
    python tools/flow_ab.py [option [values...]]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench  # noqa: E402
from coupe.dvsg_amd import _lib  # noqa: E402
from warp_bench import timeit  # noqa: E402

opt = sys.argv[1] if len(sys.argv) > 1 else "warp_xcd"
values = [int(v) for v in sys.argv[2:]] or [0, 1]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
B, H, W = 64, 720, 1280
g = torch.Generator(device=dev).manual_seed(0)
U, flow3 = bench.make_flow_inputs(B, H, W, 0, dev)
out = torch.empty_like(U)
const = torch.zeros((B, H, W, 2), device=dev)
const[..., 0], const[..., 1] = 3.3, 1.7
noise = 4.0 * torch.randn((B, H, W, 2), generator=g, device=dev)
th = torch.tensor([1, 0, 0.01, 0, 1, 0.02, 0.01, 0.0], device=dev).repeat(B, 1).contiguous()
lin = torch.linspace(-1, 1, 5, device=dev)
coord = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), -1).reshape(1, 25, 2).repeat(B, 1, 1).contiguous()
vec = 0.05 * torch.randn((B, 25, 2), generator=g, device=dev)
T = torch.empty((B, 2, 28), device=dev)
_lib.call("dvsg_tps_solve_f32", coord.data_ptr(), vec.data_ptr(), 1, B, 25, T.data_ptr(), s)
cases = [("tf_warp cfg-3 flow", 32.0, lambda: _lib.call("dvsg_flow_warp_f32", U.data_ptr(), flow3.data_ptr(), B, H, W, 3, out.data_ptr(), s)),
         ("tf_warp constant flow", 32.0, lambda: _lib.call("dvsg_flow_warp_f32", U.data_ptr(), const.data_ptr(), B, H, W, 3, out.data_ptr(), s)),
         ("tf_warp white-noise flow", 32.0, lambda: _lib.call("dvsg_flow_warp_f32", U.data_ptr(), noise.data_ptr(), B, H, W, 3, out.data_ptr(), s)),
         ("projective STN", 24.0, lambda: _lib.call("dvsg_grid_projective_f32", th.data_ptr(), U.data_ptr(), B, H, W, 3, H, W, out.data_ptr(), 0, 0, s)),
         ("TPS warp", 24.0, lambda: _lib.call("dvsg_tps_warp_f32", U.data_ptr(), coord.data_ptr(), T.data_ptr(), B, H, W, 3, 25, H, W, out.data_ptr(), 0, 0, s))]
for name, bpp, fn in cases:
    ref = None
    for rep in range(2):          # twice each, interleaved: same-box A/B
        for v in values:
            _lib.call("dvsg_debug_set_option", opt.encode(), v)
            med, mn = timeit(fn, rounds=30)
            same = ""
            if ref is None:
                ref = out.clone()
            else:
                same = "same bits" if torch.equal(out, ref) else "DIFFERENT (max %g)" % float((out - ref).abs().max())
            print("%-26s %s=%d  %8.1f us (min %8.1f)  %7.1f GB/s = %.3f of 8 TB/s  %s"
                  % (name, opt, v, med, mn, bpp * B * H * W / med / 1e3, bpp * B * H * W / med / 1e3 / 8000, same))
_lib.call("dvsg_debug_set_option", opt.encode(), values[-1])
