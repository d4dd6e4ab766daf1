#!/usr/bin/env python3
"""Micro-benchmark of the HBM-bound warps at BASELINE sizes: TPS grid+sampler A (configs[1]
shape, B=16) and tf_warp (configs[2], B=64), 1280x720.  Reports GB/s of ALGORITHMIC bytes
(24 B/px and 32 B/px) against the 8 TB/s HBM peak."""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib  # noqa: E402


def timeit(fn, rounds=20):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return float(np.median(ts)), float(np.min(ts))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    args = ap.parse_args()
    H, W = args.height, args.width
    dev = torch.device("cuda:0")
    s = torch.cuda.current_stream().cuda_stream
    g = torch.Generator(device=dev).manual_seed(0)
    for B in (16, 64):
        U = torch.rand((B, H, W, 3), generator=g, device=dev)
        out = torch.empty_like(U)
        lin = torch.linspace(-1, 1, 5, device=dev)
        coord = torch.stack(torch.meshgrid(lin, lin, indexing="xy"), -1).reshape(1, 25, 2).repeat(B, 1, 1).contiguous()
        vec = 0.05 * torch.randn((B, 25, 2), generator=g, device=dev)
        T = torch.empty((B, 2, 28), device=dev)
        _lib.call("dvsg_tps_solve_f32", coord.data_ptr(), vec.data_ptr(), 1, B, 25, T.data_ptr(), s)
        med, mn = timeit(lambda: _lib.call("dvsg_tps_warp_f32", U.data_ptr(), coord.data_ptr(), T.data_ptr(), B, H, W,
                                           3, 25, H, W, out.data_ptr(), 0, 0, s))
        nbytes = 24.0 * B * H * W
        print("tps_warp  B=%2d  %8.1f us (min %8.1f)  %7.1f GB/s  frac of 8 TB/s %.3f" % (B, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000))
        flow = 4.0 * torch.randn((B, H, W, 2), generator=g, device=dev)
        med, mn = timeit(lambda: _lib.call("dvsg_flow_warp_f32", U.data_ptr(), flow.data_ptr(), B, H, W, 3, out.data_ptr(), s))
        nbytes = 32.0 * B * H * W
        print("tf_warp   B=%2d  %8.1f us (min %8.1f)  %7.1f GB/s  frac of 8 TB/s %.3f" % (B, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000))
        xs = torch.rand((B * H * W,), generator=g, device=dev) * 2 - 1
        ys = torch.rand((B * H * W,), generator=g, device=dev) * 2 - 1
        del flow
        th = torch.tensor([1, 0, 0.01, 0, 1, 0.02, 0.01, 0.0], device=dev).repeat(B, 1).contiguous()
        med, mn = timeit(lambda: _lib.call("dvsg_grid_projective_f32", th.data_ptr(), U.data_ptr(), B, H, W, 3, H, W,
                                           out.data_ptr(), 0, 0, s))
        nbytes = 24.0 * B * H * W
        print("projective B=%2d %8.1f us (min %8.1f)  %7.1f GB/s  frac of 8 TB/s %.3f" % (B, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000))
        del xs, ys, U, out


if __name__ == "__main__":
    main()
