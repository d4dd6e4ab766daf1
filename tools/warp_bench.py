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
        # SURVEY.md 8d cfg 3: flow ~ N(0, 4 px) smoothed with a 15-px box, 1 % of the pixels pushed out of bounds
        flow = 4.0 * 15.0 * torch.randn((B, 2, H, W), generator=g, device=dev)
        flow = torch.nn.functional.avg_pool2d(flow, 15, stride=1, padding=7).permute(0, 2, 3, 1).contiguous()
        oob = torch.rand((B, H, W, 1), generator=g, device=dev) < 0.01
        flow = torch.where(oob, flow + (max(H, W) + 5.0), flow).contiguous()
        med, mn = timeit(lambda: _lib.call("dvsg_flow_warp_f32", U.data_ptr(), flow.data_ptr(), B, H, W, 3, out.data_ptr(), s))
        print("tf_warp   B=%2d  %8.1f us (min %8.1f)  %7.1f GB/s  frac of 8 TB/s %.3f   (cfg-3 smooth flow, sigma %.1f px)"
              % (B, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000, float(flow[~oob.expand_as(flow)].std())))
        xs = torch.rand((B * H * W,), generator=g, device=dev) * 2 - 1
        ys = torch.rand((B * H * W,), generator=g, device=dev) * 2 - 1
        del flow
        th = torch.tensor([1, 0, 0.01, 0, 1, 0.02, 0.01, 0.0], device=dev).repeat(B, 1).contiguous()
        med, mn = timeit(lambda: _lib.call("dvsg_grid_projective_f32", th.data_ptr(), U.data_ptr(), B, H, W, 3, H, W,
                                           out.data_ptr(), 0, 0, s))
        nbytes = 24.0 * B * H * W
        print("projective B=%2d %8.1f us (min %8.1f)  %7.1f GB/s  frac of 8 TB/s %.3f" % (B, med, mn, nbytes / med / 1e3, nbytes / med / 1e3 / 8000))
        del xs, ys, out
        # frame formats either side of the path (frames.hip): bytes moved once each
        N = B
        u8 = (U * 255).to(torch.uint8)
        f32 = torch.empty_like(U)
        med, mn = timeit(lambda: _lib.call("dvsg_frames_u8_to_f32", u8.data_ptr(), N * H * W, 1, f32.data_ptr(), s))
        print("u8_to_f32 N=%2d  %8.1f us  %7.1f GB/s (15 B/px)" % (N, med, 15.0 * N * H * W / med / 1e3))
        med, mn = timeit(lambda: _lib.call("dvsg_frames_f32_to_u8", U.data_ptr(), N, H, W, 1, u8.data_ptr(), W, 0, s))
        print("f32_to_u8 N=%2d  %8.1f us  %7.1f GB/s (15 B/px)" % (N, med, 15.0 * N * H * W / med / 1e3))
        nb = max(1, N // 8)
        idx = torch.randint(0, N, (nb, 7), generator=g, device=dev, dtype=torch.int32)
        pat = torch.empty((nb, H, W, 21), device=dev)
        med, mn = timeit(lambda: _lib.call("dvsg_window_gather_f32", U.data_ptr(), N, H, W, idx.data_ptr(), nb, 7, pat.data_ptr(), s))
        print("gather    B=%2d  %8.1f us  %7.1f GB/s (168 B/px)" % (nb, med, 168.0 * nb * H * W / med / 1e3))
        big = torch.randint(0, 256, (max(1, N // 4), 1080, 1920, 3), generator=g, device=dev, dtype=torch.uint8)
        small = torch.empty((big.shape[0], H, W, 3), device=dev)
        med, mn = timeit(lambda: _lib.call("dvsg_frames_resize_u8_f32", big.data_ptr(), big.shape[0], 1080, 1920, 1, small.data_ptr(),
                                           H, W, 0, 0, 0, s))
        print("resize 1080p->%dx%d n=%2d  %8.1f us  %7.1f GB/s (src 3 B/px + dst 12 B/px)"
              % (W, H, big.shape[0], med, big.shape[0] * (1080 * 1920 * 3.0 + 12.0 * H * W) / med / 1e3))
        del U, u8, f32, pat, big, small


if __name__ == "__main__":
    main()
