#!/usr/bin/env python3
"""Time dvsg_conv_gemm_f32 on hand-picked shapes (B,H,W,Cin,Cout,k): is a launch whose tiles fill whole
rounds of the 512 resident workgroups as fast as the bare main loop (tools/loop_bench.hip)?"""
import sys, os
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
ZERO = os.environ.get("PROBE_ZERO") == "1"     # all-zero operands: the same cycles at whatever clock the chip then holds
shapes = [tuple(int(v) for v in a.split(",")) for a in sys.argv[1:]] or [
    (16, 64, 64, 256, 256, 3), (16, 64, 64, 1024, 256, 1), (16, 64, 64, 256, 1024, 1), (16, 64, 64, 512, 512, 3),
    (16, 128, 64, 128, 128, 3), (32, 64, 64, 256, 256, 3), (16, 45, 80, 256, 256, 3)]
for B, H, W, cin, cout, k in shapes:
    x = torch.rand((B, H, W, cin), generator=g, device=dev) - 0.3
    K = k * k * cin
    wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    y = torch.empty((B, H, W, cout), device=dev)
    if ZERO:
        x.zero_(); wt.zero_()
    ts = []
    for rnd in range(8):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), 0, y.data_ptr(), B, H, W, cin, cout, k,
                  1, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
        e1.record(); e1.synchronize()
        if rnd: ts.append(e0.elapsed_time(e1) * 1e3)
    us = float(np.median(ts))
    M = B * H * W
    tiles = -(-M // 128) * (cout // 128)
    print("B=%d %dx%d %d->%d k%d  M=%d tiles=%d (%.2f rounds)  %.1f us  %.1f TFLOP/s" %
          (B, H, W, cin, cout, k, M, tiles, tiles / 512, us, 2.0 * M * cout * K / us / 1e6))
