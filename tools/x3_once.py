#!/usr/bin/env python3
"""A few launches of MFMA-bound conv layers in f32 / f32x3 for PMC passes (tools/pmc_x3.sh):  x3_once.py [f32|f32x3]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "f32x3"
dev = torch.device("cuda:0")
s = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
g = torch.Generator(device=dev).manual_seed(0)
for B, h, w, cin, cout, k in [(16, 45, 80, 256, 256, 3), (16, 90, 160, 128, 128, 3), (16, 45, 80, 1024, 256, 1)]:
    K = k * k * cin
    x = torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3
    wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    y = torch.empty((B, h, w, cout), device=dev)
    if prec == "f32x3":
        packed = torch.empty((cout * K * 6,), dtype=torch.uint8, device=dev)
        _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), cout, K, s)
        wt = packed
    for _ in range(12):
        _lib.call("dvsg_conv_gemm_" + prec, x.data_ptr(), wt.data_ptr(), bias.data_ptr(), 0, y.data_ptr(), B, h, w, cin, cout, k, 1, 1, 1,
                  scratch.data_ptr(), scratch.numel(), s)
    torch.cuda.synchronize()
