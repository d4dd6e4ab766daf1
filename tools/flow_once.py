#!/usr/bin/env python3
"""A few launches of tf_warp at BASELINE configs[2] (B=64 720p) for rocprofv3 PMC passes: `python tools/flow_once.py [cfg3|const|noise]`."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from coupe.dvsg_amd import _lib  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
dev = torch.device("cuda:0")
B, H, W = 64, 720, 1280
U, flow = bench.make_flow_inputs(B, H, W, 0, dev)
if kind == "const":
    flow = torch.zeros_like(flow)
    flow[..., 0], flow[..., 1] = 3.3, 1.7
elif kind == "noise":
    flow = 4.0 * torch.randn((B, H, W, 2), device=dev)
out = torch.empty_like(U)
for _ in range(4):
    _lib.call("dvsg_flow_warp_f32", U.data_ptr(), flow.data_ptr(), B, H, W, 3, out.data_ptr(), torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
