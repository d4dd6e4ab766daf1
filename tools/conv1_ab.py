#!/usr/bin/env python3
"""A/B the conv1 kernel variants inside the full network (class-0 hipEvent timing)."""
import ctypes, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

dev = torch.device("cuda:0")
B, H, W = 16, 720, 1280
net = LocNet(make_synthetic_weights(0))
x = bench.gpu_windows(B, H, W, 1234, dev)
ref = None
for rnd in range(3):
    for v in (0, 1):
        _lib.call("dvsg_debug_set_option", b"conv1_variant", v)
        net.forward(x)
        torch.cuda.synchronize()
        _lib.call("dvsg_prof_begin", 0)
        for _ in range(3):
            F = net.forward(x)
        ms, n, fl, by = ctypes.c_double(), ctypes.c_int(), ctypes.c_double(), ctypes.c_double()
        _lib.call("dvsg_prof_end", ctypes.byref(ms), ctypes.byref(n), ctypes.byref(fl), ctypes.byref(by))
        if ref is None:
            ref = F.clone()
        print("round %d conv1_variant=%d  %.3f ms/launch  %.1f TFLOP/s  maxdiff vs first %.2e" %
              (rnd, v, ms.value / n.value, fl.value / ms.value / 1e9, float((F - ref).abs().max())))
