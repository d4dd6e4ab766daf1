#!/usr/bin/env python3
"""Per-workgroup phase times of conv1_kernel from a -DDVSG_STAMPS build (see tools/README.md):
   DVSG_AMD_LIB=build/lib_stamps.so python tools/stamp_probe_conv1.py [B H W]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 720, 1280)
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(seed=0))
x = bench.gpu_windows(B, H, W, 1, dev)
for _ in range(3):
    net.tap(x, 0)       # stops after conv1
torch.cuda.synchronize()
lib = _lib.load()
n = min(65536, B * ((H + 1) // 2) * -(-((W + 1) // 2) // 128))
buf = np.zeros((65536, 8), dtype=np.uint64)
assert lib.dvsg_debug_read_conv1_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
st = buf[:n].astype(np.float64)
life, real = st[:, 6], st[:, 7] / 100.0
clk = np.median(life / real)
q = lambda a: "med %8.0f  p10 %8.0f  p90 %8.0f" % (np.median(a), np.percentile(a, 10), np.percentile(a, 90))
print("conv1_kernel B=%d %dx%d: %d workgroups sampled, in-kernel clock %.0f MHz" % (B, W, H, n, clk))
names = ["prologue + first load issue", "barrier 1 (7x: mates' MFMAs + own loads landed)", "scale + LDS stores (7x)", "barrier 2 (7x)",
         "load issue (6x)", "start -> end of MFMA loops", "lifetime"]
for i, nm in enumerate(names):
    print("%-48s %s" % (nm, q(st[:, i])))
mf = 7 * 150 * 64.0
print("MFMA issue cycles per wave: %.0f alone, %.0f with the partner workgroup's wave on the same SIMD" % (mf, 2 * mf))
print("MFMA share of the lifetime if the pipe were never idle: %.3f" % (2 * mf / np.median(life)))
