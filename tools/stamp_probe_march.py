#!/usr/bin/env python3
"""Where a step of conv1_f16_march_kernel goes, from a -DDVSG_STAMPS build (s_memtime stamps of one multiplying and one
staging wave per workgroup):  DVSG_AMD_LIB=build/lib_stamps.so python tools/stamp_probe_march.py [B H W]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

B, H, W = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (8, 2160, 3840)
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(seed=0))
x = bench.gpu_windows(B, H, W, 1, dev)
for _ in range(4):
    net.tap(x, 0, precision="f16")       # stops after conv1
torch.cuda.synchronize()
lib = _lib.load()
buf = np.zeros((65536, 8), dtype=np.uint64)
assert lib.dvsg_debug_read_conv1_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
st = buf[buf[:, 3] > 0].astype(np.float64)
steps = st[:, 3]
print("conv1_f16_march_kernel B=%d %dx%d: %d workgroups sampled, %.0f steps each (s_memtime ticks: shader-clock cycles on this part)"
      % (B, W, H, len(st), np.median(steps)))
for i, nm in ((0, "multiplying wave: fragment reads + 80 MFMAs"), (1, "multiplying wave: at the step barrier"),
              (2, "multiplying wave: flush (per step, amortised)"), (4, "staging wave: loads + scale + stores + DMA issue"),
              (5, "staging wave: at the step barrier"), (7, "staging wave: waiting for the rows fetched a step ago")):
    v = st[:, i] / steps
    print("%-52s per step: med %7.2f  p10 %7.2f  p90 %7.2f ticks" % (nm, np.median(v), np.percentile(v, 10), np.percentile(v, 90)))
print("lifetime per step: %.2f ticks" % np.median(st[:, 6] / steps))
