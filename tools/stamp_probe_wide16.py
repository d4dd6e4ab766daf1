#!/usr/bin/env python3
"""Where a tile of conv_wide16_kernel goes, per layer shape, from a -DDVSG_STAMPS build (s_memtime stamps of wave 0 of
every workgroup):  DVSG_AMD_LIB=build/lib_stamps.so python tools/stamp_probe_wide16.py [B H W]"""
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
lib = _lib.load()
LAYERS = [(1, 128, 512, "block 2 conv3 (expand)"), (1, 512, 128, "block 2 conv1 (reduce)"), (3, 128, 128, "block 2 conv2 3x3"),
          (1, 256, 512, "block 2 shortcut"),
          (1, 256, 1024, "block 3 conv3 (expand)"), (1, 1024, 256, "block 3 conv1 (reduce)"), (3, 256, 256, "block 3 conv2 3x3"),
          (1, 512, 2048, "block 4 conv3 (expand)"), (1, 2048, 512, "block 4 conv1 (reduce)"), (3, 512, 512, "block 4 conv2 3x3")]
for ks, cin, cout, name in LAYERS:
    lib.dvsg_debug_wide16_stamp_select(ks * 100000000 + cin * 10000 + cout)
    for _ in range(2):
        net.forward(x, precision="f16")
    torch.cuda.synchronize()
    buf = np.zeros((65536, 8), dtype=np.uint64)
    assert lib.dvsg_debug_read_wide16_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
    st = buf[buf[:, 3] > 0]
    life = (st[:, 7] >> np.uint64(24)).astype(np.float64)
    drain = (st[:, 7] & np.uint64(0xffffff)).astype(np.float64)
    f = st.astype(np.float64)
    stages = ks * ks * cin / 32
    med = lambda v: np.median(v)
    print("%-26s K %5d (%3d stages)  lifetime %7.0f ticks = 100 %%" % (name, ks * ks * cin, stages, med(life)))
    rows = (("first stage (setup + round trip)", f[:, 0]), ("K loop", f[:, 3]), ("   of it: own DMA not landed", f[:, 1]),
            ("   of it: at the stage barrier", f[:, 2]), ("epilogue: barrier before transpose", f[:, 4]),
            ("epilogue: transpose", f[:, 5]), ("epilogue: residual / earlier stores", f[:, 6]),
            ("drain of the last stores", drain))
    for nm, v in rows:
        print("    %-38s med %7.0f  p10 %7.0f  p90 %7.0f   %5.1f %%" % (nm, med(v), np.percentile(v, 10), np.percentile(v, 90), 100 * med(v) / med(life)))
    sys.stdout.flush()
