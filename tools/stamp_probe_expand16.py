#!/usr/bin/env python3
"""Where a workgroup of conv_expand16_kernel spends its time, from a -DDVSG_STAMPS build (wave 0 of every workgroup):
DVSG_AMD_LIB=build/lib_stamps.so python tools/stamp_probe_expand16.py [cin,cout,B,h,w,resmode ...]"""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib
lib = _lib.load()
layers = [tuple(int(t) for t in a.split(",")) for a in sys.argv[1:]] or [(128, 512, 32, 270, 480, 1), (256, 1024, 32, 135, 240, 1), (256, 512, 32, 270, 480, 0)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
st = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)
for cin, cout, B, h, w, rmode in layers:
    x = (torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3).half()
    ws = ((torch.rand((2 * cout, cin), generator=g, device=dev) - 0.5) * (2.0 / cin ** 0.5)).half()
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    res = (torch.rand((B, h, w, cout), generator=g, device=dev) - 0.5).half() if rmode else None
    y = torch.empty((B, h, w, cout), device=dev, dtype=torch.float16)
    for _ in range(3):
        _lib.call("dvsg_conv_gemm_f16s", x.data_ptr(), ws.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else 0,
                  y.data_ptr(), B, h, w, cin, cout, 1, 1, 1, 1, scratch.data_ptr(), scratch.numel(), st)
    torch.cuda.synchronize()
    buf = np.zeros((65536, 8), dtype=np.uint64)
    assert lib.dvsg_debug_read_expand16_stamps(ctypes.c_void_p(buf.ctypes.data), ctypes.c_size_t(buf.nbytes)) == 0
    f = buf[buf[:, 5] > 0].astype(np.float64)
    groups, stages = cout // 64, (cout // 64) * (cin // 32)
    life = np.median(f[:, 5])
    print("%d -> %d, M = %d, %d workgroups sampled: lifetime %.0f ticks = %d groups x %.0f" % (cin, cout, B * h * w, len(f), life, groups, life / groups))
    for i, nm in ((0, "group top (stores acknowledged / tile landed)"), (1, "K loop: own weight stage not landed"), (2, "K loop: at the stage barrier"),
                  (3, "K loop: DMA issue + reads + MFMA issue"), (4, "epilogue")):
        v = f[:, i]
        print("    %-48s med %8.0f  per group %7.0f  %5.1f %%" % (nm, np.median(v), np.median(v) / groups, 100 * np.median(v) / life))
    sys.stdout.flush()
