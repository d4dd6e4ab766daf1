#!/usr/bin/env python3
"""float16 mode: which layers need the lo weight piece?  For a list of pair masks (bit 4 * kind + block; kind 0 = conv1 of a
unit, 1 = conv2 (3x3), 2 = conv3, 3 = shortcut; dvsg_debug_set_option("f16_pair_mask")) prints the F_t error of two 720p
windows against the float32 CPU oracle and the step time at B=16 720p and B=8 4K.  Run on the GPU box."""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import inputs
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
from oracle.cnn_torch import TorchLocNet
import bench

w = make_synthetic_weights(0)
net = LocNet(w)
H, W = 720, 1280
x = inputs.window_frames(7, 2, H, W)
torch.set_num_threads(16)
ref = TorchLocNet(w).forward(x)
xd = torch.from_numpy(x).cuda()
dev = torch.device("cuda")
big = bench.gpu_windows(16, H, W, 1, dev)
big4k = bench.gpu_windows(8, 2160, 3840, 2, dev)


def timeit(p, n=5):
    B, h, ww, _ = p.shape
    u = p[..., 18:].contiguous()
    o = torch.empty((B, h, ww, 3), device=dev); F = torch.empty((B, 25, 2), device=dev)
    for _ in range(2): net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): net.stabilize(p, u, o, F, precision="f16")
    torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


def bit(kind, block): return 1 << (4 * kind + block)
ALL = 0xFFFF
B1 = bit(0, 0) | bit(1, 0) | bit(2, 0) | bit(3, 0)          # block 1 keeps its pairs (fused kernel)
masks = [("all pairs", ALL), ("no pairs (block 1 fused kept)", B1)]
for kind, kname in enumerate(("conv1", "conv2", "conv3", "shortcut")):
    for block in range(1, 4):
        masks.append(("drop %s of block %d" % (kname, block + 1), ALL & ~bit(kind, block)))
masks += [("drop block 1 conv1", ALL & ~bit(0, 0)),
          ("drop all conv2 (3x3) of blocks 2-4", ALL & ~(bit(1, 1) | bit(1, 2) | bit(1, 3))),
          ("drop all 1x1 of blocks 2-4", ALL & ~sum(bit(k, b) for k in (0, 2, 3) for b in (1, 2, 3))),
          ("pairs only in blocks 3-4", ALL & ~sum(bit(k, 1) for k in range(4)) | B1),
          ("pairs only in block 4", (ALL & ~sum(bit(k, b) for k in range(4) for b in (1, 2))) | B1)]
if len(sys.argv) > 1:
    masks = [("mask %s" % a, int(a, 0)) for a in sys.argv[1:]]
for name, m in masks:
    _lib.call("dvsg_debug_set_option", b"f16_pair_mask", m)
    F = net.forward(xd, precision="f16").cpu().numpy()
    err = np.abs(F - ref).max()
    print("%-40s mask 0x%04x  F_t err %.2e (%.2f px at 720p)  720p B=16 %.2f ms  4K B=8 %.2f ms"
          % (name, m, err, err * W / 2, timeit(big), timeit(big4k, 3)), flush=True)
_lib.call("dvsg_debug_set_option", b"f16_pair_mask", ALL)
