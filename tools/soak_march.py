#!/usr/bin/env python3
"""Soak of conv1_f16_march_kernel's hand-rolled synchronisation (role-split waves, s_waitcnt vmcnt(N) + s_barrier, ring
slot reuse): its conv1 output must equal conv1_f16_pair_kernel's bit for bit -- same products, same order -- on every one
of many launches over fresh inputs, window and ring sources, natural (big) and forced (small, ragged) shapes.
Usage: tools/soak_march.py [rounds]"""
import os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 40
net = LocNet(make_synthetic_weights(0))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
shapes = [(16, 720, 1280, 0), (2, 2160, 3840, 0), (3, 200, 333, 5), (1, 97, 640, 5), (2, 64, 96, 5), (5, 36, 1000, 5)]
bad = 0
t0 = time.time()
for rnd in range(rounds):
    for B, H, W, force in shapes:
        for source in ("window", "ring_u8", "ring_f32"):
            if source == "window":
                x = torch.rand((B, H, W, 21), generator=g, device=dev)
                run = lambda: net.tap(x, 0, precision="f16")
            else:
                n = 7 * B
                pool = (torch.randint(0, 256, (n, H, W, 3), generator=g, device=dev, dtype=torch.uint8) if source == "ring_u8"
                        else torch.rand((n, H, W, 3), generator=g, device=dev))
                table = torch.randint(0, n, (B, 7), generator=g, device=dev, dtype=torch.int32)
                run = lambda: net.forward_ring(pool, table, precision="f16", stage=0)
            _lib.call("dvsg_debug_set_option", b"conv1_variant", 4)
            ref = run().clone()
            _lib.call("dvsg_debug_set_option", b"conv1_variant", force)
            for rep in range(3):
                got = run()
                if not torch.equal(got, ref):
                    bad += 1
                    print("MISMATCH round %d shape %s %s rep %d: max diff %g, %d values" %
                          (rnd, (B, H, W), source, rep, float((got - ref).abs().max()), int((got != ref).sum())), flush=True)
            _lib.call("dvsg_debug_set_option", b"conv1_variant", 0)
    if rnd % 10 == 9:
        print("round %d done, %.0f s, mismatches so far %d" % (rnd + 1, time.time() - t0, bad), flush=True)
print("soak done: %d rounds x %d shapes x 3 sources x 3 launches, %d mismatches" % (rounds, len(shapes), bad))
sys.exit(1 if bad else 0)
