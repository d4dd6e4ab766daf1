#!/usr/bin/env python3
"""Where does conv1_f16_march_kernel on a ring source differ from the gathered window?  Prints mismatch coordinates."""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
net = LocNet(make_synthetic_weights(0))
dev = torch.device("cuda")
g = torch.Generator(device=dev).manual_seed(1)
for (B, H, W) in ((16, 720, 1280), (2, 2160, 3840), (4, 720, 1280)):
    for source in ("ring_f32", "ring_u8"):
        for tabkind in ("random", "sliding"):
            n = 7 * B
            pool = (torch.randint(0, 256, (n, H, W, 3), generator=g, device=dev, dtype=torch.uint8) if source == "ring_u8"
                    else torch.rand((n, H, W, 3), generator=g, device=dev))
            if tabkind == "random":
                table = torch.randint(0, n, (B, 7), generator=g, device=dev, dtype=torch.int32)
            else:
                table = (torch.arange(B, device=dev)[:, None] + torch.arange(7, device=dev)[None, :]).to(torch.int32)
            fr = pool.float() / 255.0 if source == "ring_u8" else pool
            x = fr[table.long()].permute(0, 2, 3, 1, 4).reshape(B, H, W, 21).contiguous()
            res = {}
            for v in (4, 0, 5):
                _lib.call("dvsg_debug_set_option", b"conv1_variant", v)
                res["win%d" % v] = net.tap(x, 0, precision="f16").clone()
                res["ring%d" % v] = net.forward_ring(pool, table, precision="f16", stage=0).clone()
            _lib.call("dvsg_debug_set_option", b"conv1_variant", 0)
            ref = res["win4"]
            print("B=%d %dx%d %s %s table" % (B, W, H, source, tabkind), flush=True)
            for k, v in res.items():
                d = (v != ref)
                nd = int(d.sum())
                if nd:
                    idx = d.any(-1).nonzero()
                    print("   %s: %d values differ, %d pixels; first %s ... last %s; rows %s cols(min,max) %d %d batches %s" % (
                        k, nd, idx.shape[0], idx[:6].tolist(), idx[-3:].tolist(), sorted(set(idx[:, 1].tolist()))[:20],
                        int(idx[:, 2].min()), int(idx[:, 2].max()), sorted(set(idx[:, 0].tolist()))), flush=True)
                else:
                    print("   %s: identical" % k, flush=True)
            del pool, x, res, fr
