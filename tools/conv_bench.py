#!/usr/bin/env python3
"""Layer-level micro-benchmark of dvsg_conv_gemm_f32 over the resnet_v1_50 shapes at a given
batch / resolution, optionally A/B-ing kernel variants in ONE process (interleaved rounds).

    python tools/conv_bench.py [--batch 16] [--variants 0,1] [--rounds 5] [--layers all|big]
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib  # noqa: E402
from coupe.dvsg_amd.weights import conv_specs  # noqa: E402


def layer_shapes(B, H, W):
    h1, w1 = (H - 1) // 2 + 1, (W - 1) // 2 + 1
    hh, ww = (h1 + 1) // 2, (w1 + 1) // 2
    out = []
    for s, k, cin, cout, stride, relu in conv_specs()[1:]:
        name = s.split("resnet_v1_50/")[1].replace("/bottleneck_v1", "")
        kind = name.rsplit("/", 1)[1]
        if kind == "conv2":
            out.append((name, hh, ww, cin, cout, 3, stride, True, False))
            hh, ww = (hh - 1) // stride + 1, (ww - 1) // stride + 1
        else:
            out.append((name, hh, ww, cin, cout, 1, 1, relu or kind == "conv3", kind == "conv3"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--variants", default="1")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--layers", default="uniq")
    ap.add_argument("--precision", default="f32", choices=["f32", "f32s", "f32x3", "f16"])
    ap.add_argument("--rep", type=int, default=20,
                    help="back-to-back launches per timing (1 = isolated launches: the clock governor has not ramped "
                         "and short kernels read 10-15 %% low, DESIGN.md 5.0)")
    args = ap.parse_args()
    variants = [int(v) for v in args.variants.split(",")]
    dev = torch.device("cuda:0")
    B = args.batch
    shapes = layer_shapes(B, args.height, args.width)
    if args.layers == "uniq":
        seen, uniq = set(), []
        for s in shapes:
            key = s[1:]
            if key not in seen:
                seen.add(key)
                uniq.append(s)
        shapes = uniq
    stream = torch.cuda.current_stream().cuda_stream
    scratch = torch.empty(66 << 20, dtype=torch.uint8, device=dev)   # split-K / stream-K tickets + slabs (variants 3 / 6 = off)
    g = torch.Generator(device=dev).manual_seed(0)
    tot = {v: 0.0 for v in variants}
    print("%-24s %8s %5s %5s %2s %2s | " % ("layer", "M", "N", "K", "k", "s") +
          " | ".join("v%d: us    TF/s  TB/s" % v for v in variants))
    for name, h, w, cin, cout, k, stride, relu, has_res in shapes:
        ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
        dt = torch.float16 if args.precision == "f16" else torch.float32
        x = (torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3).to(dt)
        K = k * k * cin
        wt = ((torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)).to(dt)
        if args.precision == "f32s":   # float16 pieces [Cout][K/32][32 hi | 32 lo]; activations: any 4-byte buffers time alike
            hi = wt.half()
            lo = (wt - hi.float()).half()
            wt = torch.cat([hi.reshape(cout, K // 32, 32), lo.reshape(cout, K // 32, 32)], 2).contiguous()
        if args.precision == "f32x3":  # bfloat16 piece stages (dvsg_pack_weights_f32x3)
            packed = torch.empty((cout * K * 6,), dtype=torch.uint8, device=dev)
            _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), cout, K, stream)
            wt = packed
        bias = torch.rand((cout,), generator=g, device=dev) - 0.5
        res = (torch.rand((B, ho, wo, cout), generator=g, device=dev) - 0.5).to(dt) if has_res else None
        y = torch.empty((B, ho, wo, cout), device=dev, dtype=dt)
        M = B * ho * wo
        flops = 2.0 * M * cout * K
        nbytes = x.element_size() * (x.numel() + cout * K + y.numel() * (2 if has_res else 1))
        times = {v: [] for v in variants}

        def run():
            _lib.call("dvsg_conv_gemm_" + args.precision, x.data_ptr(), wt.data_ptr(), bias.data_ptr(),
                      res.data_ptr() if has_res else 0, y.data_ptr(), B, h, w, cin, cout, k, stride, int(relu), 1,
                      scratch.data_ptr(), scratch.numel(), stream)
        for rnd in range(args.rounds + 1):
            for v in variants:
                _lib.call("dvsg_debug_set_option", b"conv_variant", v)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(args.rep):
                    run()
                e1.record()
                e1.synchronize()
                if rnd > 0:
                    times[v].append(e0.elapsed_time(e1) * 1e3 / args.rep)
        cells = []
        for v in variants:
            us = float(np.median(times[v]))
            tot[v] += us * sum(1 for s in layer_shapes(B, args.height, args.width) if s[1:] == (h, w, cin, cout, k, stride, relu, has_res)) \
                if args.layers == "uniq" else us
            cells.append("%8.1f %6.1f %5.2f" % (us, flops / us / 1e6, nbytes / us / 1e6))
        print("%-24s %8d %5d %5d %2d %2d | " % (name, M, cout, K, k, stride) + " | ".join(cells))
        del x, wt, y, res
    print("sum over the 52 conv_gemm layers of one step: " + "  ".join("v%d %.2f ms" % (v, tot[v] / 1e3) for v in variants))


if __name__ == "__main__":
    main()
