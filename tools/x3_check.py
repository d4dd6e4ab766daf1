#!/usr/bin/env python3
"""f32x3 (bfloat16 x 3 products) first checks: single layers through dvsg_conv_gemm_f32x3 against float64 torch convolutions
next to the exact float32 kernel's error, every work decomposition (plain tiles, split-K, stream-K), then the whole
network's F_t against the exact path and the step time at B=16 1280x720.   tools/x3_check.py [layers|net|time ...]"""
import os, sys, statistics
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib

dev = torch.device("cuda:0")
what = sys.argv[1:] or ["layers", "net", "time"]


def layer(B, h, w, cin, cout, k, stride, relu=1, with_res=True, scale=1.0, seed=3):
    g = torch.Generator(device=dev).manual_seed(seed)
    x = (torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3) * scale
    K = k * k * cin
    wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = (torch.rand((B, ho, wo, cout), generator=g, device=dev) - 0.5) if with_res else None
    scratch = torch.empty(((80 << 20),), dtype=torch.uint8, device=dev)
    packed = torch.empty((cout * K * 6,), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), cout, K, s)
    y32 = torch.empty((B, ho, wo, cout), device=dev)
    y3 = torch.empty((B, ho, wo, cout), device=dev)
    rp = res.data_ptr() if with_res else 0
    _lib.call("dvsg_conv_gemm_f32", x.data_ptr(), wt.data_ptr(), bias.data_ptr(), rp, y32.data_ptr(), B, h, w, cin, cout, k,
              stride, relu, 1, scratch.data_ptr(), scratch.numel(), s)
    _lib.call("dvsg_conv_gemm_f32x3", x.data_ptr(), packed.data_ptr(), bias.data_ptr(), rp, y3.data_ptr(), B, h, w, cin, cout, k,
              stride, relu, 1, scratch.data_ptr(), scratch.numel(), s)
    y3b = torch.empty_like(y3)
    _lib.call("dvsg_conv_gemm_f32x3", x.data_ptr(), packed.data_ptr(), bias.data_ptr(), rp, y3b.data_ptr(), B, h, w, cin, cout, k,
              stride, relu, 1, scratch.data_ptr(), scratch.numel(), s)
    torch.cuda.synchronize()
    w4 = wt.reshape(cout, k, k, cin).permute(0, 3, 1, 2).double()
    ref = torch.nn.functional.conv2d(x.permute(0, 3, 1, 2).double(), w4, bias.double(), stride=stride, padding=k // 2).permute(0, 2, 3, 1)
    if with_res:
        ref = ref + res.double()
    if relu:
        ref = torch.relu(ref)
    sc = float(ref.abs().max())
    e32 = float((y32.double() - ref).abs().max()) / sc
    e3 = float((y3.double() - ref).abs().max()) / sc
    r32 = float(((y32.double() - ref) ** 2).mean().sqrt()) / sc
    r3 = float(((y3.double() - ref) ** 2).mean().sqrt()) / sc
    print("B=%d %dx%d %d->%d k%d s%d relu%d res%d scale %g: max err / max|y|  f32 %.2e  f32x3 %.2e   rms f32 %.2e  f32x3 %.2e   repeat bits %s"
          % (B, h, w, cin, cout, k, stride, relu, with_res, scale, e32, e3, r32, r3, torch.equal(y3, y3b)), flush=True)
    assert e3 <= 2.0 * e32 + 1e-7 and torch.equal(y3, y3b)


if "layers" in what:
    for cfg in [(2, 45, 80, 64, 64, 1, 1), (2, 45, 80, 64, 256, 1, 1), (16, 90, 160, 256, 64, 1, 1), (16, 90, 160, 128, 128, 3, 1),
                (16, 90, 160, 128, 128, 3, 2), (16, 45, 80, 256, 256, 3, 1), (16, 45, 80, 1024, 256, 1, 1), (16, 23, 40, 512, 512, 3, 1),
                (16, 23, 40, 2048, 512, 1, 1), (1, 23, 40, 512, 512, 3, 1), (1, 23, 40, 512, 2048, 1, 1), (1, 45, 80, 256, 256, 3, 1),
                (3, 37, 53, 128, 512, 1, 1), (1, 1, 1, 64, 64, 3, 1), (1, 9, 16, 2048, 512, 1, 1), (4, 90, 160, 256, 640, 1, 1)]:
        layer(*cfg)
    layer(2, 45, 80, 64, 256, 1, 1, relu=0, with_res=False)
    layer(2, 45, 80, 256, 256, 3, 1, scale=1e-4)
    layer(2, 45, 80, 256, 256, 3, 1, scale=1e4)

if "net" in what or "time" in what:
    import bench
    from coupe.dvsg_amd.networks import LocNet
    from coupe.dvsg_amd.weights import make_synthetic_weights
    net = LocNet(make_synthetic_weights(0))

if "net" in what:
    for B, H, W in [(2, 64, 96), (2, 288, 512), (1, 720, 1280), (4, 720, 1280)]:
        x = bench.gpu_windows(B, H, W, 50, dev)
        u = x[..., 18:].contiguous()
        out = torch.empty((B, H, W, 3), device=dev)
        Fs = {}
        for prec in ("f32", "f32x3", "f32s"):
            F = torch.empty((B, 25, 2), device=dev)
            net.stabilize(x, u, out, F, precision=prec)
            torch.cuda.synchronize()
            Fs[prec] = F.clone()
        print("B=%d %dx%d F_t: |f32x3 - f32| %.3g   |f32s - f32| %.3g   max|F_t| %.3g" % (
            B, W, H, float((Fs["f32x3"] - Fs["f32"]).abs().max()), float((Fs["f32s"] - Fs["f32"]).abs().max()),
            float(Fs["f32"].abs().max())), flush=True)

if "time" in what:
    B, H, W = 16, 720, 1280
    x = torch.cat([bench.gpu_windows(8, H, W, 50 + i, dev) for i in range(2)], 0).contiguous()
    u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    times = {}
    for rnd in range(4):
        for prec in ("f32", "f32x3", "f32s"):
            net.stabilize(x, u, out, F, precision=prec)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(6):
                net.stabilize(x, u, out, F, precision=prec)
            e1.record()
            torch.cuda.synchronize()
            times.setdefault(prec, []).append(e0.elapsed_time(e1) / 6)
    for prec, t in times.items():
        print("B=16 1280x720 %s: %.3f ms/step  %.1f frames/s" % (prec, statistics.median(t), 16e3 / statistics.median(t)), flush=True)

if "conv1" in what:
    # conv1 tap (stage 0) of the f32x3 kernel against the exact kernel and a float64 torch convolution of the scaled input
    import numpy as np
    import bench
    from coupe.dvsg_amd.networks import LocNet
    from coupe.dvsg_amd.weights import make_synthetic_weights
    net = LocNet(make_synthetic_weights(0))
    for B, H, W in [(2, 64, 96), (1, 70, 101), (2, 288, 512), (1, 720, 1280), (1, 7, 5)]:
        x = bench.gpu_windows(B, H, W, 77, dev)
        a32 = net.tap(x, 0, precision="f32")
        a3 = net.tap(x, 0, precision="f32x3")
        _lib.call("dvsg_debug_set_option", b"x3_conv1", 0)
        a3off = net.tap(x, 0, precision="f32x3")
        _lib.call("dvsg_debug_set_option", b"x3_conv1", 1)
        print("conv1 B=%d %dx%d: |x3 - f32| %.3g (max |y| %.3g), x3_conv1=0 equals f32: %s" % (
            B, W, H, float((a3 - a32).abs().max()), float(a32.abs().max()), torch.equal(a3off, a32)), flush=True)
