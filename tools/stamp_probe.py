#!/usr/bin/env python3
"""Per-workgroup phase times of conv_gemm_kernel (mode 0) from a -DDVSG_STAMPS build of the library:
   DVSG_AMD_LIB=build/lib_stamps.so python tools/stamp_probe.py B,H,W,Cin,Cout,k [...]
   (PROBE_PREC=f32s: the pieces kernels; PROBE_RES=1: with a residual, a unit's conv3; PROBE_REP=20: warmed-up clock)"""
import os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib

dev = torch.device("cuda:0")
stream = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device=dev).manual_seed(0)
for arg in sys.argv[1:]:
    B, H, W, cin, cout, k = (int(v) for v in arg.split(","))
    x = torch.rand((B, H, W, cin), generator=g, device=dev) - 0.3
    K = k * k * cin
    wt = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    y = torch.empty((B, H, W, cout), device=dev)
    res = torch.rand((B, H, W, cout), generator=g, device=dev) if os.environ.get("PROBE_RES") == "1" else None   # a unit's conv3
    RES = res.data_ptr() if res is not None else 0
    FN = "dvsg_conv_gemm_f32"
    if os.environ.get("PROBE_PREC") == "f32s":   # float16 pieces [Cout][K/32][32 hi | 32 lo]
        FN = "dvsg_conv_gemm_f32s"
        hi = wt.half()
        lo = (wt - hi.float()).half()
        wt = torch.cat([hi.reshape(cout, K // 32, 32), lo.reshape(cout, K // 32, 32)], 2).contiguous()
    if os.environ.get("PROBE_PREC") == "f32x3":  # bfloat16 piece stages (dvsg_pack_weights_f32x3)
        FN = "dvsg_conv_gemm_f32x3"
        packed = torch.empty((cout * K * 6,), dtype=torch.uint8, device=dev)
        _lib.call("dvsg_pack_weights_f32x3", wt.data_ptr(), packed.data_ptr(), cout, K, stream)
        wt = packed
    scratch = torch.zeros(66 << 20, dtype=torch.uint8, device=dev)
    _lib.call("dvsg_debug_set_option", b"conv_variant", 6)   # no stream-K: every tile is a mode-0 workgroup
    REP = int(os.environ.get("PROBE_REP", "1"))     # back-to-back launches before the measured one (clock governor settles)
    for rnd in range(3):
        for _ in range(REP - 1):
            _lib.call(FN, x.data_ptr(), wt.data_ptr(), bias.data_ptr(), RES, y.data_ptr(), B, H, W, cin, cout, k,
                      1, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        _lib.call(FN, x.data_ptr(), wt.data_ptr(), bias.data_ptr(), RES, y.data_ptr(), B, H, W, cin, cout, k,
                  1, 1, 1, scratch.data_ptr(), scratch.numel(), stream)
        e1.record(); e1.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    M = B * H * W
    mt = -(-M // 128)
    tiles = mt * (cout // 128) if cout % 128 == 0 and mt * (cout // 128) >= 512 else mt * (cout // 64)
    st = scratch[2048:2048 + tiles * 64].view(torch.int64).reshape(tiles, 8).cpu().numpy().astype(np.uint64)
    t0 = st[:, 0].astype(np.float64)
    rt = st[:, 6].astype(np.float64) / 100.0       # us
    rt -= rt.min()
    pro, first, loop, epi, vmw = (st[:, i].astype(np.float64) for i in (1, 2, 3, 4, 5))
    barw = (st[:, 7] & np.uint64((1 << 40) - 1)).astype(np.float64)
    xcc = (st[:, 7] >> np.uint64(56)).astype(int)
    tot = pro + first + loop + epi
    life_us = st[:, 0].astype(np.float64) / 100.0
    clk_each = tot / life_us                     # shader cycles per us, per workgroup
    clk = float(np.median(clk_each))
    print("in-kernel shader clock: median %.0f MHz (p10 %.0f, p90 %.0f)" % (clk, np.percentile(clk_each, 10), np.percentile(clk_each, 90)))
    q = lambda a: "med %8.0f  p10 %8.0f  p90 %8.0f  max %8.0f" % (np.median(a), np.percentile(a, 10), np.percentile(a, 90), a.max())
    print("== B=%d %dx%d %d->%d k%d: %d tiles, %.1f us, %.1f TFLOP/s" % (B, H, W, cin, cout, k, tiles, us, 2.0 * M * cout * K / us / 1e6))
    print("start time (us after first)  ", q(rt))
    print("prologue (cycles)            ", q(pro))
    print("first stage wait             ", q(first))
    print("main loop                    ", q(loop), " = %.1f us" % (np.median(loop) / clk))
    print("  of which own-DMA wait      ", q(vmw))
    print("  of which barrier wait      ", q(barw))
    print("epilogue                     ", q(epi))
    print("workgroup lifetime           ", q(tot), " = %.1f us" % (np.median(tot) / clk))
    order = np.argsort(rt)
    first_round = order[:512]
    print("first 512 workgroups: lifetime", q(tot[first_round]), "; the rest:", q(tot[order[512:]]) if tiles > 512 else "")
    for xc in range(8):
        m = xcc == xc
        if m.any():
            print("  XCC %d: %4d wgs, lifetime med %.0f, loop med %.0f, vm wait med %.0f, barrier wait med %.0f" %
                  (xc, m.sum(), np.median(tot[m]), np.median(loop[m]), np.median(vmw[m]), np.median(barw[m])))
