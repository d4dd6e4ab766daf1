#!/usr/bin/env python3
"""One float16 (hi / lo weights) conv layer at a time through dvsg_conv_gemm_f16s, timed with HIP events, for each value
of a debug option:  tools/layer_bench_f16.py <option> <v0,v1,..> [k,cin,cout,B,h,w,resmode,relu ...]
Prints ms per launch, algorithmic TFLOP/s and the HBM-side rate of the layer's algorithmic bytes."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coupe.dvsg_amd import _lib
opt = sys.argv[1].encode()
vals = [int(v) for v in sys.argv[2].split(",")]
layers = [tuple(int(t) for t in a.split(",")) for a in sys.argv[3:]] or [
    (1, 128, 512, 32, 270, 480, 1, 1), (1, 256, 1024, 32, 135, 240, 1, 1), (1, 256, 512, 32, 270, 480, 0, 0),
    (1, 512, 128, 32, 270, 480, 0, 1), (1, 1024, 256, 32, 135, 240, 0, 1), (3, 128, 128, 32, 270, 480, 0, 1),
    (1, 128, 512, 16, 90, 160, 1, 1), (1, 256, 1024, 16, 45, 80, 1, 1)]
dev = torch.device("cuda:0")
g = torch.Generator(device=dev).manual_seed(1)
st = torch.cuda.current_stream().cuda_stream
scratch = torch.empty(160 << 20, dtype=torch.uint8, device=dev)   # tickets + slabs + the packed weight copies
for k, cin, cout, B, h, w, rmode, relu in layers:
    K = k * k * cin
    x = (torch.rand((B, h, w, cin), generator=g, device=dev) - 0.3).half()
    w32 = (torch.rand((cout, K), generator=g, device=dev) - 0.5) * (2.0 / K ** 0.5)
    hi = w32.half()
    lo = ((w32 - hi.float()) * 2048.0).half()
    ws = torch.stack([hi.reshape(cout // 64, 64, K), lo.reshape(cout // 64, 64, K)], 1).reshape(2 * cout, K).contiguous()
    bias = torch.rand((cout,), generator=g, device=dev) - 0.5
    res = (torch.rand((B, h, w, cout), generator=g, device=dev) - 0.5).half() if rmode else None
    y = torch.empty((B, h, w, cout), device=dev, dtype=torch.float16)
    M = B * h * w
    flops = 2.0 * M * cout * K
    nbytes = 2.0 * (M * cin + M * cout * (2 if rmode else 1))
    line = "k%d %4d->%4d M=%8d res %d:" % (k, cin, cout, M, rmode)
    ref = None
    for v in vals:
        _lib.call("dvsg_debug_set_option", opt, v)
        run = lambda: _lib.call("dvsg_conv_gemm_f16s", x.data_ptr(), ws.data_ptr(), bias.data_ptr(), res.data_ptr() if res is not None else 0,
                                y.data_ptr(), B, h, w, cin, cout, k, 1, relu, 1, scratch.data_ptr(), scratch.numel(), st)
        for _ in range(3):
            run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        same = "" if ref is None else (" same bits" if torch.equal(ref, y) else " DIFFERS %.3g" % float((ref.float() - y.float()).abs().max()))
        ref = y.clone() if ref is None else ref
        line += "   %s=%d %.3f ms %5.0f TFLOP/s %.2f TB/s%s" % (opt.decode(), v, ms, flops / ms / 1e9, nbytes / ms / 1e9, same)
    print(line, flush=True)
    del x, y, res
    torch.cuda.empty_cache()
_lib.call("dvsg_debug_set_option", opt, 1)
