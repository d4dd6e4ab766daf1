import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights
dev = torch.device("cuda:0")
net = LocNet(make_synthetic_weights(0))
for (B, H, W, n) in ((16, 720, 1280, 20), (1, 720, 1280, 40), (2, 720, 1280, 40), (4, 720, 1280, 30), (1, 288, 512, 40), (1, 2160, 3840, 10)):
    x = bench.gpu_windows(B, H, W, 1234, dev); u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev); F = torch.empty((B, 25, 2), device=dev)
    res = {}
    for rnd in range(2):
        for thr in (1 << 30, 256, 128, 64, 1):
            _lib.call("dvsg_debug_set_option", b"wide16_min_tiles", thr)
            for _ in range(3): net.stabilize(x, u, out, F, precision="f16")
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(n): net.stabilize(x, u, out, F, precision="f16")
            torch.cuda.synchronize(); res[thr] = (time.perf_counter() - t0) / n * 1e3
    print((B, H, W), "  ".join("%s: %.3f" % ("off" if k > 1e9 else ">=%d" % k, v) for k, v in res.items()))
    del x, u, out
