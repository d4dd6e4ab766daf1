#!/usr/bin/env python3
"""dvsg_stabilize_* replayed from a captured HIP graph against plain stream launches (same step, same buffers)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

prec = sys.argv[1] if len(sys.argv) > 1 else "f32"
dev = torch.device("cuda:0")
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
H = int(sys.argv[3]) if len(sys.argv) > 3 else 720
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1280
net = LocNet(make_synthetic_weights(0))
x = bench.gpu_windows(B, H, W, 1234, dev)
u = x[..., 18:].contiguous()
out = torch.empty((B, H, W, 3), device=dev)
F = torch.empty((B, 25, 2), device=dev)

def step():
    net.stabilize(x, u, out, F, precision=prec)

def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

plain = timed(step)
ref, ref_F = out.clone(), F.clone()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    step()
side.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g, stream=side):
    step()
graph = timed(g.replay)
print("B=%d %dx%d " % (B, H, W), end="")
print("%s: plain %.3f ms/step, graph replay %.3f ms/step (%.2f %%), identical output: %s" %
      (prec, plain, graph, 100 * (plain / graph - 1), bool(torch.equal(out, ref) and torch.equal(F, ref_F))))
