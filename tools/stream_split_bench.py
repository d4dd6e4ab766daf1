#!/usr/bin/env python3
"""Experiment: does splitting the batch over S concurrent HIP streams hide the per-kernel tail
(tile-count quantisation)?  Same total work per step (B=16 720p windows)."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from coupe.dvsg_amd import _lib
from coupe.dvsg_amd.networks import LocNet
from coupe.dvsg_amd.weights import make_synthetic_weights

def run(S, B=16, H=720, W=1280, steps=8, warmup=3):
    dev = torch.device("cuda:0")
    weights = make_synthetic_weights(0)
    nets = [LocNet(weights) for _ in range(S)]
    streams = [torch.cuda.Stream() for _ in range(S)]
    x = bench.gpu_windows(B, H, W, 1234, dev)
    u = x[..., 18:].contiguous()
    out = torch.empty((B, H, W, 3), device=dev)
    F = torch.empty((B, 25, 2), device=dev)
    per = B // S
    wss = [n.workspace(per, H, W) for n in nets]
    torch.cuda.synchronize()
    def step():
        for s in range(S):
            b0 = s * per
            _lib.call("dvsg_stabilize_f32", nets[s].handle, x[b0:b0+per].data_ptr(), u[b0:b0+per].data_ptr(), per, H, W,
                      out[b0:b0+per].data_ptr(), F[b0:b0+per].data_ptr(), 0, 0, wss[s][0].data_ptr(), wss[s][1],
                      streams[s].cuda_stream)
    for _ in range(warmup): step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print("streams=%d  %.2f ms/step  %.1f frames/s" % (S, 1e3 * dt / steps, B * steps / dt))

if __name__ == "__main__":
    for S in (1, 2, 4, 1, 2):
        run(S)
