#!/usr/bin/env python3
"""Numerics of the "3 x bfloat16" representation of float32 GEMM operands (DESIGN.md section 8, next
steps): a = a1 + a2 + a3 with bfloat16 pieces is EXACT for float32 a, every bf16 x bf16 product is
exact in float32, and keeping the 6 largest of the 9 cross terms leaves an error at the level of
float32's own rounding.  CPU only (torch bf16 casts, float32 matmuls standing in for the f32
accumulation of v_mfma_f32_32x32x16_bf16); shapes of block 3's 3x3 conv."""
import numpy as np
import torch

rng = np.random.default_rng(0)
M, K, N = 2048, 2304, 256
A = np.maximum(rng.standard_normal((M, K)), 0).astype(np.float32)
W = (rng.standard_normal((K, N)) * np.sqrt(2.0 / K)).astype(np.float32)
ref = A.astype(np.float64) @ W.astype(np.float64)


def split3(x):
    t = torch.from_numpy(x)
    p1 = t.to(torch.bfloat16).to(torch.float32)
    r1 = t - p1
    p2 = r1.to(torch.bfloat16).to(torch.float32)
    r2 = r1 - p2
    p3 = r2.to(torch.bfloat16).to(torch.float32)
    return p1.numpy(), p2.numpy(), p3.numpy(), float((r2 - p3).abs().max())


a1, a2, a3, ra = split3(A)
w1, w2, w3, rw = split3(W)
print("what the three pieces miss: A %.1e  W %.1e" % (ra, rw))
six = sum((x @ y).astype(np.float32) for x, y in ((a1, w1), (a1, w2), (a2, w1), (a2, w2), (a1, w3), (a3, w1)))
three = sum((x @ y).astype(np.float32) for x, y in ((a1, w1), (a1, w2), (a2, w1)))
for name, v in (("float32 matmul", A @ W), ("bf16 x 3, 6 products", six), ("bf16 x 2, 3 products", three)):
    e = np.abs(v - ref)
    print("%-22s max err %.2e  rms %.2e  (|C| max %.2f)" % (name, e.max(), np.sqrt((e ** 2).mean()), np.abs(ref).max()))
