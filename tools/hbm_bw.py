#!/usr/bin/env python3
"""HBM read / write / copy bandwidth reference points (torch fill, sum and copy on 944 MB: the size
of block 1's [B=16,180,320,256] float32 activations), to judge the HBM-bound kernels against."""
import torch
n = 16 * 180 * 320 * 256
x = torch.empty(n, device="cuda"); y = torch.empty(n, device="cuda")
def t(fn, reps=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); e1.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = t(lambda: x.fill_(1.0)); print("write 944 MB: %7.1f us  %5.2f TB/s" % (us, 4.0 * n / us / 1e6))
us = t(lambda: x.sum()); print("read  944 MB: %7.1f us  %5.2f TB/s" % (us, 4.0 * n / us / 1e6))
us = t(lambda: y.copy_(x)); print("copy  944 MB: %7.1f us  %5.2f TB/s (read + write)" % (us, 8.0 * n / us / 1e6))
us = t(lambda: torch.add(x, y, out=y)); print("add (2 reads + 1 write): %7.1f us  %5.2f TB/s" % (us, 12.0 * n / us / 1e6))
