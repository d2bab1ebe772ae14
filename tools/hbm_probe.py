"""Practical HBM rates on this box next to the FFT-product kernels (GPU box): a device-to-device copy and a read-only
reduction of the same footprint as the 2^16-product batch, then the product kernels at several batch sizes.
usage: python tools/hbm_probe.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Engine
dev = torch.device("cuda", 0)
x = torch.randint(-2**31, 2**31 - 1, (1 << 16, 1024), dtype=torch.int32, device=dev)
y = torch.empty_like(x)
def timed(fn, n=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
ms = timed(lambda: y.copy_(x))
print("copy 268 MB -> 268 MB: %.3f ms = %.2f TB/s (read + write)" % (ms, 2 * x.numel() * 4 / ms / 1e9))
ms = timed(lambda: x.sum())
print("read-only sum over 268 MB: %.3f ms = %.2f TB/s" % (ms, x.numel() * 4 / ms / 1e9))
z = torch.empty_like(x)
ms = timed(lambda: torch.add(x, y, out=z))
print("z = x + y (2 reads, 1 write, the product kernel's traffic): %.3f ms = %.2f TB/s" % (ms, 3 * x.numel() * 4 / ms / 1e9))
eng = Engine(0); eng.set_arithmetic("fast")
for n in (512, 1024, 2048):
    for logB in (14, 16, 18):
        B = 1 << logB
        if B * n * 4 * 3 > 8 << 30: continue
        a = torch.randint(-2 * n, 2 * n, (B, n), dtype=torch.int32, device=dev)
        b = torch.randint(-2**31, 2**31 - 1, (B, n), dtype=torch.int32, device=dev)
        o = torch.empty_like(a); torch.cuda.synchronize()
        ks = []
        for _ in range(6):
            eng.negacyclic_mul_n_dev(n, o, a, b, B); ks.append(eng.last_kernel_ms()[0])
        ms = float(np.median(ks[1:]))
        print("N=%d B=2^%d: %.3f ms, %.1f M products/s, %.2f TB/s algorithmic" % (n, logB, ms, B / ms / 1e3, 3 * n * 4 * B / ms / 1e9))
        del a, b, o
eng.close()
