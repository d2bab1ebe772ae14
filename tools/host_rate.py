"""PCIe-inclusive rate of the host-buffer entry point (tfhe_hip_gate_batch: H2D of both operands, kernels, D2H of the
result) beside the device-resident rate, for DESIGN.md section 6.  usage: python tools/host_rate.py [B]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, Gate  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
client = Client()
keys = client.keygen(42)
eng = Engine(0)
eng.load_keys_coef(keys.bk_coef, keys.ks)
rng = np.random.default_rng(1)
ba, bb = rng.integers(0, 2, B).astype(np.uint8), rng.integers(0, 2, B).astype(np.uint8)
ca, cb = client.encrypt(keys, ba, seed=11), client.encrypt(keys, bb, seed=12)
out = eng.gate_batch(Gate.NAND, ca, cb)
t0 = time.perf_counter()
for _ in range(3):
    out = eng.gate_batch(Gate.NAND, ca, cb)
dt = (time.perf_counter() - t0) / 3
bits, _ = client.decrypt(keys, out)
assert np.array_equal(bits, 1 - (ba & bb))
br, ks = eng.last_kernel_ms()     # calls above 4,096 gates run in chunks: these are the last chunk's kernels
print("host buffers, B=%d: %.1f ms per call = %.0f gates/s (last chunk's kernels: %.1f + %.1f ms)"
      % (B, dt * 1e3, B / dt, br, ks))
