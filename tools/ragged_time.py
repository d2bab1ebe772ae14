"""time of batches just above whole rounds of 2,048 (GPU box); usage: python tools/ragged_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tfhe_amd import Client, Engine, Gate
client = Client(); keys = client.keygen(42)
N = 16384 + 1024
bits = np.random.default_rng(0).integers(0, 2, (2, N)).astype(np.uint8)
a, b = client.encrypt(keys, bits[0], seed=1), client.encrypt(keys, bits[1], seed=2)
for arith in ("fast", "exact"):
    eng = Engine(0); eng.set_arithmetic(arith); eng.load_keys_coef(keys.bk_coef, keys.ks)
    d_a, d_b, d_o = (eng.dev_alloc(a.nbytes) for _ in range(3))
    eng.to_dev(d_a, a); eng.to_dev(d_b, b)
    out = []
    for count in (4096, 4096 + 37, 4096 + 600, 16384, 16384 + 37, 16384 + 1024):
        eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
        t0 = time.perf_counter()
        for _ in range(3):
            eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
        out.append("%d: %.1f" % (count, (time.perf_counter() - t0) / 3 * 1e3))
    print(arith, " ".join(out), flush=True)
    eng.close()
