"""Blind rotation + keyswitch time of small batches per launch shape and arithmetic mode (GPU box).
usage: python tools/lat_time.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tfhe_amd import Client, Engine, Gate
client = Client(); keys = client.keygen(42)
rng = np.random.default_rng(0)
N = 2048
bits = rng.integers(0, 2, (2, N)).astype(np.uint8)
a, b = client.encrypt(keys, bits[0], seed=1), client.encrypt(keys, bits[1], seed=2)
for arith in (sys.argv[1:] or ("fast", "exact")):
    for shape in ("6", "1", "4", "8"):
        os.environ["TFHE_HIP_BR_WAVES"] = shape
        eng = Engine(0); eng.set_arithmetic(arith); eng.load_keys_coef(keys.bk_coef, keys.ks)
        d_a, d_b, d_o = (eng.dev_alloc(a.nbytes) for _ in range(3))
        eng.to_dev(d_a, a); eng.to_dev(d_b, b)
        out = []
        for count in (1, 64, 256, 384, 512, 640, 768, 1024, 1280, 1536, 2048):
            eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
            t0 = time.perf_counter()
            for _ in range(5):
                eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
            out.append("%d: %.2f" % (count, (time.perf_counter() - t0) / 5 * 1e3))
        print(arith, "shape", shape, " ".join(out), flush=True)
        eng.close()
