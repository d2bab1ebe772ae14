import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from tfhe_amd import Client, Engine, Gate
client = Client(); keys = client.keygen(42)
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, (2, 512)).astype(np.uint8)
a, b = client.encrypt(keys, bits[0], seed=1), client.encrypt(keys, bits[1], seed=2)
eng = Engine(0); eng.set_arithmetic("fast"); eng.load_keys_coef(keys.bk_coef, keys.ks)
d_a, d_b, d_o = (eng.dev_alloc(a.nbytes) for _ in range(3))
eng.to_dev(d_a, a); eng.to_dev(d_b, b)
out = []
for count in (1, 64, 256, 512):
    eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
    t0 = time.perf_counter()
    for _ in range(8):
        eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count); eng.sync()
    out.append("%d: %.3f ms (kernel %.3f)" % (count, (time.perf_counter() - t0) / 8 * 1e3, eng.last_kernel_ms()[0]))
print(os.environ.get("TFHE_HIP_LIB", "default"), " ".join(out), flush=True)
