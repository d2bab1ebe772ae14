"""diagnostic: times the keyswitch kernels alone (no correctness check); usage ks_time.py B"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, PARAMS as P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
client = Client(); keys = client.keygen(42)
eng = Engine(0); eng.load_keys_coef(keys.bk_coef, keys.ks)
u = np.random.default_rng(1).integers(-2**31, 2**31, (B, P.N + 1)).astype(np.int32)
ts = []
for _ in range(6):
    eng.keyswitch_batch(u)
    ts.append(eng.last_kernel_ms()[1])
print("B", B, "keyswitch ms (transpose + mfma [+ reduce])", " ".join("%.3f" % t for t in ts), flush=True)
