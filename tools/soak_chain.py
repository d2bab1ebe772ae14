"""Chained soak in the shape of the reference's src/test/test-long-run.cpp:61-183, batched: W pools of 50 ciphertexts,
`steps` random gates per pool, every step one netlist level reading outputs of earlier steps.  Default 2,500 pools x 400
steps = 10^6 chained gates.  Every ciphertext is decrypted at every check; prints the residual |phase| - MU of the rows last
written by a bootstrapped gate.  usage: python tools/soak_chain.py [fast|exact] [pools] [steps]   (GPU box)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests.test_fast_paths import soak  # noqa: E402
from tfhe_amd import Client, Engine  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fast"
W = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
client = Client()
keys = client.keygen(42)
eng = Engine(0)
eng.set_arithmetic(mode)
eng.load_keys_coef(keys.bk_coef, keys.ks)
t0 = time.perf_counter()
stats = soak(eng, client, keys, W=W, R=50, steps=steps, seed=4001, check_every=40)
print("%s arithmetic: %d pools x 50 ciphertexts, %d steps = %d chained gates, every check decrypts to the plaintext simulation; %.1f s"
      % (mode.upper(), W, steps, W * steps, time.perf_counter() - t0))
for step, (rms, mx, dep) in sorted(stats.items()):
    print("step %4d: residual rms 2^%.2f max 2^%.2f LSB (margin 2^29), deepest chain %d bootstraps" % (step, np.log2(rms), np.log2(mx + 1), dep))
eng.close()
