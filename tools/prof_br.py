"""Small driver for rocprofv3 counter passes: one warm-up and two timed launches of the blind-rotation kernel
(bootstrap without keyswitch) over B ciphertexts."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
with_ks = len(sys.argv) > 2 and sys.argv[2] == "ks"
client = Client()
keys = client.keygen(42)
eng = Engine(0)
eng.load_keys_coef(keys.bk_coef, keys.ks)
x = client.encrypt(keys, np.random.default_rng(0).integers(0, 2, B).astype(np.uint8), seed=3)
for _ in range(3):
    out = eng.bootstrap_batch(x, with_ks=with_ks)
    print("kernel ms", eng.last_kernel_ms())
bits, _ = client.decrypt(keys, out, extracted=not with_ks)
print("ok", bool((bits == np.random.default_rng(0).integers(0, 2, B)).all()))
