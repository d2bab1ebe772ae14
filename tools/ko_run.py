"""diagnostic: times one blind-rotation launch (no correctness check) for knock-out builds; usage ko_run.py B"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine
B = int(sys.argv[1])
client = Client(); keys = client.keygen(42)
eng = Engine(0); eng.load_keys_coef(keys.bk_coef, keys.ks)
x = client.encrypt(keys, np.zeros(B, np.uint8), seed=3)
for _ in range(3):
    eng.bootstrap_batch(x, with_ks=False)
    print("B", B, "kernel ms", eng.last_kernel_ms()[0], flush=True)
