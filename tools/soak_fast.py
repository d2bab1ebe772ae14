"""Soak of the FAST arithmetic mode: 2^20 bootsNAND + 2^18 bootsMUX gates on fresh encryptions (BASELINE configs[2] size on one
GPU), every output decrypted; reports wrong bits and the output noise.  usage: python tools/soak_fast.py [log2_count]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, Gate
L = int(sys.argv[1]) if len(sys.argv) > 1 else 20
client = Client(); keys = client.keygen(42)
eng = Engine(0); eng.set_arithmetic("fast"); eng.load_keys_coef(keys.bk_coef, keys.ks)
chunk, total, wrong, worst, sq = 1 << 17, 1 << L, 0, 0, 0.0
t0 = time.perf_counter()
for k in range(total // chunk):
    rng = np.random.default_rng(5000 + k)
    a, b = (rng.integers(0, 2, chunk).astype(np.uint8) for _ in range(2))
    ca, cb = client.encrypt(keys, a, 7000 + 2 * k), client.encrypt(keys, b, 7001 + 2 * k)
    out = eng.gate_batch(Gate.NAND, ca, cb)
    dec, ph = client.decrypt(keys, out)
    wrong += int((dec != 1 - (a & b)).sum())
    err = np.abs(ph.astype(np.int64)) - (1 << 29)
    worst = max(worst, int(np.abs(err).max())); sq += float((err.astype(np.float64) ** 2).sum())
    print("chunk %d: wrong so far %d, max |phase -+ MU| 2^%.2f" % (k, wrong, np.log2(worst)), flush=True)
print("NAND: %d gates, %d wrong, noise rms 2^%.2f max 2^%.2f LSB (margin 2^29), %.1f s" %
      (total, wrong, 0.5 * np.log2(sq / total), np.log2(worst), time.perf_counter() - t0))
m = 1 << (L - 2)
rng = np.random.default_rng(99)
a, b, c = (rng.integers(0, 2, m).astype(np.uint8) for _ in range(3))
ct = [client.encrypt(keys, x, 8000 + i) for i, x in enumerate((a, b, c))]
out = eng.gate_batch(Gate.MUX, *ct)
dec, ph = client.decrypt(keys, out)
err = np.abs(ph.astype(np.int64)) - (1 << 29)
print("MUX: %d gates, %d wrong, noise rms 2^%.2f max 2^%.2f LSB" % (m, int((dec != np.where(a == 1, b, c)).sum()),
      np.log2(err.std()), np.log2(np.abs(err).max())))
