"""Full-size parity, once: EVERY output ciphertext of a BASELINE configs[1] batch (16,384 bootsNAND) and of 4,096 bootsMUX
against the reference CPU path run on the box's host cores (oracle/_ref, else the C restatement), coefficient by
coefficient.  Takes a minute or two of CPU time per thousand gates and core; prints progress.
usage: python tools/parity_full.py [count]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfhe_amd import Client, Engine, Gate
from bench import cpu_gates, _cpu_threads
count = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
client = Client(); keys = client.keygen(42)
eng = Engine(0); eng.load_keys_coef(keys.bk_coef, keys.ks)
assert eng.arithmetic == "exact"
rng = np.random.default_rng(4242)
bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
ct = [client.encrypt(keys, bits[i], 9100 + i) for i in range(3)]
threads = _cpu_threads()
for gate, n in ((Gate.NAND, count), (Gate.MUX, count // 4)):
    got = eng.gate_batch(gate, ct[0][:n], ct[1][:n], ct[2][:n] if gate == Gate.MUX else None)
    differ, t0 = 0, time.perf_counter()
    for lo in range(0, n, 1024):
        hi = min(n, lo + 1024)
        want, _, kind = cpu_gates(keys, gate, ct[0][lo:hi], ct[1][lo:hi], ct[2][lo:hi] if gate == Gate.MUX else None, threads)
        differ += int((want != got[lo:hi]).any(axis=1).sum())
        print("%s rows %d..%d compared, %d differ so far (%.0f s)" % (gate.name, lo, hi, differ, time.perf_counter() - t0), flush=True)
    print("%s: %d ciphertexts x 631 coefficients against the CPU %s on %d threads: %d differ" % (gate.name, n, kind, threads, differ))
