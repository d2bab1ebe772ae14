"""Latency A/B: gate_batch_dev on 1 and 256 gates with the library named by TFHE_HIP_LIB; outputs checked against the
oracle.  usage: TFHE_HIP_LIB=$PWD/tfhe_amd/ab_x.so [LAT_PARAMS=80] python tools/lat_bench.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, Gate, PARAMS, PARAMS_80  # noqa: E402
import tfhe_amd.engine as _eng  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402

P = PARAMS_80 if os.environ.get("LAT_PARAMS") == "80" else PARAMS
if os.environ.get("TFHE_HIP_LIB") and P.lam != 128:
    _eng.lib_path = lambda params=P: os.environ["TFHE_HIP_LIB"]   # A/B builds of the other parameter set
client = Client(P)
keys = client.keygen(42)
eng = Engine(0, P)
eng.load_keys_coef(keys.bk_coef, keys.ks)
rng = np.random.default_rng(0)
bits = rng.integers(0, 2, (2, 256)).astype(np.uint8)
a, b = client.encrypt(keys, bits[0], seed=1), client.encrypt(keys, bits[1], seed=2)
d_a, d_b, d_o = (eng.dev_alloc(a.nbytes) for _ in range(3))
eng.to_dev(d_a, a)
eng.to_dev(d_b, b)
res = {}
for count, reps in ((1, 30), (256, 8)):
    eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count)
    eng.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        eng.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count)
        eng.sync()
    res[count] = (time.perf_counter() - t0) / reps * 1e3
out = eng.from_dev(d_o, a.shape)
orc = Oracle(P)
ok = orc.keys_from_coef(keys.bk_coef, keys.ks)
want = orc.gate_batch(ok, int(Gate.NAND), a[:4], b[:4], nthreads=4)
good = np.array_equal(out[:4], want) and np.array_equal(client.decrypt(keys, out)[0], 1 - (bits[0] & bits[1]))
print("%s: 1 gate %.3f ms, 256 gates %.3f ms, parity %s" % (os.path.basename(os.environ.get("TFHE_HIP_LIB", "default")),
                                                          res[1], res[256], "ok" if good else "FAILED"))
