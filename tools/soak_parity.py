"""One-off soak: random gate kinds and batch sizes (all launch shapes, ragged tails) against the oracle, bit for bit.
usage: python tools/soak_parity.py [rounds]"""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, Gate, PARAMS, PARAMS_80  # noqa: E402
from tests.oracle_lib import Oracle  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 12
for P in (PARAMS, PARAMS_80):
    client = Client(P)
    keys = client.keygen(1234)
    eng = Engine(0, P)
    eng.load_keys_coef(keys.bk_coef, keys.ks)
    orc = Oracle(P)
    ok = orc.keys_from_coef(keys.bk_coef, keys.ks)
    rng = np.random.default_rng(5)
    checked = 0
    for rnd in range(rounds):
        count = int(rng.choice([1, 2, 63, 64, 65, 255, 256, 257, 511, 513, 767, 769, 1023, 1025, 2047, 2049, 3000, 4097, 5000]))
        gate = Gate(int(rng.integers(0, 11)))
        bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
        a, b, c = (client.encrypt(keys, bits[i], seed=1000 * rnd + i) for i in range(3))
        if rnd % 5 == 0:  # ciphertexts with many zero rotation amounts (the skipped iterations): coarse masks
            a = a.copy()
            a[:, : P.n // 2] = 0
        out = eng.gate_batch(gate, a, b, c if gate == Gate.MUX else None)
        pick = np.unique(np.concatenate([np.arange(min(count, 3)), rng.integers(0, count, 12), [count - 1]]))
        want = orc.gate_batch(ok, gate, a[pick], b[pick], c[pick] if gate == Gate.MUX else None, nthreads=16)
        assert np.array_equal(out[pick], want), (P, rnd, gate, count)
        checked += len(pick)
    print("%r: %d rounds, %d ciphertexts equal the oracle's" % (P, rounds, checked), flush=True)
    eng.close()
