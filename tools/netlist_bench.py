"""32-bit full_adder_MUX circuits (src/test/test-addition-boot.cpp:30-68), W independent instances: level-by-level
(tfhe_hip_gate_level_dev, one host round-trip per level) against the compiled netlist (tfhe_hip_netlist_*, one graph
launch per evaluation).  usage: python tools/netlist_bench.py [W ...]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tfhe_amd import Client, Engine, PARAMS as P  # noqa: E402
from tfhe_amd.netlist import Netlist, full_adder_mux  # noqa: E402

client = Client()
keys = client.keygen(42)
eng = Engine(0)
eng.load_keys_coef(keys.bk_coef, keys.ks)
nbits = 32
nl = Netlist()
x, y = nl.input("x", nbits), nl.input("y", nbits)
nl.output("out", full_adder_mux(nl, x, y))
for W in [int(a) for a in sys.argv[1:]] or [1, 64, 1024]:
    rng = np.random.default_rng(W)
    xs, ys = rng.integers(0, 1 << nbits, W, dtype=np.int64), rng.integers(0, 1 << nbits, W, dtype=np.int64)
    pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    bits_x = ((xs[:, None] >> np.arange(nbits)) & 1).astype(np.uint8)
    bits_y = ((ys[:, None] >> np.arange(nbits)) & 1).astype(np.uint8)
    rows_x = (np.arange(W)[:, None] * nl.n_wires + np.array(nl.inputs["x"])[None, :]).reshape(-1)
    rows_y = (np.arange(W)[:, None] * nl.n_wires + np.array(nl.inputs["y"])[None, :]).reshape(-1)
    pool[rows_x] = client.encrypt(keys, bits_x.reshape(-1), seed=1)
    pool[rows_y] = client.encrypt(keys, bits_y.reshape(-1), seed=2)
    d_pool = eng.dev_alloc(pool.nbytes)
    levels = nl.level_arrays(instances=W)
    compiled = eng.netlist_create(d_pool, pool.shape[0], levels)
    res = {}
    for name, fn in (("level by level", lambda: nl.evaluate(lambda lv: eng.gate_level_dev(d_pool, pool.shape[0], lv), instances=W)),
                     ("compiled", lambda: compiled.run())):
        eng.to_dev(d_pool, pool)
        fn()
        eng.to_dev(d_pool, pool)
        eng.sync()
        t0 = time.perf_counter()
        fn()
        eng.sync()
        res[name] = time.perf_counter() - t0
        out = eng.from_dev(d_pool, pool.shape)
        rows_o = (np.arange(W)[:, None] * nl.n_wires + np.array(nl.outputs["out"])[None, :]).reshape(-1)
        got = client.decrypt(keys, out[rows_o])[0].reshape(W, -1)
        val = (got.astype(np.int64) << np.arange(got.shape[1])).sum(axis=1)
        assert np.array_equal(val, xs + ys), name
    print("W=%d: %d levels, %d bootstrapped gates; level by level %.1f ms, compiled (%s) %.1f ms"
          % (W, len(levels), compiled.n_bootstrapped, res["level by level"] * 1e3,
             "hipGraph" if compiled.is_graph else "direct", res["compiled"] * 1e3), flush=True)
    compiled.close()
    eng.dev_free(d_pool)
