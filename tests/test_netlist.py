"""Levelized netlists (tfhe_amd/netlist.py): levelizer logic on the CPU; the reference's adder / comparator circuits
(src/test/test-addition-boot.cpp:30-115) evaluated level by level on the GPU over a device-resident pool."""
import numpy as np
import pytest

from tfhe_amd import Gate, PARAMS as P
from tfhe_amd.netlist import Netlist, comparison_mux, full_adder, full_adder_mux


def build(kind, nbits):
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    fn = {"add": full_adder, "addmux": full_adder_mux, "cmp": comparison_mux}[kind]
    nl.output("out", fn(nl, x, y))
    return nl


@pytest.mark.parametrize("kind", ["add", "addmux", "cmp"])
def test_plain_semantics_and_levels(kind):
    nbits = 8
    nl = build(kind, nbits)
    rng = np.random.default_rng(1)
    xs, ys = rng.integers(0, 1 << nbits, 200), rng.integers(0, 1 << nbits, 200)
    vals = {w: (xs >> i) & 1 for i, w in enumerate(nl.inputs["x"])}
    vals.update({w: (ys >> i) & 1 for i, w in enumerate(nl.inputs["y"])})
    v = nl.evaluate_plain(vals)
    out = sum(v[w].astype(np.int64) << i for i, w in enumerate(nl.outputs["out"]))
    assert np.array_equal(out, xs + ys if kind != "cmp" else (xs <= ys).astype(np.int64))
    levels = nl.levelize()
    assert sum(len(lv) for lv in levels) == len(nl.gates)
    written = set()
    for lv in levels:                      # every level only reads wires produced by earlier levels or inputs
        outs = {g[1] for g in lv}
        assert len(outs) == len(lv)
        for gate, o, a, b, c in lv:
            from tfhe_amd.netlist import n_inputs
            for w in (a, b, c)[:n_inputs(gate)]:
                assert w not in outs and (w in written or w < 2 * nbits)
        written |= outs
    arrs = nl.level_arrays(instances=3)    # replicated instances use disjoint row ranges
    assert arrs[0].shape[0] == 3 * len(levels[0]) and arrs[0][:, 1].max() < 3 * nl.n_wires


@pytest.mark.gpu
@pytest.mark.parametrize("kind,nbits", [("add", 4), ("addmux", 4), ("cmp", 4), ("addmux", 32)])
def test_circuits_on_gpu(kind, nbits, engine, client, keys, oracle, okeys):
    nl = build(kind, nbits)
    W = 5 if nbits <= 4 else 2
    rng = np.random.default_rng(nbits)
    xs = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    ys = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    for k in range(W):
        xb = ((xs[k] >> np.arange(nbits)) & 1).astype(np.uint8)
        yb = ((ys[k] >> np.arange(nbits)) & 1).astype(np.uint8)
        pool[k * nl.n_wires + np.array(nl.inputs["x"])] = client.encrypt(keys, xb, seed=100 + k)
        pool[k * nl.n_wires + np.array(nl.inputs["y"])] = client.encrypt(keys, yb, seed=200 + k)
    d_pool = engine.dev_alloc(pool.nbytes)          # the library's own device-memory helpers (no torch needed)
    engine.to_dev(d_pool, pool)
    nboot = nl.evaluate(lambda lv: engine.gate_level_dev(d_pool, pool.shape[0], lv), instances=W)
    assert nboot == W * sum(1 for g in nl.gates if g[0] <= int(Gate.MUX))
    res = engine.from_dev(d_pool, pool.shape)
    engine.dev_free(d_pool)
    for k in range(W):
        bits, _ = client.decrypt(keys, res[k * nl.n_wires + np.array(nl.outputs["out"])])
        val = int(sum(int(b) << i for i, b in enumerate(bits)))
        assert val == (int(xs[k] + ys[k]) if kind != "cmp" else int(xs[k] <= ys[k])), (kind, k)
    if nbits <= 4:
        # bit-exact against the reference's sequential evaluation order (oracle, instance 0)
        rows = {w: pool[w] for w in nl.inputs["x"] + nl.inputs["y"]}
        for gate, o, a, b, c in nl.gates:
            rows[o] = oracle.gate(okeys, gate, rows.get(a), rows.get(b), rows.get(c))
        for w in nl.outputs["out"]:
            assert np.array_equal(res[w], rows[w])


@pytest.mark.gpu
def test_level_contract_is_enforced(engine):
    from tfhe_amd import TfheHipError
    d_pool = engine.dev_alloc(4 * (P.n + 1) * 4)
    with pytest.raises(TfheHipError):   # row 2 written twice
        engine.gate_level_dev(d_pool, 4, [[0, 2, 0, 1, -1], [0, 2, 1, 0, -1]])
    with pytest.raises(TfheHipError):   # row 1 read and written
        engine.gate_level_dev(d_pool, 4, [[0, 1, 0, 1, -1]])
    with pytest.raises(TfheHipError):   # out of range
        engine.gate_level_dev(d_pool, 4, [[0, 4, 0, 1, -1]])
    engine.dev_free(d_pool)


@pytest.mark.gpu
def test_long_run_of_random_gates(engine, client, keys):
    """the reference's noise-drift soak test (src/test/test-long-run.cpp:61-183) in batched form: W independent pools
    of 50 ciphertexts; every step applies one random gate (all 11 bootstrapped kinds incl. MUX, plus NOT) per pool to
    random rows of that pool, all W gates of a step as one level.  Every ciphertext must decrypt to the plaintext
    simulation throughout, and the outputs of bootstrapped gates must stay within 1/16 of +-1/8 (:179-183)."""
    from tfhe_amd.truth import TRUTH
    W, R, steps = 32, 50, 120
    rng = np.random.default_rng(2024)
    plain = rng.integers(0, 2, (W, R)).astype(np.uint8)
    pool = client.encrypt(keys, plain.reshape(-1), seed=99)
    d_pool = engine.dev_alloc(pool.nbytes)
    engine.to_dev(d_pool, pool)
    fresh = np.zeros((W, R), bool)                       # rows last written by a bootstrapped gate
    kinds = list(range(0, 11)) + [int(Gate.NOT)]
    for step in range(steps):
        level = np.full((W, 5), -1, np.int32)
        for k in range(W):
            gate = kinds[rng.integers(0, len(kinds))]
            o, a, b, c = rng.choice(R, 4, replace=False)  # output row distinct from the inputs, as in the reference
            level[k] = (gate, k * R + o, k * R + a, k * R + b, k * R + c)
            plain[k, o] = TRUTH[gate](plain[k, a:a + 1], plain[k, b:b + 1], plain[k, c:c + 1])[0]
            fresh[k, o] = gate <= int(Gate.MUX)
        engine.gate_level_dev(d_pool, W * R, level)
        if step % 40 == 39 or step == steps - 1:
            bits, phases = client.decrypt(keys, engine.from_dev(d_pool, pool.shape))
            assert np.array_equal(bits.reshape(W, R), plain), "step %d" % step
            resid = np.abs(np.abs(phases.astype(np.int64)) - (1 << 29)).reshape(W, R)[fresh]
            assert resid.max() < (1 << 28), resid.max()
    engine.dev_free(d_pool)


@pytest.mark.gpu
def test_compiled_netlist_equals_level_by_level(engine, client, keys):
    """tfhe_hip_netlist_*: the whole circuit validated and uploaded once, replayed as one hipGraph (or, where the
    runtime cannot capture, enqueued without host round-trips).  Same pool contents, bit for bit, as the level-by-level
    path; a second run on fresh inputs in the same pool works; contract violations are caught at creation."""
    from tfhe_amd import TfheHipError
    nbits, W = 8, 6
    nl = build("addmux", nbits)
    rng = np.random.default_rng(77)

    def fresh_pool(seed):
        xs = rng.integers(0, 1 << nbits, W, dtype=np.int64)
        ys = rng.integers(0, 1 << nbits, W, dtype=np.int64)
        pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
        for k in range(W):
            xb = ((xs[k] >> np.arange(nbits)) & 1).astype(np.uint8)
            yb = ((ys[k] >> np.arange(nbits)) & 1).astype(np.uint8)
            pool[k * nl.n_wires + np.array(nl.inputs["x"])] = client.encrypt(keys, xb, seed=seed + 2 * k)
            pool[k * nl.n_wires + np.array(nl.inputs["y"])] = client.encrypt(keys, yb, seed=seed + 2 * k + 1)
        return pool, xs, ys

    levels = nl.level_arrays(instances=W)
    d_a, d_b = engine.dev_alloc(W * nl.n_wires * (P.n + 1) * 4), engine.dev_alloc(W * nl.n_wires * (P.n + 1) * 4)
    compiled = engine.netlist_create(d_b, W * nl.n_wires, levels)
    print("hipGraph" if compiled.is_graph else "direct enqueue", "levels", len(levels), "bootstrapped gates", compiled.n_bootstrapped)
    for seed in (300, 400):
        pool, xs, ys = fresh_pool(seed)
        engine.to_dev(d_a, pool)
        engine.to_dev(d_b, pool)
        nl.evaluate(lambda lv: engine.gate_level_dev(d_a, pool.shape[0], lv), instances=W)
        compiled.run()
        ra, rb = engine.from_dev(d_a, pool.shape), engine.from_dev(d_b, pool.shape)
        assert np.array_equal(ra, rb)
        for k in range(W):
            bits, _ = client.decrypt(keys, rb[k * nl.n_wires + np.array(nl.outputs["out"])])
            assert int(sum(int(b) << i for i, b in enumerate(bits))) == int(xs[k] + ys[k])
    compiled.close()
    with pytest.raises(TfheHipError):   # second level reads and writes row 1
        engine.netlist_create(d_b, 4, [[[0, 2, 0, 1, -1]], [[0, 1, 1, 2, -1]]])
    empty = engine.netlist_create(d_b, 4, [])
    empty.run()
    empty.close()
    engine.dev_free(d_a)
    engine.dev_free(d_b)
