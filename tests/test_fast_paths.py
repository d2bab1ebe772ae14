"""The FAST arithmetic mode (the bench headline's mode) through every product path that carries it, on the GPU through
the C-ABI: netlist levels (tfhe_hip_gate_level_dev), compiled netlists (tfhe_hip_netlist_*), the entry point that
allocates nothing, device groups, the full-size bootsMUX batch of BASELINE configs[4], and a chained soak in the shape of
the reference's src/test/test-long-run.cpp.

What FAST promises (include/tfhe_hip.h, tests/test_fast_mode.py): decrypted bits identical to the reference path's,
|phase -+ MU| < 2^28 LSB (1/16) on every bootstrapped output, and bit-for-bit determinism within the mode.  So the checks
here are: decrypted bits of EVERY wire equal the oracle's sequential evaluation of the same circuit (and the plain
computation), residuals below 2^28, and the different entry points equal each other bit for bit within the mode.
EXACT-mode twins of the same paths compare ciphertexts with the oracle coefficient by coefficient."""
import numpy as np
import pytest

from tests.truth import TRUTH
from tfhe_amd import Gate, PARAMS as P
from tfhe_amd.netlist import Netlist, comparison_mux, full_adder, full_adder_mux

pytestmark = pytest.mark.gpu

CIRCUITS = {"add": full_adder, "addmux": full_adder_mux, "cmp": comparison_mux}
MU = 1 << 29


def build(kind, nbits):
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    nl.output("out", CIRCUITS[kind](nl, x, y))
    return nl, x, y


def make_pool(nl, x, y, nbits, W, client, keys, seed, rng):
    xs = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    ys = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    ys[:3] = np.clip(xs[:3] + np.array([0, 1, -1]), 0, (1 << nbits) - 1)   # equal, just above, just below
    sh = np.arange(nbits)
    base = (np.arange(W) * nl.n_wires)[:, None]
    pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    pool[(base + np.array(x)[None, :]).reshape(-1)] = client.encrypt(keys, ((xs[:, None] >> sh) & 1).reshape(-1).astype(np.uint8), seed)
    pool[(base + np.array(y)[None, :]).reshape(-1)] = client.encrypt(keys, ((ys[:, None] >> sh) & 1).reshape(-1).astype(np.uint8), seed + 1)
    return pool, xs, ys, base


def plain_wires(nl, x, y, xs, ys, nbits):
    """bit of every wire of every instance: [W][n_wires]"""
    vals = {w: ((xs >> i) & 1).astype(np.int64) for i, w in enumerate(x)}
    vals.update({w: ((ys >> i) & 1).astype(np.int64) for i, w in enumerate(y)})
    v = nl.evaluate_plain(vals)
    return np.stack([v[w] for w in range(nl.n_wires)], axis=1).astype(np.uint8)


def bootstrapped_wires(nl):
    return np.array([g[1] for g in nl.gates if g[0] <= int(Gate.MUX)])


@pytest.mark.parametrize("kind", ["add", "addmux", "cmp"])
def test_4bit_circuits_fast_levels_and_compiled(kind, fast, client, keys, oracle, okeys):
    """src/test/test-addition-boot.cpp:30-115 at 4 bits on a FAST context: level by level and as a compiled netlist.
    After EVERY level the wires written so far decrypt to the oracle's sequential evaluation (instance 0) and to the plain
    computation (all instances); bootstrapped outputs stay within 2^28 of +-MU; compiled == level by level bit for bit."""
    nbits, W = 4, 5
    nl, x, y = build(kind, nbits)
    pool, xs, ys, base = make_pool(nl, x, y, nbits, W, client, keys, 2100, np.random.default_rng(len(kind) + 40))
    want_bits = plain_wires(nl, x, y, xs, ys, nbits)
    # the oracle's sequential evaluation, instance 0 (the reference's order of bootsXXX calls)
    rows = {w: pool[w] for w in x + y}
    for gate, o, a, b, c in nl.gates:
        rows[o] = oracle.gate(okeys, gate, rows.get(a), rows.get(b), rows.get(c))
    obits, _ = client.decrypt(keys, np.stack([rows[w] for w in range(nl.n_wires)]))
    assert np.array_equal(obits, want_bits[0])
    levels = nl.level_arrays(instances=W)
    d_a, d_b = fast.dev_alloc(pool.nbytes), fast.dev_alloc(pool.nbytes)
    try:
        fast.to_dev(d_a, pool)
        fast.to_dev(d_b, pool)
        written = np.zeros(nl.n_wires, bool)
        written[x + y] = True
        boot = np.zeros(nl.n_wires, bool)
        boot[bootstrapped_wires(nl)] = True
        for lv, lv0 in zip(levels, nl.levels_cached()):
            fast.gate_level_dev(d_a, pool.shape[0], lv)
            written[[g[1] for g in lv0]] = True
            got = fast.from_dev(d_a, pool.shape)
            bits, phases = client.decrypt(keys, got)
            bits, phases = bits.reshape(W, -1), phases.reshape(W, -1).astype(np.int64)
            assert np.array_equal(bits[:, written], want_bits[:, written])
            assert np.array_equal(bits[0, written], obits[written])
            new = [g[1] for g in lv0 if g[0] <= int(Gate.MUX)]
            if new:
                assert np.abs(np.abs(phases[:, new]) - MU).max() < (1 << 28)
        compiled = fast.netlist_create(d_b, pool.shape[0], levels)
        compiled.run()
        compiled.close()
        assert np.array_equal(fast.from_dev(d_b, pool.shape), got)           # bit for bit within the mode
    finally:
        fast.dev_free(d_a)
        fast.dev_free(d_b)
    outs = np.array(nl.outputs["out"])
    val = (bits[:, outs].astype(np.int64) << np.arange(outs.shape[0])).sum(axis=1)
    assert np.array_equal(val, (xs <= ys).astype(np.int64) if kind == "cmp" else xs + ys)


@pytest.mark.parametrize("kind", ["add", "addmux", "cmp"])
def test_32bit_circuits_fast_wide_levels(kind, fast, client, keys):
    """32 bits, 150-300 instances per level: levels are 300 to 19,200 bootstraps wide, i.e. both the latency kernel and the
    job-list branch of the throughput kernel (k_blind_rotate<MODE_GATE, 8, true> with A.jobs) run in FAST.  Every wire of
    every instance decrypts to the plain computation; compiled == level by level bit for bit."""
    nbits, W = 32, {"add": 300, "addmux": 150, "cmp": 200}[kind]
    nl, x, y = build(kind, nbits)
    pool, xs, ys, base = make_pool(nl, x, y, nbits, W, client, keys, 2200, np.random.default_rng(len(kind) + 50))
    want_bits = plain_wires(nl, x, y, xs, ys, nbits)
    levels = nl.level_arrays(instances=W)
    widths = [int((lv[:, 0] < int(Gate.MUX)).sum() + 2 * (lv[:, 0] == int(Gate.MUX)).sum()) for lv in levels]
    assert max(widths) > 768 and min(w for w in widths if w) <= 768     # both the throughput and the latency kernel
    d_a, d_b = fast.dev_alloc(pool.nbytes), fast.dev_alloc(pool.nbytes)
    try:
        fast.to_dev(d_a, pool)
        fast.to_dev(d_b, pool)
        for lv in levels:
            fast.gate_level_dev(d_a, pool.shape[0], lv)
        compiled = fast.netlist_create(d_b, pool.shape[0], levels)
        compiled.run()
        compiled.close()
        got = fast.from_dev(d_a, pool.shape)
        assert np.array_equal(fast.from_dev(d_b, pool.shape), got)
    finally:
        fast.dev_free(d_a)
        fast.dev_free(d_b)
    bits, phases = client.decrypt(keys, got)
    bits, phases = bits.reshape(W, -1), phases.reshape(W, -1).astype(np.int64)
    assert np.array_equal(bits, want_bits)
    assert np.abs(np.abs(phases[:, bootstrapped_wires(nl)]) - MU).max() < (1 << 28)


def test_32bit_adder_exact_equals_the_oracle_level_by_level(engine, client, keys, oracle, okeys):
    """EXACT mode, 32-bit full_adder_MUX: EVERY wire of instance 0 (and the outputs of a second instance) equals the
    oracle's evaluation of the same levels coefficient by coefficient - the 32-bit form of test_netlist.py's 4-bit check"""
    nbits, W = 32, 3
    nl, x, y = build("addmux", nbits)
    pool, xs, ys, base = make_pool(nl, x, y, nbits, W, client, keys, 2300, np.random.default_rng(60))
    d = engine.dev_alloc(pool.nbytes)
    try:
        engine.to_dev(d, pool)
        compiled = engine.netlist_create(d, pool.shape[0], nl.level_arrays(instances=W))
        compiled.run()
        compiled.close()
        got = engine.from_dev(d, pool.shape)
    finally:
        engine.dev_free(d)
    ref = pool[: 2 * nl.n_wires].copy()                # instances 0 and 1 through the oracle, one level at a time
    for lv in nl.level_arrays(instances=2):
        outs = {}
        for g in np.unique(lv[:, 0]):
            sel = lv[lv[:, 0] == g]
            if g == int(Gate.COPY):
                outs[g] = (sel[:, 1], ref[sel[:, 2]].copy())
            elif g in (int(Gate.CONST0), int(Gate.CONST1)):
                v = np.zeros((sel.shape[0], P.n + 1), np.int32)
                v[:, P.n] = MU if g == int(Gate.CONST1) else -MU
                outs[g] = (sel[:, 1], v)
            else:
                c_ = ref[sel[:, 4]] if g == int(Gate.MUX) else None
                outs[g] = (sel[:, 1], oracle.gate_batch(okeys, int(g), ref[sel[:, 2]], ref[sel[:, 3]], c_, nthreads=4))
        for r_, v_ in outs.values():
            ref[r_] = v_
    assert np.array_equal(got[: 2 * nl.n_wires], ref)
    bits, _ = client.decrypt(keys, got[(base + np.array(nl.outputs["out"])[None, :]).reshape(-1)])
    val = (bits.reshape(W, -1).astype(np.int64) << np.arange(nbits + 1)).sum(axis=1)
    assert np.array_equal(val, xs + ys)


def test_noalloc_entry_point_in_fast_mode(fast, client, keys):
    """tfhe_hip_gate_batch_dev_noalloc == tfhe_hip_gate_batch_dev bit for bit on a FAST context (1,000 / 37 / 1 gates)"""
    rng = np.random.default_rng(2026)
    first = 1000
    bits = rng.integers(0, 2, (3, first)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], seed=2400 + i) for i in range(3)]
    d = []
    for c in ct:
        d.append(fast.dev_alloc(c.nbytes))
        fast.to_dev(d[-1], c)
    d_ref, d_out = fast.dev_alloc(ct[0].nbytes), fast.dev_alloc(ct[0].nbytes)
    try:
        for gate, count in ((Gate.NAND, 1000), (Gate.MUX, 37), (Gate.XOR, 1), (Gate.MUX, 1000)):
            fast.gate_batch_dev(gate, d_ref, d[0], d[1], d[2], count=count)
            fast.sync()
            fast.to_dev(d_out, np.full((first, P.n + 1), 0x5a5a5a5a, np.int32))
            fast.gate_batch_dev_noalloc(gate, d_out, d[0], d[1], d[2], count=count)
            fast.sync()
            want, got = fast.from_dev(d_ref, (count, P.n + 1)), fast.from_dev(d_out, (count, P.n + 1))
            assert np.array_equal(got, want), (gate, count)
            a, b, c = (bits[i][:count].astype(np.int64) for i in range(3))
            assert np.array_equal(client.decrypt(keys, got)[0], TRUTH[int(gate)](a, b, c))
    finally:
        for p in d + [d_ref, d_out]:
            fast.dev_free(p)


def test_device_group_in_fast_mode_equals_single_context(fast, client, keys):
    """Group([0, 0, 0]) in FAST == a single FAST context bit for bit on 16,384 + 37 NAND gates (three shards of 5,474 /
    5,474 / 5,473 against one batch with a 37-gate ragged tail) and on a MUX batch"""
    from tfhe_amd import Group
    rng = np.random.default_rng(101)
    count = 16384 + 37
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], 2500 + i) for i in range(3)]
    g = Group([0, 0, 0])
    try:
        g.set_arithmetic("fast")
        g.load_keys_coef(keys.bk_coef, keys.ks)
        out = g.gate_batch(Gate.NAND, ct[0], ct[1])
        single = fast.gate_batch(Gate.NAND, ct[0], ct[1])
        assert np.array_equal(client.decrypt(keys, out)[0], 1 - (bits[0] & bits[1]))
        assert np.array_equal(out, single)
        m = 2000
        assert np.array_equal(g.gate_batch(Gate.MUX, ct[0][:m], ct[1][:m], ct[2][:m]),
                              fast.gate_batch(Gate.MUX, ct[0][:m], ct[1][:m], ct[2][:m]))
    finally:
        g.close()


def test_config4_mux_batch_16384(engine, fast, client, keys, oracle, okeys):
    """BASELINE configs[4]: 16,384 bootsMUX gates in one call (two blind rotations + one keyswitch each,
    boot-gates.cpp:329-370).  EXACT: every output decrypts to a ? b : c, 96+ rows equal the oracle's ciphertexts; FAST: the
    same bits, residuals < 2^28, noise of the EXACT outputs' size."""
    count = 16384
    rng = np.random.default_rng(404)
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], 2600 + i) for i in range(3)]
    want_bits = np.where(bits[0] == 1, bits[1], bits[2])
    ex = engine.gate_batch(Gate.MUX, *ct)
    dec, ph = client.decrypt(keys, ex)
    assert np.array_equal(dec, want_bits)
    pick = np.unique(np.concatenate([np.arange(40), rng.integers(0, count, 48), np.arange(count - 16, count)]))
    assert pick.shape[0] >= 96
    want = oracle.gate_batch(okeys, Gate.MUX, ct[0][pick], ct[1][pick], ct[2][pick], nthreads=16)
    assert np.array_equal(ex[pick], want)
    fa = fast.gate_batch(Gate.MUX, *ct)
    dec_f, ph_f = client.decrypt(keys, fa)
    assert np.array_equal(dec_f, want_bits)
    err, err_f = np.abs(ph.astype(np.int64)) - MU, np.abs(ph_f.astype(np.int64)) - MU
    assert np.abs(err_f).max() < (1 << 28) and np.abs(err).max() < (1 << 28)
    assert 0.9 < err_f.std() / err.std() < 1.1


def soak(eng, client, keys, W, R, steps, seed, check_every):
    """src/test/test-long-run.cpp:61-183 in batched form: W pools of R ciphertexts, one random gate per pool and step (the
    11 bootstrapped kinds incl. MUX, plus NOT), every step one level; returns {step: (rms, max) of the residual of the rows
    last written by a bootstrapped gate} and asserts decryption == plaintext simulation at every check"""
    rng = np.random.default_rng(seed)
    plain = rng.integers(0, 2, (W, R)).astype(np.uint8)
    pool = client.encrypt(keys, plain.reshape(-1), seed=seed)
    d_pool = eng.dev_alloc(pool.nbytes)
    eng.to_dev(d_pool, pool)
    fresh = np.zeros((W, R), bool)
    depth = np.zeros((W, R), np.int32)                       # bootstraps on the longest path behind each ciphertext
    kinds = np.array(list(range(0, 11)) + [int(Gate.NOT)])
    stats = {}
    ar = np.arange(W)
    try:
        for step in range(steps):
            gate = kinds[rng.integers(0, len(kinds), W)]
            sel = np.argsort(rng.random((W, R)), axis=1)[:, :4]   # four distinct rows per pool: out, a, b, c
            o, a, b, c = sel.T
            level = np.stack([gate, ar * R + o, ar * R + a, ar * R + b, ar * R + c], axis=1).astype(np.int32)
            va, vb, vc = (plain[ar, k].astype(np.int64) for k in (a, b, c))
            res = np.zeros(W, np.uint8)
            for gk in np.unique(gate):
                m = gate == gk
                res[m] = TRUTH[int(gk)](va[m], vb[m], vc[m])
            boot = gate <= int(Gate.MUX)
            nin = np.where(gate == int(Gate.MUX), 3, np.where(gate == int(Gate.NOT), 1, 2))
            dmax = np.maximum(depth[ar, a], np.where(nin >= 2, depth[ar, b], 0))
            dmax = np.maximum(dmax, np.where(nin >= 3, depth[ar, c], 0))
            plain[ar, o], fresh[ar, o], depth[ar, o] = res, boot, dmax + boot
            eng.gate_level_dev(d_pool, W * R, level)
            if (step + 1) % check_every == 0 or step == steps - 1:
                bits, phases = client.decrypt(keys, eng.from_dev(d_pool, pool.shape))
                assert np.array_equal(bits.reshape(W, R), plain), "step %d" % step
                resid = (np.abs(phases.astype(np.int64)) - MU).reshape(W, R)[fresh].astype(np.float64)
                stats[step + 1] = (float(np.sqrt((resid ** 2).mean())), float(np.abs(resid).max()), int(depth.max()))
    finally:
        eng.dev_free(d_pool)
    return stats


def test_chained_soak_fast(fast, client, keys):
    """102,400 chained gates in FAST (256 pools x 50 rows x 400 steps, each step reading outputs of earlier steps): every
    ciphertext decrypts to the plaintext simulation at every check, residuals stay below 2^28, and the residual rms does
    not drift: step 400 within 10 % of step 40 (a bootstrap's output noise does not depend on its inputs' noise as long as
    they decrypt).  tools/soak_chain.py keeps a 10^6-gate record of the same run under profiles/."""
    stats = soak(fast, client, keys, W=256, R=50, steps=400, seed=3001, check_every=40)
    for step, (rms, mx, dep) in sorted(stats.items()):
        print("step %4d: residual rms 2^%.2f max 2^%.2f LSB, deepest chain %d bootstraps" % (step, np.log2(rms), np.log2(mx + 1), dep))
        assert mx < (1 << 28)
    assert 0.9 < stats[400][0] / stats[40][0] < 1.1
    assert stats[400][2] >= 25                               # the run is a chain, not a set of depth-one gates


def test_fast_results_do_not_depend_on_the_batch_size(fast, client, keys):
    """the latency kernel (batches up to 768: one workgroup per ciphertext) accumulates the row products in the throughput
    kernel's order (one FMA chain per spectrum value over the rows), so a gate's ciphertext is the same in a batch of 1, 9,
    600 or 2,100 - e.g. the same circuit at level widths 400 and 600"""
    rng = np.random.default_rng(808)
    count = 2100
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], 2700 + i) for i in range(3)]
    big = fast.gate_batch(Gate.MUX, *ct)                               # 2,048 in the 8-wave kernel + 52 in the latency kernel
    assert np.array_equal(client.decrypt(keys, big)[0], np.where(bits[0] == 1, bits[1], bits[2]))
    for lo, n in ((0, 1), (5, 9), (100, 400), (300, 600), (200, 800), (1000, 1024)):
        part = fast.gate_batch(Gate.MUX, ct[0][lo:lo + n], ct[1][lo:lo + n], ct[2][lo:lo + n])
        assert np.array_equal(part, big[lo:lo + n]), (lo, n)
    x = ct[0][:700]
    woks = fast.bootstrap_batch(x, with_ks=False)
    assert np.array_equal(fast.bootstrap_batch(x[:3], with_ks=False), woks[:3])
    rngb = np.random.default_rng(809)                                  # the reference seam (MODE_RAW), zero rotations among them
    acc = rngb.integers(-2**31, 2**31, (520, 2, P.N)).astype(np.int32)
    bara = rngb.integers(0, 2 * P.N, (520, P.n)).astype(np.int32)
    bara[:, ::7] = 0
    out = fast.blind_rotate_batch(acc, bara)
    assert np.array_equal(fast.blind_rotate_batch(acc[:4], bara[:4]), out[:4])
