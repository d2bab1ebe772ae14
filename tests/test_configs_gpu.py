"""BASELINE.json configs at their full per-GPU sizes, on the GPU through the C-ABI:
  configs[2]  one GPU's shard of the 2^20-gate batch (131,072 gates),
  configs[3]  the 32-bit ripple-carry adder and comparator netlists of src/test/test-addition-boot.cpp:30-115, and the
              level-by-level sharding over ranks (two ranks on the one GPU of the test box, gloo collectives).
Bit-exact rows against the oracle; size-independent properties (decrypt = truth table, phase margin) on everything."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from tfhe_amd import Gate, PARAMS as P
from tfhe_amd.netlist import Netlist, comparison_mux, full_adder, full_adder_mux

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_config2_shard_of_131072_gates(engine, client, keys, oracle, okeys):
    count = (1 << 20) // 8
    rng = np.random.default_rng(2020)
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    ca, cb = client.encrypt(keys, bits[0], 1201), client.encrypt(keys, bits[1], 1202)
    d_a, d_b, d_o = (engine.dev_alloc(ca.nbytes) for _ in range(3))
    try:
        engine.to_dev(d_a, ca)
        engine.to_dev(d_b, cb)
        engine.gate_batch_dev(Gate.NAND, d_o, d_a, d_b, None, count=count)
        got = engine.from_dev(d_o, ca.shape)
    finally:
        for p in (d_a, d_b, d_o):
            engine.dev_free(p)
    dec, phases = client.decrypt(keys, got)
    assert np.array_equal(dec, 1 - (bits[0] & bits[1]))
    assert np.abs(np.abs(phases.astype(np.int64)) - (1 << 29)).max() < (1 << 28)
    pick = np.unique(np.concatenate([np.arange(64), rng.integers(0, count, 112), np.arange(count - 16, count)]))
    assert pick.shape[0] >= 180
    want = oracle.gate_batch(okeys, Gate.NAND, ca[pick], cb[pick], None, nthreads=16)
    assert np.array_equal(got[pick], want)


@pytest.mark.parametrize("kind", ["cmp", "add", "addmux"])
def test_config3_32bit_circuits(kind, engine, client, keys):
    """32-bit comparison_MUX / full_adder / full_adder_MUX, W instances per level, as compiled netlists; decrypted
    results equal the plain computation (the reference's own check, src/test/test-addition-boot.cpp:177-254)"""
    nbits, W = 32, 24
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    nl.output("out", {"add": full_adder, "addmux": full_adder_mux, "cmp": comparison_mux}[kind](nl, x, y))
    rng = np.random.default_rng(len(kind))
    xs = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    ys = rng.integers(0, 1 << nbits, W, dtype=np.int64)
    ys[:3] = xs[:3] + np.array([0, 1, -1])       # equal, just above, just below: the comparator's hard cases
    ys = np.clip(ys, 0, (1 << nbits) - 1)
    pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    sh = np.arange(nbits)
    base = (np.arange(W) * nl.n_wires)[:, None]
    pool[(base + np.array(x)[None, :]).reshape(-1)] = client.encrypt(keys, ((xs[:, None] >> sh) & 1).reshape(-1), 1301)
    pool[(base + np.array(y)[None, :]).reshape(-1)] = client.encrypt(keys, ((ys[:, None] >> sh) & 1).reshape(-1), 1302)
    d_pool = engine.dev_alloc(pool.nbytes)
    try:
        engine.to_dev(d_pool, pool)
        compiled = engine.netlist_create(d_pool, pool.shape[0], nl.level_arrays(instances=W))
        compiled.run()
        compiled.close()
        res = engine.from_dev(d_pool, pool.shape)
    finally:
        engine.dev_free(d_pool)
    outs = np.array(nl.outputs["out"])
    bits, _ = client.decrypt(keys, res[(base + outs[None, :]).reshape(-1)])
    got = (bits.reshape(W, -1).astype(np.int64) << np.arange(outs.shape[0])).sum(axis=1)
    assert np.array_equal(got, (xs <= ys).astype(np.int64) if kind == "cmp" else xs + ys), kind


def test_two_ranks_on_one_gpu(tmp_path, engine, client, keys):
    """BASELINE configs[2]/[3], the N > 1 path with the GPU engine as the per-rank compute: two processes, both on GPU 0,
    gloo collectives.  The gathered batch and both ranks' pools must equal the single-process results bit for bit."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker_gpu.py"), str(tmp_path)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    r0, r1 = (np.load(tmp_path / ("rank%d.npz" % k)) for k in range(2))
    assert np.array_equal(r0["out"], r1["out"]) and np.array_equal(r0["pool"], r1["pool"])
    # single process, same inputs (the worker's generator and seeds)
    rng = np.random.default_rng(31)
    count = 1537
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    ca, cb = client.encrypt(keys, bits[0], 501), client.encrypt(keys, bits[1], 502)
    assert np.array_equal(engine.gate_batch(Gate.XOR, ca, cb), r0["out"])
    nbits, W = 4, 9
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    nl.output("out", full_adder_mux(nl, x, y))
    xs, ys = rng.integers(0, 1 << nbits, W), rng.integers(0, 1 << nbits, W)
    assert np.array_equal(xs, r0["xs"]) and np.array_equal(ys, r0["ys"])
    host = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    for k in range(W):
        host[k * nl.n_wires + np.array(x)] = client.encrypt(keys, ((xs[k] >> np.arange(nbits)) & 1).astype(np.uint8), 600 + k)
        host[k * nl.n_wires + np.array(y)] = client.encrypt(keys, ((ys[k] >> np.arange(nbits)) & 1).astype(np.uint8), 700 + k)
    d_pool = engine.dev_alloc(host.nbytes)
    try:
        engine.to_dev(d_pool, host)
        nl.evaluate(lambda lv: engine.gate_level_dev(d_pool, host.shape[0], lv), instances=W)
        single = engine.from_dev(d_pool, host.shape)
    finally:
        engine.dev_free(d_pool)
    assert np.array_equal(single, r0["pool"])
    for k in range(W):
        b, _ = client.decrypt(keys, single[k * nl.n_wires + np.array(nl.outputs["out"])])
        assert int(sum(int(v) << i for i, v in enumerate(b))) == int(xs[k] + ys[k])


def test_key_buffers_adopted_from_another_context(engine, client, keys):
    """tfhe_hip_keys_device_buffers / tfhe_hip_keys_adopt: a context that never saw the host keys takes over the three
    device-resident key buffers of another one (here by plain device copies; tfhe_amd.parallel.broadcast_keys_device
    moves the same buffers with a broadcast, covered by test_two_ranks_on_one_gpu) and computes the same ciphertexts"""
    import ctypes
    from tfhe_amd import Engine, TfheHipError
    hip = ctypes.CDLL("libamdhip64.so")
    e = Engine(0)
    try:
        with pytest.raises(TfheHipError):
            e.adopt_keys()                                    # no buffers yet
        engine.sync()
        for (src, n1), (dst, n2) in zip(engine.key_buffers(), e.key_buffers()):
            assert n1 == n2 and src != dst
            assert hip.hipMemcpy(ctypes.c_void_p(dst), ctypes.c_void_p(src), ctypes.c_size_t(n1), 3) == 0   # device to device
        bits = np.random.default_rng(3).integers(0, 2, (3, 700)).astype(np.uint8)
        ct = [client.encrypt(keys, bits[i], 1500 + i) for i in range(3)]
        with pytest.raises(TfheHipError):
            e.gate_batch(Gate.NAND, ct[0], ct[1])             # filled but not adopted: still "keys not loaded"
        e.adopt_keys()
        assert np.array_equal(e.gate_batch(Gate.MUX, *ct), engine.gate_batch(Gate.MUX, *ct))
        assert np.array_equal(e.gate_batch(Gate.NAND, ct[0][:3], ct[1][:3]), engine.gate_batch(Gate.NAND, ct[0][:3], ct[1][:3]))
        e.set_arithmetic("fast")                              # the FAST key is derived from the adopted buffers
        assert np.array_equal(client.decrypt(keys, e.gate_batch(Gate.XOR, ct[0], ct[1]))[0], bits[0] ^ bits[1])
    finally:
        e.close()


def test_device_group_equals_single_context(engine, client, keys):
    """tfhe_hip_group_*: the C-level multi-GPU path (one context, stream and host thread per member, keys replicated
    device to device, contiguous shards).  The test box has one GPU, so the group is {0, 0, 0}: three contexts on it.
    16,384 + 37 NAND gates and a MUX batch must equal the single-context results bit for bit, in both arithmetic modes"""
    from tfhe_amd import Group
    rng = np.random.default_rng(99)
    count = 16384 + 37
    bits = rng.integers(0, 2, (3, count)).astype(np.uint8)
    ct = [client.encrypt(keys, bits[i], 1400 + i) for i in range(3)]
    g = Group([0, 0, 0])
    try:
        assert len(g) == 3
        g.load_keys_coef(keys.bk_coef, keys.ks)
        out = g.gate_batch(Gate.NAND, ct[0], ct[1])
        assert np.array_equal(out, engine.gate_batch(Gate.NAND, ct[0], ct[1]))
        m = 3000 + 1
        mux = g.gate_batch(Gate.MUX, ct[0][:m], ct[1][:m], ct[2][:m])
        assert np.array_equal(mux, engine.gate_batch(Gate.MUX, ct[0][:m], ct[1][:m], ct[2][:m]))
        assert g.gate_batch(Gate.NAND, ct[0][:0], ct[1][:0]).shape == (0, P.n + 1)
        two = g.gate_batch(Gate.XOR, ct[0][:2], ct[1][:2])         # fewer gates than members: empty shards
        assert np.array_equal(two, engine.gate_batch(Gate.XOR, ct[0][:2], ct[1][:2]))
        g.set_arithmetic("fast")
        f = g.gate_batch(Gate.NAND, ct[0], ct[1])
        assert np.array_equal(client.decrypt(keys, f)[0], 1 - (bits[0] & bits[1]))
    finally:
        g.close()


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it (the command form the driver uses for N = 1): bench.py starts
    torch.distributed.run as a child process before touching the GPU and relays rank 0's line.  Two ranks share the one
    GPU of the test box, so the collectives run over gloo (RCCL needs one device per rank); the key broadcast goes through
    the verified path (fingerprints compared across the ranks)."""
    import json
    env = dict(os.environ, TFHE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "1100", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["keys_via"] == "gloo-staged" and line["rccl_ranks"] == 0
    assert line["config"]["outputs_decrypt_correctly"] and line["value"] > 0
    assert line["kernel_ms"]["blind_rotate_per_rank_min"] <= line["kernel_ms"]["blind_rotate_per_rank_max"]
    assert line["exact_arithmetic_same_run"]["value"] > 0


def test_group_records_how_keys_travelled(engine, keys):
    """tfhe_hip_group_key_path: member 0 uploaded from the host, members on the same device copied within it (3); on a
    multi-GPU host 1 = peer access enabled (xGMI), 2 = staged.  Replicas are fingerprint-checked against member 0."""
    from tfhe_amd import Group
    g = Group([0, 0])
    try:
        g.load_keys_coef(keys.bk_coef, keys.ks)
        assert g.key_paths() == [0, 3]
    finally:
        g.close()
    fp = engine.key_fingerprints()
    assert len(set(fp)) == 3 and all(0 < v < 1 << 64 for v in fp)
    assert engine.key_fingerprints() == fp


@pytest.mark.parametrize("mode", ["exact", "fast"])
def test_group_netlist_levels_equal_single_context(mode, engine, fast, client, keys):
    """tfhe_hip_group_gate_level: BASELINE configs[3] through the C-level group - every member holds a replica of the pool,
    evaluates its block of a level and the written rows travel device to device into the other replicas.  After every
    circuit all three replicas must equal the pool a single context computes level by level, bit for bit (EXACT and FAST:
    FAST results do not depend on how a level is split), and a level that breaks the level contract is refused whole."""
    from tfhe_amd import Group, TfheHipError
    single = engine if mode == "exact" else fast
    nbits, W = 8, 7
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    nl.output("sum", full_adder_mux(nl, x, y))
    nl.output("le", comparison_mux(nl, x, y))
    rng = np.random.default_rng(31)
    xs, ys = rng.integers(0, 1 << nbits, W), rng.integers(0, 1 << nbits, W)
    pool = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    sh = np.arange(nbits)
    base = (np.arange(W) * nl.n_wires)[:, None]
    pool[(base + np.array(x)[None, :]).reshape(-1)] = client.encrypt(keys, ((xs[:, None] >> sh) & 1).reshape(-1), 1501)
    pool[(base + np.array(y)[None, :]).reshape(-1)] = client.encrypt(keys, ((ys[:, None] >> sh) & 1).reshape(-1), 1502)
    levels = nl.level_arrays(instances=W)
    assert min(lv.shape[0] for lv in levels) >= 3 and len(levels) > 10
    d_ref = single.dev_alloc(pool.nbytes)
    g = Group([0, 0, 0])
    members, d_pools = [], []
    try:
        g.load_keys_coef(keys.bk_coef, keys.ks)
        g.set_arithmetic(mode)
        members = [g.member(i) for i in range(3)]
        d_pools = [m.dev_alloc(pool.nbytes) for m in members]
        for m, d in zip(members, d_pools):
            m.to_dev(d, pool)
        single.to_dev(d_ref, pool)
        for lv in levels:
            g.gate_level(d_pools, pool.shape[0], lv)
            single.gate_level_dev(d_ref, pool.shape[0], lv)
        want = single.from_dev(d_ref, pool.shape)
        for m, d in zip(members, d_pools):
            assert np.array_equal(m.from_dev(d, pool.shape), want)
        g.gate_level(d_pools, pool.shape[0], levels[0][:2])          # fewer gates than members: an empty block
        g.gate_level(d_pools, pool.shape[0], levels[0][:0])
        bad = np.array(levels[1][:4])
        bad[3, 1] = bad[0, 2]                                         # writes a row another gate of the level reads
        with pytest.raises(TfheHipError):
            g.gate_level(d_pools, pool.shape[0], bad)
        for m, d in zip(members, d_pools):                            # the refused level left every replica untouched
            assert np.array_equal(m.from_dev(d, pool.shape), want)
    finally:
        for m, d in zip(members, d_pools):
            m.dev_free(d)
        single.dev_free(d_ref)
        g.close()
    outs = np.array(nl.outputs["sum"])
    bits, _ = client.decrypt(keys, want[(base + outs[None, :]).reshape(-1)])
    got = (bits.reshape(W, -1).astype(np.int64) << np.arange(outs.shape[0])).sum(axis=1)
    assert np.array_equal(got, xs + ys)
    le, _ = client.decrypt(keys, want[(base + np.array(nl.outputs["le"])[None, :]).reshape(-1)])
    assert np.array_equal(le.reshape(-1), (xs <= ys).astype(np.uint8))


def test_damaged_key_broadcast_is_detected_and_replaced():
    """the verified key broadcast: rank 1 damages eight bytes of what it received (failure injection in
    parallel.broadcast_keys_device); the fingerprints differ, every rank reports "host", bench.py redistributes the keys
    through host memory and the run's outputs still decrypt correctly on both ranks"""
    import json
    env = dict(os.environ, TFHE_BENCH_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0", TFHE_TEST_CORRUPT_KEYS_RANK="1")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--batch", "600", "--no-cpu-baseline", "--arith", "fast"], env=env, capture_output=True, text=True,
                       timeout=420)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert line["keys_via"] == "host" and line["config"]["outputs_decrypt_correctly"]
    assert "falling back to the host path" in r.stderr
