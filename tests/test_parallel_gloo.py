"""N > 1 host logic on CPU: world_size 2, gloo.  Rank 0 owns the keys and broadcasts them; gates shard in contiguous
blocks; outputs are all-gathered.  The per-rank compute is the oracle (a GPU is not available here); the sharded
result must equal the single-process result bit for bit."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shard_bounds_partition():
    from tfhe_amd.parallel import shard_bounds
    for count in (0, 1, 7, 8, 9, 16384, 1 << 20):
        for world in (1, 2, 3, 4, 8):
            spans = [shard_bounds(count, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == count
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    from tests.oracle_lib import Oracle
    from tfhe_amd import Client, Gate
    from tfhe_amd.parallel import broadcast_keys, sharded_gate_batch
    dist.init_process_group("gloo", rank=rank, world_size=world)
    client = Client()
    keys = broadcast_keys(client.keygen(7) if rank == 0 else None)
    orc = Oracle()
    ok = orc.keys_from_coef(keys.bk_coef, keys.ks)
    bits_a = np.array([0, 1, 1, 0, 1], np.uint8)
    bits_b = np.array([1, 1, 0, 0, 1], np.uint8)
    ca, cb = client.encrypt(keys, bits_a, 1), client.encrypt(keys, bits_b, 2)
    out = sharded_gate_batch(lambda g, a, b, c: orc.gate_batch(ok, g, a, b, c, nthreads=2, count=a.shape[0]),
                             Gate.NAND, ca, cb)
    dec, _ = client.decrypt(keys, out)
    single = orc.gate(ok, Gate.NAND, ca[4], cb[4])     # last row belongs to rank 1
    # a levelized netlist over both ranks: three 2-bit adders (the reference's full_adder_MUX gate sequence), every
    # level split between the ranks, output rows all-gathered per level
    import torch
    from tfhe_amd.netlist import Netlist, full_adder_mux
    from tfhe_amd.parallel import run_levels_sharded
    nl = Netlist()
    x, y = nl.input("x", 2), nl.input("y", 2)
    nl.output("out", full_adder_mux(nl, x, y))
    W, xs, ys = 3, [1, 2, 3], [3, 1, 3]
    pool = torch.zeros((W * nl.n_wires, 631), dtype=torch.int32)
    for k in range(W):
        xb = np.array([(xs[k] >> i) & 1 for i in range(2)], np.uint8)
        yb = np.array([(ys[k] >> i) & 1 for i in range(2)], np.uint8)
        pool[k * nl.n_wires + torch.tensor(x)] = torch.from_numpy(client.encrypt(keys, xb, 10 + k))
        pool[k * nl.n_wires + torch.tensor(y)] = torch.from_numpy(client.encrypt(keys, yb, 20 + k))

    def level_fn(gates):            # the oracle stands in for Engine.gate_level_dev (level semantics: read, then write)
        rows = pool.numpy()
        outs = [orc.gate(ok, int(g[0]), rows[g[2]] if g[2] >= 0 else None, rows[g[3]] if g[3] >= 0 else None,
                         rows[g[4]] if g[4] >= 0 else None) for g in gates]
        for g, o in zip(gates, outs):
            rows[g[1]] = o
    run_levels_sharded(level_fn, pool, nl.level_arrays(instances=W))
    sums = []
    for k in range(W):
        bits, _ = client.decrypt(keys, pool[k * nl.n_wires + torch.tensor(nl.outputs["out"])].numpy())
        sums.append(int(sum(int(b) << i for i, b in enumerate(bits))))
    q.put((rank, keys.seed, dec.tolist(), bool(np.array_equal(out[4], single)), out.tobytes(), sums,
           pool.numpy().tobytes()))
    dist.destroy_process_group()


def test_world_size_2_broadcast_shard_gather():
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=600) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, seed0, dec0, same0, raw0, sums0, pool0), (r1, seed1, dec1, same1, raw1, sums1, pool1) = res
    assert (r0, r1) == (0, 1) and seed0 == seed1 == 7
    assert dec0 == dec1 == [1, 0, 1, 1, 0]          # NAND truth table
    assert same0 and same1 and raw0 == raw1          # both ranks hold the identical full result
    assert sums0 == sums1 == [4, 3, 6]               # 1+3, 2+1, 3+3 through the sharded netlist
    assert pool0 == pool1                            # and identical pools, every wire included
