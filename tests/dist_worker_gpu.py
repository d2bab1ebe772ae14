"""Worker of tests/test_configs_gpu.py::test_two_ranks_on_one_gpu - launched by torch.distributed.run with two ranks
that BOTH open Engine(0) (the one GPU of the test box); collectives over gloo on host tensors.  Runs the sharded gate
batch and the sharded levelized netlist with Engine.gate_level_dev as the per-rank compute and writes what each rank
ends with to <outdir>/rank<r>.npz for the parent test to compare with the single-process result."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(outdir):
    import torch
    import torch.distributed as dist
    from tfhe_amd import Client, Engine, Gate, PARAMS as P
    from tfhe_amd.netlist import Netlist, full_adder_mux
    from tfhe_amd.parallel import broadcast_keys_device, broadcast_secret, run_levels_sharded, sharded_gate_batch
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo")
    client = Client()
    # rank 0 loads and converts the cloud keys; rank 1 receives only the secret part (to encrypt its inputs) and the
    # three device-resident key buffers, and never sees the coefficient-domain keys
    keys = broadcast_secret(client.keygen(42) if rank == 0 else None)
    eng = Engine(0)
    if rank == 0:
        eng.load_keys_coef(keys.bk_coef, keys.ks)
    else:
        assert keys.bk_coef is None and keys.ks is None
    broadcast_keys_device(eng)
    # a batch of independent gates, contiguous shards, all-gathered
    rng = np.random.default_rng(31)
    count = 1537
    bits = rng.integers(0, 2, (2, count)).astype(np.uint8)
    ca, cb = client.encrypt(keys, bits[0], 501), client.encrypt(keys, bits[1], 502)
    out = sharded_gate_batch(lambda g, a, b, c: eng.gate_batch(g, a, b, c), Gate.XOR, ca, cb)
    # a levelized netlist over a device-resident pool, every level split between the ranks
    nbits, W = 4, 9
    nl = Netlist()
    x, y = nl.input("x", nbits), nl.input("y", nbits)
    nl.output("out", full_adder_mux(nl, x, y))
    xs, ys = rng.integers(0, 1 << nbits, W), rng.integers(0, 1 << nbits, W)
    host = np.zeros((W * nl.n_wires, P.n + 1), np.int32)
    for k in range(W):
        xb = ((xs[k] >> np.arange(nbits)) & 1).astype(np.uint8)
        yb = ((ys[k] >> np.arange(nbits)) & 1).astype(np.uint8)
        host[k * nl.n_wires + np.array(x)] = client.encrypt(keys, xb, 600 + k)
        host[k * nl.n_wires + np.array(y)] = client.encrypt(keys, yb, 700 + k)
    pool = torch.from_numpy(host).to(dev)
    rows = pool.shape[0]
    run_levels_sharded(lambda lv: eng.gate_level_dev(pool, rows, lv), pool, nl.level_arrays(instances=W))
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), out=out, pool=pool.cpu().numpy(), xs=xs, ys=ys,
             outputs=np.array(nl.outputs["out"]), n_wires=nl.n_wires)
    eng.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
