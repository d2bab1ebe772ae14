"""Multi-GPU host logic: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the GPU box,
"gloo" in CPU tests).  The reference has no parallel code at all; what shards here is what is
independent in the reference: the gates of one batch / one netlist level.

  * keys are generated once and REPLICATED: rank 0 loads and converts them, then its three device-resident key buffers
    (62 MB FFT key, 62 + 84 MB keyswitch key) are broadcast straight into the other ranks' engines over RCCL
    (broadcast_keys_device; broadcast_keys is the host-side form that the CPU tests use),
  * gates shard in contiguous blocks, no collective on the data path,
  * results are all-gathered only when the caller needs the whole array on every rank (next level of a netlist).
"""
import os

import numpy as np

from .client import KeySet
from .engine import PARAMS as P


def shard_bounds(count, rank, world):
    """contiguous block [lo, hi) of `count` items owned by `rank`; sizes differ by at most one"""
    base, rem = divmod(int(count), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _dist():
    import torch.distributed as dist
    return dist


def broadcast_keys(keys, device=None, src=0, params=None):
    """rank `src` passes its KeySet, the others pass None; returns the KeySet on every rank"""
    P = params if params is not None else globals()["P"]
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return keys
    shapes = {"lwe_key": (P.n,), "tlwe_key": (P.N,), "bk_coef": (P.n, P.kpl, 2, P.N),
              "ks": (P.N, P.ks_t, P.ks_base, P.n + 1)}
    dev = device if device is not None else torch.device("cpu")
    seed = torch.tensor([keys.seed if dist.get_rank() == src else 0], dtype=torch.int64, device=dev)
    dist.broadcast(seed, src=src)
    parts = {}
    for name, shp in shapes.items():
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(getattr(keys, name))).to(dev)
        else:
            t = torch.empty(shp, dtype=torch.int32, device=dev)
        dist.broadcast(t, src=src)
        parts[name] = t.cpu().numpy()
    return KeySet(int(seed.item()), **parts)


class _DevBytes:
    """zero-copy view of library-owned device memory for torch.as_tensor (uint8[nbytes])"""

    def __init__(self, ptr, nbytes):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def broadcast_secret(keys, src=0, params=None, device=None):
    """the small secret part of a KeySet (seed, LWE and TLWE keys) to every rank; cloud keys stay None elsewhere"""
    P = params if params is not None else globals()["P"]
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return keys
    dev = device if device is not None else torch.device("cpu")
    t = torch.zeros(1 + P.n + P.N, dtype=torch.int64, device=dev)
    if dist.get_rank() == src:
        t[0] = keys.seed
        t[1:1 + P.n] = torch.from_numpy(keys.lwe_key.astype(np.int64)).to(dev)
        t[1 + P.n:] = torch.from_numpy(keys.tlwe_key.astype(np.int64)).to(dev)
    dist.broadcast(t, src=src)
    if dist.get_rank() == src:
        return keys
    h = t.cpu().numpy()
    return KeySet(int(h[0]), h[1:1 + P.n].astype(np.int32), h[1 + P.n:].astype(np.int32), None, None)


def broadcast_keys_device(engine, src=0, verify=False, device=None):
    """replicates the cloud keys of rank `src`'s engine (already loaded there) into every other rank's engine: one RCCL
    broadcast per device-resident key buffer, no host copy and no second conversion (tfhe_hip_keys_device_buffers /
    tfhe_hip_keys_adopt).  With a backend that takes no device tensors (gloo, the one-GPU rehearsal) the same buffers
    are staged through the host.
    verify=True: every rank fingerprints its three buffers on its GPU (tfhe_hip_keys_fingerprint) and the ranks compare
    before anyone adopts; an exception on any rank or a mismatch makes ALL ranks return "host" without adopting (the
    caller then distributes the keys through host memory: broadcast_keys + load_keys_coef).  Returns "rccl" when the
    buffers travelled as device tensors, "gloo-staged" when through host tensors, "host" on failure, None if not
    distributed."""
    import torch
    dist = _dist()
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return None
    on_device = dist.get_backend() == "nccl"
    cdev = device if device is not None else (torch.device("cuda", torch.cuda.current_device()) if on_device
                                              else torch.device("cpu"))
    ok, err = 1, None
    try:
        engine.sync()                                    # the engine's stream is not torch's
        for ptr, nbytes in engine.key_buffers():
            t = torch.as_tensor(_DevBytes(ptr, nbytes), device=torch.device("cuda", torch.cuda.current_device()))
            if on_device:
                dist.broadcast(t, src=src)
            else:
                h = t.cpu()
                dist.broadcast(h, src=src)
                if dist.get_rank() != src:
                    t.copy_(h)
            # failure injection for the tests: the named rank damages what it received, so that the verification below
            # has something to find (TFHE_TEST_CORRUPT_KEYS_RANK; never set in production)
            if os.environ.get("TFHE_TEST_CORRUPT_KEYS_RANK") == str(dist.get_rank()) and dist.get_rank() != src:
                t[:8] ^= 0x55
        torch.cuda.synchronize()
    except Exception as e:                               # noqa: BLE001 - any failure here means "use the host path"
        if not verify:
            raise
        ok, err = 0, e
    if verify:
        prints = engine.key_fingerprints() if ok else [0, 0, 0]
        mine = torch.tensor([ok] + [p - (1 << 64) if p >= (1 << 63) else p for p in prints], dtype=torch.int64, device=cdev)
        parts = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(parts, mine)
        rows = [p.cpu().tolist() for p in parts]
        if any(r[0] != 1 for r in rows) or any(r[1:] != rows[src][1:] for r in rows):
            if dist.get_rank() == src:
                import sys
                print("[broadcast_keys_device] device-side key broadcast failed (%r; fingerprints %r): falling back to "
                      "the host path" % (err, rows), file=sys.stderr, flush=True)
            return "host"
    if dist.get_rank() != src:
        engine.adopt_keys()
    return "rccl" if on_device else "gloo-staged"


def all_gather_rows(local, count, device=None):
    """local: int32 array [hi-lo, width] of this rank's shard (shard_bounds order) -> full [count, width] everywhere"""
    import torch
    dist = _dist()
    local = np.ascontiguousarray(local, dtype=np.int32)
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world, width = dist.get_world_size(), local.shape[1]
    dev = device if device is not None else torch.device("cpu")
    cap = -(-int(count) // world)
    buf = torch.zeros((cap, width), dtype=torch.int32, device=dev)
    buf[:local.shape[0]] = torch.from_numpy(local).to(dev)
    parts = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(parts, buf)
    out = np.empty((count, width), np.int32)
    for r in range(world):
        lo, hi = shard_bounds(count, r, world)
        out[lo:hi] = parts[r][:hi - lo].cpu().numpy()
    return out


def sharded_gate_batch(gate_fn, gate, a, b=None, c=None, gather=True, device=None):
    """evaluates one batch of independent gates over all ranks.  gate_fn(gate, a, b, c) -> outputs for the rows it is
    given (Engine.gate_batch on the GPU box).  Every rank holds the full inputs; returns the full outputs (gather) or
    the local shard."""
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    count = a.shape[0]
    lo, hi = shard_bounds(count, rank, world)
    sl = slice(lo, hi)
    local = gate_fn(gate, a[sl], None if b is None else b[sl], None if c is None else c[sl])
    return all_gather_rows(local, count, device) if gather else local


def run_levels_sharded(level_fn, pool, levels):
    """Levelized netlist over all ranks (BASELINE configs[3]).  Every rank holds the whole ciphertext pool (`pool`: torch
    int32 tensor [rows, n+1], on this rank's GPU with the nccl backend, on the CPU with gloo); per level each rank
    evaluates its contiguous block of the level's gates on its own pool - level_fn(gates int32[k][5]), i.e.
    Engine.gate_level_dev on the GPU box - and the output rows are all-gathered (one collective per level,
    2,524 B per gate) so that every pool is complete before the next level.  Returns the number of gates evaluated."""
    import torch
    dist = _dist()
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_initialized() else (0, 1)
    total = 0

    def order_streams():
        # STREAM CONTRACT of every *_dev entry point: the engine launches on its own non-blocking stream
        # (tfhe_hip_stream), which is not ordered with torch's current stream nor with the collective's.  Whatever torch
        # wrote into `pool` (the scatter of the gathered rows below, the caller's input rows) must have landed before
        # level_fn reads it; Engine.gate_level_dev itself returns only after its level has completed.
        if pool.is_cuda:
            torch.cuda.current_stream(pool.device).synchronize()

    for lv in levels:
        lv = np.ascontiguousarray(lv, dtype=np.int32).reshape(-1, 5)
        count = lv.shape[0]
        total += count
        lo, hi = shard_bounds(count, rank, world)
        order_streams()
        if hi > lo:
            level_fn(lv[lo:hi])
        if world == 1:
            continue
        cap = -(-count // world)
        buf = torch.zeros((cap, pool.shape[1]), dtype=pool.dtype, device=pool.device)
        if hi > lo:
            buf[:hi - lo] = pool[torch.from_numpy(lv[lo:hi, 1].astype(np.int64)).to(pool.device)]
        parts = [torch.empty_like(buf) for _ in range(world)]
        dist.all_gather(parts, buf)
        for r in range(world):
            rlo, rhi = shard_bounds(count, r, world)
            if r != rank and rhi > rlo:
                pool[torch.from_numpy(lv[rlo:rhi, 1].astype(np.int64)).to(pool.device)] = parts[r][:rhi - rlo]
    order_streams()
    return total
