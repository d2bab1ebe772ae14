import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.getcwd())
import torch
from tfhe_amd import Client, Engine, Gate
from tfhe_amd.engine import _p
B = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
client = Client(); keys = client.keygen(42)
torch.cuda.set_device(0)
eng = Engine(0); eng.load_keys_coef(keys.bk_coef, keys.ks)
rng = np.random.default_rng(1)
ba, bb = rng.integers(0, 2, B).astype(np.uint8), rng.integers(0, 2, B).astype(np.uint8)
ca, cb = client.encrypt(keys, ba, seed=11), client.encrypt(keys, bb, seed=12)
for pinned in (False, True):
    if pinned:
        ta = torch.empty(ca.shape, dtype=torch.int32, pin_memory=True); ta.numpy()[:] = ca
        tb = torch.empty(cb.shape, dtype=torch.int32, pin_memory=True); tb.numpy()[:] = cb
        to = torch.empty(ca.shape, dtype=torch.int32, pin_memory=True)
        a, b, o = ta.numpy(), tb.numpy(), to.numpy()
    else:
        a, b, o = ca, cb, np.empty_like(ca)
    def call():
        rc = eng.lib.tfhe_hip_gate_batch(eng.ctx, 0, _p(o), _p(a), _p(b), None, ctypes.c_int64(B))
        assert rc == 0
    call()
    t0 = time.perf_counter()
    for _ in range(3): call()
    dt = (time.perf_counter() - t0) / 3
    assert np.array_equal(client.decrypt(keys, o)[0], 1 - (ba & bb))
    print("pinned" if pinned else "pageable", "B=%d: %.1f ms per call = %.0f gates/s" % (B, dt * 1e3, B / dt), flush=True)
