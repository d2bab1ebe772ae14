"""rate of the struct-level path (bootsNAND_batch on LweSample arrays) for INTEGRATION.md; usage: compat_rate.py [B]"""
import ctypes, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfhe_amd import Client, Engine
from tests.oracle_lib import P
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
client = Client(); keys = client.keygen(42)
eng = Engine(0); eng.load_keys_coef(keys.bk_coef, keys.ks); bkfft = eng.get_bkfft(); eng.close()
lib = ctypes.CDLL(os.path.join(ROOT, "tests", "compat", "libcompat_harness.so"))
lib.harness_make_keyset.restype = ctypes.c_void_p
lib.harness_time_nand.restype = ctypes.c_double
h = ctypes.c_void_p(lib.harness_make_keyset(P(np.ascontiguousarray(bkfft)), P(np.ascontiguousarray(keys.ks))))
rng = np.random.default_rng(1)
a, b = rng.integers(0, 2, B).astype(np.uint8), rng.integers(0, 2, B).astype(np.uint8)
ca, cb = client.encrypt(keys, a, 1), client.encrypt(keys, b, 2)
out = np.zeros((B, 631), np.int32)
ms = lib.harness_time_nand(h, P(ca), P(cb), P(out), B, 3)
assert np.array_equal(client.decrypt(keys, out)[0], 1 - (a & b))
print("struct-level bootsNAND_batch, B=%d: %.1f ms per call = %.0f gates/s" % (B, ms, B / ms * 1e3))
