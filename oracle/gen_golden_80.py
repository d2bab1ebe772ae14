#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - tests/golden/stages_80.npz: the 80-bit parameter set (n = 500, l = 2, Bgbit = 10) pinned STAGE BY
STAGE against the reference itself (oracle/_ref/libtfhe_ref.so, functions ref80_* of oracle/ref_harness.cpp).  The fork's
composed bootstrap hard-codes the default set, but lweKeySwitch, the generic gadget decomposition
Torus32PolynomialDecompH_old, tGswToFFTConvert and the transform-domain pieces of tGswFFTExternMulToTLwe take their
parameters at run time; the fixtures hold inputs and the reference's outputs of exactly those.  Runs only in the
authoring container.        python oracle/gen_golden_80.py

Keys: the portable client keygen for the 80-bit set, seed 42 (regenerated identically by the tests)."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfhe_amd import Client, PARAMS_80 as P80  # noqa: E402

N = 1024
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so"))


def P(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def main():
    sizes = np.zeros(8, np.int32)
    R.ref80_get_sizes(P(sizes))
    assert sizes.tolist() == [P80.n, N, 1, P80.l, P80.Bgbit, P80.kpl, 8, 2], sizes
    rng = np.random.default_rng(80)
    g = {"key_seed": np.array([42])}
    # gadget decomposition, l = 2, Bgbit = 10 (tgsw-functions.cpp:277-293; the check of src/test/test-decomp-tgsw.cpp:72-79
    # is repeated by the test that replays this)
    polys = rng.integers(-2**31, 2**31, (6, N)).astype(np.int32)
    polys[0, :6] = [0, -1, 2**31 - 1, -2**31, 2**11, -2**11]
    g["decomp_in"] = polys
    g["decomp_out"] = np.zeros((6, P80.l, N), np.int32)
    for i in range(6):
        R.ref80_decomp(P(g["decomp_out"][i]), P(polys[i]))
    keys = Client(P80).keygen(42)
    # keyswitch LWE_1024 -> LWE_500 (lwe-keyswitch-functions.cpp:114-129,228-238)
    u = rng.integers(-2**31, 2**31, (70, N + 1)).astype(np.int32)
    u[0, :10] = 0
    g["ks_in"] = u
    g["ks_out"] = np.zeros((70, P80.n + 1), np.int32)
    R.ref80_keyswitch(P(g["ks_out"]), P(u), P(np.ascontiguousarray(keys.ks)), 70)
    # key conversion of selected TGSW samples and the external product with them (tgsw-fft-operations.cpp:52-57,92-120)
    idx = np.array([0, 1, 250, 499], np.int32)
    acc = rng.integers(-2**31, 2**31, (4, 2, N)).astype(np.int32)
    g["extmul_index"], g["extmul_acc"] = idx, acc
    g["extmul_out"] = acc.copy()
    g["bkfft_rows"] = np.zeros((4, P80.kpl, 2, N // 2, 2))
    for i in range(4):
        R.ref80_extern_mul(P(g["extmul_out"][i]), P(g["bkfft_rows"][i]), P(np.ascontiguousarray(keys.bk_coef[idx[i]])))
    out = os.path.join(ROOT, "tests", "golden", "stages_80.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
