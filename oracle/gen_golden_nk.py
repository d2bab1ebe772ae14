#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - tests/golden/nk.npz: what the reference itself computes at other parameters than its compiled-in
set, from oracle/_ref/libtfhe_ref.so:
  refn_decomp      Torus32PolynomialDecompH_old (tgsw-functions.cpp:277-293) with TGswParams(l, Bgbit) of
                   src/test/tgsw_fft_test.cpp:28-35 at N = 512 and 1024 (its structs hold 1024 coefficients)
  refn_mul_naive   torusPolynomialMultNaive_aux (multiplication.cpp:31-43) at N = 512, 1024, 2048, operands as
                   src/test/test-multiplication.cpp:41-44
Inputs + the reference's outputs; authoring container only.      python oracle/gen_golden_nk.py"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so"))


def P(a):
    return ctypes.c_void_p(a.ctypes.data)


def main():
    rng = np.random.default_rng(5150)
    g = {}
    for n, l, bgbit in ((512, 4, 8), (512, 3, 10), (1024, 3, 10), (1024, 4, 8), (1024, 2, 16)):
        poly = rng.integers(-2**31, 2**31, n).astype(np.int32)
        poly[:4] = [0, -1, 2**31 - 1, -2**31]
        out = np.zeros((l, n), np.int32)
        assert R.refn_decomp(P(out), P(poly), n, l, bgbit) == 0
        g["decomp_%d_%d_%d_in" % (n, l, bgbit)] = poly
        g["decomp_%d_%d_%d_out" % (n, l, bgbit)] = out
    for n in (512, 1024, 2048):
        a = rng.integers(-2 * n, 2 * n, (3, n)).astype(np.int32)
        b = rng.integers(-2**31, 2**31, (3, n)).astype(np.int32)
        a[1, 1:] = 0
        out = np.zeros((3, n), np.int32)
        for i in range(3):
            R.refn_mul_naive(P(out[i]), P(a[i]), P(b[i]), n)
        g["mul_%d_a" % n], g["mul_%d_b" % n], g["mul_%d_out" % n] = a, b, out
    path = os.path.join(ROOT, "tests", "golden", "nk.npz")
    np.savez_compressed(path, **g)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
