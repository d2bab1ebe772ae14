#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - tests/golden/leveled.npz: the reference's leveled TGSW / TLWE operations on arbitrary operands
(SURVEY.md 8(f) N4), default parameter set, from oracle/_ref/libtfhe_ref.so (functions ref_tgsw_extern_mul,
ref_tgsw_fft_extern_mul, ref_tlwe_addmul_r of oracle/ref_harness.cpp = tGswExternMulToTLwe tgsw-functions.cpp:157-170,
tGswToFFTConvert + tGswFFTExternMulToTLwe tgsw-fft-operations.cpp:52-57,92-120, tLweAddMulRTo tlwe-functions.cpp:191-197).
Inputs + the reference's outputs; authoring container only.      python oracle/gen_golden_leveled.py"""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N, KPL = 1024, 6
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so"))


def P(a):
    return ctypes.c_void_p(a.ctypes.data)


def main():
    rng = np.random.default_rng(4242)
    g = {}
    tgsw = rng.integers(-2**31, 2**31, (2, KPL, 2, N)).astype(np.int32)     # TGSW samples look uniform on the torus
    acc = rng.integers(-2**31, 2**31, (3, 2, N)).astype(np.int32)
    idx = np.array([0, 1, 0], np.int32)
    g["tgsw"], g["acc"], g["index"] = tgsw, acc, idx
    g["extern_mul_out"] = acc.copy()
    g["fft_extern_mul_out"] = acc.copy()
    g["gsw_fft"] = np.zeros((2, KPL, 2, N // 2, 2))
    for i in range(3):
        R.ref_tgsw_extern_mul(P(g["extern_mul_out"][i]), P(np.ascontiguousarray(tgsw[idx[i]])))
        R.ref_tgsw_fft_extern_mul(P(g["fft_extern_mul_out"][i]), P(g["gsw_fft"][idx[i]]), P(np.ascontiguousarray(tgsw[idx[i]])))
    poly = rng.integers(-512, 512, (3, N)).astype(np.int32)
    poly[0] = 0
    res = rng.integers(-2**31, 2**31, (3, 2, N)).astype(np.int32)
    g["addmul_poly"], g["addmul_sample"], g["addmul_in"] = poly, acc, res
    g["addmul_out"] = res.copy()
    for i in range(3):
        R.ref_tlwe_addmul_r(P(g["addmul_out"][i]), P(poly[i]), P(np.ascontiguousarray(acc[i])))
    out = os.path.join(ROOT, "tests", "golden", "leveled.npz")
    np.savez_compressed(out, **g)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
