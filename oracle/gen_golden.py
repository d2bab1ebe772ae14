#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - generates tests/golden/*.npz from the reference itself (oracle/_ref/libtfhe_ref.so, built
from /root/reference by `make -C oracle ref`).  Runs only in the authoring container; the fixtures it writes are
inputs + the reference's outputs (data, no reference code) and are what pins the oracle where the reference is absent.

    python oracle/gen_golden.py

Keys: the build's portable client keygen, seed 42 (tfhe_amd/csrc/client_keys.c) - regenerated identically by the
tests - installed into the reference's own structs (ref_set_keys), with bkFFT computed BY THE REFERENCE."""
import ctypes
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfhe_amd import Client  # noqa: E402

n, N = 630, 1024
R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so"))
R.ref_phase.restype = ctypes.c_int32
R.ref_modswitch_from.restype = ctypes.c_int32
R.ref_modswitch_to.restype = ctypes.c_int32


def P(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    rng = np.random.default_rng(20250117)

    # ---------------- primitives (no keys) ----------------
    g = {}
    ints = rng.integers(-64, 64, (6, N)).astype(np.int32)
    ints[0] = 0
    ints[1] = -64
    g["int_polys"] = ints
    g["int_ifft"] = np.zeros((6, N // 2, 2))
    for i in range(6):
        R.ref_int_ifft(P(g["int_ifft"][i]), P(ints[i]))
    tors = rng.integers(-2**31, 2**31, (6, N)).astype(np.int32)
    tors[0] = -2**31
    tors[1] = 2**31 - 1
    g["torus_polys"] = tors
    g["torus_ifft"] = np.zeros((6, N // 2, 2))
    for i in range(6):
        R.ref_torus_ifft(P(g["torus_ifft"][i]), P(tors[i]))
    spec = g["torus_ifft"] * rng.integers(1, 90000, (6, 1, 1))
    g["spectra"] = spec
    g["torus_fft"] = np.zeros((6, N), np.int32)
    for i in range(6):
        R.ref_torus_fft(P(g["torus_fft"][i]), P(np.ascontiguousarray(spec[i])))
    g["decomp"] = np.zeros((6, 3, N), np.int32)
    for i in range(6):
        R.ref_decomp(P(g["decomp"][i]), P(tors[i]))
    # reference src/test/test-multiplication.cpp:41-44 distribution
    ma = rng.integers(-2 * N, 2 * N, (6, N)).astype(np.int32)
    mb = rng.integers(-2**31, 2**31, (6, N)).astype(np.int32)
    g["mul_a"], g["mul_b"] = ma, mb
    g["mul_fft"] = np.zeros((6, N), np.int32)
    g["mul_naive"] = np.zeros((6, N), np.int32)
    for i in range(6):
        R.ref_mul_fft(P(g["mul_fft"][i]), P(ma[i]), P(mb[i]))
        R.ref_mul_naive(P(g["mul_naive"][i]), P(ma[i]), P(mb[i]))
    rot_a = np.array([0, 1, 5, 1023, 1024, 1025, 2047, 1500], np.int32)
    g["rot_a"] = rot_a
    g["xai"] = np.zeros((8, N), np.int32)
    g["xai_minus_one"] = np.zeros((8, N), np.int32)
    for i, a in enumerate(rot_a):
        if a != 0:  # torusPolynomialMulByXai is only called with a != 0 (lwe-bootstrapping-functions-fft.cpp:199)
            R.ref_mul_by_xai(P(g["xai"][i]), int(a), P(tors[2]))
        else:
            g["xai"][i] = tors[2]
        R.ref_mul_by_xai_minus_one(P(g["xai_minus_one"][i]), int(a), P(tors[2]))
    phases = np.concatenate([rng.integers(-2**31, 2**31, 50),
                             [0, 1, -1, 2**20 - 1, 2**20, 2**20 + 1, 2**31 - 1, -2**31, -2**20, 2**31 - 2**20,
                              2**31 - 2**20 - 1]]).astype(np.int32)
    g["ms_phases"] = phases
    g["ms_from_2N"] = np.array([R.ref_modswitch_from(int(p), 2 * N) for p in phases], np.int32)
    g["ms_to"] = np.array([[R.ref_modswitch_to(mu, ms) for mu in (-3, -1, 0, 1, 3)] for ms in (4, 8)], np.int32)
    np.savez_compressed(os.path.join(out_dir, "primitives.npz"), **g)

    # ---------------- keyed stages ----------------
    client = Client()
    keys = client.keygen(42)
    R.ref_set_keys(P(keys.lwe_key), P(keys.tlwe_key), P(keys.bk_coef), P(keys.ks))
    b = {"key_seed": np.array([42])}
    bkfft = np.zeros((n, 6, 2, N // 2, 2))
    R.ref_get_bkfft(P(bkfft))
    sel = np.array([0, 313, 629], np.int32)
    b["bkfft_index"] = sel
    b["bkfft_rows"] = bkfft[sel]
    acc = rng.integers(-2**31, 2**31, (4, 2, N)).astype(np.int32)
    idx = np.array([0, 1, 313, 629], np.int32)
    b["extmul_acc"], b["extmul_index"] = acc, idx
    b["extmul_out"] = acc.copy()
    for i in range(4):
        R.ref_extern_mul(P(b["extmul_out"][i]), int(idx[i]))
    bar = np.array([1, 1024, 2047, 700], np.int32)
    b["mux_bara"] = bar
    b["mux_out"] = np.zeros_like(acc)
    for i in range(4):
        R.ref_mux_rotate(P(b["mux_out"][i]), P(acc[i]), int(idx[i]), int(bar[i]))
    bara = rng.integers(0, 2 * N, (2, n)).astype(np.int32)
    bara[1, :9] = 0
    b["br_bara"] = bara
    b["br_acc"] = acc[:2]
    b["br_out"] = acc[:2].copy()
    for i in range(2):
        R.ref_blind_rotate(P(b["br_out"][i]), P(bara[i]))
    b["extract"] = np.zeros((2, N + 1), np.int32)
    for i in range(2):
        R.ref_extract(P(b["extract"][i]), P(b["br_out"][i]))
    bits = np.array([0, 1, 1, 0], np.uint8)
    x = client.encrypt(keys, bits, seed=77)
    b["bs_bits"], b["bs_in"] = bits, x
    b["bs_woks"] = np.zeros((4, N + 1), np.int32)
    b["bs_out"] = np.zeros((4, n + 1), np.int32)
    b["ks_out"] = np.zeros((4, n + 1), np.int32)
    for i in range(4):
        R.ref_bootstrap_woKS(P(b["bs_woks"][i]), P(x[i]), 1 << 29)
        R.ref_bootstrap(P(b["bs_out"][i]), P(x[i]), 1 << 29)
        R.ref_keyswitch(P(b["ks_out"][i]), P(b["bs_woks"][i]))
    u = rng.integers(-2**31, 2**31, (3, N + 1)).astype(np.int32)
    b["ks_rand_in"] = u
    b["ks_rand_out"] = np.zeros((3, n + 1), np.int32)
    for i in range(3):
        R.ref_keyswitch(P(b["ks_rand_out"][i]), P(u[i]))
    np.savez_compressed(os.path.join(out_dir, "bootstrap.npz"), **b)

    # ---------------- gates: full truth tables through the reference's bootsXXX ----------------
    a_bits = np.array([0, 1, 0, 1, 0, 1, 0, 1], np.uint8)
    b_bits = np.array([0, 0, 1, 1, 0, 0, 1, 1], np.uint8)
    c_bits = np.array([0, 0, 0, 0, 1, 1, 1, 1], np.uint8)
    ca, cb, cc = client.encrypt(keys, a_bits, 31), client.encrypt(keys, b_bits, 32), client.encrypt(keys, c_bits, 33)
    gt = {"a_bits": a_bits, "b_bits": b_bits, "c_bits": c_bits, "ca": ca, "cb": cb, "cc": cc}
    outs = np.zeros((15, 8, n + 1), np.int32)
    dec = np.zeros((15, 8), np.uint8)
    for gate in range(15):
        cnt = 8 if gate >= 10 else 4   # binary gates: the 4 (a,b) combinations
        for i in range(cnt):
            R.ref_gate(gate, P(outs[gate, i]), P(ca[i]), P(cb[i]), P(cc[i]))
            dec[gate, i] = 1 if R.ref_phase(P(outs[gate, i])) > 0 else 0
    gt["out"], gt["dec"] = outs, dec
    np.savez_compressed(os.path.join(out_dir, "gates.npz"), **gt)
    for f in ("primitives.npz", "bootstrap.npz", "gates.npz"):
        print(f, os.path.getsize(os.path.join(out_dir, f)), "bytes")


if __name__ == "__main__":
    main()
