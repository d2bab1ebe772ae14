"""Client-side key material (tfhe_amd/csrc/client_keys.c): determinism and the algebraic relations the hot path
relies on (reference lwe-keyswitch-functions.cpp:163-215, lwe-bootstrapping-functions.cpp:185-217)."""
import hashlib

import numpy as np

from tfhe_amd import PARAMS as P


def test_keygen_is_deterministic(client, keys):
    again = client.keygen(42)
    for f in ("lwe_key", "tlwe_key", "bk_coef", "ks"):
        assert np.array_equal(getattr(keys, f), getattr(again, f))
    assert set(np.unique(keys.lwe_key)) == {0, 1} and set(np.unique(keys.tlwe_key)) == {0, 1}
    # fingerprint pins the fixtures in tests/golden to this generator
    h = hashlib.sha256(keys.lwe_key.tobytes() + keys.tlwe_key.tobytes() + keys.bk_coef[:2].tobytes()
                       + keys.ks[:2].tobytes()).hexdigest()
    assert h == open(__file__.replace("test_client.py", "golden/keys_seed42.sha256")).read().strip()


def test_encrypt_decrypt_roundtrip(client, keys):
    bits = np.random.default_rng(3).integers(0, 2, 500).astype(np.uint8)
    cts = client.encrypt(keys, bits, seed=9)
    dec, phases = client.decrypt(keys, cts)
    assert np.array_equal(dec, bits)
    err = phases.astype(np.int64) - np.where(bits == 1, 1 << 29, -(1 << 29))
    assert np.abs(err).max() < 2**32 * 2**-15 * 8          # noise stdev 2^-15
    assert 0.5 < np.std(err) / (2**32 * 2**-15) < 1.5


def test_keyswitch_rows_encrypt_digit_times_key(client, keys):
    # ks[i][j][d] = LWE_n(d * s'_i / 4^(j+1)), d = 0 trivial
    assert not keys.ks[:, :, 0, :].any()
    rng = np.random.default_rng(4)
    for _ in range(200):
        i, j, d = int(rng.integers(P.N)), int(rng.integers(8)), int(rng.integers(1, 4))
        _, ph = client.decrypt(keys, keys.ks[i, j, d][None])
        want = (int(keys.tlwe_key[i]) * d) << (32 - 2 * (j + 1))
        diff = (int(ph[0]) - want + 2**31) % 2**32 - 2**31
        assert abs(diff) < 2**32 * 2**-15 * 8


def test_bootstrapping_key_rows_are_tgsw_of_key_bits(keys, oracle):
    # phase of row p of bk[i] = s_i * h_p on coefficient 0 of its block (+ the -s''*... term for block 0), noise 2^-25
    s2 = keys.tlwe_key
    for i in (0, 17, 629):
        for p in range(6):
            a, b = keys.bk_coef[i, p, 0], keys.bk_coef[i, p, 1]
            phase = (b.astype(np.int64) - oracle.mul_naive(s2, a).astype(np.int64)) % 2**32
            h = 1 << (32 - (p % 3 + 1) * 7)
            msg = np.zeros(P.N, np.int64)
            if p >= 3:
                msg[0] = int(keys.lwe_key[i]) * h
            else:   # block 0: message sits on the mask polynomial: phase = -s'' * (s_i h)
                msg = (-(s2.astype(np.int64)) * int(keys.lwe_key[i]) * h) % 2**32
            diff = (phase - msg + 2**31) % 2**32 - 2**31
            assert np.abs(diff).max() < 2**32 * 2**-25 * 8


def test_noise_law_is_the_references_gaussian(client):
    """gaussian32 (src/libtfhe/numeric-functions.cpp:22-34): normal_distribution(0, sigma), truncated toward zero to a
    Torus32.  The client samples it by integer table inversion (gauss_cdt.h): standard deviation, symmetry, shape
    (kurtosis 3, mass beyond 1 / 2 / 3 sigma) on four million samples; the tail beyond 6 sigma through the inversion itself"""
    from math import erfc, sqrt
    for m, sigma in ((1 << 25, 2.0 ** 17), (1 << 15, 128.0)):          # 2^-15 (LWE, keyswitch) and 2^-25 (bootstrapping key)
        x = client.noise(11, m, 4_000_000).astype(np.float64)
        # truncation toward zero shrinks |x| by 1/2 LSB on average: variance sigma^2 - sigma sqrt(2/pi) + 1/3
        want_std = sqrt(sigma * sigma - sigma * sqrt(2 / np.pi) + 1 / 3)
        assert abs(x.std() / want_std - 1) < 2e-3
        assert abs(x.mean()) < 4 * sigma / sqrt(x.size)
        z = x / sigma
        assert abs((z ** 4).mean() / (z ** 2).mean() ** 2 - 3) < 0.02
        for k in (1, 2, 3, 4):
            p = erfc((np.floor(k * sigma) + 1) / sigma / sqrt(2))     # |trunc(y)| > k sigma  <=>  |y| >= floor(k sigma) + 1
            assert abs((np.abs(x) > k * sigma).mean() - p) < 5 * sqrt(p / x.size) + 2e-5, k
        assert (x > 0).mean() - (x < 0).mean() < 2e-3
    # the tail: uniform words just below 2^64 P(|z| < 6) / at the top of the range land at 6 / beyond 9 sigma
    m, sigma = 1 << 25, 2.0 ** 17
    w6 = int((1 - erfc(6 / sqrt(2))) * 2.0 ** 64)
    assert 5.99 < client.noise_from_words(w6, 0, m) / sigma < 6.01
    assert -6.01 < client.noise_from_words(w6, 1 << 63, m) / sigma < -5.99
    assert 9.1 < client.noise_from_words(2 ** 64 - 1, 0, m) / sigma < 9.2            # where the 64-bit table runs out
    assert client.noise_from_words(0, 0, m) == 0 and abs(client.noise_from_words(0, (1 << 63) - 1, m)) < sigma / 1024 + 1
