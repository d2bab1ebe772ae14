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
