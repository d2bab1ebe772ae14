"""The oracle replays the committed reference outputs (tests/golden/*.npz, produced by oracle/gen_golden.py from the
reference itself) BIT FOR BIT.  This is what pins the oracle on machines where /root/reference does not exist."""
import os

import numpy as np
import pytest

from tfhe_amd import PARAMS as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def prim():
    return np.load(os.path.join(GOLD, "primitives.npz"))


@pytest.fixture(scope="module")
def boot():
    return np.load(os.path.join(GOLD, "bootstrap.npz"))


@pytest.fixture(scope="module")
def gates():
    return np.load(os.path.join(GOLD, "gates.npz"))


def test_transforms(oracle, prim):
    for i in range(6):
        assert np.array_equal(oracle.int_ifft(prim["int_polys"][i]), prim["int_ifft"][i])
        assert np.array_equal(oracle.torus_ifft(prim["torus_polys"][i]), prim["torus_ifft"][i])
        assert np.array_equal(oracle.torus_fft(prim["spectra"][i]), prim["torus_fft"][i])
        assert np.array_equal(oracle.decomp(prim["torus_polys"][i]), prim["decomp"][i])
        assert np.array_equal(oracle.mul_fft(prim["mul_a"][i], prim["mul_b"][i]), prim["mul_fft"][i])
        assert np.array_equal(oracle.mul_naive(prim["mul_a"][i], prim["mul_b"][i]), prim["mul_naive"][i])


def test_rotations_and_modswitch(oracle, prim):
    src = prim["torus_polys"][2]
    for i, a in enumerate(prim["rot_a"]):
        if a != 0:
            assert np.array_equal(oracle.mul_by_xai(int(a), src), prim["xai"][i])
        assert np.array_equal(oracle.mul_by_xai_minus_one(int(a), src), prim["xai_minus_one"][i])
    got = [oracle.modswitch_from(int(p), 2 * P.N) for p in prim["ms_phases"]]
    assert np.array_equal(np.array(got, np.int32), prim["ms_from_2N"])
    for r, ms in enumerate((4, 8)):
        for c, mu in enumerate((-3, -1, 0, 1, 3)):
            assert oracle.modswitch_to(mu, ms) == prim["ms_to"][r, c]


def test_key_conversion(oracle, okeys, boot):
    bk = oracle.bkfft(okeys)
    assert np.array_equal(bk[boot["bkfft_index"]], boot["bkfft_rows"])


def test_stages(oracle, okeys, boot):
    for i in range(4):
        assert np.array_equal(oracle.extern_mul(okeys, boot["extmul_acc"][i], int(boot["extmul_index"][i])),
                              boot["extmul_out"][i])
        assert np.array_equal(oracle.mux_rotate(okeys, boot["extmul_acc"][i], int(boot["extmul_index"][i]),
                                                int(boot["mux_bara"][i])), boot["mux_out"][i])
    for i in range(2):
        got = oracle.blind_rotate(okeys, boot["br_acc"][i], boot["br_bara"][i])
        assert np.array_equal(got, boot["br_out"][i])
        assert np.array_equal(oracle.extract(got), boot["extract"][i])
    for i in range(3):
        assert np.array_equal(oracle.keyswitch(okeys, boot["ks_rand_in"][i]), boot["ks_rand_out"][i])


def test_bootstrap(oracle, okeys, boot, client, keys):
    # the fixture's inputs are what the portable client produces today
    assert np.array_equal(client.encrypt(keys, boot["bs_bits"], seed=77), boot["bs_in"])
    for i in range(2):
        woks = oracle.bootstrap_woks(okeys, boot["bs_in"][i])
        assert np.array_equal(woks, boot["bs_woks"][i])
        assert np.array_equal(oracle.keyswitch(okeys, woks), boot["ks_out"][i])
    assert np.array_equal(boot["ks_out"], boot["bs_out"])
    dec, _ = client.decrypt(keys, boot["bs_out"])
    assert np.array_equal(dec, boot["bs_bits"])


def test_gates(oracle, okeys, gates):
    from tests.truth import TRUTH
    a, b, c = (gates[k].astype(np.int64) for k in ("a_bits", "b_bits", "c_bits"))
    for gate in range(15):
        cnt = 8 if gate >= 10 else 4
        want_bits = TRUTH[gate](a, b, c)
        assert np.array_equal(gates["dec"][gate, :cnt], want_bits[:cnt]), gate
        check = range(cnt) if gate >= 11 else (0, cnt - 1)   # bootstrapped gates: 2 per gate keeps the suite fast
        for i in check:
            got = oracle.gate(okeys, gate, gates["ca"][i], gates["cb"][i], gates["cc"][i])
            assert np.array_equal(got, gates["out"][gate, i]), (gate, i)
