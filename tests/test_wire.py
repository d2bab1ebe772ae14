"""Wire formats (include/tfhe_hip_io.h) against files written by the reference's own tfhe_io.cpp.

Fixtures (tests/golden/wire*, made by oracle/gen_golden_wire.py in the authoring container): the reference's parameter
file, four ciphertexts as the reference exports them, and size + sha256 + text header of the reference's cloud-keyset
export of the seed-42 client keys.  The `ref` tests additionally run the reference's reader/writer live."""
import ctypes
import hashlib
import json
import os
import re

import numpy as np
import pytest

from tests.oracle_lib import REF_SO, P, have_ref
from tfhe_amd import TfheHipError, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLD, "wire.json")) as f:
        return json.load(f)


@pytest.fixture(scope="module")
def key_image(keys, meta):
    return wire.write_cloud_keyset(keys.bk_coef, keys.ks, meta["cloud_keyset_variances"])


def test_header_declares_what_the_library_exports():
    from tfhe_amd.engine import load_library
    lib = load_library()
    text = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, "include", "tfhe_hip_io.h")).read(), flags=re.S)
    names = sorted(set(re.findall(r"\b(tfhe_hip_\w+)\s*\(", text)))
    assert names == sorted(wire.EXPORTS)
    for nm in names:
        assert hasattr(lib, nm), nm


def test_params_file_is_the_references_byte_for_byte():
    ref_bytes = open(os.path.join(GOLD, "wire_params.txt"), "rb").read()
    assert wire.write_params() == ref_bytes
    assert wire.read_params(ref_bytes) == len(ref_bytes)
    assert wire.read_params(ref_bytes.replace(b"\n", b"\r\n")) == len(ref_bytes) + ref_bytes.count(b"\n")


def test_other_parameter_sets_are_refused_with_a_reason():
    ref_bytes = open(os.path.join(GOLD, "wire_params.txt"), "rb").read()
    with pytest.raises(TfheHipError, match="n = 500"):
        wire.read_params(ref_bytes.replace(b"n:        630", b"n:        500"))   # the 80-bit set
    with pytest.raises(TfheHipError, match="Bgbit"):
        wire.read_params(ref_bytes.replace(b"Bgbit:          7", b"Bgbit:         10"))
    with pytest.raises(TfheHipError, match="truncated or missing"):
        wire.read_params(ref_bytes[:200])
    with pytest.raises(TfheHipError, match="expected section"):
        wire.read_params(ref_bytes[ref_bytes.index(b"-----BEGIN LWEPARAMS"):])


def test_ciphertexts_written_by_the_reference(client, keys, meta):
    data = open(os.path.join(GOLD, "wire_ciphertexts.bin"), "rb").read()
    assert len(data) == 4 * wire.SAMPLE_BYTES
    samples, var = wire.read_samples(data)
    expect = client.encrypt(keys, np.array(meta["ciphertext_bits"], np.uint8), seed=meta["ciphertext_seed"])
    assert np.array_equal(samples, expect)
    assert np.array_equal(var, np.array(meta["ciphertext_variances"]))
    assert wire.write_samples(samples, var) == data                      # and back, byte for byte
    one, _ = wire.read_samples(data[wire.SAMPLE_BYTES:], count=1)        # a stream is a sequence of samples
    assert np.array_equal(one[0], expect[1])
    empty, _ = wire.read_samples(b"", count=0)
    assert empty.shape == (0, 631)
    with pytest.raises(TfheHipError, match="truncated"):
        wire.read_samples(data[:-1], count=4)
    bad = bytearray(data)
    bad[wire.SAMPLE_BYTES] = 43                                          # LWE_KEY_TYPE_UID where a sample should be
    with pytest.raises(TfheHipError, match="ciphertext 1: wrong type id 43"):
        wire.read_samples(bytes(bad))


def test_cloud_keyset_image_equals_the_references_file(key_image, meta):
    assert key_image.size == meta["cloud_keyset_bytes"]
    head = meta["cloud_keyset_header"].encode()
    assert key_image[:len(head)].tobytes() == head
    assert hashlib.sha256(key_image).hexdigest() == meta["cloud_keyset_sha256"]


def test_cloud_keyset_round_trip_and_errors(key_image, keys, tmp_path):
    bk, ks, var, used = wire.read_cloud_keyset(key_image)
    assert used == key_image.size
    assert np.array_equal(bk, keys.bk_coef) and np.array_equal(ks, keys.ks) and not var.any()
    path = tmp_path / "cloud.key"
    wire.write_cloud_keyset_file(path, keys.bk_coef, keys.ks, [0.5, 0.25])
    bk2, ks2, var2 = wire.read_cloud_keyset_file(path)
    assert np.array_equal(bk2, keys.bk_coef) and np.array_equal(ks2, keys.ks) and var2.tolist() == [0.5, 0.25]
    with pytest.raises(TfheHipError, match="bootstrapping key: truncated"):
        wire.read_cloud_keyset(key_image[:-4])
    with pytest.raises(TfheHipError, match="keyswitch key: truncated"):
        wire.read_cloud_keyset(key_image[:10_000_000])
    head_len = key_image.tobytes().index(b"-----END LWEKSPARAMS-----\n") + 26
    bad = key_image.copy()
    bad[head_len] = 201
    with pytest.raises(TfheHipError, match="keyswitch key: wrong type id 201"):
        wire.read_cloud_keyset(bad)
    with pytest.raises(TfheHipError, match="cannot open"):
        wire.read_cloud_keyset_file(tmp_path / "missing.key")


@pytest.mark.ref
@pytest.mark.skipif(not have_ref(), reason="reference build oracle/_ref absent")
def test_live_against_the_reference_reader_and_writer(keys, client, tmp_path):
    R = ctypes.CDLL(REF_SO)
    # keys generated BY THE REFERENCE (its own RNG and variances), written by the reference, read here
    R.ref_keygen(7)
    lwe, tlwe = np.empty(630, np.int32), np.empty(1024, np.int32)
    bk, ks = np.empty((630, 6, 2, 1024), np.int32), np.empty((1024, 8, 4, 631), np.int32)
    R.ref_get_keys(P(lwe), P(tlwe), P(bk), P(ks))
    path = str(tmp_path / "ref.key")
    assert R.ref_export_cloud_keyset(path.encode()) == 0
    bk_r, ks_r, var = wire.read_cloud_keyset_file(path)
    assert np.array_equal(bk_r, bk) and np.array_equal(ks_r, ks)
    assert var[0] > 0 and var[1] > 0
    # written here with the same variances == the reference's file
    assert wire.write_cloud_keyset(bk, ks, var).tobytes() == open(path, "rb").read()
    # a file written here is a working cloud key for the reference's reader
    mine = str(tmp_path / "mine.key")
    wire.write_cloud_keyset_file(mine, keys.bk_coef, keys.ks)
    bk_i, ks_i = np.empty_like(bk), np.empty_like(ks)
    assert R.ref_import_cloud_keyset(mine.encode(), P(bk_i), P(ks_i)) == 0
    assert np.array_equal(bk_i, keys.bk_coef) and np.array_equal(ks_i, keys.ks)
    a = client.encrypt(keys, np.array([1, 1], np.uint8), seed=5)
    out = np.empty(631, np.int32)
    R.ref_imported_nand(P(out), P(a[0]), P(a[1]))
    bits, _ = client.decrypt(keys, out[None, :])
    assert bits[0] == 0
    # ciphertexts: written here, read by the reference
    ct = str(tmp_path / "ct.bin")
    var_ct = np.array([0.125, 3e-9])
    open(ct, "wb").write(wire.write_samples(a, var_ct))
    back, var_back = np.empty_like(a), np.empty(2)
    assert R.ref_import_ciphertexts(ct.encode(), P(back), P(var_back), 2) == 0
    assert np.array_equal(back, a) and np.array_equal(var_back, var_ct)


@pytest.mark.gpu
def test_gpu_loads_a_cloud_keyset_file_and_computes_the_same_gates(key_image, keys, client, tmp_path):
    """file -> GPU in one call; same ciphertexts as with keys handed over as arrays (tests/golden/gates.npz holds the
    reference's outputs for these inputs)"""
    from tfhe_amd import Engine, Gate
    g = np.load(os.path.join(GOLD, "gates.npz"))
    path = tmp_path / "cloud.key"
    path.write_bytes(key_image.tobytes())
    eng = Engine(0)
    try:
        with pytest.raises(TfheHipError):
            eng.gate_batch(Gate.NAND, g["ca"][:4], g["cb"][:4])          # no keys yet
        wire.load_cloud_keyset_file(eng, path)
        out = eng.gate_batch(Gate.NAND, g["ca"][:4], g["cb"][:4])
        assert np.array_equal(out, g["out"][int(Gate.NAND), :4])
        with pytest.raises(TfheHipError, match="truncated"):
            wire.load_cloud_keyset_buffer(eng, key_image[:1000])
        wire.load_cloud_keyset_buffer(eng, key_image)
        out = eng.gate_batch(Gate.XOR, g["ca"][:4], g["cb"][:4])
        assert np.array_equal(out, g["out"][int(Gate.XOR), :4])
    finally:
        eng.close()
