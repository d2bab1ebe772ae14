#!/usr/bin/env python3
"""TEST INFRASTRUCTURE - wire-format fixtures written BY THE REFERENCE's own tfhe_io.cpp (oracle/_ref/libtfhe_ref.so).
Authoring container only.  Produces data only:

  tests/golden/wire_params.txt       export_tfheGateBootstrappingParameterSet_toFile of the default parameter set
  tests/golden/wire_ciphertexts.bin  export_gate_bootstrapping_ciphertext_toFile x4 (client encryptions, seed 77)
  tests/golden/wire.json             size, sha256 and text header of export_tfheGateBootstrappingCloudKeySet_toFile for
                                     the seed-42 client keys (the 114 MB file itself is not committed), variances used

    python oracle/gen_golden_wire.py"""
import ctypes
import hashlib
import json
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tfhe_amd import Client  # noqa: E402

R = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so"))


def P(a):
    return None if a is None else ctypes.c_void_p(a.ctypes.data)


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    client = Client()
    keys = client.keygen(42)
    R.ref_set_keys(P(keys.lwe_key), P(keys.tlwe_key), P(keys.bk_coef), P(keys.ks))
    meta = {"key_seed": 42}
    assert R.ref_export_params(os.path.join(out_dir, "wire_params.txt").encode()) == 0
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "cloud.key")
        assert R.ref_export_cloud_keyset(path.encode()) == 0
        data = open(path, "rb").read()
    cut = data.index(b"-----END LWEKSPARAMS-----\n") + len(b"-----END LWEKSPARAMS-----\n")
    meta["cloud_keyset_bytes"] = len(data)
    meta["cloud_keyset_sha256"] = hashlib.sha256(data).hexdigest()
    meta["cloud_keyset_header"] = data[:cut].decode("ascii")
    meta["cloud_keyset_variances"] = [0.0, 0.0]   # ref_set_keys installs the keys with current_variance = 0
    bits = np.array([0, 1, 1, 0], np.uint8)
    x = client.encrypt(keys, bits, seed=77)
    var = np.array([0.0, 2.0 ** -30, 1.5e-9, 0.25])
    assert R.ref_export_ciphertexts(os.path.join(out_dir, "wire_ciphertexts.bin").encode(), P(x), P(var), 4) == 0
    meta["ciphertext_seed"] = 77
    meta["ciphertext_bits"] = bits.tolist()
    meta["ciphertext_variances"] = var.tolist()
    with open(os.path.join(out_dir, "wire.json"), "w") as f:
        json.dump(meta, f, indent=1)
    print(json.dumps({k: v for k, v in meta.items() if k != "cloud_keyset_header"}, indent=1))
    print(meta["cloud_keyset_header"])


if __name__ == "__main__":
    main()
