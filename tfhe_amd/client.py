"""Client side (key generation, encryption, decryption): ctypes binding of libtfhe_client.so (csrc/client_keys.c).
Host CPU code; produces the inputs of the accelerated path in the flat layouts of include/tfhe_hip.h."""
import ctypes
import os
from dataclasses import dataclass

import numpy as np

from .engine import PARAMS

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libtfhe_client.so")


@dataclass
class KeySet:
    seed: int
    lwe_key: np.ndarray   # int32[n]      secret
    tlwe_key: np.ndarray  # int32[N]      secret (= extracted LWE_N key)
    bk_coef: np.ndarray   # int32[n][6][2][N]   cloud
    ks: np.ndarray        # int32[N][t][4][n+1] cloud


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


class Client:
    def __init__(self, params=PARAMS):
        self.params = params
        path = LIB_PATH if params.lam == 128 else os.path.join(_HERE, "libtfhe_client%s.so" % params.suffix)
        if not os.path.exists(path):
            raise RuntimeError("%s not built: run `python -m tfhe_amd.build`" % path)
        self.lib = ctypes.CDLL(path)
        if self.lib.tfhe_client_selftest() != 1:
            raise RuntimeError("noise table self check failed: this host's libm rounds erfcl differently (keys would differ)")

    def noise(self, seed, sigma_times_2_40, count):
        """`count` centred noise samples (Torus32) of standard deviation sigma_times_2_40 / 2^40: the law of every noise term
        this client draws (gauss_cdt.h), exposed so that tests can check it against the reference's gaussian32"""
        out = np.zeros(count, np.int32)
        self.lib.tfhe_client_noise(ctypes.c_uint64(seed), ctypes.c_int64(sigma_times_2_40), ctypes.c_int64(count), _p(out))
        return out

    def noise_from_words(self, w_mag, w_pos, sigma_times_2_40):
        self.lib.tfhe_client_noise_from_words.restype = ctypes.c_int32
        return self.lib.tfhe_client_noise_from_words(ctypes.c_uint64(w_mag), ctypes.c_uint64(w_pos), ctypes.c_int64(sigma_times_2_40))

    def keygen(self, seed=42):
        P = self.params
        lk = np.zeros(P.n, np.int32)
        tk = np.zeros(P.N, np.int32)
        bk = np.zeros((P.n, P.kpl, 2, P.N), np.int32)
        ks = np.zeros((P.N, P.ks_t, P.ks_base, P.n + 1), np.int32)
        self.lib.tfhe_client_keygen(ctypes.c_uint64(seed), _p(lk), _p(tk), _p(bk), _p(ks))
        return KeySet(seed, lk, tk, bk, ks)

    def encrypt(self, keys, bits, seed=1):
        bits = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
        out = np.zeros((bits.shape[0], self.params.n + 1), np.int32)
        self.lib.tfhe_client_encrypt(ctypes.c_uint64(seed), _p(keys.lwe_key), _p(bits), ctypes.c_int64(bits.shape[0]),
                                     _p(out))
        return out

    def decrypt(self, keys, cts, extracted=False):
        """returns (bits, phases); extracted=True decrypts LWE_N samples under the TLWE key"""
        key = keys.tlwe_key if extracted else keys.lwe_key
        n = key.shape[0]
        cts = np.ascontiguousarray(cts, dtype=np.int32).reshape(-1, n + 1)
        bits = np.zeros(cts.shape[0], np.uint8)
        phases = np.zeros(cts.shape[0], np.int32)
        self.lib.tfhe_client_decrypt(_p(key), ctypes.c_int32(n), _p(cts), ctypes.c_int64(cts.shape[0]), _p(bits),
                                     _p(phases))
        return bits, phases
