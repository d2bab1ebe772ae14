"""ctypes binding of oracle/liboracle.so (the CPU restatement; TEST INFRASTRUCTURE) and, where it exists, of
oracle/_ref/libtfhe_ref.so (the reference itself; authoring container only)."""
import ctypes
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_SO = os.path.join(ROOT, "oracle", "liboracle.so")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libtfhe_ref.so")
n, N, KPL, T, BASE = 630, 1024, 6, 8, 4


class _Ptr(ctypes.c_void_p):
    """pointer that keeps its array alive (a converted temporary would otherwise be freed before the call)"""


def P(a):
    if a is None:
        return None
    p = _Ptr(a.ctypes.data)
    p._keep = a
    return p


def i32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.int32)


class Oracle:
    def __init__(self, params=None):
        """params: a tfhe_amd.ParamSet; None = the default 128-bit set.  The 80-bit oracle is pinned per stage against
        the reference (tests/golden/stages_80.npz); its composed bootstrap is unpinned (the reference hard-codes the
        default set in its composed hot path, src/include/tfhe_core.h:13-24)."""
        self.n, self.kpl, self.l = (630, 6, 3) if params is None else (params.n, params.kpl, params.l)
        so = ORACLE_SO if params is None or params.lam == 128 else ORACLE_SO.replace(".so", "%s.so" % params.suffix)
        self.lib = ctypes.CDLL(so)
        self.lib.oracle_keys_from_coef.restype = ctypes.c_void_p
        self.lib.oracle_keys_from_fft.restype = ctypes.c_void_p
        self.lib.oracle_trig_table_hash.restype = ctypes.c_uint64

    # keys
    def keys_from_coef(self, bk_coef, ks):
        return ctypes.c_void_p(self.lib.oracle_keys_from_coef(P(i32(bk_coef)), P(i32(ks))))

    def keys_from_fft(self, bkfft, ks):
        return ctypes.c_void_p(self.lib.oracle_keys_from_fft(P(np.ascontiguousarray(bkfft, np.float64)), P(i32(ks))))

    def bkfft(self, k):
        out = np.empty((self.n, self.kpl, 2, N // 2, 2), np.float64)
        self.lib.oracle_keys_get_bkfft(k, P(out))
        return out

    # primitives
    def int_ifft(self, poly):
        out = np.empty((N // 2, 2), np.float64)
        self.lib.oracle_IntPolynomial_ifft(P(out), P(i32(poly)))
        return out

    def torus_ifft(self, poly):
        out = np.empty((N // 2, 2), np.float64)
        self.lib.oracle_TorusPolynomial_ifft(P(out), P(i32(poly)))
        return out

    def torus_fft(self, spec):
        out = np.empty(N, np.int32)
        self.lib.oracle_TorusPolynomial_fft(P(out), P(np.ascontiguousarray(spec, np.float64)))
        return out

    def decomp(self, poly):
        out = np.empty((self.l, N), np.int32)
        self.lib.oracle_decompH(P(out), P(i32(poly)))
        return out

    def mul_fft(self, ip, tp):
        out = np.empty(N, np.int32)
        self.lib.oracle_mulFFT(P(out), P(i32(ip)), P(i32(tp)))
        return out

    def mul_naive(self, ip, tp):
        ip, tp = i32(ip), i32(tp)
        out = np.empty(ip.shape[0], np.int32)
        self.lib.oracle_mulNaive(P(out), P(ip), P(tp), ctypes.c_int32(ip.shape[0]))
        return out

    def mul_by_xai(self, a, poly):
        out = np.empty(N, np.int32)
        self.lib.oracle_mulByXai(P(out), ctypes.c_int32(a), P(i32(poly)))
        return out

    def mul_by_xai_minus_one(self, a, poly):
        out = np.empty(N, np.int32)
        self.lib.oracle_mulByXaiMinusOne(P(out), ctypes.c_int32(a), P(i32(poly)))
        return out

    def modswitch_from(self, phase, msize):
        return self.lib.oracle_modSwitchFromTorus32(ctypes.c_int32(phase), ctypes.c_int32(msize))

    def modswitch_to(self, mu, msize):
        return self.lib.oracle_modSwitchToTorus32(ctypes.c_int32(mu), ctypes.c_int32(msize))

    # stages
    def extern_mul(self, k, acc, i):
        acc = i32(acc).copy()
        self.lib.oracle_externMul(k, P(acc), ctypes.c_int32(i))
        return acc

    def extern_mul_fft(self, acc, gsw_fft):
        acc = i32(acc).copy()
        self.lib.oracle_externMulFFT(P(acc), P(np.ascontiguousarray(gsw_fft, np.float64)))
        return acc

    def tgsw_extern_mul(self, acc, tgsw):
        acc = i32(acc).copy()
        self.lib.oracle_tgswExternMulToTLwe(P(acc), P(i32(tgsw)))
        return acc

    def tlwe_addmul_r(self, result, poly, sample):
        res = i32(result).copy()
        self.lib.oracle_tLweAddMulRTo(P(res), P(i32(poly)), P(i32(sample)))
        return res

    def tgsw_to_fft(self, tgsw):
        out = np.empty((self.kpl, 2, N // 2, 2), np.float64)
        self.lib.oracle_tgsw_to_fft(P(out), P(i32(tgsw)))
        return out

    def mux_rotate(self, k, acc, i, barai):
        out = np.empty((2, N), np.int32)
        self.lib.oracle_muxRotate(k, P(out), P(i32(acc)), ctypes.c_int32(i), ctypes.c_int32(barai))
        return out

    def blind_rotate(self, k, acc, bara):
        acc = i32(acc).copy()
        self.lib.oracle_blindRotate(k, P(acc), P(i32(bara)))
        return acc

    def extract(self, acc):
        out = np.empty(N + 1, np.int32)
        self.lib.oracle_extract(P(out), P(i32(acc)))
        return out

    def bootstrap_woks(self, k, x, mu=1 << 29):
        out = np.empty(N + 1, np.int32)
        self.lib.oracle_bootstrap_woKS(k, P(out), P(i32(x)), ctypes.c_int32(mu))
        return out

    def keyswitch(self, k, u):
        out = np.empty(self.n + 1, np.int32)
        self.lib.oracle_keyswitch(k, P(out), P(i32(u)))
        return out

    def bootstrap(self, k, x, mu=1 << 29):
        out = np.empty(self.n + 1, np.int32)
        self.lib.oracle_bootstrap(k, P(out), P(i32(x)), ctypes.c_int32(mu))
        return out

    def gate(self, k, gate, a=None, b=None, c=None):
        out = np.empty(self.n + 1, np.int32)
        self.lib.oracle_gate(k, ctypes.c_int32(int(gate)), P(out), P(i32(a)), P(i32(b)), P(i32(c)))
        return out

    def gate_batch(self, k, gate, a=None, b=None, c=None, nthreads=8, count=None):
        a, b, c = i32(a), i32(b), i32(c)
        if count is None:
            count = a.shape[0]
        out = np.empty((count, self.n + 1), np.int32)
        self.lib.oracle_gate_batch(k, ctypes.c_int32(int(gate)), P(out), P(a), P(b), P(c), ctypes.c_int64(count),
                                   ctypes.c_int32(nthreads))
        return out

    # run-time N, k, l, Bgbit (exact products)
    def decomp_n(self, poly, n_, l, bgbit):
        poly = i32(poly)
        out = np.empty((l, n_), np.int32)
        self.lib.oracle_decompH_n(P(out), P(poly), ctypes.c_int32(n_), ctypes.c_int32(l), ctypes.c_int32(bgbit))
        return out

    def tlwe_addmul_r_n(self, result, poly, sample, n_, k):
        res = i32(result).copy()
        self.lib.oracle_tLweAddMulRTo_n(P(res), P(i32(poly)), P(i32(sample)), ctypes.c_int32(n_), ctypes.c_int32(k))
        return res

    def tgsw_extern_mul_n(self, acc, tgsw, n_, k, l, bgbit):
        acc = i32(acc).copy()
        self.lib.oracle_tgswExternMulToTLwe_n(P(acc), P(i32(tgsw)), ctypes.c_int32(n_), ctypes.c_int32(k), ctypes.c_int32(l),
                                              ctypes.c_int32(bgbit))
        return acc

    def phase(self, ct, key):
        ct, key = i32(ct), i32(key)
        self.lib.oracle_lwePhase.restype = ctypes.c_int32
        return self.lib.oracle_lwePhase(P(ct), P(key), ctypes.c_int32(key.shape[0]))


def have_ref():
    return os.path.exists(REF_SO)
