"""ctypes binding of libtfhe_hip.so (include/tfhe_hip.h).  No CPU fallback: if the library or a GPU is missing,
constructing an Engine raises."""
import ctypes
import enum
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("TFHE_HIP_LIB") or os.path.join(_HERE, "libtfhe_hip.so")  # override: A/B builds


class ParamSet:
    """a gate-bootstrapping parameter set (tfhe_gate_bootstrapping.cpp:23-75); one native library per set"""

    def __init__(self, lam, n, l, Bgbit, ks_stdev, bk_stdev):
        self.lam, self.n, self.l, self.Bgbit = lam, n, l, Bgbit
        self.N, self.k, self.kpl = 1024, 1, 2 * l
        self.ks_t, self.ks_basebit, self.ks_base = 8, 2, 4
        self.ks_stdev, self.bk_stdev = ks_stdev, bk_stdev
        self.MU = 1 << 29
        self.suffix = "" if lam == 128 else "_%d" % lam     # libtfhe_hip.so / libtfhe_hip_80.so

    def __repr__(self):
        return "ParamSet(lambda=%d, n=%d, l=%d, Bgbit=%d)" % (self.lam, self.n, self.l, self.Bgbit)


PARAMS = ParamSet(128, 630, 3, 7, 2.0 ** -15, 2.0 ** -25)    # the default set; every function defaults to it
PARAMS_80 = ParamSet(80, 500, 2, 10, 2.44e-5, 7.18e-9)


class Gate(enum.IntEnum):
    NAND = 0
    OR = 1
    AND = 2
    XOR = 3
    XNOR = 4
    NOR = 5
    ANDNY = 6
    ANDYN = 7
    ORNY = 8
    ORYN = 9
    MUX = 10
    NOT = 11
    COPY = 12
    CONST1 = 13
    CONST0 = 14


class TfheHipError(RuntimeError):
    pass


_libs = {}

EXPORTS = [
    "tfhe_hip_keys_device_buffers", "tfhe_hip_keys_adopt", "tfhe_hip_tgsw_decomp_batch",
    "tfhe_hip_create", "tfhe_hip_destroy", "tfhe_hip_last_error", "tfhe_hip_load_keys_coef", "tfhe_hip_load_keys_fft",
    "tfhe_hip_get_bkfft", "tfhe_hip_gate_batch", "tfhe_hip_bootstrap_batch", "tfhe_hip_keyswitch_batch",
    "tfhe_hip_gate_batch_dev", "tfhe_hip_bootstrap_batch_dev", "tfhe_hip_keyswitch_batch_dev", "tfhe_hip_sync",
    "tfhe_hip_stream", "tfhe_hip_last_kernel_ms", "tfhe_hip_blind_rotate_batch", "tfhe_hip_extern_mul_batch",
    "tfhe_hip_int_ifft_batch", "tfhe_hip_torus_ifft_batch", "tfhe_hip_torus_fft_batch",
    "tfhe_hip_negacyclic_mul_batch", "tfhe_hip_negacyclic_mul_batch_dev", "tfhe_hip_gate_level_dev",
    "tfhe_hip_dev_alloc", "tfhe_hip_dev_free", "tfhe_hip_copy_to_dev", "tfhe_hip_copy_from_dev",
    "tfhe_hip_negacyclic_mul2048_batch", "tfhe_hip_negacyclic_mul2048_batch_dev",
    "tfhe_hip_encrypt_batch", "tfhe_hip_encrypt_batch_dev", "tfhe_hip_phase_batch", "tfhe_hip_keygen_bk",
    "tfhe_hip_param_set", "tfhe_hip_netlist_create", "tfhe_hip_netlist_run", "tfhe_hip_netlist_is_graph",
    "tfhe_hip_netlist_destroy", "tfhe_hip_keygen_cloud", "tfhe_hip_gate_batch_dev_noalloc",
    "tfhe_hip_set_arithmetic", "tfhe_hip_get_arithmetic",
    "tfhe_hip_group_create", "tfhe_hip_group_destroy", "tfhe_hip_group_size", "tfhe_hip_group_ctx",
    "tfhe_hip_group_load_keys_coef", "tfhe_hip_group_load_keys_fft", "tfhe_hip_group_set_arithmetic",
    "tfhe_hip_group_gate_batch", "tfhe_hip_group_key_path", "tfhe_hip_keys_fingerprint",
    "tfhe_hip_group_gate_level",
    "tfhe_hip_tgsw_fft_extern_mul_batch", "tfhe_hip_tgsw_extern_mul_batch", "tfhe_hip_tlwe_addmul_r_batch",
    "tfhe_hip_negacyclic_mul_n_batch", "tfhe_hip_negacyclic_mul_n_batch_dev", "tfhe_hip_int_ifft_n_batch",
    "tfhe_hip_torus_ifft_n_batch", "tfhe_hip_torus_fft_n_batch", "tfhe_hip_tgsw_decomp_n_batch",
    "tfhe_hip_tgsw_fft_extern_mul_n_batch", "tfhe_hip_tgsw_extern_mul_n_batch", "tfhe_hip_tlwe_addmul_r_n_batch",
]


def lib_path(params=PARAMS):
    if params.lam == 128:
        return LIB_PATH
    return os.path.join(_HERE, "libtfhe_hip%s.so" % params.suffix)


def load_library(params=PARAMS):
    if params.lam not in _libs:
        path = lib_path(params)
        if not os.path.exists(path):
            raise TfheHipError("%s not built: run `python -m tfhe_amd.build`" % path)
        lib = ctypes.CDLL(path)
        lib.tfhe_hip_last_error.restype = ctypes.c_char_p
        lib.tfhe_hip_stream.restype = ctypes.c_void_p
        lib.tfhe_hip_param_set.restype = None
        lib.tfhe_hip_netlist_destroy.restype = None
        lib.tfhe_hip_netlist_destroy.argtypes = [ctypes.c_void_p]
        lib.tfhe_hip_netlist_run.argtypes = [ctypes.c_void_p]
        lib.tfhe_hip_netlist_is_graph.argtypes = [ctypes.c_void_p]
        got = (ctypes.c_int32 * 8)()
        lib.tfhe_hip_param_set(got)
        want = [params.lam, params.n, params.N, params.k, params.l, params.Bgbit, params.ks_t, params.ks_basebit]
        if list(got) != want:
            raise TfheHipError("%s is built for %r, expected %r" % (path, list(got), want))
        _libs[params.lam] = lib
    return _libs[params.lam]


def _p(arr):
    return None if arr is None else ctypes.c_void_p(arr.ctypes.data)


def _i32(arr, shape_tail):
    if arr is None:
        return None
    a = np.ascontiguousarray(arr, dtype=np.int32)
    if a.ndim < 1 or tuple(a.shape[-len(shape_tail):]) != tuple(shape_tail):
        raise ValueError("expected trailing shape %r, got %r" % (shape_tail, a.shape))
    return a.reshape((-1,) + tuple(shape_tail))


class CompiledNetlist:
    """tfhe_hip_netlist: every level validated and resident on the device, replayed as one hipGraph"""

    def __init__(self, engine, pool, pool_rows, levels):
        self.engine, self.lib = engine, engine.lib
        arrs = [np.ascontiguousarray(lv, dtype=np.int32).reshape(-1, 5) for lv in levels]
        gates = np.concatenate(arrs) if arrs else np.zeros((0, 5), np.int32)
        counts = np.array([a.shape[0] for a in arrs], np.int64)
        self.n_bootstrapped = int((gates[:, 0] <= int(Gate.MUX)).sum())
        h = ctypes.c_void_p()
        engine._chk(self.lib.tfhe_hip_netlist_create(engine.ctx, ctypes.byref(h), engine._dptr(pool),
                                                     ctypes.c_int64(pool_rows), _p(gates), _p(counts),
                                                     ctypes.c_int64(len(arrs))), "netlist_create")
        self.handle = h

    @property
    def is_graph(self):
        return bool(self.lib.tfhe_hip_netlist_is_graph(self.handle))

    def run(self, sync=True):
        self.engine._chk(self.lib.tfhe_hip_netlist_run(self.handle), "netlist_run")
        if sync:
            self.engine.sync()

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tfhe_hip_netlist_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Group:
    """tfhe_hip_group: several GPUs behind one handle (one context, stream and host thread per listed device); keys
    uploaded once and replicated device to device, batches split into contiguous shards"""

    def __init__(self, devices, params=PARAMS):
        self.params = params
        self.lib = load_library(params)
        self.lib.tfhe_hip_group_destroy.restype = None
        self.lib.tfhe_hip_group_destroy.argtypes = [ctypes.c_void_p]
        devs = (ctypes.c_int * len(devices))(*devices)
        h = ctypes.c_void_p()
        rc = self.lib.tfhe_hip_group_create(ctypes.byref(h), devs, len(devices))
        if rc != 0:
            raise TfheHipError("tfhe_hip_group_create(%r): %s" % (list(devices), self.lib.tfhe_hip_last_error().decode()))
        self.handle = h

    def _chk(self, rc, what):
        if rc != 0:
            raise TfheHipError("%s failed (%d): %s" % (what, rc, self.lib.tfhe_hip_last_error().decode()))

    def __len__(self):
        return self.lib.tfhe_hip_group_size(self.handle)

    def load_keys_coef(self, bk_coef, ks):
        P = self.params
        bk = _i32(bk_coef, (P.n, P.kpl, 2, P.N))
        k = _i32(ks, (P.N, P.ks_t, P.ks_base, P.n + 1))
        self._chk(self.lib.tfhe_hip_group_load_keys_coef(self.handle, _p(bk), _p(k)), "group_load_keys_coef")

    def set_arithmetic(self, mode):
        code = {"exact": 0, "fast": 1}[mode] if isinstance(mode, str) else int(mode)
        self._chk(self.lib.tfhe_hip_group_set_arithmetic(self.handle, code), "group_set_arithmetic")

    def member(self, i):
        """borrowed Engine view of member i's context (device-memory helpers, *_dev entry points); do not close it"""
        self.lib.tfhe_hip_group_ctx.restype = ctypes.c_void_p
        e = Engine.__new__(Engine)
        e.params, e.lib, e.device = self.params, self.lib, None
        e.ctx = ctypes.c_void_p(self.lib.tfhe_hip_group_ctx(self.handle, i))
        e.close = lambda: None
        return e

    def gate_level(self, pools, pool_rows, gates):
        """tfhe_hip_group_gate_level: pools = one device pointer per member (replicas of the ciphertext pool)"""
        g = np.ascontiguousarray(gates, dtype=np.int32).reshape(-1, 5)
        arr = (ctypes.c_void_p * len(pools))(*[ctypes.c_void_p(p) for p in pools])
        self._chk(self.lib.tfhe_hip_group_gate_level(self.handle, arr, ctypes.c_int64(pool_rows), _p(g),
                                                     ctypes.c_int64(g.shape[0])), "group_gate_level")

    def key_paths(self):
        """per member: 0 host upload, 1 peer copy with peer access, 2 peer copy staged by the runtime, 3 same device"""
        return [self.lib.tfhe_hip_group_key_path(self.handle, i) for i in range(len(self))]

    def gate_batch(self, gate, a=None, b=None, c=None, count=None):
        P = self.params
        a, b, c = (_i32(x, (P.n + 1,)) for x in (a, b, c))
        if count is None:
            count = a.shape[0]
        out = np.empty((count, P.n + 1), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_group_gate_batch(self.handle, int(gate), _p(out), _p(a), _p(b), _p(c),
                                                     ctypes.c_int64(count)), "group_gate_batch")
        return out

    def close(self):
        if getattr(self, "handle", None):
            self.lib.tfhe_hip_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Engine:
    """One engine = one tfhe_hip_ctx = one GPU."""

    def __init__(self, device=0, params=PARAMS):
        self.params = params
        self.lib = load_library(params)
        ctx = ctypes.c_void_p()
        rc = self.lib.tfhe_hip_create(ctypes.byref(ctx), int(device))
        if rc != 0:
            raise TfheHipError("tfhe_hip_create(%d): %s" % (device, self.lib.tfhe_hip_last_error().decode()))
        self.ctx = ctx
        self.device = device

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.tfhe_hip_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc, what):
        if rc != 0:
            raise TfheHipError("%s failed (%d): %s" % (what, rc, self.lib.tfhe_hip_last_error().decode()))

    # ---- arithmetic mode of the FP64 stage: "exact" (bit-identical to the reference) or "fast" ----
    def set_arithmetic(self, mode):
        code = {"exact": 0, "fast": 1}[mode] if isinstance(mode, str) else int(mode)
        self._chk(self.lib.tfhe_hip_set_arithmetic(self.ctx, code), "set_arithmetic")

    @property
    def arithmetic(self):
        return "fast" if self.lib.tfhe_hip_get_arithmetic(self.ctx) == 1 else "exact"

    # ---- keys ----
    def load_keys_coef(self, bk_coef, ks):
        P = self.params
        bk = _i32(bk_coef, (P.n, P.kpl, 2, P.N))
        k = _i32(ks, (P.N, P.ks_t, P.ks_base, P.n + 1))
        self._chk(self.lib.tfhe_hip_load_keys_coef(self.ctx, _p(bk), _p(k)), "load_keys_coef")

    def load_keys_fft(self, bkfft, ks):
        P = self.params
        bk = np.ascontiguousarray(bkfft, dtype=np.float64)
        assert bk.size == P.n * P.kpl * 2 * P.N
        k = _i32(ks, (P.N, P.ks_t, P.ks_base, P.n + 1))
        self._chk(self.lib.tfhe_hip_load_keys_fft(self.ctx, _p(bk), _p(k)), "load_keys_fft")

    def decomp_batch(self, polys):
        """tGswTorus32PolynomialDecompH per polynomial: int32[count][N] -> int32[count][l][N]"""
        P = self.params
        x = np.ascontiguousarray(polys, dtype=np.int32).reshape(-1, P.N)
        out = np.empty((x.shape[0], P.l, P.N), np.int32)
        self._chk(self.lib.tfhe_hip_tgsw_decomp_batch(self.ctx, _p(out), _p(x), ctypes.c_int64(x.shape[0])), "tgsw_decomp_batch")
        return out

    def key_buffers(self):
        """[(device pointer, bytes)] of the three device-resident key buffers (allocated if absent): the source or the
        destination of a key broadcast between ranks (tfhe_amd.parallel.broadcast_keys_device)"""
        ptrs, sizes = (ctypes.c_void_p * 3)(), (ctypes.c_size_t * 3)()
        self._chk(self.lib.tfhe_hip_keys_device_buffers(self.ctx, ptrs, sizes), "keys_device_buffers")
        return [(int(ptrs[i]), int(sizes[i])) for i in range(3)]

    def adopt_keys(self):
        self._chk(self.lib.tfhe_hip_keys_adopt(self.ctx), "keys_adopt")

    def key_fingerprints(self):
        """64-bit content fingerprints of the three key buffers, computed on the device"""
        out = (ctypes.c_uint64 * 3)()
        self._chk(self.lib.tfhe_hip_keys_fingerprint(self.ctx, out), "keys_fingerprint")
        return [int(v) for v in out]

    def get_bkfft(self):
        P = self.params
        out = np.empty((P.n, P.kpl, 2, P.N // 2, 2), dtype=np.float64)
        self._chk(self.lib.tfhe_hip_get_bkfft(self.ctx, _p(out)), "get_bkfft")
        return out

    # ---- gates, host buffers ----
    def gate_batch(self, gate, a=None, b=None, c=None, count=None):
        P = self.params
        a, b, c = (_i32(x, (P.n + 1,)) for x in (a, b, c))
        if count is None:
            count = a.shape[0]
        out = np.empty((count, P.n + 1), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_gate_batch(self.ctx, int(gate), _p(out), _p(a), _p(b), _p(c),
                                               ctypes.c_int64(count)), "gate_batch")
        return out

    def bootstrap_batch(self, x, mu=PARAMS.MU, with_ks=True):
        P = self.params
        x = _i32(x, (P.n + 1,))
        out = np.empty((x.shape[0], (P.n if with_ks else P.N) + 1), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_bootstrap_batch(self.ctx, _p(out), _p(x), ctypes.c_int32(mu),
                                                    ctypes.c_int64(x.shape[0]), int(bool(with_ks))), "bootstrap_batch")
        return out

    def keyswitch_batch(self, u):
        P = self.params
        u = _i32(u, (P.N + 1,))
        out = np.empty((u.shape[0], P.n + 1), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_keyswitch_batch(self.ctx, _p(out), _p(u), ctypes.c_int64(u.shape[0])),
                  "keyswitch_batch")
        return out

    # ---- device-resident (torch tensors or raw device pointers) ----
    # STREAM CONTRACT: these launch on the context's own non-blocking stream (self.stream()), not on torch's current
    # stream.  Inputs written by torch must have landed first (torch.cuda.current_stream().synchronize()), and
    # self.sync() must return before torch reads the outputs.
    @staticmethod
    def _dptr(t):
        if t is None:
            return None
        if isinstance(t, int):
            return ctypes.c_void_p(t)
        return ctypes.c_void_p(t.data_ptr())

    def gate_batch_dev(self, gate, out, a=None, b=None, c=None, count=None):
        self._chk(self.lib.tfhe_hip_gate_batch_dev(self.ctx, int(gate), self._dptr(out), self._dptr(a), self._dptr(b),
                                                   self._dptr(c), ctypes.c_int64(count)), "gate_batch_dev")

    def gate_batch_dev_noalloc(self, gate, out, a=None, b=None, c=None, count=None):
        """tfhe_hip_gate_batch_dev_noalloc: same results, work buffers in the library's static device array"""
        self._chk(self.lib.tfhe_hip_gate_batch_dev_noalloc(self.ctx, int(gate), self._dptr(out), self._dptr(a),
                                                           self._dptr(b), self._dptr(c), ctypes.c_int64(count)),
                  "gate_batch_dev_noalloc")

    def bootstrap_batch_dev(self, out, x, count, mu=PARAMS.MU, with_ks=True):
        self._chk(self.lib.tfhe_hip_bootstrap_batch_dev(self.ctx, self._dptr(out), self._dptr(x), ctypes.c_int32(mu),
                                                        ctypes.c_int64(count), int(bool(with_ks))),
                  "bootstrap_batch_dev")

    def keyswitch_batch_dev(self, out, u, count):
        self._chk(self.lib.tfhe_hip_keyswitch_batch_dev(self.ctx, self._dptr(out), self._dptr(u),
                                                        ctypes.c_int64(count)), "keyswitch_batch_dev")

    def negacyclic_mul_dev(self, out, ipolys, tpolys, count):
        self._chk(self.lib.tfhe_hip_negacyclic_mul_batch_dev(self.ctx, self._dptr(out), self._dptr(ipolys),
                                                             self._dptr(tpolys), ctypes.c_int64(count)),
                  "negacyclic_mul_batch_dev")

    def gate_level_dev(self, pool, pool_rows, gates):
        """one netlist level on a device-resident pool; gates: int32[count][5] = gate, out, a, b, c (host)"""
        g = np.ascontiguousarray(gates, dtype=np.int32).reshape(-1, 5)
        self._chk(self.lib.tfhe_hip_gate_level_dev(self.ctx, self._dptr(pool), ctypes.c_int64(pool_rows), _p(g),
                                                   ctypes.c_int64(g.shape[0])), "gate_level_dev")

    def netlist_create(self, pool, pool_rows, levels):
        """compile a levelized circuit (list of int32[count][5] arrays) over a device-resident pool; -> CompiledNetlist"""
        return CompiledNetlist(self, pool, pool_rows, levels)

    # device memory without torch (C callers use the same four functions)
    def dev_alloc(self, nbytes):
        ptr = ctypes.c_void_p()
        self._chk(self.lib.tfhe_hip_dev_alloc(self.ctx, ctypes.c_size_t(nbytes), ctypes.byref(ptr)), "dev_alloc")
        return ptr.value

    def dev_free(self, ptr):
        self._chk(self.lib.tfhe_hip_dev_free(self.ctx, ctypes.c_void_p(ptr)), "dev_free")

    def to_dev(self, ptr, arr):
        arr = np.ascontiguousarray(arr)
        self._chk(self.lib.tfhe_hip_copy_to_dev(self.ctx, ctypes.c_void_p(ptr), _p(arr), ctypes.c_size_t(arr.nbytes)),
                  "copy_to_dev")

    def from_dev(self, ptr, shape, dtype=np.int32):
        out = np.empty(shape, dtype=dtype)
        self._chk(self.lib.tfhe_hip_copy_from_dev(self.ctx, _p(out), ctypes.c_void_p(ptr), ctypes.c_size_t(out.nbytes)),
                  "copy_from_dev")
        return out

    def sync(self):
        self._chk(self.lib.tfhe_hip_sync(self.ctx), "sync")

    def stream(self):
        return self.lib.tfhe_hip_stream(self.ctx)

    def last_kernel_ms(self):
        br, ks = ctypes.c_float(), ctypes.c_float()
        self._chk(self.lib.tfhe_hip_last_kernel_ms(self.ctx, ctypes.byref(br), ctypes.byref(ks)), "last_kernel_ms")
        return br.value, ks.value

    # ---- client-side bulk operations ----
    def encrypt_batch(self, seed, lwe_key, bits, first_index=0):
        key = _i32(lwe_key, (self.params.n,)).reshape(-1)
        b = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
        out = np.empty((b.shape[0], self.params.n + 1), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_encrypt_batch(self.ctx, _p(out), ctypes.c_uint64(seed), ctypes.c_int64(first_index),
                                                  _p(key), _p(b), ctypes.c_int64(b.shape[0])), "encrypt_batch")
        return out

    def encrypt_batch_dev(self, out, seed, lwe_key, bits, first_index=0):
        key = _i32(lwe_key, (self.params.n,)).reshape(-1)
        b = np.ascontiguousarray(bits, dtype=np.uint8).reshape(-1)
        self._chk(self.lib.tfhe_hip_encrypt_batch_dev(self.ctx, self._dptr(out), ctypes.c_uint64(seed),
                                                      ctypes.c_int64(first_index), _p(key), _p(b),
                                                      ctypes.c_int64(b.shape[0])), "encrypt_batch_dev")

    def keygen_bk(self, seed, lwe_key, tlwe_key, ks, want_coef=False):
        """draw, transform and install the bootstrapping key on the device; returns its coefficient form if asked"""
        lk = _i32(lwe_key, (self.params.n,)).reshape(-1)
        tk = _i32(tlwe_key, (self.params.N,)).reshape(-1)
        ks = None if ks is None else np.ascontiguousarray(ks, dtype=np.int32)   # None: keep the key on the device
        out = np.empty((self.params.n, self.params.kpl, 2, self.params.N), np.int32) if want_coef else None
        self._chk(self.lib.tfhe_hip_keygen_bk(self.ctx, ctypes.c_uint64(seed), _p(lk), _p(tk), _p(ks),
                                              _p(out) if want_coef else None), "keygen_bk")
        return out

    def keygen_cloud(self, seed, lwe_key, tlwe_key, want_keys=False):
        """draw both cloud keys on the device from the secret keys and install them; returns (bk_coef, ks) if asked"""
        P = self.params
        lk = _i32(lwe_key, (P.n,)).reshape(-1)
        tk = _i32(tlwe_key, (P.N,)).reshape(-1)
        bk = np.empty((P.n, P.kpl, 2, P.N), np.int32) if want_keys else None
        ks = np.empty((P.N, P.ks_t, P.ks_base, P.n + 1), np.int32) if want_keys else None
        self._chk(self.lib.tfhe_hip_keygen_cloud(self.ctx, ctypes.c_uint64(seed), _p(lk), _p(tk), _p(bk), _p(ks)),
                  "keygen_cloud")
        return (bk, ks) if want_keys else None

    def phase_batch(self, lwe_key, samples):
        key = _i32(lwe_key, (self.params.n,)).reshape(-1)
        s = _i32(samples, (self.params.n + 1,))
        out = np.empty(s.shape[0], dtype=np.int32)
        self._chk(self.lib.tfhe_hip_phase_batch(self.ctx, _p(out), _p(key), _p(s), ctypes.c_int64(s.shape[0])),
                  "phase_batch")
        return out

    # ---- stages ----
    def blind_rotate_batch(self, acc, bara):
        P = self.params
        acc = _i32(acc, (2, P.N)).copy()
        bara = _i32(bara, (P.n,))
        self._chk(self.lib.tfhe_hip_blind_rotate_batch(self.ctx, _p(acc), _p(bara), ctypes.c_int64(acc.shape[0])),
                  "blind_rotate_batch")
        return acc

    def extern_mul_batch(self, acc, key_index):
        P = self.params
        acc = _i32(acc, (2, P.N)).copy()
        idx = np.ascontiguousarray(key_index, dtype=np.int32).reshape(-1)
        assert idx.shape[0] == acc.shape[0]
        self._chk(self.lib.tfhe_hip_extern_mul_batch(self.ctx, _p(acc), _p(idx), ctypes.c_int64(acc.shape[0])),
                  "extern_mul_batch")
        return acc

    def tgsw_fft_extern_mul(self, acc, gsw_fft, index):
        """tGswFFTExternMulToTLwe with caller-supplied TGswSampleFFT operands [ngsw][2l][2][N/2][2]"""
        P = self.params
        acc = _i32(acc, (2, P.N)).copy()
        g = np.ascontiguousarray(gsw_fft, dtype=np.float64).reshape(-1, P.kpl, 2, P.N // 2, 2)
        idx = np.ascontiguousarray(index, dtype=np.int32).reshape(-1)
        self._chk(self.lib.tfhe_hip_tgsw_fft_extern_mul_batch(self.ctx, _p(acc), _p(g), _p(idx), ctypes.c_int64(g.shape[0]),
                                                              ctypes.c_int64(acc.shape[0])), "tgsw_fft_extern_mul_batch")
        return acc

    def tgsw_extern_mul(self, acc, tgsw, index):
        """tGswExternMulToTLwe with caller-supplied coefficient-domain TGSW operands [ntgsw][2l][2][N]"""
        P = self.params
        acc = _i32(acc, (2, P.N)).copy()
        g = _i32(tgsw, (P.kpl, 2, P.N))
        idx = np.ascontiguousarray(index, dtype=np.int32).reshape(-1)
        self._chk(self.lib.tfhe_hip_tgsw_extern_mul_batch(self.ctx, _p(acc), _p(g), _p(idx), ctypes.c_int64(g.shape[0]),
                                                          ctypes.c_int64(acc.shape[0])), "tgsw_extern_mul_batch")
        return acc

    def tlwe_addmul_r(self, result, poly, sample):
        P = self.params
        res = _i32(result, (2, P.N)).copy()
        self._chk(self.lib.tfhe_hip_tlwe_addmul_r_batch(self.ctx, _p(res), _p(_i32(poly, (P.N,))), _p(_i32(sample, (2, P.N))),
                                                        ctypes.c_int64(res.shape[0])), "tlwe_addmul_r_batch")
        return res

    def int_ifft(self, polys):
        return self._fwd(polys, self.lib.tfhe_hip_int_ifft_batch, "int_ifft_batch")

    def torus_ifft(self, polys):
        return self._fwd(polys, self.lib.tfhe_hip_torus_ifft_batch, "torus_ifft_batch")

    def _fwd(self, polys, fn, what):
        P = self.params
        polys = _i32(polys, (P.N,))
        out = np.empty((polys.shape[0], P.N // 2, 2), dtype=np.float64)
        self._chk(fn(self.ctx, _p(out), _p(polys), ctypes.c_int64(polys.shape[0])), what)
        return out

    def torus_fft(self, spectra):
        P = self.params
        s = np.ascontiguousarray(spectra, dtype=np.float64).reshape(-1, P.N // 2, 2)
        out = np.empty((s.shape[0], P.N), dtype=np.int32)
        self._chk(self.lib.tfhe_hip_torus_fft_batch(self.ctx, _p(out), _p(s), ctypes.c_int64(s.shape[0])),
                  "torus_fft_batch")
        return out

    def negacyclic_mul2048(self, ipolys, tpolys):
        a, b = _i32(ipolys, (2 * self.params.N,)), _i32(tpolys, (2 * self.params.N,))
        assert a.shape == b.shape
        out = np.empty_like(a)
        self._chk(self.lib.tfhe_hip_negacyclic_mul2048_batch(self.ctx, _p(out), _p(a), _p(b),
                                                             ctypes.c_int64(a.shape[0])), "negacyclic_mul2048_batch")
        return out

    def negacyclic_mul2048_dev(self, out, ipolys, tpolys, count):
        self._chk(self.lib.tfhe_hip_negacyclic_mul2048_batch_dev(self.ctx, self._dptr(out), self._dptr(ipolys),
                                                                 self._dptr(tpolys), ctypes.c_int64(count)),
                  "negacyclic_mul2048_batch_dev")

    def negacyclic_mul(self, ipolys, tpolys):
        P = self.params
        a, b = _i32(ipolys, (P.N,)), _i32(tpolys, (P.N,))
        assert a.shape == b.shape
        out = np.empty_like(a)
        self._chk(self.lib.tfhe_hip_negacyclic_mul_batch(self.ctx, _p(out), _p(a), _p(b), ctypes.c_int64(a.shape[0])),
                  "negacyclic_mul_batch")
        return out

    # ---- run-time N, k, l, Bgbit ----
    def negacyclic_mul_n(self, N, ipolys, tpolys):
        a = np.ascontiguousarray(ipolys, np.int32).reshape(-1, N)
        b = np.ascontiguousarray(tpolys, np.int32).reshape(-1, N)
        assert a.shape == b.shape
        out = np.empty_like(a)
        self._chk(self.lib.tfhe_hip_negacyclic_mul_n_batch(self.ctx, ctypes.c_int32(N), _p(out), _p(a), _p(b),
                                                           ctypes.c_int64(a.shape[0])), "negacyclic_mul_n_batch")
        return out

    def negacyclic_mul_n_dev(self, N, out, ipolys, tpolys, count):
        self._chk(self.lib.tfhe_hip_negacyclic_mul_n_batch_dev(self.ctx, ctypes.c_int32(N), self._dptr(out), self._dptr(ipolys),
                                                               self._dptr(tpolys), ctypes.c_int64(count)),
                  "negacyclic_mul_n_batch_dev")

    def ifft_n(self, N, polys, torus=True):
        x = np.ascontiguousarray(polys, np.int32).reshape(-1, N)
        out = np.empty((x.shape[0], N // 2, 2), np.float64)
        fn = self.lib.tfhe_hip_torus_ifft_n_batch if torus else self.lib.tfhe_hip_int_ifft_n_batch
        self._chk(fn(self.ctx, ctypes.c_int32(N), _p(out), _p(x), ctypes.c_int64(x.shape[0])), "ifft_n_batch")
        return out

    def fft_n(self, N, spectra):
        s = np.ascontiguousarray(spectra, np.float64).reshape(-1, N // 2, 2)
        out = np.empty((s.shape[0], N), np.int32)
        self._chk(self.lib.tfhe_hip_torus_fft_n_batch(self.ctx, ctypes.c_int32(N), _p(out), _p(s), ctypes.c_int64(s.shape[0])),
                  "torus_fft_n_batch")
        return out

    def decomp_n(self, N, l, Bgbit, polys):
        x = np.ascontiguousarray(polys, np.int32).reshape(-1, N)
        out = np.empty((x.shape[0], l, N), np.int32)
        self._chk(self.lib.tfhe_hip_tgsw_decomp_n_batch(self.ctx, ctypes.c_int32(N), ctypes.c_int32(l), ctypes.c_int32(Bgbit), _p(out),
                                                        _p(x), ctypes.c_int64(x.shape[0])), "tgsw_decomp_n_batch")
        return out

    def tgsw_extern_mul_n(self, N, k, l, Bgbit, acc, gsw, index, fft=False):
        """acc int32[count][k+1][N]; gsw: int32[ngsw][(k+1)l][k+1][N] (fft=False) or float64[ngsw][(k+1)l][k+1][N/2][2]"""
        acc = np.ascontiguousarray(acc, np.int32).reshape(-1, k + 1, N).copy()
        idx = np.ascontiguousarray(index, np.int32).reshape(-1)
        if fft:
            g = np.ascontiguousarray(gsw, np.float64).reshape(-1, (k + 1) * l, k + 1, N // 2, 2)
            fn = self.lib.tfhe_hip_tgsw_fft_extern_mul_n_batch
        else:
            g = np.ascontiguousarray(gsw, np.int32).reshape(-1, (k + 1) * l, k + 1, N)
            fn = self.lib.tfhe_hip_tgsw_extern_mul_n_batch
        self._chk(fn(self.ctx, ctypes.c_int32(N), ctypes.c_int32(k), ctypes.c_int32(l), ctypes.c_int32(Bgbit), _p(acc), _p(g),
                     _p(idx), ctypes.c_int64(g.shape[0]), ctypes.c_int64(acc.shape[0])), "tgsw_extern_mul_n_batch")
        return acc

    def tlwe_addmul_r_n(self, N, k, result, poly, sample):
        res = np.ascontiguousarray(result, np.int32).reshape(-1, k + 1, N).copy()
        p = np.ascontiguousarray(poly, np.int32).reshape(-1, N)
        s = np.ascontiguousarray(sample, np.int32).reshape(-1, k + 1, N)
        self._chk(self.lib.tfhe_hip_tlwe_addmul_r_n_batch(self.ctx, ctypes.c_int32(N), ctypes.c_int32(k), _p(res), _p(p), _p(s),
                                                          ctypes.c_int64(res.shape[0])), "tlwe_addmul_r_n_batch")
        return res
