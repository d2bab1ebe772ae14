"""ctypes binding of include/tfhe_hip_io.h: the reference's wire formats (tfhe_io.cpp) over flat numpy arrays.
Host code of libtfhe_hip.so; needs no GPU.  Every function takes params= (default: the 128-bit set) and uses the
library built for that set."""
import ctypes

import numpy as np

from .engine import PARAMS, TfheHipError, _p, load_library

SAMPLE_BYTES = 2536   # default set; sample_bytes(params) in general


def sample_bytes(params=PARAMS):
    return 4 + 4 * (params.n + 1) + 8

EXPORTS = [
    "tfhe_hip_io_read_params", "tfhe_hip_io_write_params", "tfhe_hip_io_cloud_keyset_bytes",
    "tfhe_hip_io_read_cloud_keyset", "tfhe_hip_io_write_cloud_keyset", "tfhe_hip_io_read_cloud_keyset_file",
    "tfhe_hip_io_write_cloud_keyset_file", "tfhe_hip_load_cloud_keyset_file", "tfhe_hip_load_cloud_keyset_buffer",
    "tfhe_hip_io_read_samples", "tfhe_hip_io_write_samples",
]


def _lib(params=PARAMS):
    lib = load_library(params)
    lib.tfhe_hip_io_cloud_keyset_bytes.restype = ctypes.c_size_t
    return lib


def _chk(lib, rc, what):
    if rc != 0:
        raise TfheHipError("%s failed (%d): %s" % (what, rc, lib.tfhe_hip_last_error().decode()))


def _buf(data):
    b = np.frombuffer(data, dtype=np.uint8) if not isinstance(data, np.ndarray) else data.view(np.uint8).reshape(-1)
    return np.ascontiguousarray(b)


def read_params(data, params=PARAMS):
    """checks that data starts with the default parameter set; returns the number of bytes it occupies"""
    lib, b, used = _lib(params), _buf(data), ctypes.c_size_t()
    _chk(lib, lib.tfhe_hip_io_read_params(_p(b), ctypes.c_size_t(b.size), ctypes.byref(used)), "read_params")
    return used.value


def write_params(params=PARAMS):
    lib, out, n = _lib(params), np.empty(1024, np.uint8), ctypes.c_size_t()
    _chk(lib, lib.tfhe_hip_io_write_params(_p(out), ctypes.c_size_t(out.size), ctypes.byref(n)), "write_params")
    return out[:n.value].tobytes()


def _key_arrays(P=PARAMS):
    return (np.empty((P.n, P.kpl, 2, P.N), np.int32), np.empty((P.N, P.ks_t, P.ks_base, P.n + 1), np.int32),
            np.empty(2, np.float64))


def read_cloud_keyset(data, params=PARAMS):
    """-> bk_coef int32[630][6][2][1024], ks int32[1024][8][4][631], variances float64[2] (ks, bk), bytes consumed"""
    lib, b, used = _lib(params), _buf(data), ctypes.c_size_t()
    bk, ks, var = _key_arrays(params)
    _chk(lib, lib.tfhe_hip_io_read_cloud_keyset(_p(b), ctypes.c_size_t(b.size), _p(bk), _p(ks), _p(var),
                                                ctypes.byref(used)), "read_cloud_keyset")
    return bk, ks, var, used.value


def read_cloud_keyset_file(path, params=PARAMS):
    lib = _lib(params)
    bk, ks, var = _key_arrays(params)
    _chk(lib, lib.tfhe_hip_io_read_cloud_keyset_file(str(path).encode(), _p(bk), _p(ks), _p(var)),
         "read_cloud_keyset_file")
    return bk, ks, var


def _key_args(bk_coef, ks, variances, P=PARAMS):
    bk = np.ascontiguousarray(bk_coef, np.int32)
    k = np.ascontiguousarray(ks, np.int32)
    if bk.shape != (P.n, P.kpl, 2, P.N) or k.shape != (P.N, P.ks_t, P.ks_base, P.n + 1):
        raise ValueError("key arrays have the wrong shape")
    var = None if variances is None else np.ascontiguousarray(variances, np.float64).reshape(2)
    return bk, k, var


def write_cloud_keyset(bk_coef, ks, variances=None, params=PARAMS):
    """-> uint8 array holding the file image"""
    lib = _lib(params)
    bk, k, var = _key_args(bk_coef, ks, variances, params)
    out, n = np.empty(lib.tfhe_hip_io_cloud_keyset_bytes(), np.uint8), ctypes.c_size_t()
    _chk(lib, lib.tfhe_hip_io_write_cloud_keyset(_p(out), ctypes.c_size_t(out.size), ctypes.byref(n), _p(bk), _p(k),
                                                 _p(var)), "write_cloud_keyset")
    return out[:n.value]


def write_cloud_keyset_file(path, bk_coef, ks, variances=None, params=PARAMS):
    lib = _lib(params)
    bk, k, var = _key_args(bk_coef, ks, variances, params)
    _chk(lib, lib.tfhe_hip_io_write_cloud_keyset_file(str(path).encode(), _p(bk), _p(k), _p(var)),
         "write_cloud_keyset_file")


def load_cloud_keyset_file(engine, path):
    """file -> GPU (read, upload, on-GPU FFT conversion)"""
    _chk(engine.lib, engine.lib.tfhe_hip_load_cloud_keyset_file(engine.ctx, str(path).encode()),
         "load_cloud_keyset_file")


def load_cloud_keyset_buffer(engine, data):
    b = _buf(data)
    _chk(engine.lib, engine.lib.tfhe_hip_load_cloud_keyset_buffer(engine.ctx, _p(b), ctypes.c_size_t(b.size)),
         "load_cloud_keyset_buffer")


def read_samples(data, count=None, params=PARAMS):
    """-> samples int32[count][631], variances float64[count]"""
    lib, b, used = _lib(params), _buf(data), ctypes.c_size_t()
    if count is None:
        count = b.size // sample_bytes(params)
    out, var = np.empty((count, params.n + 1), np.int32), np.empty(count, np.float64)
    _chk(lib, lib.tfhe_hip_io_read_samples(_p(b), ctypes.c_size_t(b.size), _p(out), _p(var), ctypes.c_int64(count),
                                           ctypes.byref(used)), "read_samples")
    return out, var


def write_samples(samples, variances=None, params=PARAMS):
    lib = _lib(params)
    s = np.ascontiguousarray(samples, np.int32).reshape(-1, params.n + 1)
    var = None if variances is None else np.ascontiguousarray(variances, np.float64).reshape(s.shape[0])
    out, n = np.empty(s.shape[0] * sample_bytes(params), np.uint8), ctypes.c_size_t()
    _chk(lib, lib.tfhe_hip_io_write_samples(_p(out), ctypes.c_size_t(out.size), ctypes.byref(n), _p(s), _p(var),
                                            ctypes.c_int64(s.shape[0])), "write_samples")
    return out[:n.value].tobytes()
