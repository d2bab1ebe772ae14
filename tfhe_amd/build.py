"""Builds the native pieces in-tree.

  tfhe_amd/libtfhe_hip.so     HIP kernels + C-ABI (include/tfhe_hip.h), cross-compiled for gfx950
  tfhe_amd/libtfhe_client.so  client-side key generation / encryption (host C)
  oracle/liboracle.so         CPU restatement of the reference path (test infrastructure)
  ..._80.so of each           the same sources built for the 80-bit parameter set (-DTFHE_HIP_LAMBDA=80)
  oracle/_ref/libtfhe_ref.so  the reference itself, only where /root/reference exists (authoring container)
"""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "tfhe_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
REFERENCE = "/root/reference"


def _run(cmd, **kw):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, **kw)


def _stale(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


LAMBDAS = (128, 80)


def _suffix(lam):
    return "" if lam == 128 else "_%d" % lam


def build_hip(force=False, lam=128):
    out = os.path.join(ROOT, "tfhe_amd", "libtfhe_hip%s.so" % _suffix(lam))
    srcs = [os.path.join(CSRC, "tfhe_hip.hip"), os.path.join(CSRC, "br_core.h"),
            os.path.join(CSRC, "fast_core.h"), os.path.join(CSRC, "gauss_cdt.h"), os.path.join(CSRC, "nfft_core.h"), os.path.join(CSRC, "nk_ops.h"),
            os.path.join(CSRC, "compat_libtfhe.cpp"), os.path.join(CSRC, "wire_io.cpp"),
            os.path.join(CSRC, "internal.h"), os.path.join(ROOT, "include", "tfhe_hip.h"),
            os.path.join(ROOT, "include", "tfhe_hip_compat.h"), os.path.join(ROOT, "include", "tfhe_hip_io.h")]
    srcs = [s for s in srcs if os.path.exists(s)]
    if force or _stale(out, srcs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
               "-DTFHE_HIP_LAMBDA=%d" % lam, "-I" + os.path.join(ROOT, "include"), "-o", out, os.path.join(CSRC, "tfhe_hip.hip")]
        for extra in ("compat_libtfhe.cpp", "wire_io.cpp"):
            if os.path.exists(os.path.join(CSRC, extra)):
                cmd.append(os.path.join(CSRC, extra))
        _run(cmd)
    return out


def build_gates(force=False, lam=128):
    """libtfhe_hip_gates.so: the reference's own single-gate symbols (bootsNAND ...) forwarding to the GPU library"""
    out = os.path.join(ROOT, "tfhe_amd", "libtfhe_hip_gates%s.so" % _suffix(lam))
    src = os.path.join(CSRC, "gates_libtfhe.cpp")
    if force or _stale(out, [src, os.path.join(ROOT, "include", "tfhe_hip_compat.h"), build_hip(False, lam)]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-fvisibility=hidden", "-DTFHE_HIP_LAMBDA=%d" % lam,
              "-I" + os.path.join(ROOT, "include"), "-o", out, src, "-L" + os.path.join(ROOT, "tfhe_amd"),
              "-ltfhe_hip%s" % _suffix(lam), "-Wl,-rpath,$ORIGIN"])
    return out


def build_client(force=False, lam=128):
    out = os.path.join(ROOT, "tfhe_amd", "libtfhe_client%s.so" % _suffix(lam))
    src = os.path.join(CSRC, "client_keys.c")
    if force or _stale(out, [src, os.path.join(CSRC, "gauss_cdt.h")]):
        _run(["gcc", "-std=gnu11", "-O2", "-fPIC", "-shared", "-DTFHE_HIP_LAMBDA=%d" % lam, "-o", out, src, "-lpthread", "-lm"])
    return out


def build_oracle(force=False):
    args = ["make", "-C", os.path.join(ROOT, "oracle"), "oracle"]
    if force:
        args.insert(1, "-B")
    _run(args)
    if os.path.isdir(REFERENCE):
        _run(["make", "-C", os.path.join(ROOT, "oracle"), "-j8", "ref"], stdout=subprocess.DEVNULL,
             stderr=subprocess.DEVNULL)


def build_test_harness(force=False):
    """tests/compat/libcompat_harness.so: a stand-in application written against libtfhe's structs (test only)"""
    out = os.path.join(ROOT, "tests", "compat", "libcompat_harness.so")
    src = os.path.join(ROOT, "tests", "compat", "compat_harness.cpp")
    if force or _stale(out, [src, os.path.join(ROOT, "include", "tfhe_hip_compat.h")]):
        _run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-I" + os.path.join(ROOT, "include"), "-o", out, src,
              "-L" + os.path.join(ROOT, "tfhe_amd"), "-ltfhe_hip", "-Wl,-rpath,$ORIGIN/../../tfhe_amd"])
    return out


def build_all(force=False):
    for lam in LAMBDAS:
        build_hip(force, lam)
        build_gates(force, lam)
        build_client(force, lam)
    build_oracle(force)
    build_test_harness(force)


if __name__ == "__main__":
    import sys
    build_all(force="--force" in sys.argv)
