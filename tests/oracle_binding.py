"""ctypes binding of oracle/liboracle.so -- the CPU oracle (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ODIR, "liboracle.so")
REF_DIR = os.path.join(ODIR, "_ref")


def build():
    """Compile the oracle (plain C, gcc) if it is not there yet."""
    if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(os.path.join(ODIR, "archon_oracle.c")):
        subprocess.run(["make", "-C", ODIR, "liboracle.so"], check=True, capture_output=True)


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


class Oracle:
    def __init__(self):
        build()
        L = ctypes.CDLL(LIB)
        vp, u32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_size_t
        L.oracle_hist256.argtypes = [vp, sz, vp, vp]
        L.oracle_hist256.restype = None
        for f in ("oracle_sa_brute", "oracle_sa"):
            getattr(L, f).argtypes = [vp, u32, vp]
            getattr(L, f).restype = ctypes.c_int
        L.oracle_sa_to_bwt.argtypes = [vp, u32, vp, vp, vp]
        L.oracle_sa_to_bwt.restype = None
        L.oracle_validate.argtypes = [vp, u32, vp]
        L.oracle_validate.restype = ctypes.c_int
        L.oracle_check_sorted.argtypes = [vp, u32, vp]
        L.oracle_check_sorted.restype = ctypes.c_int
        L.oracle_lf_build.argtypes = [vp, u32, u32, vp]
        L.oracle_lf_build.restype = None
        L.oracle_lf_walk.argtypes = [vp, u32, u32, vp, vp]
        L.oracle_lf_walk.restype = ctypes.c_int
        L.oracle_inverse.argtypes = [vp, u32, u32, vp]
        L.oracle_inverse.restype = ctypes.c_int
        L.oracle_radix_scatter.argtypes = [vp, sz, vp]
        L.oracle_radix_scatter.restype = None
        L.oracle_lms_select.argtypes = [vp, u32, vp, vp]
        L.oracle_lms_select.restype = u32
        L.oracle_clock_seconds.restype = ctypes.c_double
        self.L = L

    @staticmethod
    def _x(x):
        if isinstance(x, (bytes, bytearray)):
            x = np.frombuffer(bytes(x), dtype=np.uint8)
        return np.ascontiguousarray(x, dtype=np.uint8)

    def hist256(self, x):
        x = self._x(x)
        c = np.zeros(256, np.uint32)
        s = np.zeros(257, np.uint32)
        self.L.oracle_hist256(_p(x), x.size, _p(c), _p(s))
        return c, s

    def sa(self, x, brute=False):
        x = self._x(x)
        P = np.empty(x.size, np.uint32)
        rc = (self.L.oracle_sa_brute if brute else self.L.oracle_sa)(_p(x), x.size, _p(P))
        assert rc == 0, rc
        return P

    def sa_to_bwt(self, x, P):
        x = self._x(x)
        P = np.ascontiguousarray(P, np.uint32)
        b = np.empty(x.size, np.uint8)
        base = ctypes.c_uint32(0)
        self.L.oracle_sa_to_bwt(_p(x), x.size, _p(P), _p(b), ctypes.cast(ctypes.byref(base), ctypes.c_void_p))
        return b, base.value

    def forward(self, x):
        P = self.sa(x)
        b, base = self.sa_to_bwt(x, P)
        return P, b, base

    def validate(self, x, P):
        x = self._x(x)
        P = np.ascontiguousarray(P, np.uint32)
        return bool(self.L.oracle_validate(_p(x), x.size, _p(P)))

    def check_sorted(self, x, P):
        x = self._x(x)
        P = np.ascontiguousarray(P, np.uint32)
        return bool(self.L.oracle_check_sorted(_p(x), x.size, _p(P)))

    def lf_build(self, bwt, base):
        bwt = self._x(bwt)
        T = np.empty(bwt.size, np.uint32)
        self.L.oracle_lf_build(_p(bwt), bwt.size, base, _p(T))
        return T

    def inverse(self, bwt, base):
        bwt = self._x(bwt)
        out = np.empty(bwt.size, np.uint8)
        rc = self.L.oracle_inverse(_p(bwt), bwt.size, base, _p(out))
        return rc, out

    def lms_select(self, x):
        x = self._x(x)
        count = np.zeros(256, np.uint32)
        items = np.zeros(max(1, x.size), np.uint32)
        n1 = self.L.oracle_lms_select(_p(x), x.size, _p(count), _p(items))
        return count, items[:n1]

    def radix_scatter(self, src):
        src = self._x(src)
        dst = np.empty_like(src)
        self.L.oracle_radix_scatter(_p(src), src.size, _p(dst))
        return dst


def ref_available(variant="a7ref"):
    return os.path.exists(os.path.join(REF_DIR, variant))


def run_ref(x, variant="a7ref", tmpdir="/tmp"):
    """Run the reference a7 binary (oracle/_ref, built from /root/reference by oracle/Makefile).
    Returns dict(P, bwt, base, validate, sa_time) or None if the reference crashed."""
    exe = os.path.join(REF_DIR, variant)
    tag = "%s/a7ref_%d" % (tmpdir, os.getpid())
    np.ascontiguousarray(x, np.uint8).tofile(tag + ".in")
    try:
        r = subprocess.run([exe, "e", tag + ".in", tag + ".bwt", tag + ".sa"], capture_output=True, text=True)
        if r.returncode != 0:
            return None
        P = np.fromfile(tag + ".sa", np.uint32)
        f = np.fromfile(tag + ".bwt", np.uint8)
        kv = dict(t.split("=") for t in r.stdout.split())
        return dict(P=P, bwt=f[:-4].copy(), base=int(f[-4:].view("<u4")[0]),
                    validate=int(kv["validate"]), sa_time=float(kv["sa_time"]))
    finally:
        for ext in (".in", ".bwt", ".sa"):
            if os.path.exists(tag + ext):
                os.remove(tag + ext)


def run_ref_lms(x, tmpdir="/tmp"):
    """The reference's own Constructor<byte>::findLMS (archon.cpp:160-172) through oracle/_ref/a7lms:
    returns (count[256], items) as a7 places them, or None when the binary is absent."""
    exe = os.path.join(REF_DIR, "a7lms")
    if not os.path.exists(exe):
        return None
    tag = "%s/a7lms_%d" % (tmpdir, os.getpid())
    np.ascontiguousarray(x, np.uint8).tofile(tag + ".in")
    try:
        r = subprocess.run([exe, tag + ".in", tag + ".out"], capture_output=True, text=True)
        if r.returncode != 0:
            return None
        d = np.fromfile(tag + ".out", np.uint32)
        return d[1:257].copy(), d[257:257 + int(d[0])].copy()
    finally:
        for ext in (".in", ".out"):
            if os.path.exists(tag + ext):
                os.remove(tag + ext)


def lms_digest(count, items):
    import hashlib
    return hashlib.sha256(np.ascontiguousarray(count, "<u4").tobytes() + np.ascontiguousarray(items, "<u4").tobytes()).hexdigest()
