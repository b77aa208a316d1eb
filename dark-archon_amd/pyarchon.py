"""ctypes binding of libarchon_hip.so (include/archon_hip.h).

Host-side plumbing for tests and bench.py only: numpy arrays for the host-buffer
entry points, torch CUDA tensors (raw device pointers + the current HIP stream)
for the device-resident ones.  There is no CPU fallback here or in the library:
if the shared object is missing, importing raises; without a GPU every compute
call returns ARCHON_E_NODEVICE and this module raises ArchonError.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARCHON_HIP_LIB") or os.path.join(_HERE, "libarchon_hip.so")   # override: A/B builds only

OK, E_ARG, E_NODEVICE, E_NOMEM, E_HIP, E_INTERNAL, E_CORRUPT = 0, -1, -2, -3, -4, -5, -6
MAX_N = 0x3FFFFF00

SYMBOLS = [
    "archon_hip_device_count", "archon_hip_last_error",
    "archon_hip_forward", "archon_hip_inverse", "archon_hip_hist256",
    "archon_hip_validate", "archon_hip_radix_scatter",
    "archon_hip_forward_keep", "archon_hip_read_bwt", "archon_hip_host_alloc", "archon_hip_host_free",
    "archon_hip_forward_dev", "archon_hip_inverse_dev", "archon_hip_hist256_dev",
    "archon_hip_validate_dev", "archon_hip_radix_scatter_dev", "archon_hip_sa_to_bwt", "archon_hip_sa_to_bwt_dev",
    "archon_hip_lms_select", "archon_hip_lms_select_dev",
    "archon_hip_reserve", "archon_hip_release", "archon_hip_get_stats",
    "archon_hip_block_create", "archon_hip_block_destroy", "archon_hip_block_forward", "archon_hip_block_read_bwt",
    "archon_hip_block_validate", "archon_hip_block_stats", "archon_hip_validate_keep",
    "archon_hip_bind_context", "archon_hip_context_of_thread", "archon_hip_set_option", "archon_hip_get_option",
    "archon_hip_post_bound", "archon_hip_post_encode_dev", "archon_hip_forward_post", "archon_hip_validate_resident_dev",
    "archon_hip_forward_batch", "archon_hip_inverse_batch", "archon_hip_forward_batch_dev", "archon_hip_inverse_batch_dev",
    "archon_hip_post_decode_dev", "archon_hip_inverse_post",
]


class Stats(ctypes.Structure):
    _fields_ = [
        ("n", ctypes.c_uint32), ("radix_passes", ctypes.c_uint32), ("doubling_rounds", ctypes.c_uint32),
        ("_pad0", ctypes.c_uint32),
        ("unresolved_initial", ctypes.c_uint64), ("unresolved_total", ctypes.c_uint64),
        ("ms_total", ctypes.c_float), ("ms_hist", ctypes.c_float), ("ms_sort", ctypes.c_float),
        ("ms_doubling", ctypes.c_float), ("ms_bwt", ctypes.c_float), ("ms_lf_build", ctypes.c_float),
        ("ms_lf_walk", ctypes.c_float), ("_pad1", ctypes.c_uint32),
        ("walk_chains", ctypes.c_uint64), ("kernel_launches", ctypes.c_uint32), ("radix_pass_timed", ctypes.c_uint32),
        ("ms_radix_pass_sum", ctypes.c_float), ("ms_local_sort", ctypes.c_float), ("ms_resolve", ctypes.c_float),
        ("path", ctypes.c_uint32), ("tie_groups", ctypes.c_uint32), ("tie_items", ctypes.c_uint32),
        ("ms_pass_text", ctypes.c_float), ("ms_pass_rec", ctypes.c_float),
        ("alphabet_bits", ctypes.c_uint32), ("period", ctypes.c_uint32),
        ("chain_items", ctypes.c_uint32), ("text_rounds", ctypes.c_uint32), ("seg_big_items", ctypes.c_uint64),
        ("chain_pairs", ctypes.c_uint64), ("break_rounds", ctypes.c_uint32), ("break_settled", ctypes.c_uint32),
        ("mid_items", ctypes.c_uint64), ("arena_bytes", ctypes.c_uint64),
        ("host_syncs", ctypes.c_uint32), ("_pad2", ctypes.c_uint32),
    ]

    def asdict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("_")}


class ArchonError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("archon_hip error %d: %s" % (code, msg))
        self.code = code


def load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("libarchon_hip.so not built: run `make lib` (hipcc, gfx950); there is no CPU fallback")
    # A process that also uses PyTorch-ROCm must load torch FIRST: torch brings its own HIP runtime, and a library that has
    # already bound to the system's runtime then finds no device ("no HIP device available") once torch has opened the GPU.
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    vp, u32, i32, sz = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int, ctypes.c_size_t
    lib.archon_hip_device_count.restype = i32
    lib.archon_hip_last_error.restype = ctypes.c_char_p
    for name, args in {
        "archon_hip_forward": [vp, u32, vp, vp, vp, i32],
        "archon_hip_inverse": [vp, u32, u32, vp, i32],
        "archon_hip_hist256": [vp, sz, vp, i32],
        "archon_hip_validate": [vp, u32, vp, i32],
        "archon_hip_sa_to_bwt": [vp, u32, vp, vp, vp, i32],
        "archon_hip_sa_to_bwt_dev": [vp, u32, vp, vp, vp, i32, vp],
        "archon_hip_radix_scatter": [vp, sz, vp, i32],
        "archon_hip_lms_select": [vp, u32, vp, vp, vp, i32],
        "archon_hip_lms_select_dev": [vp, u32, vp, vp, vp, i32, vp],
        "archon_hip_forward_dev": [vp, u32, vp, vp, vp, i32, vp],
        "archon_hip_inverse_dev": [vp, u32, u32, vp, i32, vp],
        "archon_hip_hist256_dev": [vp, sz, vp, i32, vp],
        "archon_hip_validate_dev": [vp, u32, vp, i32, vp],
        "archon_hip_validate_resident_dev": [vp, u32, vp, vp, u32, i32, vp],
        "archon_hip_radix_scatter_dev": [vp, sz, vp, i32, vp],
        "archon_hip_reserve": [u32, i32, vp],
        "archon_hip_release": [i32],
        "archon_hip_get_stats": [i32, ctypes.POINTER(Stats)],
        "archon_hip_block_create": [i32, vp],
        "archon_hip_block_forward": [vp, vp, u32, vp, vp],
        "archon_hip_block_read_bwt": [vp, u32, u32, vp],
        "archon_hip_block_validate": [vp],
        "archon_hip_block_stats": [vp, ctypes.POINTER(Stats)],
        "archon_hip_forward_keep": [vp, u32, vp, vp, i32],
        "archon_hip_read_bwt": [i32, u32, u32, vp],
        "archon_hip_validate_keep": [i32],
        "archon_hip_bind_context": [i32, i32],
        "archon_hip_context_of_thread": [i32],
        "archon_hip_set_option": [i32, ctypes.c_char_p, ctypes.c_long],
        "archon_hip_get_option": [i32, ctypes.c_char_p, ctypes.POINTER(ctypes.c_long)],
        "archon_hip_post_decode_dev": [vp, sz, vp, u32, vp, i32, vp],
        "archon_hip_inverse_post": [vp, sz, u32, vp, u32, vp, i32],
        "archon_hip_forward_batch": [vp, vp, u32, vp, vp, i32, i32],
        "archon_hip_inverse_batch": [vp, vp, vp, u32, vp, i32, i32],
        "archon_hip_forward_batch_dev": [vp, vp, u32, vp, vp, vp, i32, i32],
        "archon_hip_inverse_batch_dev": [vp, vp, vp, u32, vp, i32, i32],
        "archon_hip_post_encode_dev": [vp, u32, vp, sz, vp, i32, vp],
        "archon_hip_forward_post": [vp, u32, vp, sz, vp, vp, i32],
    }.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = i32
    lib.archon_hip_block_destroy.argtypes = [vp]
    lib.archon_hip_block_destroy.restype = None
    lib.archon_hip_post_bound.argtypes = [u32]
    lib.archon_hip_post_bound.restype = sz
    lib.archon_hip_test_route.argtypes = [ctypes.c_char_p, ctypes.c_long]      # include/archon_hip_test.h (tests only)
    lib.archon_hip_test_route.restype = i32
    return lib


_lib = None
_routes_seen = None
# Test routing (include/archon_hip_test.h): the library reads nothing from the environment; the tests, bench.py and the
# tools keep saying ARCHON_<NAME>=<value> in os.environ, and this binding hands what it finds to archon_hip_test_route
# before the next call into the library.
_ROUTE_NAMES = ("FORCE_PATH", "SMALL_BLOCK", "PASS_RANGES", "INV_ROWS", "INV_SLAB", "INV_SBITS", "INV_WALK_WGS", "NO_ALIGNED", "NO_CHAINS", "NO_DEEP_HINT",
                "NO_PACK", "NO_PACK_STREAM", "NO_PAIR_CHAINS", "NO_PERIOD_HINT", "NO_BREAK_ROUND", "NO_PERIOD_PROBE", "NO_PERIOD_STREAM", "NO_PROBE",
                "NO_RANK_WRITER", "NO_TEXT_ROUNDS", "NO_MID", "NO_SHALLOW", "NO_CLOSED_FORM", "NO_REL_RECORDS", "ALIGNED_MIN", "REL_MIN_SEG", "KEY_BYTES")


def _sync_routes(L):
    global _routes_seen
    now = tuple(os.environ.get("ARCHON_" + k) for k in _ROUTE_NAMES)
    if now == _routes_seen or (_routes_seen is None and not any(v is not None for v in now)):
        return          # (a process that names no route never touches the test hook)
    _routes_seen = now
    L.archon_hip_test_route(b"RESET", 0)
    for k, v in zip(_ROUTE_NAMES, now):
        if v is None:
            continue
        try:
            iv = int(v)
        except ValueError:
            iv = 1
        if L.archon_hip_test_route(k.encode(), iv) < 0:
            raise ArchonError(-1, L.archon_hip_last_error().decode("utf-8", "replace"))


def lib():
    global _lib
    if _lib is None:
        _lib = load()
    _sync_routes(_lib)
    return _lib


def _check(rc):
    if rc < 0:
        raise ArchonError(rc, lib().archon_hip_last_error().decode("utf-8", "replace"))
    return rc


def _p(a):
    return ctypes.c_void_p(a.ctypes.data)


def device_count():
    return lib().archon_hip_device_count()


# ---------------------------------------------------------------- host buffers (numpy)
def forward(x, want_sa=True, dev=0):
    """x: uint8 array -> (sa or None, bwt, base_id).  Archon::enCompute + enWrite semantics."""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    n = x.size
    sa = np.empty(n, dtype=np.uint32) if want_sa else None
    bwt = np.empty(n, dtype=np.uint8)
    base = ctypes.c_uint32(0)
    _check(lib().archon_hip_forward(_p(x), n, _p(sa) if want_sa else None, _p(bwt),
                                    ctypes.cast(ctypes.byref(base), ctypes.c_void_p), dev))
    return sa, bwt, base.value


def inverse(bwt, base_id, dev=0):
    """bwt + base_id -> x.  Archon::deCompute + deWrite semantics."""
    bwt = np.ascontiguousarray(bwt, dtype=np.uint8)
    out = np.empty(bwt.size, dtype=np.uint8)
    _check(lib().archon_hip_inverse(_p(bwt), bwt.size, int(base_id), _p(out), dev))
    return out


def hist256(x, dev=0):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    out = np.zeros(256, dtype=np.uint32)
    _check(lib().archon_hip_hist256(_p(x), x.size, _p(out), dev))
    return out


def validate(x, sa, dev=0):
    x = np.ascontiguousarray(x, dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    return bool(_check(lib().archon_hip_validate(_p(x), x.size, _p(sa), dev)))


def sa_to_bwt(x, sa, dev=0):
    """(bwt, base_id) for a suffix array the caller holds (Archon::enWrite's gather, archon.cpp:887-900)"""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    sa = np.ascontiguousarray(sa, dtype=np.uint32)
    bwt = np.empty(x.size, np.uint8)
    base = ctypes.c_uint32(0)
    _check(lib().archon_hip_sa_to_bwt(_p(x), x.size, _p(sa), _p(bwt), ctypes.byref(base), dev))
    return bwt, int(base.value)


def lms_select(x, dev=0):
    """(count[256], items) of a7's findLMS (archon.cpp:160-172): the subset the reference sorts directly"""
    x = np.ascontiguousarray(x, dtype=np.uint8)
    count = np.zeros(256, np.uint32)
    items = np.zeros(x.size // 2 + 8, np.uint32)
    n1 = ctypes.c_uint32(0)
    _check(lib().archon_hip_lms_select(_p(x), x.size, _p(count), _p(items), ctypes.cast(ctypes.byref(n1), ctypes.c_void_p), dev))
    return count, items[:n1.value]


def radix_scatter(src, dev=0):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    dst = np.empty_like(src)
    _check(lib().archon_hip_radix_scatter(_p(src), src.size, _p(dst), dev))
    return dst


def stats(dev=0):
    return stats_raw(dev).asdict()


def stats_raw(dev=0):
    """the Stats structure itself (asdict() later): what a timed loop takes per step"""
    s = Stats()
    _check(_lib.archon_hip_get_stats(dev, ctypes.byref(s)) if _lib is not None else lib().archon_hip_get_stats(dev, ctypes.byref(s)))
    return s


def reserve(n, dev=0):
    b = ctypes.c_size_t(0)
    _check(lib().archon_hip_reserve(n, dev, ctypes.cast(ctypes.byref(b), ctypes.c_void_p)))
    return b.value


def set_option(name, value, dev=0):
    """product option of a device (include/archon_hip.h: "pass_ranges", "pass_b_buckets")"""
    _check(lib().archon_hip_set_option(dev, name.encode(), int(value)))


def get_option(name, dev=0):
    v = ctypes.c_long(0)
    _check(lib().archon_hip_get_option(dev, name.encode(), ctypes.byref(v)))
    return int(v.value)


def bind_context(slot, dev=0):
    _check(lib().archon_hip_bind_context(dev, slot))


def context_of_thread(dev=0):
    return _check(lib().archon_hip_context_of_thread(dev))


class Block:
    """archon_hip_block: the device side of one block-coder object (x, SA and BWT resident between compute, validate, write)"""

    def __init__(self, dev=0):
        h = ctypes.c_void_p(None)
        _check(lib().archon_hip_block_create(dev, ctypes.byref(h)))
        self.h = h

    def close(self):
        if self.h:
            lib().archon_hip_block_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def forward(self, x, want_sa=True):
        x = np.ascontiguousarray(x, dtype=np.uint8)
        self.n = x.size
        sa = np.empty(x.size, dtype=np.uint32) if want_sa else None
        base = ctypes.c_uint32(0)
        _check(lib().archon_hip_block_forward(self.h, _p(x), x.size, _p(sa) if want_sa else None,
                                              ctypes.cast(ctypes.byref(base), ctypes.c_void_p)))
        return sa, base.value

    def read_bwt(self, offset=0, length=None):
        length = self.n - offset if length is None else length
        out = np.empty(length, dtype=np.uint8)
        _check(lib().archon_hip_block_read_bwt(self.h, offset, length, _p(out)))
        return out

    def validate(self):
        return bool(_check(lib().archon_hip_block_validate(self.h)))

    def stats(self):
        s = Stats()
        _check(lib().archon_hip_block_stats(self.h, ctypes.byref(s)))
        return s.asdict()


# ---------------------------------------------------------------- device resident (torch)
def _stream_ptr():
    import torch
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def forward_dev(x_t, sa_t, bwt_t, base_t):
    """torch CUDA tensors: x uint8[n], sa int32[n] or None, bwt uint8[n], base int32[1]."""
    dev = x_t.device.index or 0
    _check(lib().archon_hip_forward_dev(ctypes.c_void_p(x_t.data_ptr()), x_t.numel(),
                                        ctypes.c_void_p(sa_t.data_ptr()) if sa_t is not None else None,
                                        ctypes.c_void_p(bwt_t.data_ptr()), ctypes.c_void_p(base_t.data_ptr()),
                                        dev, _stream_ptr()))


def inverse_dev(bwt_t, base_id, out_t):
    dev = bwt_t.device.index or 0
    _check(lib().archon_hip_inverse_dev(ctypes.c_void_p(bwt_t.data_ptr()), bwt_t.numel(), int(base_id),
                                        ctypes.c_void_p(out_t.data_ptr()), dev, _stream_ptr()))


def hist256_dev(x_t, out_t):
    dev = x_t.device.index or 0
    _check(lib().archon_hip_hist256_dev(ctypes.c_void_p(x_t.data_ptr()), x_t.numel(),
                                        ctypes.c_void_p(out_t.data_ptr()), dev, _stream_ptr()))


def validate_dev(x_t, sa_t):
    dev = x_t.device.index or 0
    return bool(_check(lib().archon_hip_validate_dev(ctypes.c_void_p(x_t.data_ptr()), x_t.numel(),
                                                     ctypes.c_void_p(sa_t.data_ptr()), dev, _stream_ptr())))


def validate_resident_dev(x_t, sa_t, bwt_t, base_id):
    """Archon::validate on the outputs of a forward pass that are still on the device (no second gather of x[sa[i]])"""
    dev = x_t.device.index or 0
    return bool(_check(lib().archon_hip_validate_resident_dev(ctypes.c_void_p(x_t.data_ptr()), x_t.numel(), ctypes.c_void_p(sa_t.data_ptr()),
                                                              ctypes.c_void_p(bwt_t.data_ptr()), int(base_id), dev, _stream_ptr())))


def _ptr_array(ptrs):
    return (ctypes.c_void_p * len(ptrs))(*ptrs)


def forward_batch(blocks, dev=0, workers=0):
    """several small blocks per call (archon_hip_forward_batch): list of uint8 arrays -> list of (bwt, base_id)"""
    xs = [np.ascontiguousarray(b, dtype=np.uint8) for b in blocks]
    outs = [np.empty(b.size, np.uint8) for b in xs]
    ns = (ctypes.c_uint32 * len(xs))(*[b.size for b in xs])
    base = (ctypes.c_uint32 * len(xs))()
    _check(lib().archon_hip_forward_batch(_ptr_array([b.ctypes.data for b in xs]), ns, len(xs), _ptr_array([o.ctypes.data for o in outs]), base, dev, workers))
    return [(o, int(base[i])) for i, o in enumerate(outs)]


def inverse_batch(bwts, bases, dev=0, workers=0):
    bs = [np.ascontiguousarray(b, dtype=np.uint8) for b in bwts]
    outs = [np.empty(b.size, np.uint8) for b in bs]
    ns = (ctypes.c_uint32 * len(bs))(*[b.size for b in bs])
    base = (ctypes.c_uint32 * len(bs))(*[int(v) for v in bases])
    _check(lib().archon_hip_inverse_batch(_ptr_array([b.ctypes.data for b in bs]), ns, base, len(bs), _ptr_array([o.ctypes.data for o in outs]), dev, workers))
    return outs


def forward_batch_dev(x_ts, bwt_ts, base_ts, sa_ts=None, workers=0):
    dev = x_ts[0].device.index or 0
    ns = (ctypes.c_uint32 * len(x_ts))(*[t.numel() for t in x_ts])
    sa = _ptr_array([t.data_ptr() if t is not None else None for t in sa_ts]) if sa_ts is not None else None
    _check(lib().archon_hip_forward_batch_dev(_ptr_array([t.data_ptr() for t in x_ts]), ns, len(x_ts), sa, _ptr_array([t.data_ptr() for t in bwt_ts]),
                                              _ptr_array([t.data_ptr() for t in base_ts]), dev, workers))


def inverse_batch_dev(bwt_ts, bases, out_ts, workers=0):
    dev = bwt_ts[0].device.index or 0
    ns = (ctypes.c_uint32 * len(bwt_ts))(*[t.numel() for t in bwt_ts])
    base = (ctypes.c_uint32 * len(bwt_ts))(*[int(v) for v in bases])
    _check(lib().archon_hip_inverse_batch_dev(_ptr_array([t.data_ptr() for t in bwt_ts]), ns, base, len(bwt_ts), _ptr_array([t.data_ptr() for t in out_ts]), dev, workers))


def post_decode_dev(stream_t, nbytes, bwt_t):
    """a block's stream of the post stage on the device -> its BWT on the device; returns the block's length"""
    dev = stream_t.device.index or 0
    n = ctypes.c_uint32(0)
    _check(lib().archon_hip_post_decode_dev(ctypes.c_void_p(stream_t.data_ptr()), int(nbytes), ctypes.c_void_p(bwt_t.data_ptr()), bwt_t.numel(),
                                            ctypes.cast(ctypes.byref(n), ctypes.c_void_p), dev, _stream_ptr()))
    return int(n.value)


def inverse_post(stream, base_id, cap, dev=0):
    """host stream of the post stage + primary index -> the block (decoded and inverted on the device)"""
    stream = np.ascontiguousarray(stream, dtype=np.uint8)
    out = np.empty(cap, dtype=np.uint8)
    n = ctypes.c_uint32(0)
    _check(lib().archon_hip_inverse_post(_p(stream), stream.size, int(base_id), _p(out), cap, ctypes.cast(ctypes.byref(n), ctypes.c_void_p), dev))
    return out[:n.value]
