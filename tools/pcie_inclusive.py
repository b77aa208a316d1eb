"""PCIe-inclusive rate of the host-buffer entry points (never the bench `value`): archon_hip_forward / _inverse on a
256 MiB random block with pinned host buffers (archon_hip_host_alloc) and with plain pageable numpy arrays."""
import ctypes, os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np
import archon_synth as S, pyarchon
L = pyarchon.lib()
L.archon_hip_host_alloc.restype = ctypes.c_void_p
L.archon_hip_host_alloc.argtypes = [ctypes.c_size_t]
L.archon_hip_host_free.argtypes = [ctypes.c_void_p]
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
x = S.gen_random(n)
def pinned(nbytes, dtype):
    p = L.archon_hip_host_alloc(nbytes)
    return p, np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(nbytes,)).view(dtype)
px, hx = pinned(n, np.uint8); psa, hsa = pinned(4 * n, np.uint32); pb, hb = pinned(n, np.uint8); po, ho = pinned(n, np.uint8)
hx[:] = x
base = ctypes.c_uint32(0)
res = {"n": n}
for name, (ax, asa, ab, ao) in {"pinned": (hx, hsa, hb, ho), "pageable": (x, np.empty(n, np.uint32), np.empty(n, np.uint8), np.empty(n, np.uint8))}.items():
    best_f = best_i = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = L.archon_hip_forward(ctypes.c_void_p(ax.ctypes.data), n, ctypes.c_void_p(asa.ctypes.data), ctypes.c_void_p(ab.ctypes.data), ctypes.byref(base), 0)
        t1 = time.perf_counter()
        assert rc == 0
        rc = L.archon_hip_inverse(ctypes.c_void_p(ab.ctypes.data), n, base.value, ctypes.c_void_p(ao.ctypes.data), 0)
        t2 = time.perf_counter()
        assert rc == 0
        best_f = min(best_f, t1 - t0); best_i = min(best_i, t2 - t1)
    assert (ao == x).all()
    best_b = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = L.archon_hip_forward(ctypes.c_void_p(ax.ctypes.data), n, None, ctypes.c_void_p(ab.ctypes.data), ctypes.byref(base), 0)
        best_b = min(best_b, time.perf_counter() - t0)
        assert rc == 0
    res[name] = {"forward_bwt_only_ms": round(best_b * 1e3, 2), "forward_ms": round(best_f * 1e3, 2), "forward_MBps": round(n / 1e6 / best_f, 1), "inverse_ms": round(best_i * 1e3, 2), "inverse_MBps": round(n / 1e6 / best_i, 1)}
for p in (px, psa, pb, po):
    L.archon_hip_host_free(ctypes.c_void_p(p))
print(json.dumps(res))
