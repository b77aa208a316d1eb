"""Archon::validate (bwt/a7/src/archon.cpp:862-874) behind a forward pass: on what the pass left resident
(archon_hip_block_validate: no upload, no second gather) against the host-buffer form (archon_hip_validate: 5N bytes up,
x[sa[i]] gathered again) -- wall clock of both, same block, same process."""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
L = pyarchon.lib()
x = S.gen_random(n)
hx = ctypes.c_void_p(L.archon_hip_host_alloc(n)); L.archon_hip_host_alloc.restype = ctypes.c_void_p
L.archon_hip_host_alloc.argtypes = [ctypes.c_size_t]
px = L.archon_hip_host_alloc(n); psa = L.archon_hip_host_alloc(4 * n)
xs = np.ctypeslib.as_array(ctypes.cast(px, ctypes.POINTER(ctypes.c_uint8)), shape=(n,)); xs[:] = x
sa = np.ctypeslib.as_array(ctypes.cast(psa, ctypes.POINTER(ctypes.c_uint32)), shape=(n,))
blk = pyarchon.Block()
base = ctypes.c_uint32(0)
out = {"n": n}
for rep in range(3):
    t0 = time.perf_counter()
    pyarchon._check(L.archon_hip_block_forward(blk.h, ctypes.c_void_p(px), n, ctypes.c_void_p(psa), ctypes.cast(ctypes.byref(base), ctypes.c_void_p)))
    t1 = time.perf_counter()
    ok_res = L.archon_hip_block_validate(blk.h)
    t2 = time.perf_counter()
    ok_host = L.archon_hip_validate(ctypes.c_void_p(px), n, ctypes.c_void_p(psa), 0)
    t3 = time.perf_counter()
    assert ok_res == 1 and ok_host == 1
    out = {"n": n, "en_compute_wall_ms": round((t1 - t0) * 1e3, 2), "validate_resident_wall_ms": round((t2 - t1) * 1e3, 2),
           "validate_host_buffers_wall_ms": round((t3 - t2) * 1e3, 2)}
# a corrupted suffix array must be refused by both
sa[[1000, 1001]] = sa[[1001, 1000]]
assert L.archon_hip_validate(ctypes.c_void_p(px), n, ctypes.c_void_p(psa), 0) == 0
print(json.dumps(out))
