"""Host-buffer entry points fed by one host thread against two (the library binds a thread to one of its two contexts per
device): blocks per second of archon_hip_forward / archon_hip_inverse on pinned host buffers.
Usage: python tools/two_ctx.py [MiB] [blocks per thread]"""
import ctypes, json, os, sys, threading, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np, pyarchon, archon_synth as S
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
L = pyarchon.lib()
L.archon_hip_host_alloc.restype = ctypes.c_void_p
L.archon_hip_host_alloc.argtypes = [ctypes.c_size_t]


def pinned(count, dtype):
    p = L.archon_hip_host_alloc(count * np.dtype(dtype).itemsize)
    assert p
    return np.ctypeslib.as_array(ctypes.cast(p, ctypes.POINTER(ctypes.c_uint8)), shape=(count * np.dtype(dtype).itemsize,)).view(dtype)


def run(threads, want_sa, inverse=False):
    bufs = []
    for t in range(threads):
        x = pinned(n, np.uint8); x[:] = S.gen_random(n, S.SEED_BASE + 2 + t)
        bufs.append((x, pinned(n, np.uint32) if want_sa else None, pinned(n, np.uint8), pinned(n, np.uint8)))
    bases = [0] * threads

    def fwd(t, k):
        # (every thread of a run -- warm-up and timed alike -- names its context: thread t of a run always computes on context t, so
        #  the timed threads find the arenas and staging buffers their warm-up twins grew.  Round 3's harness left the binding to
        #  the order the threads were created in; its one-thread figure (19.7 ms) was a timed thread on a cold context.)
        L.archon_hip_bind_context(0, t)
        x, sa, bwt, back = bufs[t]
        for _ in range(k):
            base = ctypes.c_uint32(0)
            rc = L.archon_hip_forward(x.ctypes.data, n, sa.ctypes.data if sa is not None else None, bwt.ctypes.data, ctypes.cast(ctypes.byref(base), ctypes.c_void_p), 0)
            assert rc == 0, pyarchon.lib().archon_hip_last_error()
            bases[t] = base.value

    def inv(t, k):
        L.archon_hip_bind_context(0, t)
        x, sa, bwt, back = bufs[t]
        for _ in range(k):
            assert L.archon_hip_inverse(bwt.ctypes.data, n, bases[t], back.ctypes.data, 0) == 0

    def timed(fn):
        for _ in range(2):                                                               # warm-up (arenas, staging buffers, first touch of the pinned pages)
            ts = [threading.Thread(target=fn, args=(t, 1)) for t in range(threads)]
            [t.start() for t in ts]; [t.join() for t in ts]
        ts = [threading.Thread(target=fn, args=(t, reps)) for t in range(threads)]
        t0 = time.perf_counter()
        [t.start() for t in ts]; [t.join() for t in ts]
        return (time.perf_counter() - t0) * 1e3 / (reps * threads)

    ms = timed(fwd)
    out = {"threads": threads, "sa": want_sa, "forward_ms_per_block": round(ms, 2), "forward_GBps": round(n / ms / 1e6, 2)}
    if inverse:
        ms = timed(inv)
        out.update(inverse_ms_per_block=round(ms, 2), inverse_GBps=round(n / ms / 1e6, 2))
        for t in range(threads):
            assert (bufs[t][3] == bufs[t][0]).all()
    return out


for threads in (1, 2):
    print(json.dumps(run(threads, False, True)))
    print(json.dumps(run(threads, True)))
