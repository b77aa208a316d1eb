"""Randomised check of the device post stage (csrc/post.hiph) against the host stage it restates (host/archon_post.cpp):
byte-identical block streams on random lengths, alphabets and run structures.  Usage: python tools/fuzz_post.py [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: F401  (first: the library then binds to the HIP runtime torch brings)
import ctypes, numpy as np, pyarchon
import test_gpu_post as T
host = ctypes.CDLL(os.path.join(ROOT, "dark-archon_amd", "libarchon.so"))
host.archon_post_bound.restype = ctypes.c_size_t; host.archon_post_bound.argtypes = [ctypes.c_size_t]
host.archon_post_encode.restype = ctypes.c_size_t; host.archon_post_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
bad = 0
seeds = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for seed in range(seeds):
    rng = np.random.default_rng(7000 + seed)
    n = int(rng.choice([0, 1, 2, 63, 2047, 2048, 2049, 32767, 32768, 32769, 65536, int(rng.integers(1, 400000))]))
    kind = int(rng.integers(0, 5))
    if kind == 0: x = rng.integers(0, 256, n).astype(np.uint8)
    elif kind == 1: x = rng.integers(0, int(rng.integers(1, 6)), n).astype(np.uint8) * int(rng.integers(1, 60))
    elif kind == 2: x = np.repeat(rng.integers(0, 256, n // 7 + 1).astype(np.uint8), rng.integers(1, 40, n // 7 + 1))[:n]
    elif kind == 3: x = np.sort(rng.integers(0, 256, n).astype(np.uint8))            # long runs, every symbol once in a while
    else: x = rng.choice(np.array([0, 1, 255], np.uint8), n, p=[0.9, 0.09, 0.01])
    x = np.ascontiguousarray(x, np.uint8)
    if x.size != n: x = np.resize(x, n) if n else np.zeros(0, np.uint8)
    want = T.host_stream(host, x)
    got = T.device_stream(pyarchon, x)
    if got != want:
        bad += 1
        print("FAIL seed", seed, "n", n, "kind", kind, len(got), len(want), flush=True)
    if seed % 50 == 49: print("... seed", seed, "failures", bad, flush=True)
print("failures:", bad)
