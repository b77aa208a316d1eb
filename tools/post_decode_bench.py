"""The post stage's way back on the device (csrc/post.hiph, k_post_decode) against the host decoder: 256 MiB BWTs of four shapes.
Usage: python tools/post_decode_bench.py [MiB]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = mib << 20
L = pyarchon.lib()
L.archon_hip_post_bound.restype = ctypes.c_size_t
L.archon_hip_post_bound.argtypes = [ctypes.c_uint32]
for shape in ("text", "dna", "prose", "random"):
    x = torch.from_numpy(S.gen_shape(shape, n)).cuda()
    bwt = torch.empty_like(x); base = torch.zeros(1, dtype=torch.int32, device="cuda")
    pyarchon.forward_dev(x, None, bwt, base)
    cap = L.archon_hip_post_bound(n)
    d_pk = torch.empty(cap, dtype=torch.uint8, device="cuda")
    got = ctypes.c_size_t(0)
    torch.cuda.synchronize()
    assert L.archon_hip_post_encode_dev(bwt.data_ptr(), n, d_pk.data_ptr(), cap, ctypes.byref(got), 0, None) == 0
    back = torch.empty_like(bwt)
    ts = []
    for _ in range(4):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        m = pyarchon.post_decode_dev(d_pk, got.value, back)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    assert m == n and torch.equal(back, bwt)
    t = sorted(ts)[1]
    print(json.dumps({"shape": shape, "n": n, "stream_bytes": got.value, "ratio": round(got.value / n, 4), "device_decode_ms": round(t * 1e3, 2),
                      "device_decode_GBps_of_output": round(n / t / 1e9, 2)}), flush=True)
