"""The MTF + zero-run + Huffman stage on the device (csrc/post.hiph) against the host stage it restates:
device time per 256 MiB BWT, packed size, and the host-buffer entry point archon_hip_forward_post against archon_hip_forward.
Usage: python tools/post_bench.py [MiB] [shape ...]"""
import ctypes, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
shapes = sys.argv[2:] or ["text", "prose", "dna", "random"]
L = pyarchon.lib()
L.archon_hip_post_bound.restype = ctypes.c_size_t; L.archon_hip_post_bound.argtypes = [ctypes.c_uint32]
L.archon_hip_post_encode_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
L.archon_hip_forward_post.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
host = ctypes.CDLL(os.path.join(ROOT, "dark-archon_amd", "libarchon.so"))
host.archon_post_bound.restype = ctypes.c_size_t; host.archon_post_bound.argtypes = [ctypes.c_size_t]
host.archon_post_encode.restype = ctypes.c_size_t; host.archon_post_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
cap = L.archon_hip_post_bound(n)
d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
h_out = np.empty(cap, np.uint8)
for shape in shapes:
    x = S.gen_shape(shape, n)
    x_t = torch.from_numpy(x).cuda()
    bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
    pyarchon.forward_dev(x_t, None, bwt, base) if False else pyarchon.forward_dev(x_t, torch.empty(n, dtype=torch.int32, device="cuda"), bwt, base)
    got = ctypes.c_size_t(0)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        assert L.archon_hip_post_encode_dev(bwt.data_ptr(), n, d_out.data_ptr(), cap, ctypes.byref(got), 0, None) == 0
        best = min(best, (time.perf_counter() - t0) * 1e3)
    # host stage on a 16 MiB sample of the same BWT, one thread
    hb = bwt[:16 << 20].cpu().numpy()
    tmp = np.empty(host.archon_post_bound(32768), np.uint8)
    t0 = time.perf_counter()
    for o in range(0, hb.size, 32768):
        host.archon_post_encode(hb[o:o + 32768].ctypes.data, 32768, tmp.ctypes.data)
    host_ms_per_256 = (time.perf_counter() - t0) * 1e3 * (n / hb.size)
    # the host-buffer entry points
    tf = 1e9; tp = 1e9
    sa = None
    for _ in range(3):
        t0 = time.perf_counter(); pyarchon.forward(x, want_sa=False); tf = min(tf, (time.perf_counter() - t0) * 1e3)
        b = ctypes.c_uint32(0); g2 = ctypes.c_size_t(0)
        t0 = time.perf_counter()
        assert L.archon_hip_forward_post(x.ctypes.data, n, h_out.ctypes.data, cap, ctypes.byref(g2), ctypes.byref(b), 0) == 0
        tp = min(tp, (time.perf_counter() - t0) * 1e3)
    print(json.dumps({"shape": shape, "n": n, "packed_bytes": got.value, "ratio": round(got.value / n, 4), "device_post_ms": round(best, 3),
                      "device_post_GBps": round(n / best / 1e6, 1), "host_post_ms_one_thread_extrapolated": round(host_ms_per_256, 0),
                      "forward_bwt_only_host_buffers_ms": round(tf, 2), "forward_post_host_buffers_ms": round(tp, 2)}), flush=True)
