"""What co-running kernels that hold CUs (RCCL send/recv during the overlapped gather) do to the forward pipeline, and what
more ranges per pass buy: a test-only kernel occupies K CUs for the duration of a forward call (tools/micro/cu_hog.hip)."""
import ctypes, os, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon
hog = ctypes.CDLL(os.path.join(ROOT, "tools", "micro", "libcu_hog.so"))
hog.hog_launch.argtypes = [ctypes.c_int, ctypes.c_double, ctypes.c_void_p]
n = 256 << 20
x_t = torch.from_numpy(S.gen_random(n)).cuda()
sa_t = torch.empty(n, dtype=torch.int32, device="cuda"); bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
for ranges, aligned in (("256", None), ("1024", "1")):
    pyarchon.set_option("pass_ranges", int(ranges))          # the product options bench.py sets at N > 1
    pyarchon.set_option("pass_b_buckets", 0 if aligned else 1)
    for k in (0, 1, 2, 8, 32):
        best = 1e9
        for rep in range(3):
            torch.cuda.synchronize()
            if k: assert hog.hog_launch(k, 12.0, None) == 0
            time.sleep(0.002)                       # let the hogs settle on their CUs
            pyarchon.forward_dev(x_t, sa_t, bwt_t, base_t)
            st = pyarchon.stats()
            hog.hog_wait()
            best = min(best, st["ms_total"])
        print(json.dumps({"pass_ranges": int(ranges), "bucket_mode": not aligned, "cus_held": k, "forward_ms": round(best, 3),
                          "ms_pass_text": round(st["ms_pass_text"], 3), "ms_pass_rec": round(st["ms_pass_rec"], 3), "ms_local_sort": round(st["ms_local_sort"], 3)}), flush=True)
assert pyarchon.validate_dev(x_t, sa_t)
