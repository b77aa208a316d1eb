"""ad-hoc bug hunt: the natural-route fuzz generator of tests/test_gpu_fuzz.py over many seeds"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pyarchon, oracle_binding
from test_gpu_fuzz import _natural_cases, _cases
orc = oracle_binding.Oracle()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    for k, x in enumerate(list(_natural_cases(rng, 10)) + list(_cases(rng, 30))):
        P, B, b0 = orc.forward(x)
        try:
            sa, bwt, base = pyarchon.forward(x)
            ok = (sa == P).all() and (bwt == B).all() and base == b0 and (pyarchon.inverse(B, b0) == x).all()
        except Exception as e:
            ok = False; print("EXC", e)
        if not ok:
            bad += 1
            print("FAIL seed", seed, "case", k, "n", x.size, x[:24].tolist(), pyarchon.stats(), flush=True)
    print("seed", seed, "done", flush=True)
print("failures:", bad)
