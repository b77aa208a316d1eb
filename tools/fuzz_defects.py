"""Randomised hunt around the period-defect rounds (rounds.hiph break_key / cont_key, k_zone_certify): motifs of many periods
with point defects, bursts, twin defects at one phase, defects in the first and the last period, two motifs in one block --
every block checked against the CPU oracle (bit-exact SA, BWT, primary index) on both first-stage routes.
Usage: python tools/fuzz_defects.py [first seed] [seeds]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, pyarchon
import test_gpu_fuzz as T
from oracle_binding import Oracle
oracle = Oracle()
s0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
cnt = int(sys.argv[2]) if len(sys.argv) > 2 else 20
bad = 0
stats = {"break_rounds": 0, "cases": 0}
for seed in range(s0, s0 + cnt):
    x = T.defect_case(seed)
    n = x.size
    P, B, b0 = oracle.forward(x)
    for route in ("", "0", "1"):
        if route: os.environ["ARCHON_FORCE_PATH"] = route
        else: os.environ.pop("ARCHON_FORCE_PATH", None)
        try:
            sa, bwt, base = pyarchon.forward(x)
            st = pyarchon.stats()
            ok = bool((sa == P).all() and (bwt == B).all() and base == b0)
        except Exception as e:      # noqa: BLE001
            ok = False; st = {}; print("EXC", e)
        stats["cases"] += 1; stats["break_rounds"] += 1 if st.get("break_rounds") else 0
        if not ok:
            bad += 1
            print("FAIL seed", seed, "route", route or "auto", "n", n, {k2: st.get(k2) for k2 in ("path", "period", "break_rounds", "doubling_rounds")}, flush=True)
    os.environ.pop("ARCHON_FORCE_PATH", None)
    if (seed - s0) % 20 == 19: print("... seed", seed, "cases", stats["cases"], "failures", bad, flush=True)
print("cases", stats["cases"], "with break rounds", stats["break_rounds"], "failures:", bad)
