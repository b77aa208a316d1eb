#!/usr/bin/env python3
"""Generate tests/golden/golden_full.json: THE REFERENCE ITSELF at the graded size.

Same procedure as make_golden.py (oracle/_ref/a7ref = shipped a7, oracle/_ref/a7ref_nt = a7 with
sTracking=false, both built from /root/reference by oracle/Makefile), on 256 MiB blocks -- the size
BASELINE.json's metric is quoted on.  Per case: SHA-256 of the little-endian P array (the reference's
Archon::P after enCompute, bwt/a7/src/archon.cpp:882-885) and of BWT||baseId (the bytes Archon::enWrite
emits, archon.cpp:887-900), plus which binary produced it and whether shipped a7 ran.

    python tests/golden/make_golden_full.py [case ...]     # case = shape:block, default = all
    python tests/golden/make_golden_full.py --merge        # collect the per-case files into golden_full.json

Each case takes 10 s .. 3 min of one core and ~2.6 GiB; cases are independent (run a few in parallel,
then --merge).  Needs /root/reference (development container only).
"""
import hashlib
import json
import os
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import archon_synth as S  # noqa: E402
import oracle_binding as OB  # noqa: E402

N = 256 << 20
# (shape, block): bench.py seeds rank r's block with block=r, configs[3] is 8 DNA blocks
CASES = ([("random", b) for b in range(8)] + [("dna", b) for b in range(8)] +
         [("text", 0), ("a", 0), ("ab", 0), ("motif", 0), ("prose", 0), ("motif_defects", 0), ("random_copy", 0)])
PART_DIR = os.path.join(HERE, "_full_parts")


def sha_chunks(a):
    h = hashlib.sha256()
    b = memoryview(np.ascontiguousarray(a)).cast("B")
    for o in range(0, len(b), 1 << 26):
        h.update(b[o:o + (1 << 26)])
    return h


def run_case(shape, block):
    x = S.gen_shape(shape, N, block=block)
    t0 = time.time()
    # shipped a7 first; it crashes or fails its own validate() on the no-LMS / repetitive shapes
    res, who = None, None
    if shape in ("random", "dna", "text", "prose", "motif_defects", "random_copy"):
        res = OB.run_ref(x, "a7ref")
        who = "a7ref"
        shipped = "crash" if res is None else ("ok" if res["validate"] == 1 else "fails_own_validate")
        if shipped != "ok":
            res = None
    else:
        shipped = "not_run (crashes on this shape at every smaller size, see golden.json)"
    if res is None:
        res = OB.run_ref(x, "a7ref_nt")
        who = "a7ref_nt"
        assert res is not None and res["validate"] == 1, (shape, block)
    h_p = sha_chunks(res["P"].astype("<u4", copy=False)).hexdigest()
    h_b = sha_chunks(res["bwt"])
    h_b.update(int(res["base"]).to_bytes(4, "little"))
    case = {"shape": shape, "block": block, "n": N, "shipped_a7": shipped, "produced_by": who,
            "base_id": int(res["base"]), "sha256_P": h_p, "sha256_bwt_base": h_b.hexdigest(),
            "reference_sa_time_s": res["sa_time"]}
    os.makedirs(PART_DIR, exist_ok=True)
    with open(os.path.join(PART_DIR, "%s_%d.json" % (shape, block)), "w") as f:
        json.dump(case, f)
    print(shape, block, who, shipped, case["base_id"], "%.0f s" % (time.time() - t0), flush=True)


def merge():
    cases = []
    for shape, block in CASES:
        p = os.path.join(PART_DIR, "%s_%d.json" % (shape, block))
        if os.path.exists(p):
            with open(p) as f:
                cases.append(json.load(f))
    out = {
        "generator": "tests/golden/make_golden_full.py",
        "reference": "kvark/dark-archon bwt/a7 (-O3 -DNDEBUG, 64-bit); a7ref_nt = sTracking=false",
        "inputs": "dark-archon_amd/archon_synth.gen_shape(shape, n, block)",
        "cases": cases,
    }
    with open(os.path.join(HERE, "golden_full.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(cases), "of", len(CASES), "cases")


if __name__ == "__main__":
    if "--merge" in sys.argv:
        merge()
    else:
        todo = [tuple(a.split(":")) for a in sys.argv[1:]] or CASES
        assert OB.ref_available("a7ref") and OB.ref_available("a7ref_nt"), "run `make -C oracle` first"
        for shape, block in todo:
            run_case(shape, int(block))
