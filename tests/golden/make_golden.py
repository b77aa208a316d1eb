#!/usr/bin/env python3
"""Generate tests/golden/golden.json by running THE REFERENCE ITSELF here.

Runs the reference a7 binaries that oracle/Makefile builds from /root/reference
(oracle/_ref/a7ref = shipped a7; oracle/_ref/a7ref_nt = a7 with its compile-time
switch sTracking=false, the route that is correct on every input -- SURVEY.md 8(c))
on the synthetic shapes of dark-archon_amd/archon_synth.py and records, per case,
either the full answer (small cases) or SHA-256 digests of the little-endian P array
and of BWT||baseId.  Inputs are regenerated from (shape, n) by the tests; only
outputs are stored.  Where shipped a7 crashes or fails its own validate() the case
records that fact and carries the sTracking=false answer only.

    python tests/golden/make_golden.py        # needs /root/reference (development container)
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402
import archon_synth as S  # noqa: E402
import oracle_binding as OB  # noqa: E402

SMALL = [257]
HASHED = [65536, 1 << 20]
BIG = {"random": 1 << 24, "dna": 1 << 24, "text": 1 << 24, "prose": 1 << 24, "motif_defects": 1 << 24, "random_copy": 1 << 24}


def digest(P, bwt, base):
    h1 = hashlib.sha256(np.ascontiguousarray(P, "<u4").tobytes()).hexdigest()
    h2 = hashlib.sha256(bwt.tobytes() + int(base).to_bytes(4, "little")).hexdigest()
    return h1, h2


def main():
    assert OB.ref_available("a7ref") and OB.ref_available("a7ref_nt"), "run `make -C oracle` first"
    cases = []
    for shape in S.SHAPES:
        sizes = SMALL + HASHED + ([BIG[shape]] if shape in BIG else [])
        for n in sizes:
            x = S.gen_shape(shape, n)
            shipped = OB.run_ref(x, "a7ref")
            nt = OB.run_ref(x, "a7ref_nt")
            assert nt is not None and nt["validate"] == 1, (shape, n)
            case = {"shape": shape, "n": n,
                    "shipped_a7": "crash" if shipped is None else ("ok" if shipped["validate"] == 1 else "fails_own_validate")}
            if shipped is not None and shipped["validate"] == 1:
                assert (shipped["P"] == nt["P"]).all() and (shipped["bwt"] == nt["bwt"]).all() and shipped["base"] == nt["base"]
            # A3: the reference's own findLMS placement (oracle/_ref/a7lms): number of LMS items + digest of count[256] || items
            lms = OB.run_ref_lms(x)
            assert lms is not None, "oracle/_ref/a7lms missing: make -C oracle"
            case["lms_n1"] = int(lms[1].size)
            case["sha256_lms"] = OB.lms_digest(lms[0], lms[1])
            case["base_id"] = nt["base"]
            case["sha256_P"], case["sha256_bwt_base"] = digest(nt["P"], nt["bwt"], nt["base"])
            if n in SMALL:
                case["P"] = [int(v) for v in nt["P"]]
                case["bwt_hex"] = nt["bwt"].tobytes().hex()
            cases.append(case)
            print(shape, n, case["shipped_a7"], case["base_id"], flush=True)
    known = {
        "abracadabra": ([6, 8, 11, 4, 1, 9, 2, 5, 7, 10, 3], "dbacbrraaaa", 2),
        "mississippi": ([2, 11, 5, 8, 1, 9, 10, 3, 6, 4, 7], "smspipissii", 1),
        "banana": ([2, 4, 6, 1, 3, 5], "nnbaaa", 2),
        "aaaa": ([4, 3, 2, 1], "aaaa", 0),
        "abab": ([3, 1, 4, 2], "bbaa", 2),
        "baba": ([4, 2, 3, 1], "bbaa", 0),
        "ab": ([1, 2], "ba", 1),
        "ba": ([2, 1], "ba", 0),
        "a": ([1], "a", 0),
    }
    out = {
        "generator": "tests/golden/make_golden.py",
        "reference": "kvark/dark-archon bwt/a7 (-O3 -DNDEBUG, 64-bit); a7ref_nt = sTracking=false",
        "known_answers_survey_8a0": [{"x_hex": k.encode().hex(), "P": v[0], "bwt_hex": v[1].encode().hex(), "base_id": v[2]}
                                     for k, v in known.items()] +
                                    [{"x_hex": "00ff00ffff", "P": [3, 1, 4, 2, 5], "bwt_hex": "ffffff0000", "base_id": 4}],
        "cases": cases,
    }
    with open(os.path.join(HERE, "golden.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", len(cases), "cases")


if __name__ == "__main__":
    main()
