"""CPU: the oracle (oracle/archon_oracle.c) is pinned against the reference's answers.

  * the ten known answers of SURVEY.md 8(a0)
  * the brute-force statement of the definition (a7 sufCompare) on exhaustive small alphabets
  * tests/golden/golden.json -- outputs of the reference a7 binaries run in the development
    container (generator: tests/golden/make_golden.py)
  * when oracle/_ref is present (development container), the live reference binaries
"""
import hashlib
import itertools
import json
import os

import numpy as np
import pytest

import archon_synth as S
import oracle_binding as OB

GOLDEN = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "golden.json")))


def test_known_answers(oracle):
    for ka in GOLDEN["known_answers_survey_8a0"]:
        x = np.frombuffer(bytes.fromhex(ka["x_hex"]), np.uint8)
        for brute in (True, False):
            P = oracle.sa(x, brute=brute)
            bwt, base = oracle.sa_to_bwt(x, P)
            assert list(P) == ka["P"]
            assert bwt.tobytes().hex() == ka["bwt_hex"] and base == ka["base_id"]
        assert oracle.validate(x, P) and oracle.check_sorted(x, P)


@pytest.mark.parametrize("alpha,maxlen", [((0, 1), 11), ((0, 1, 2), 7), ((254, 255), 9), ((0, 128, 255), 6)])
def test_exhaustive_vs_definition(oracle, alpha, maxlen):
    for n in range(1, maxlen + 1):
        for t in itertools.product(alpha, repeat=n):
            x = np.array(t, np.uint8)
            P = oracle.sa(x)
            assert (P == oracle.sa(x, brute=True)).all(), t
            bwt, base = oracle.sa_to_bwt(x, P)
            rc, back = oracle.inverse(bwt, base)
            assert rc == 0 and (back == x).all(), t


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "%s-%d" % (c["shape"], c["n"]))
def test_golden_reference_outputs(oracle, case):
    """bit-exact against what the reference a7 produced on the same bytes"""
    x = S.gen_shape(case["shape"], case["n"])
    P, bwt, base = oracle.forward(x)
    assert base == case["base_id"]
    assert hashlib.sha256(np.ascontiguousarray(P, "<u4").tobytes()).hexdigest() == case["sha256_P"]
    assert hashlib.sha256(bwt.tobytes() + int(base).to_bytes(4, "little")).hexdigest() == case["sha256_bwt_base"]
    if "P" in case:
        assert list(P) == case["P"] and bwt.tobytes().hex() == case["bwt_hex"]
    if case["n"] <= (1 << 20):
        assert oracle.validate(x, P)
        rc, back = oracle.inverse(bwt, base)
        assert rc == 0 and (back == x).all()
    # A3: oracle_lms_select against the placement the reference's own findLMS produced (oracle/_ref/a7lms, make_golden.py)
    count, items = oracle.lms_select(x)
    assert items.size == case["lms_n1"] and OB.lms_digest(count, items) == case["sha256_lms"]


def test_live_reference_lms(oracle):
    """the reference's findLMS itself (when oracle/_ref/a7lms is present: development container and GPU box)"""
    if not OB.ref_available("a7lms"):
        pytest.skip("oracle/_ref/a7lms not built")
    rng = np.random.default_rng(5)
    for x in [rng.integers(0, 256, 5000).astype(np.uint8), rng.integers(0, 2, 777).astype(np.uint8), S.gen_text(30000),
              np.arange(256, dtype=np.uint8).repeat(3), np.frombuffer(b"a" * 100, np.uint8), np.frombuffer(b"ba", np.uint8)]:
        ref = OB.run_ref_lms(x)
        count, items = oracle.lms_select(x)
        assert ref is not None and (ref[0] == count).all() and ref[1].size == items.size and (ref[1] == items).all()


def test_random_small_vs_definition(oracle):
    rng = np.random.default_rng(1)
    for _ in range(300):
        n = int(rng.integers(1, 200))
        k = int(rng.choice([1, 2, 3, 4, 16, 256]))
        x = rng.integers(0, k, size=n, dtype=np.uint8)
        if rng.random() < 0.3:
            x = 255 - x
        assert (oracle.sa(x) == oracle.sa(x, brute=True)).all()


def test_validate_detects_errors(oracle):
    x = S.gen_text(5000)
    P = oracle.sa(x)
    assert oracle.validate(x, P)
    Q = P.copy()
    Q[[100, 101]] = Q[[101, 100]]
    assert not oracle.validate(x, Q)
    assert not oracle.check_sorted(x, Q)


def test_lf_build_base_last(oracle):
    """row baseId ranks last in its bucket (archon.cpp:931-933)"""
    bwt = np.frombuffer(b"aaaa", np.uint8)
    T = oracle.lf_build(bwt, 1)
    assert list(T) == [0, 3, 1, 2]


def test_hist_and_scatter(oracle):
    x = S.gen_random(1 << 15)
    c, s = oracle.hist256(x)
    assert (c == np.bincount(x, minlength=256)).all()
    assert s[256] == x.size and (np.diff(s.astype(np.int64)) == c).all()
    assert (oracle.radix_scatter(x) == np.sort(x, kind="stable")).all()


@pytest.mark.skipif(not OB.ref_available("a7ref_nt"), reason="oracle/_ref not built (no /root/reference here)")
def test_live_reference(oracle):
    """development container only: the oracle against the reference binary on fresh seeds"""
    for shape, n, block in (("random", 300000, 5), ("dna", 200001, 6), ("text", 123457, 7), ("motif", 99999, 8)):
        x = S.gen_shape(shape, n, block=block)
        r = OB.run_ref(x, "a7ref_nt")
        assert r is not None and r["validate"] == 1
        P, bwt, base = oracle.forward(x)
        assert (P == r["P"]).all() and (bwt == r["bwt"]).all() and base == r["base"]
