"""GPU parity against THE REFERENCE's own outputs, directly (no oracle in between).

tests/golden/golden.json       outputs of the reference a7 on the synthetic shapes up to 16 Mi (make_golden.py)
tests/golden/golden_full.json  the same at 256 MiB -- the size BASELINE.json's metric is quoted on
                               (make_golden_full.py: shipped a7 where it validates, a7 sTracking=false elsewhere)

The HIP output (suffix array P as little-endian u32, and BWT || baseId as Archon::enWrite emits them,
bwt/a7/src/archon.cpp:887-900) is hashed with SHA-256 and compared with the recorded digests.
"""
import hashlib
import json
import os

import numpy as np
import pytest

import archon_synth as S

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
with open(os.path.join(HERE, "golden", "golden.json")) as f:
    GOLDEN = json.load(f)
with open(os.path.join(HERE, "golden", "golden_full.json")) as f:
    GOLDEN_FULL = json.load(f)


def _sha(*parts):
    h = hashlib.sha256()
    for p in parts:
        b = memoryview(p).cast("B") if not isinstance(p, bytes) else p
        for o in range(0, len(b), 1 << 26):
            h.update(b[o:o + (1 << 26)])
    return h.hexdigest()


@pytest.mark.small_block_default
@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "%s-%d" % (c["shape"], c["n"]))
def test_hip_vs_reference_small_product_route(archon, case):
    """the same cases on the route the product takes by itself: blocks below 8 MiB skip the streaming stage"""
    x = S.gen_shape(case["shape"], case["n"])
    sa, bwt, base = archon.forward(x)
    assert base == case["base_id"]
    assert _sha(np.ascontiguousarray(sa, "<u4")) == case["sha256_P"]
    assert _sha(bwt, int(base).to_bytes(4, "little")) == case["sha256_bwt_base"]
    if 8 <= case["n"] < (8 << 20):
        periodic = case["shape"] in ("a", "ab", "motif") and case["n"] >= (1 << 16)      # clean periodic blocks: the closed form (path 2)
        assert archon.stats()["path"] == (2 if periodic else 0)


@pytest.mark.parametrize("case", GOLDEN["cases"], ids=lambda c: "%s-%d" % (c["shape"], c["n"]))
def test_hip_vs_reference_small(archon, case):
    """every committed reference case up to 16 Mi, through the host-buffer C ABI"""
    x = S.gen_shape(case["shape"], case["n"])
    sa, bwt, base = archon.forward(x)
    assert base == case["base_id"]
    if "P" in case:
        assert list(sa) == case["P"] and bwt.tobytes().hex() == case["bwt_hex"]
    assert _sha(np.ascontiguousarray(sa, "<u4")) == case["sha256_P"]
    assert _sha(bwt, int(base).to_bytes(4, "little")) == case["sha256_bwt_base"]


def test_known_answers_from_golden(archon):
    for k in GOLDEN["known_answers_survey_8a0"]:
        x = np.frombuffer(bytes.fromhex(k["x_hex"]), np.uint8)
        sa, bwt, base = archon.forward(x)
        assert list(sa) == k["P"] and bwt.tobytes().hex() == k["bwt_hex"] and base == k["base_id"]


@pytest.mark.parametrize("case", GOLDEN_FULL["cases"], ids=lambda c: "%s-b%d" % (c["shape"], c["block"]))
def test_hip_vs_reference_full_size(archon, case):
    """BASELINE.json configs[1] (random), configs[2] (a / ab / motif), configs[3] (all 8 DNA blocks) and the text block of
    configs[4] at the full 256 MiB: device-resident forward, outputs hashed against what the reference produced."""
    import torch
    n = case["n"]
    x = S.gen_shape(case["shape"], n, block=case["block"])
    x_t = torch.from_numpy(x).cuda()
    del x
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    out_t = torch.empty(n + 4, dtype=torch.uint8, device="cuda")       # BWT || baseId (LE), the a7 file layout
    archon.forward_dev(x_t, sa_t, out_t[:n], out_t[n:].view(torch.int32))
    out = out_t.cpu().numpy()
    assert int(out[n:].view("<u4")[0]) == case["base_id"]
    assert _sha(out) == case["sha256_bwt_base"]
    del out
    sa = sa_t.cpu().numpy()
    assert _sha(sa) == case["sha256_P"]


@pytest.mark.parametrize("shape,route", [("random", "0"), ("dna", "0"), ("a", "0"), ("ab", "0"), ("random_copy", "0"),
                                         ("text", "1"), ("prose", "1"), ("motif", "1"), ("motif_defects", "1"),
                                         ("a", "shortcut"), ("ab", "shortcut"), ("motif", "shortcut"),
                                         ("random", "plain_records"), ("dna", "plain_records")])
def test_hip_vs_reference_full_size_other_route(archon, shape, route, monkeypatch):
    """the full-size reference digests again with the first stage the block would NOT take by itself: the 7-pass route on
    the blocks that stream by themselves (ARCHON_FORCE_PATH=0: random, DNA on packed keys, the periodic ones, the block with a
    long copy), the streaming stage -- oversized two-byte buckets handed on as groups tied over two bytes -- on the skewed
    ones (=1): every route gives the reference's bytes at the graded size, not only at n / 8"""
    import torch
    case = [c for c in GOLDEN_FULL["cases"] if c["shape"] == shape and c["block"] == 0][0]
    if route == "shortcut":          # the clean periodic blocks without their closed form: a shallow first stage + the run shortcut
        monkeypatch.setenv("ARCHON_NO_CLOSED_FORM", "1")
    elif route == "plain_records":   # bucket mode of pass B with the byte stream beside the records (by themselves these blocks take the range-relative ones)
        monkeypatch.setenv("ARCHON_NO_REL_RECORDS", "1")
    else:
        monkeypatch.setenv("ARCHON_FORCE_PATH", route)
    n = case["n"]
    x_t = torch.from_numpy(S.gen_shape(shape, n, block=0)).cuda()
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    out_t = torch.empty(n + 4, dtype=torch.uint8, device="cuda")
    archon.forward_dev(x_t, sa_t, out_t[:n], out_t[n:].view(torch.int32))
    st = archon.stats()
    if route == "shortcut":
        assert st["path"] != 2 and st["period"] in (1, 2, 1000) and st["doubling_rounds"] == 0
    elif route == "plain_records":
        assert st["path"] == 1
    else:
        assert st["path"] == int(route)
    assert _sha(out_t.cpu().numpy()) == case["sha256_bwt_base"]
    assert _sha(sa_t.cpu().numpy()) == case["sha256_P"]
