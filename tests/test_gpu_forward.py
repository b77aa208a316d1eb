"""GPU parity: the HIP forward path through the C ABI vs the CPU oracle (bit-exact)."""
import itertools

import numpy as np
import pytest

import archon_synth as S

pytestmark = pytest.mark.gpu

KNOWN = {
    b"abracadabra": ([6, 8, 11, 4, 1, 9, 2, 5, 7, 10, 3], b"dbacbrraaaa", 2),
    b"mississippi": ([2, 11, 5, 8, 1, 9, 10, 3, 6, 4, 7], b"smspipissii", 1),
    b"banana": ([2, 4, 6, 1, 3, 5], b"nnbaaa", 2),
    b"aaaa": ([4, 3, 2, 1], b"aaaa", 0),
    b"abab": ([3, 1, 4, 2], b"bbaa", 2),
    b"baba": ([4, 2, 3, 1], b"bbaa", 0),
    bytes([0, 255, 0, 255, 255]): ([3, 1, 4, 2, 5], bytes([255, 255, 255, 0, 0]), 4),
    b"ab": ([1, 2], b"ba", 1),
    b"ba": ([2, 1], b"ba", 0),
    b"a": ([1], b"a", 0),
}


def test_known_answers(archon):
    """SURVEY.md 8(a0) known answers (cross-checked against a7 / a6 there)."""
    for s, (P, B, base) in KNOWN.items():
        x = np.frombuffer(s, np.uint8)
        sa, bwt, b = archon.forward(x)
        assert list(sa) == P and bwt.tobytes() == B and b == base, s


def test_exhaustive_small(archon, oracle):
    """every string over {0,1}^<=8, {0,1,2}^<=5, {254,255}^<=8: SA equals the definition"""
    for alpha, maxlen in (((0, 1), 8), ((0, 1, 2), 5), ((254, 255), 8)):
        for n in range(1, maxlen + 1):
            for t in itertools.product(alpha, repeat=n):
                x = np.array(t, np.uint8)
                sa, bwt, b = archon.forward(x)
                assert (sa == oracle.sa(x, brute=True)).all(), t


@pytest.mark.parametrize("shape", S.SHAPES)
@pytest.mark.parametrize("n", [1000, 65536, 1 << 20])
def test_shapes_vs_oracle(archon, oracle, shape, n):
    x = S.gen_shape(shape, n)
    sa, bwt, base = archon.forward(x)
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all()
    assert (bwt == B).all() and base == b0
    assert archon.validate(x, sa)


@pytest.mark.parametrize("records", ["relative", "plain"])
@pytest.mark.parametrize("ranges", ["0", "77", "1024"])
@pytest.mark.parametrize("shape,n", [("random", 1 << 20), ("random", (3 << 20) + 12345), ("dna", 1 << 21), ("random", 300007), ("text", 1 << 20)])
def test_bucket_mode_record_formats(archon, oracle, shape, n, ranges, records, monkeypatch):
    """pass B in bucket mode (a balanced block: workgroup c takes second-byte bucket c) on blocks the oracle finishes in seconds
    (ALIGNED_MIN lowers the 16 MiB limit), with the range-relative records of passes.hiph -- the item of a record from its place
    in the bucket and pass A's range table -- and with the plain ones (NO_REL_RECORDS), for pass A cut into 256 / 77 / 1024 ranges;
    text is not balanced: the count turns bucket mode down on the device and the plain format's instantiations run"""
    monkeypatch.setenv("ARCHON_ALIGNED_MIN", "65536")
    monkeypatch.setenv("ARCHON_REL_MIN_SEG", "1")
    monkeypatch.setenv("ARCHON_FORCE_PATH", "1")
    if ranges != "0":
        monkeypatch.setenv("ARCHON_PASS_RANGES", ranges)
    if records == "plain":
        monkeypatch.setenv("ARCHON_NO_REL_RECORDS", "1")
    x = S.gen_shape(shape, n)
    sa, bwt, base = archon.forward(x)
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all() and (bwt == B).all() and base == b0
    assert archon.stats()["path"] == 1


def test_ff_heavy(archon, oracle):
    """0xFF runs exercise the end-of-string-above-255 rule (padding ties)."""
    rng = np.random.default_rng(5)
    for n in (1, 2, 7, 8, 9, 15, 64, 1000, 5000):
        x = np.full(n, 255, np.uint8)
        sa, _, _ = archon.forward(x)
        assert (sa == oracle.sa(x)).all(), n
        y = rng.choice(np.array([254, 255], np.uint8), size=n, p=[0.1, 0.9])
        sa, _, _ = archon.forward(y)
        assert (sa == oracle.sa(y)).all(), n


def test_ragged_sizes(archon, oracle):
    rng = np.random.default_rng(7)
    for n in (3, 5, 63, 64, 65, 255, 257, 4095, 4097, 8191, 8193, 16385, 100003):
        x = rng.integers(0, 4, size=n, dtype=np.uint8)
        sa, bwt, base = archon.forward(x)
        P, B, b0 = oracle.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, n


def test_bwt_without_sa(archon, oracle):
    x = S.gen_text(50000)
    sa, bwt, base = archon.forward(x, want_sa=False)
    assert sa is None
    P, B, b0 = oracle.forward(x)
    assert (bwt == B).all() and base == b0


def test_hist256(archon, oracle):
    for shape in ("random", "a", "text"):
        for n in (1, 15, 17, 4096, 1 << 20, (1 << 20) + 13):
            x = S.gen_shape(shape, n)
            assert (archon.hist256(x) == oracle.hist256(x)[0]).all()


def test_radix_scatter(archon, oracle):
    x = S.gen_random(1 << 15)
    assert (archon.radix_scatter(x) == oracle.radix_scatter(x)).all()


def test_validate_rejects(archon, oracle):
    x = S.gen_text(10000)
    P = oracle.sa(x)
    assert archon.validate(x, P)
    Q = P.copy()
    Q[[10, 11]] = Q[[11, 10]]
    assert not archon.validate(x, Q)
    Q = P.copy()
    Q[5] = 0
    assert not archon.validate(x, Q)


def test_errors(archon):
    import ctypes
    L = archon.lib()
    assert L.archon_hip_forward(None, 10, None, None, None, 0) == archon.E_ARG
    x = np.zeros(4, np.uint8)
    b = np.zeros(4, np.uint8)
    base = ctypes.c_uint32()
    bp = ctypes.cast(ctypes.byref(base), ctypes.c_void_p)
    assert L.archon_hip_forward(ctypes.c_void_p(x.ctypes.data), 0, None, ctypes.c_void_p(b.ctypes.data), bp, 0) == archon.E_ARG
    assert L.archon_hip_forward(ctypes.c_void_p(x.ctypes.data), 4, None, ctypes.c_void_p(b.ctypes.data), bp, 99) == archon.E_NODEVICE


def test_full_block_properties(archon):
    """BASELINE size (256 MiB uniform random, config 2): size-independent properties only --
    LF-consistency of the SA on the device, BWT is a permutation of the block, and the
    encode -> decode round trip; everything through the C ABI with device-resident buffers."""
    import torch
    n = 256 << 20
    x = S.gen_random(n)
    x_t = torch.from_numpy(x).cuda()
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(x_t, sa_t, bwt_t, base_t)
    st = archon.stats()
    assert st["path"] == 1 and st["doubling_rounds"] == 0
    assert archon.validate_dev(x_t, sa_t)
    base = int(base_t.item())
    assert int(sa_t[base].item()) == n
    assert torch.equal(torch.bincount(bwt_t.int(), minlength=256), torch.bincount(x_t.int(), minlength=256))
    out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    archon.inverse_dev(bwt_t, base, out_t)
    assert torch.equal(out_t, x_t)
    # the 7-pass route (taken by heavily skewed blocks) must produce the same SA on the same bytes
    import os
    os.environ["ARCHON_FORCE_PATH"] = "0"
    try:
        sa2_t = torch.empty(n // 8, dtype=torch.int32, device="cuda")
        bwt2_t = torch.empty(n // 8, dtype=torch.uint8, device="cuda")
        archon.forward_dev(x_t[: n // 8], sa2_t, bwt2_t, base_t)
        assert archon.stats()["path"] == 0
        os.environ["ARCHON_FORCE_PATH"] = "1"
        sa3_t = torch.empty(n // 8, dtype=torch.int32, device="cuda")
        archon.forward_dev(x_t[: n // 8], sa3_t, bwt2_t, base_t)
        assert torch.equal(sa2_t, sa3_t)
    finally:
        del os.environ["ARCHON_FORCE_PATH"]


@pytest.mark.parametrize("shape", ["dna", "a", "ab", "motif", "text"])
def test_large_shapes_lf_consistent(archon, shape):
    """64 MiB blocks of the skewed shapes (configs 3-5 at reduced size): LF-consistency + round trip."""
    import torch
    n = 64 << 20
    x_t = torch.from_numpy(S.gen_shape(shape, n)).cuda()
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(x_t, sa_t, bwt_t, base_t)
    assert archon.validate_dev(x_t, sa_t)
    out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    archon.inverse_dev(bwt_t, int(base_t.item()), out_t)
    assert torch.equal(out_t, x_t)


def test_alphabet_compaction(archon, oracle):
    """with <= 32 distinct bytes the 7-pass route packs 56/bits symbols per key (17 ... 32 symbols: five bits,
    eleven symbols -- the key's last bit stays empty); packed and byte keys must give the same SA (and the oracle's), also at
    the 16/17- and the 32/33-symbol edges."""
    import os
    rng = np.random.default_rng(11)
    cases = [S.gen_dna(200000), S.gen_repeat(70001, b"ab"), S.gen_repeat(5000, b"a"),
             rng.choice(np.arange(100, 116, dtype=np.uint8), size=150000),          # 16 symbols -> 4 bits
             rng.choice(np.arange(100, 117, dtype=np.uint8), size=150000),          # 17 symbols -> 5 bits
             rng.choice(np.arange(200, 232, dtype=np.uint8), size=150000),          # 32 symbols -> 5 bits, code 31 = the pad code
             rng.choice(np.arange(223, 256, dtype=np.uint8), size=150000),          # 33 symbols -> bytes (0xFF present)
             S.gen_shape("prose", 300000),                                          # 28 symbols, deep ties
             rng.choice(np.array([0, 255], np.uint8), size=100000),                 # codes 0/1 with 0xFF present
             rng.choice(np.array([7, 9, 200], np.uint8), size=90000, p=[0.9, 0.05, 0.05])]
    for x in cases:
        os.environ["ARCHON_FORCE_PATH"] = "0"
        try:
            sa1, bwt1, b1 = archon.forward(x)
            bits = archon.stats()["alphabet_bits"]
            os.environ["ARCHON_NO_PACK"] = "1"
            sa2, bwt2, b2 = archon.forward(x)
            assert archon.stats()["alphabet_bits"] == 0
        finally:
            os.environ.pop("ARCHON_NO_PACK", None)
            os.environ.pop("ARCHON_FORCE_PATH", None)
        nsym = len(np.unique(x))
        assert (bits > 0) == (nsym <= 32), (nsym, bits)
        if bits: assert bits == max(1, int(np.ceil(np.log2(nsym)))), (nsym, bits)
        P, B, b0 = oracle.forward(x)
        assert (sa1 == P).all() and (sa2 == P).all() and (bwt1 == B).all() and b1 == b0 == b2


def _repeat_cases():
    rng = np.random.default_rng(31)
    motif = rng.integers(0, 256, size=1000, dtype=np.uint8)
    short = np.frombuffer(b"abcabd", np.uint8)
    a = lambda k, c=97: np.full(k, c, np.uint8)
    r = lambda k: rng.integers(0, 256, size=k, dtype=np.uint8)
    lit = lambda s: np.frombuffer(s, np.uint8)
    return {
        "a": a(300000),
        "ab": np.tile(lit(b"ab"), 150000),
        "ba_odd": np.tile(lit(b"ba"), 150000)[:-1],
        "motif1000": np.tile(motif, 300),
        "motif1000_ragged": np.tile(motif, 300)[137:-451],
        "motif6": np.tile(short, 50000),
        "ff": a(200000, 255),
        "feff": np.tile(np.array([254, 255], np.uint8), 100000),
        "three_runs": np.concatenate([a(100000), lit(b"b"), a(100000), lit(b"c"), a(70000), lit(b"\x00"), a(30000)]),
        "run_inside_random": np.concatenate([r(5000), np.tile(motif, 250), r(5000)]),
        "two_periods": np.concatenate([np.tile(lit(b"ab"), 80000), np.tile(lit(b"abc"), 60000)]),
        "run_signs": np.concatenate([lit(b"z"), a(60000), lit(b"A"), a(60000), lit(b"z"), a(60001), lit(b"a"), a(5)]),
        "motif_with_glitches": np.concatenate([np.tile(motif, 100), motif[:500], lit(b"!"), np.tile(motif, 100), motif[:3]]),
        "nested": np.tile(np.concatenate([a(500), lit(b"b")]), 400),
    }


@pytest.mark.parametrize("closed", ["closed_form", "shortcut"])
@pytest.mark.parametrize("name", sorted(_repeat_cases()))
def test_long_repeats(archon, oracle, name, closed, monkeypatch):
    """Gauntlet-style periodic inputs (BASELINE.json configs[2]): the closed form of clean periodic blocks (periodic.hiph),
    the run shortcut (k_chain_*) and the doubling rounds behind it must give the a7 order whatever mix of runs, periods
    and run boundaries the block holds."""
    if closed == "shortcut":
        monkeypatch.setenv("ARCHON_NO_CLOSED_FORM", "1")
    x = _repeat_cases()[name]
    sa, bwt, base = archon.forward(x)
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all()
    assert (bwt == B).all() and base == b0


@pytest.mark.streaming_machinery
@pytest.mark.parametrize("route", ["streaming", "lsb"])
def test_long_repeats_use_the_shortcut(archon, monkeypatch, route):
    """periodic blocks without the closed form: two streaming passes + the run shortcut, or three LSB passes + the shortcut"""
    monkeypatch.setenv("ARCHON_NO_CLOSED_FORM", "1")
    if route == "lsb":
        monkeypatch.setenv("ARCHON_NO_PERIOD_STREAM", "1")
    x = np.tile(np.frombuffer(b"ab", np.uint8), 1 << 20)
    archon.forward(x)
    st = archon.stats()
    assert st["period"] == 2 and st["chain_items"] > x.size * 0.99 and st["doubling_rounds"] == 0
    assert st["path"] == (1 if route == "streaming" else 0)


@pytest.mark.parametrize("hint", ["probe", "gaps"])
@pytest.mark.parametrize("name", ["a", "ab", "motif1000", "motif_with_glitches", "three_runs", "nested"])
def test_long_repeats_lsb_route(archon, oracle, name, hint, monkeypatch):
    """the same gauntlet with the periodic blocks kept on the 7-pass route (ordered groups): the period from the driver's
    probe, or -- probe's answer withheld -- from a sample of the neighbour gaps inside the tied groups"""
    monkeypatch.setenv("ARCHON_NO_PERIOD_STREAM", "1")
    monkeypatch.setenv("ARCHON_NO_CLOSED_FORM", "1")
    if hint == "gaps":
        monkeypatch.setenv("ARCHON_NO_PERIOD_HINT", "1")
    x = _repeat_cases()[name]
    sa, bwt, base = archon.forward(x)
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all() and (bwt == B).all() and base == b0


def _clean_periodic(p, n, alphabet, seed):
    """a block of n bytes with minimal period p over the given byte values (the motif is made primitive: its last byte
    differs from every other when the draw happens to repeat)"""
    rng = np.random.default_rng(seed)
    alpha = np.array(alphabet, np.uint8)
    motif = alpha[rng.integers(0, alpha.size, size=p)]
    if p > 1:
        if alpha.size < 2:
            raise ValueError("a period above 1 needs two byte values")
        # a^(p-1) b style tail guarantees primitivity: make position p-1 differ from position 0 and break any inner period
        motif[: p - 1] = alpha[rng.integers(0, alpha.size, size=p - 1)]
        for d in range(1, p):
            if p % d == 0 and np.array_equal(np.tile(motif[:d], p // d), motif):
                motif[p - 1] = alpha[(int(np.where(alpha == motif[p - 1])[0][0]) + 1) % alpha.size]
                break
    return np.tile(motif, n // p + 2)[:n].copy()


_CLOSED = [(p, extra, alpha_name) for p in (1, 2, 3, 7, 1000, 4099, 65521) for extra in (0, 1, "p-1", "half")
           for alpha_name in ("bytes", "ff", "ab")]


@pytest.mark.parametrize("route", ["big", "small"])
@pytest.mark.parametrize("p,extra,alpha_name", _CLOSED)
def test_clean_periodic_closed_form(archon, oracle, p, extra, alpha_name, route, request, monkeypatch):
    """a block with x[i] == x[i-p] everywhere is written down from the suffix array of its first 2p bytes
    (periodic.hiph) -- every period class, ragged ends (n mod p = 0, 1, p-1, p/2), alphabets with 0xFF (the end of the
    block sorts above it) and with two letters (classes that agree for long), against the oracle; on the route of big
    blocks (two-byte count in front) and on the product's small-block route (byte count in front)."""
    if p == 1 and extra in ("p-1", "half"):
        pytest.skip("n mod 1 is 0")
    if p == 1 and alpha_name == "ab":
        pytest.skip("one letter")
    alphabet = {"bytes": list(range(256)), "ff": [0, 254, 255], "ab": [97, 98]}[alpha_name]
    periods = 40 if p < 4099 else 17
    n = max(p * periods, 70000 // p * p)
    n += {0: 0, 1: 1, "p-1": p - 1, "half": p // 2}[extra]
    if route == "small":
        monkeypatch.setenv("ARCHON_SMALL_BLOCK", "-1")
    x = _clean_periodic(p, n, alphabet, 1000 * p + len(alphabet))
    sa, bwt, base = archon.forward(x)
    st = archon.stats()
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all()
    assert (bwt == B).all() and base == b0
    assert st["path"] == 2 and st["period"] == p and st["chain_items"] == n and st["doubling_rounds"] == 0, st
    # ... and the same block with one defect takes the general route (the closed form certifies the WHOLE text)
    y = x.copy()
    y[n - 1 - (n // 3)] ^= 1
    sa, bwt, base = archon.forward(y)
    P, B, b0 = oracle.forward(y)
    assert archon.stats()["path"] != 2
    assert (sa == P).all() and (bwt == B).all() and base == b0


def test_closed_form_without_sa_and_unaligned(archon, oracle):
    """the expansion with no suffix array asked for, and on a text that starts at an odd address (the aligned copy lives in
    the arena the nested transform also uses)"""
    import torch
    x = _clean_periodic(1000, 1000 * 300 + 17, list(range(256)), 5)
    _, bwt, base = archon.forward(x, want_sa=False)
    P, B, b0 = oracle.forward(x)
    assert (bwt == B).all() and base == b0 and archon.stats()["path"] == 2
    buf = torch.zeros(x.size + 64, dtype=torch.uint8, device="cuda")
    for shift in (1, 7):
        xt = buf[shift: shift + x.size]
        xt.copy_(torch.from_numpy(x))
        sa_t = torch.empty(x.size, dtype=torch.int32, device="cuda")
        bwt_t = torch.empty(x.size, dtype=torch.uint8, device="cuda")
        base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
        archon.forward_dev(xt, sa_t, bwt_t, base_t)
        assert archon.stats()["path"] == 2
        assert (sa_t.cpu().numpy().astype(np.uint32) == P).all() and (bwt_t.cpu().numpy() == B).all() and int(base_t.item()) == b0


def _defect_cases():
    """a motif repeated with point defects: groups that straddle the defects (never ONE clean run of the period)"""
    rng = np.random.default_rng(4099)
    out = {}
    for p, n, flips in ((1, 300000, 3), (2, 400001, 4), (3, 299999, 2), (7, 500000, 5), (100, 600000, 3), (1000, 1200000, 3),
                        (4099, 2000000, 3), (257, 700000, 12)):
        motif = rng.integers(0, 256, size=p, dtype=np.uint8)
        x = np.tile(motif, n // p + 1)[:n].copy()
        for q in rng.integers(0, n, size=flips):
            x[q] ^= np.uint8(1 + rng.integers(0, 255))
        out["p%d_%dflips" % (p, flips)] = x
    motif = rng.integers(0, 256, size=50, dtype=np.uint8)
    x = np.tile(motif, 8000)
    x[100000] = 0; x[200000] = 255; x[300000] = 0          # same phase, leaving below / above / below the periodic continuation
    out["same_phase_signs"] = x.copy()
    x = np.tile(motif, 8000)
    x[120025] ^= 1; x[240025] ^= 1                          # the same defect twice: items at equal distances behind them stay tied for the rounds
    out["twin_defects"] = x.copy()
    x = np.tile(motif, 8000)
    x[5] ^= 7; x[399990] ^= 9; x[200000:200003] ^= 3        # defects inside the first and the last period, and three in a row
    out["edges_and_burst"] = x.copy()
    x = np.concatenate([np.tile(motif, 4000), np.tile(rng.integers(0, 256, size=50, dtype=np.uint8), 4000)])   # two motifs of one period
    x[77777] ^= 1
    out["two_motifs"] = x
    return out


@pytest.mark.parametrize("name", sorted(_defect_cases()))
def test_period_defects(archon, oracle, name, monkeypatch):
    """N3 beyond exact periods (a4's anchors / tandem test, direct.c:90-161,198-221): groups that straddle defects of the
    period are settled by ONE round keyed on the distance to the last defect (rounds.hiph, break_key) once every group is tied
    over a whole period -- same a7 order as the oracle, with that round and with it switched off."""
    x = _defect_cases()[name]
    P, B, b0 = oracle.forward(x)
    sa, bwt, base = archon.forward(x)
    st = archon.stats()
    assert (sa == P).all(), name
    assert (bwt == B).all() and base == b0
    if name not in ("two_motifs", "p257_12flips"):          # (those two: parity only -- the period probe need not name a period for them)
        assert st["period"] > 0 and st["break_rounds"] >= 1 and st["break_settled"] > 0, st
    monkeypatch.setenv("ARCHON_NO_BREAK_ROUND", "1")
    sa2, bwt2, base2 = archon.forward(x)
    st2 = archon.stats()
    assert (sa2 == P).all() and (bwt2 == B).all() and base2 == b0
    assert st2["break_rounds"] == 0
    if st["break_rounds"]:
        assert st["doubling_rounds"] < st2["doubling_rounds"], (st, st2)


def _seg_cases():
    rng = np.random.default_rng(77)
    r = lambda k, hi=256: rng.integers(0, hi, size=k, dtype=np.uint8)
    R = r(200000)
    words = [bytes(r(int(k), 26) + 97) for k in rng.integers(2, 9, size=300)]
    prose = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, 300, size=120000)), np.uint8)
    runs = np.concatenate([np.concatenate([np.full(int(k), 32, np.uint8), r(40, 26) + 97]) for k in rng.integers(1, 400, size=2500)])
    return {
        "duplicated_block": np.concatenate([R, r(7), R, r(3), R[:150000]]),                 # pairs and triples, deep repeats
        "prose": prose,                                                                     # skewed groups of every size
        "prose_twice": np.concatenate([prose[:300000], prose[:300000]]),
        "space_runs": runs,                                                                 # long groups next to short ones
        "dups_and_runs": np.concatenate([R[:60000], runs[:200000], R[:60000], np.full(5000, 7, np.uint8), R[:999]]),
    }


@pytest.mark.parametrize("name", sorted(_seg_cases()))
def test_segmented_rounds(archon, oracle, name, monkeypatch):
    """Refinement rounds (text rounds and prefix doubling) with short groups sorted in LDS (k_round_fused over the S list)
    and long groups through the global sort (B list), in every mix, with and without the shortcuts: same order as the oracle."""
    x = _seg_cases()[name]
    P, B, b0 = oracle.forward(x)
    for env in ({}, {"ARCHON_FORCE_PATH": "0"}, {"ARCHON_NO_TEXT_ROUNDS": "1", "ARCHON_NO_CHAINS": "1"}, {"ARCHON_NO_PAIR_CHAINS": "1"},
                {"ARCHON_FORCE_PATH": "0", "ARCHON_NO_TEXT_ROUNDS": "1", "ARCHON_NO_PAIR_CHAINS": "1", "ARCHON_NO_CHAINS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        for k in env:
            monkeypatch.delenv(k)
        assert (sa == P).all(), (name, env)
        assert (bwt == B).all() and base == b0
    assert st["text_rounds"] + st["doubling_rounds"] > 0


@pytest.mark.small_block_default
@pytest.mark.parametrize("n", [1, 2, 7, 8, 9, 255, 4097, 65536, 1 << 20, (4 << 20) + 3, (8 << 20) - 1, 8 << 20])
def test_small_blocks_product_route(archon, oracle, n):
    """blocks below 8 MiB on the product's own route (byte count + LSB passes, no two-byte count): every shape, the sizes
    around the limits (n < 8: keys made by k_init_keys; the last size below the limit; the first one that streams)"""
    for shape in ("random", "text", "dna", "a", "ab", "motif", "prose"):
        if (n >= (1 << 22) and shape not in ("random", "text", "dna")) or (n < 255 and shape == "prose"):
            continue
        x = S.gen_shape(shape, n)
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (shape, n)
        if 8 <= n < (8 << 20):
            closed = shape in ("a", "ab", "motif") and n >= (1 << 16)        # clean periodic blocks: the closed form (path 2)
            assert archon.stats()["path"] == (2 if closed else 0), (shape, n)


@pytest.mark.small_block_default
@pytest.mark.parametrize("no_shallow", ["0", "1"])
def test_key_depth_follows_the_byte_counts(archon, oracle, no_shallow, monkeypatch):
    """a block whose byte counts promise hardly any tie at fewer than seven key bytes (incompressible data) sorts on fewer:
    3 or 4 LSB passes instead of 7, whatever stays tied goes to the rounds; text keeps seven; NO_SHALLOW switches it off"""
    monkeypatch.setenv("ARCHON_NO_SHALLOW", no_shallow)
    for shape, n, passes in (("random", 1 << 16, 3), ("random", (1 << 20) + 1, 4), ("random", 4 << 20, 4), ("random_copy", 3 << 20, 4),
                             ("text", 1 << 20, 7), ("prose", 1 << 20, 7), ("dna", 1 << 20, None)):
        x = S.gen_shape(shape, n)
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (shape, n)
        st = archon.stats()
        if passes is not None:
            assert st["radix_passes"] == (7 if no_shallow == "1" else passes), (shape, n, st["radix_passes"])
    # ties that the short keys leave: a random block with a copied region is tied deep whatever the counts say
    x = S.gen_random(2 << 20)
    x[(1 << 20):(1 << 20) + 300000] = x[1000:301000]
    P, B, b0 = oracle.forward(x)
    sa, bwt, base = archon.forward(x)
    assert (sa == P).all() and (bwt == B).all() and base == b0


@pytest.mark.small_block_default
@pytest.mark.parametrize("key_bytes", ["3", "4", "5", "6"])
def test_forced_key_depth(archon, oracle, key_bytes, monkeypatch):
    """route KEY_BYTES (tools/key_bytes_sweep.py): the LSB passes of the 7-pass route sort on that many key bytes, the rounds take
    the ties at that depth; the order never depends on it (on real text every byte less costs 2.4 - 3.6 ms of rounds and saves
    1.35 ms of passes at 256 MiB: profiles/r05_final/key_bytes_sweep.txt -- seven stays)"""
    monkeypatch.setenv("ARCHON_KEY_BYTES", key_bytes)
    for shape, n in (("text", (1 << 20) + 3), ("random_copy", 1 << 20), ("text", 70001)):
        x = S.gen_shape(shape, n)
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (shape, n)
        if shape == "text":
            assert archon.stats()["radix_passes"] == int(key_bytes), archon.stats()["radix_passes"]


def test_workspace_follows_the_block(archon, oracle):
    """the device workspace a forward call uses (archon_hip_stats.arena_bytes).  A block the streaming stage
    settles stays inside the first tier -- 26.2 bytes per input byte + 145 MB of fixed tables and slack (26.8 N at 256 MiB);
    a block that needs the general stage takes the second tier too; both give the oracle's bytes."""
    n = 32 << 20
    x = S.gen_random(n)
    sa, bwt, base = archon.forward(x)
    st = archon.stats()
    assert st["path"] == 1 and st["doubling_rounds"] == 0
    assert st["arena_bytes"] <= 27 * n + (160 << 20), st["arena_bytes"] / n          # 26.2 N + the fixed tables and the buffers' slack (145 MB)
    assert archon.validate(x, sa)
    y = S.gen_text(1 << 20)
    P, B, b0 = oracle.forward(y)
    sa, bwt, base = archon.forward(y)
    st2 = archon.stats()
    assert (sa == P).all() and (bwt == B).all() and base == b0
    assert st2["arena_bytes"] > 60 * y.size


def _word_soup(n, vocab, word_len, seed, skew=1.3, alphabet=12):
    """tokens drawn (Zipf-like) from a small vocabulary of fixed-length words over a small alphabet: after the first stage the
    tied groups are the (word, offset) classes -- thousands to tens of thousands of rows each -- and every doubling round
    splits them by the words in front: groups of every size class, each handing groups down to the smaller ones"""
    rng = np.random.default_rng(seed)
    words = rng.integers(97, 97 + alphabet, size=(vocab, word_len)).astype(np.uint8)
    pr = 1.0 / np.arange(1, vocab + 1) ** skew
    pr /= pr.sum()
    toks = rng.choice(vocab, size=n // word_len + 1, p=pr)
    return words[toks].reshape(-1)[:n].copy()


@pytest.mark.parametrize("n,vocab,wl", [(3 << 20, 24, 16), ((5 << 20) + 12345, 7, 24), (1 << 21, 150, 9)])
def test_mid_groups(archon, oracle, monkeypatch, n, vocab, wl):
    """mid_rounds.hiph: groups of 1025 .. 16384 rows sorted by one workgroup in LDS, longer ones through the global sort,
    survivors handed down from class to class (big -> large -> small -> S list) -- on both first-stage routes, with the
    text rounds on and off, and against the old route (NO_MID) that sends every long group through the global sort."""
    x = _word_soup(n, vocab, wl, 7 + vocab)
    P, B, b0 = oracle.forward(x)
    seen_mid = 0
    for env in ({}, {"ARCHON_FORCE_PATH": "0"}, {"ARCHON_NO_TEXT_ROUNDS": "1"}, {"ARCHON_NO_RANK_WRITER": "1", "ARCHON_NO_PAIR_CHAINS": "1"},
                {"ARCHON_NO_MID": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        for k in env:
            monkeypatch.delenv(k)
        assert (sa == P).all(), (env, int(np.argmax(sa != P)))
        assert (bwt == B).all() and base == b0, env
        if "ARCHON_NO_MID" in env:
            assert st["mid_items"] == 0
        else:
            seen_mid += st["mid_items"]
    assert seen_mid > 0


def test_mid_groups_at_the_class_limits(archon, oracle, monkeypatch):
    """groups of EXACTLY 512 / 513 / 4096 / 4097 / 16384 / 16385 rows (and their neighbours; 1024 / 1025: the S list's limit in round 3): distinct 16-byte words that occur
    exactly that often, in random order -- the rows of a word's offsets 7 .. 15 are tied in groups of the word's count after the
    first stage, on the boundaries between the S list, the two mid classes (a full LDS image: the end mark of the group falls
    behind the last bitmap word) and the global path"""
    rng = np.random.default_rng(16384)
    counts = [16384, 16385, 16383, 4096, 4097, 4095, 1025, 1024, 1023, 513, 512, 511, 8192, 2048, 2, 3, 1, 40000]
    words = rng.integers(0, 256, size=(len(counts), 16)).astype(np.uint8)
    toks = np.repeat(np.arange(len(counts)), counts)
    rng.shuffle(toks)
    x = words[toks].reshape(-1).copy()
    P, B, b0 = oracle.forward(x)
    for env in ({}, {"ARCHON_FORCE_PATH": "0"}, {"ARCHON_NO_TEXT_ROUNDS": "1"}, {"ARCHON_FORCE_PATH": "0", "ARCHON_NO_TEXT_ROUNDS": "1", "ARCHON_NO_RANK_WRITER": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        for k in env:
            monkeypatch.delenv(k)
        assert (sa == P).all(), (env, int(np.argmax(sa != P)))
        assert (bwt == B).all() and base == b0, env
    assert st["mid_items"] > 0 and st["seg_big_items"] > 0


@pytest.mark.streaming_machinery
@pytest.mark.parametrize("sigma", [2, 3, 4, 5, 9, 16])
@pytest.mark.parametrize("n", [70001, 1 << 20])
def test_compacted_alphabet_streaming(archon, oracle, sigma, n):
    """<= 16 distinct bytes -> key bytes of 8/4/2 symbols (k_build_y) through the streaming
    stage.  Symbols are spread over the byte range so that the order-preserving recode matters; the block
    starts with a run of the largest symbol (ties with the end-of-string padding)."""
    rng = np.random.default_rng(100 + sigma)
    alpha = np.sort(rng.choice(256, size=sigma, replace=False)).astype(np.uint8)
    alpha[-1] = 255 if sigma % 2 else alpha[-1]
    x = alpha[rng.integers(0, sigma, size=n)]
    x[:37] = alpha[-1]
    sa, bwt, base = archon.forward(x)
    st = archon.stats()
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all()
    assert (bwt == B).all() and base == b0
    # at most 4 distinct bytes: the count's probe sends the block to the recode at once; otherwise two-byte buckets
    # that are small enough already (n / sigma^2 under the 4608-item cap) keep plain bytes
    want_bits = (1 if sigma == 2 else 2) if sigma <= 4 else (0 if n / sigma**2 < 4400 else 4)
    assert st["path"] == 1 and st["alphabet_bits"] == want_bits


@pytest.mark.streaming_machinery
def test_compacted_alphabet_with_repeats(archon, oracle):
    """small alphabet + repeated material: streaming stage on packed bytes, then oversize buckets / deep ties
    go to the doubling stage with the depth counted in symbols"""
    rng = np.random.default_rng(77)
    acgt = np.frombuffer(b"ACGT", np.uint8)
    chunk = acgt[rng.integers(0, 4, size=50000)]
    x = np.concatenate([acgt[rng.integers(0, 4, size=400000)], chunk, acgt[rng.integers(0, 4, size=1000)], chunk,
                        np.full(3000, ord("A"), np.uint8), chunk[:20000], acgt[rng.integers(0, 4, size=300000)]])
    sa, bwt, base = archon.forward(x)
    st = archon.stats()
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all()
    assert (bwt == B).all() and base == b0
    assert st["path"] == 1 and st["alphabet_bits"] == 2 and st["doubling_rounds"] > 0


@pytest.mark.streaming_machinery
def test_alphabet_hint_between_blocks(archon, oracle):
    """a context whose last block had <= 4 distinct bytes looks at the next block's alphabet before counting it (archon_hip.hip,
    hint_poor_alphabet): the same order and the same route whatever block came before -- DNA after DNA, then blocks that break the
    promise (256 symbols, 5 symbols, 4 symbols and one stray byte at the end), clean periodic blocks, DNA again"""
    rng = np.random.default_rng(5)
    n = (1 << 20) + 77
    acgt = np.frombuffer(b"ACGT", np.uint8)
    dna = lambda: acgt[rng.integers(0, 4, size=n)]
    stray = dna()
    stray[-3] = 0x80
    blocks = [("dna", dna(), 1, 2), ("dna", dna(), 1, 2), ("random", rng.integers(0, 256, size=n, dtype=np.uint8), 1, 0),
              ("dna", dna(), 1, 2), ("five", np.frombuffer(b"ACGTN", np.uint8)[rng.integers(0, 5, size=n)], 1, None),
              ("dna", dna(), 1, 2), ("stray", stray, 1, None), ("dna", dna(), 1, 2), ("a", np.full(n, 65, np.uint8), 2, None),
              ("dna", dna(), 1, 2), ("ab", np.tile(np.frombuffer(b"ab", np.uint8), n // 2 + 1)[:n].copy(), 2, None),
              ("binary", acgt[rng.integers(0, 2, size=n)], 1, 1), ("dna", dna(), 1, 2)]
    routes = {}
    for name, x, path, bits in blocks:
        x = np.ascontiguousarray(x)
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        P, B, b0 = oracle.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, name
        assert st["path"] == path and (bits is None or st["alphabet_bits"] == bits), (name, st)
        routes.setdefault(name, []).append((st["path"], st["alphabet_bits"]))
    assert len(set(routes["dna"])) == 1
    # the second DNA block of a pair skips the count it would abandon
    archon.forward(np.ascontiguousarray(blocks[2][1]))
    x = np.ascontiguousarray(dna())
    archon.forward(x)
    first = archon.stats()["kernel_launches"]
    archon.forward(x)
    assert archon.stats()["kernel_launches"] < first


@pytest.mark.parametrize("shape", ["random", "a"])
def test_max_block(archon, shape):
    """the largest block the boundary accepts (MAX_N = 0x3FFFFF00 bytes, just under 1 GiB; the reference's own
    tracking path needs N < 2^30, archon.cpp:802): size-independent properties, device-resident buffers."""
    import torch
    n = archon.MAX_N
    x = S.gen_random(n) if shape == "random" else np.full(n, 97, np.uint8)
    x_t = torch.from_numpy(x).cuda()
    del x
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(x_t, sa_t, bwt_t, base_t)
    st = archon.stats()
    # beyond ~300 MB the two-byte buckets of even a uniform block exceed the in-LDS sort (4608 items): 7-pass route;
    # a clean periodic block is written down in closed form whatever its size (path 2)
    assert st["path"] == (0 if shape == "random" else 2)
    base = int(base_t.item())
    assert int(sa_t[base].item()) == n
    if shape == "a":      # a^N: the order is N, N-1, ..., 1
        assert torch.equal(sa_t, torch.arange(n, 0, -1, dtype=torch.int32, device="cuda")) and base == 0
        assert st["doubling_rounds"] == 0 and st["period"] == 1 and st["chain_items"] == n
    assert archon.validate_dev(x_t, sa_t)
    del sa_t
    out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    archon.inverse_dev(bwt_t, base, out_t)
    assert torch.equal(out_t, x_t)


def test_sa_to_bwt_standalone(archon, oracle):
    """A7 on its own (Archon::enWrite's gather): BWT + primary index from a caller's SA; foreign SAs are rejected."""
    for shape, n in (("text", 100003), ("random", 1 << 20), ("a", 777)):
        x = S.gen_shape(shape, n)
        P, B, b0 = oracle.forward(x)
        bwt, base = archon.sa_to_bwt(x, P)
        assert (bwt == B).all() and base == b0
    x = S.gen_text(1000)
    P, _, _ = oracle.forward(x)
    for bad in (np.where(P == 1000, 999, P), np.where(P == 5, 0, P), np.where(P == 7, 1001, P)):
        with pytest.raises(archon.ArchonError) as e:
            archon.sa_to_bwt(x, bad.astype(np.uint32))
        assert e.value.code == archon.E_CORRUPT


def test_pass_b_bucket_mode_matches_range_mode(archon, oracle, monkeypatch):
    """pass B deals whole second-byte buckets to workgroups when they are balanced (n >= 16 Mi), equal tile ranges
    otherwise: both must give the oracle's SA, also on a block whose buckets are balanced but not uniform."""
    rng = np.random.default_rng(12)
    n = (1 << 24) + 12345
    x_uniform = S.gen_random(n)
    # second bytes balanced (every value equally often), first bytes skewed towards small values
    x_mixed = np.where(np.arange(n) % 2 == 0, rng.integers(0, 256, n), np.minimum(rng.integers(0, 256, n), rng.integers(0, 256, n))).astype(np.uint8)
    for x in (x_uniform, x_mixed):
        P = oracle.sa(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all()
        monkeypatch.setenv("ARCHON_NO_ALIGNED", "1")
        sa2, bwt2, base2 = archon.forward(x)
        monkeypatch.delenv("ARCHON_NO_ALIGNED")
        assert (sa2 == P).all() and (bwt2 == bwt).all() and base2 == base


@pytest.mark.parametrize("shape", ["random", "dna"])
@pytest.mark.parametrize("n", [1 << 24, (1 << 24) + 1, 23456789, (1 << 25) + 16383, 100000007])
def test_mid_sizes_lf_consistent(archon, shape, n):
    """block sizes between the oracle-checked ones and the full block, tile counts that do not divide evenly among
    the 256 workgroups, bucket mode of pass B on byte and on recoded (2-bit) keys: LF-consistency + round trip"""
    import torch
    x_t = torch.from_numpy(S.gen_shape(shape, n)).cuda()
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(x_t, sa_t, bwt_t, base_t)
    assert archon.stats()["path"] == 1
    assert archon.validate_dev(x_t, sa_t)
    out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    archon.inverse_dev(bwt_t, int(base_t.item()), out_t)
    assert torch.equal(out_t, x_t)


@pytest.mark.streaming_machinery
@pytest.mark.parametrize("n,heavy", [(1 << 20, 0), (1 << 20, 700), (1 << 20, 3000), (24 << 20, 0), (24 << 20, 600), (70 << 20, 0), (70 << 20, 1700),
                                     (70 << 20, 6000)])
def test_bucket_sort_instances(archon, oracle, n, heavy):
    """the bucket sort's short instances (k_local_sort<1>, <3>: blocks whose largest two-byte bucket holds at most 512 / 1536 rows) and
    the general one behind them: uniform blocks of 16, 384 and 1120 rows per bucket, and the same blocks with ONE bucket pushed over the
    short instance's limit (the count's largest bucket sends the block to the general instance) or over the sort's capacity"""
    import torch
    rng = np.random.default_rng(n + heavy)
    x = rng.integers(0, 256, size=n, dtype=np.uint8)
    if heavy:
        at = rng.choice(n // 8 - 1, size=heavy, replace=False) * 8 + 3
        x[at] = 0x41
        x[at + 1] = 0x42                      # `heavy` more rows in the bucket of "AB"
    if n <= (1 << 20):
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0
    else:
        x_t = torch.from_numpy(x).cuda()
        sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
        bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
        base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
        archon.forward_dev(x_t, sa_t, bwt_t, base_t)
        assert archon.validate_dev(x_t, sa_t)
    assert archon.stats()["path"] == 1


@pytest.mark.streaming_machinery
@pytest.mark.parametrize("ranges", ["1024", "512", "300", "7"])
def test_many_pass_ranges(archon, oracle, monkeypatch, ranges):
    """bench.py cuts the passes into 1024 ranges for N > 1 (shorter tails when RCCL holds CUs): prefix-summed range
    tables (k_col_prefix), the two-byte count covering several pass ranges per workgroup, pass B in range mode."""
    monkeypatch.setenv("ARCHON_PASS_RANGES", ranges)
    monkeypatch.setenv("ARCHON_NO_ALIGNED", "1")
    for shape, n in (("random", (1 << 24) + 4321), ("random", 3000001), ("dna", (1 << 23) + 99), ("random", 70000)):
        x = S.gen_shape(shape, n)
        P = oracle.sa(x)
        sa, bwt, base = archon.forward(x)
        assert archon.stats()["path"] == 1
        assert (sa == P).all(), (shape, n)
    # a size whose tile count is a multiple of 256: the two-byte count then covers several pass ranges per workgroup
    x = S.gen_random(1 << 25)
    sa, bwt, base = archon.forward(x)
    assert archon.stats()["path"] == 1 and archon.validate(x, sa)
    assert (archon.inverse(bwt, base) == x).all()


def test_lms_select(archon, oracle):
    """the subset a7 sorts directly (Constructor::findLMS, archon.cpp:160-172) as a GPU operator -- per-bucket
    counts and the items in a7's placement order, against the oracle's restatement of the serial scan."""
    rng = np.random.default_rng(21)
    cases = [S.gen_shape(sh, n) for sh in ("random", "dna", "text", "a", "ab", "motif") for n in (1, 2, 3, 1000, 65537, 1 << 20)]
    cases += [rng.integers(0, 3, 100001).astype(np.uint8), np.arange(256, dtype=np.uint8).repeat(5), np.arange(255, -1, -1, dtype=np.uint8).repeat(7)]
    for x in cases:
        count, items = archon.lms_select(x)
        c0, i0 = oracle.lms_select(x)
        assert (count == c0).all() and items.size == i0.size and (items == i0).all(), (x.size, x[:8])
    x = S.gen_random(1 << 22)
    _, items = archon.lms_select(x)
    assert abs(items.size / x.size - 1 / 3) < 0.01          # LMS density 0.33 on random bytes
    # ... and against the reference itself: digests of the placement a7's own findLMS produced (tests/golden/golden.json,
    # written by make_golden.py from oracle/_ref/a7lms), no oracle in between
    import json, os
    import oracle_binding as OB
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden.json")) as f:
        golden = json.load(f)
    for case in golden["cases"]:
        count, items = archon.lms_select(S.gen_shape(case["shape"], case["n"]))
        assert items.size == case["lms_n1"] and OB.lms_digest(count, items) == case["sha256_lms"], (case["shape"], case["n"])


def test_two_byte_count_hot_bins(archon, oracle):
    """k_hist16 keeps 16-bit LDS counters that are cut back (and booked in a spill table) whenever they cross a multiple of
    16 384: blocks whose two-byte bins hold far more than 65 535 items per workgroup must still be counted exactly --
    five symbols (the low-entropy probe lets them through) with one of them 97 % of the block, and a skewed 30-symbol text."""
    rng = np.random.default_rng(8)
    n = 3 << 20
    x = rng.choice(np.array([3, 50, 51, 52, 250], np.uint8), size=n, p=[0.97, 0.01, 0.01, 0.005, 0.005])
    sa, bwt, base = archon.forward(x)
    P, B, b0 = oracle.forward(x)
    assert (sa == P).all() and (bwt == B).all() and base == b0
    p = 0.5 ** np.arange(1, 31); p /= p.sum()
    y = rng.choice(np.arange(60, 90, dtype=np.uint8), size=n, p=p)
    sa, bwt, base = archon.forward(y)
    P, B, b0 = oracle.forward(y)
    assert (sa == P).all() and (bwt == B).all() and base == b0


@pytest.mark.parametrize("shape,n", [("text", 300001), ("random", 1 << 20), ("dna", 250000), ("a", 90001), ("motif", 200003)])
def test_misaligned_device_buffers(archon, oracle, shape, n):
    """caller buffers at odd addresses (text + 3 bytes, SA + 1 word, BWT + 1 byte): every route has to fall back from its
    16-byte loads and stores (the library copies a misaligned text; SA and BWT are written where they are)"""
    import torch
    x = S.gen_shape(shape, n)
    if shape == "dna":
        x = np.concatenate([x[:100000], x[:100000], x[:50000]])      # repeats: the streaming stage leaves ties for the rounds
    xb = torch.zeros(n + 16, dtype=torch.uint8, device="cuda")
    xb[3:3 + n] = torch.from_numpy(x).cuda()
    sab = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
    bwb = torch.zeros(n + 8, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(xb[3:3 + n], sab[1:1 + n], bwb[1:1 + n], base_t)
    P, B, b0 = oracle.forward(x)
    assert (sab[1:1 + n].cpu().numpy().view(np.uint32) == P).all()
    assert (bwb[1:1 + n].cpu().numpy() == B).all() and int(base_t.item()) == b0
    assert int(sab[0].item()) == 0 and int(sab[1 + n].item()) == 0 and int(bwb[0].item()) == 0 and int(bwb[1 + n].item()) == 0


@pytest.mark.parametrize("ranges", ["300", "1024"])
def test_periodic_blocks_with_many_pass_ranges(archon, oracle, monkeypatch, ranges):
    """the N > 1 configuration of bench.py (many pass ranges, no bucket mode) on the periodic route: hot-digit ranking,
    deferred buckets and the run shortcut with the passes cut into odd ranges"""
    monkeypatch.setenv("ARCHON_PASS_RANGES", ranges)
    monkeypatch.setenv("ARCHON_NO_ALIGNED", "1")
    monkeypatch.setenv("ARCHON_NO_CLOSED_FORM", "1")
    rng = np.random.default_rng(5)
    for x in (np.tile(np.frombuffer(b"ab", np.uint8), 1 << 20), np.full(3000001, 97, np.uint8),
              np.tile(rng.integers(0, 256, size=37, dtype=np.uint8), 60000)):
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        P, B, b0 = oracle.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0
        assert st["period"] in (1, 2, 37) and st["doubling_rounds"] == 0


def test_tie_heavy_block_full_size(archon):
    """two copies of a 128 MiB random block: every row is tied with its twin beyond the streaming stage's five key bytes, so
    every 16-bit bucket lists ~2048 tie groups -- nearly as many as its LDS list can hold: the per-bucket booking, the
    tie-list cap, the run shortcut at distance N/2 and the rounds behind it.  Size-independent checks."""
    import torch
    half = S.gen_random(128 << 20)
    x = np.concatenate([half, half])
    n = x.size
    x_t = torch.from_numpy(x).cuda()
    del x, half
    sa_t = torch.empty(n, dtype=torch.int32, device="cuda")
    bwt_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    base_t = torch.zeros(1, dtype=torch.int32, device="cuda")
    archon.forward_dev(x_t, sa_t, bwt_t, base_t)
    st = archon.stats()
    assert st["path"] == 1 and st["tie_items"] >= n // 2 - 16
    assert st["ms_local_sort"] < 10.0          # one global atomic per tie GROUP made this 70+ ms
    assert archon.validate_dev(x_t, sa_t)
    out_t = torch.empty(n, dtype=torch.uint8, device="cuda")
    archon.inverse_dev(bwt_t, int(base_t.item()), out_t)
    assert torch.equal(out_t, x_t)


def _dup_cases():
    rng = np.random.default_rng(91)
    r = lambda k, hi=256: rng.integers(0, hi, size=k, dtype=np.uint8)
    R = r(1 << 20)
    words = [bytes(r(int(k), 26) + 97) for k in rng.integers(2, 9, size=400)]
    prose = np.frombuffer(b" ".join(words[i] for i in rng.integers(0, 400, size=200000)), np.uint8)
    return {
        # random block with one long duplicate: every tied group is a pair {s, s + d} of ONE passage
        "one_duplicate": np.concatenate([R, r(11), R[100000:900000], r(5)]),
        # three copies of a passage (groups of three at first, pairs once the copies' left contexts differ), nested copies
        "three_copies": np.concatenate([R[:300000], r(3), R[:300000], r(7), R[:200000], R[50000:250000]]),
        # text with copied passages next to many short repeats: pairs are listed while longer groups keep doubling
        "prose_with_copies": np.concatenate([prose[:400000], prose[100000:350000], prose[400000:600000], prose[120000:300000]]),
        # a duplicate whose predecessors sit in a LONGER group (the run's head cannot be settled at first): 0-runs in front of the copies
        "duplicate_behind_runs": np.concatenate([np.zeros(5000, np.uint8), R[:200000], np.zeros(5000, np.uint8), R[:200000], np.zeros(4000, np.uint8), R[:100000]]),
    }


@pytest.mark.parametrize("name", sorted(_dup_cases()))
def test_pair_chains(archon, oracle, name, monkeypatch):
    """Long duplicates (a4's anchors, bwt/a4/src/direct.c:90-161): the pairs of a repeated passage are listed, sorted by item
    and settled passage by passage (k_pair_heads / k_pair_apply) -- same order as the oracle, with and without the shortcut
    and with the run shortcut of the periodic blocks out of the way."""
    x = _dup_cases()[name]
    P, B, b0 = oracle.forward(x)
    used = 0
    for env in ({}, {"ARCHON_NO_CHAINS": "1"}, {"ARCHON_NO_CHAINS": "1", "ARCHON_FORCE_PATH": "0"}, {"ARCHON_NO_PAIR_CHAINS": "1", "ARCHON_NO_CHAINS": "1"}):
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        sa, bwt, base = archon.forward(x)
        st = archon.stats()
        for k in env:
            monkeypatch.delenv(k)
        assert (sa == P).all(), (name, env)
        assert (bwt == B).all() and base == b0, (name, env)
        if "ARCHON_NO_PAIR_CHAINS" in env:
            assert st["chain_pairs"] == 0
        else:
            used += st["chain_pairs"]
    if name == "one_duplicate":
        assert used > 0          # the passage was settled as pairs on at least one route


def test_rank_writer_matches_direct_stores(archon, monkeypatch):
    """rank updates dealt by item into windows of the table (rank_writer.hiph) against one store per update: the same
    suffix array, on a block that keeps the writer busy for several rounds (the reference digest of prose-16Mi is in
    golden.json; here the two routes are compared directly)."""
    x = S.gen_shape("prose", 1 << 24)
    sa1, bwt1, base1 = archon.forward(x)
    assert archon.stats()["doubling_rounds"] > 4
    monkeypatch.setenv("ARCHON_NO_RANK_WRITER", "1")
    sa2, bwt2, base2 = archon.forward(x)
    monkeypatch.delenv("ARCHON_NO_RANK_WRITER")
    assert (sa1 == sa2).all() and (bwt1 == bwt2).all() and base1 == base2
    assert archon.validate(x, sa1)
