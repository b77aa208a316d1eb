"""GPU: randomized parity sweep -- many small and medium blocks of varied alphabets, run lengths and
sizes through BOTH first stages (streaming and 7-pass), SA/BWT/baseId bit-exact with the oracle,
inverse round trip on the device."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _cases(rng, count):
    for _ in range(count):
        n = int(rng.choice([1, 2, 3, 7, 8, 9, 63, 64, 65, 255, 4097, 16385, 50000, 200001]))
        n = max(1, n + int(rng.integers(-3, 4)))
        kind = rng.integers(0, 6)
        if kind == 0:
            x = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1:
            k = int(rng.choice([1, 2, 3, 4, 5, 16, 17]))
            x = rng.choice(rng.integers(0, 256, size=k, dtype=np.uint8), size=n)
        elif kind == 2:      # long runs
            x = np.repeat(rng.integers(0, 256, size=max(1, n // 50 + 1), dtype=np.uint8), 50)[:n]
        elif kind == 3:      # periodic with a defect
            m = int(rng.integers(1, 40))
            x = np.tile(rng.integers(0, 4, size=m, dtype=np.uint8) + 250, n // m + 1)[:n].copy()
            x[int(rng.integers(0, n))] ^= 1
        elif kind == 4:      # 0xFF-heavy (the end-of-string rule)
            x = rng.choice(np.array([253, 254, 255], np.uint8), size=n, p=[0.05, 0.15, 0.8])
        else:                # two-byte-context skew: a few hot pairs
            x = rng.choice(np.frombuffer(b"ab", np.uint8), size=n, p=[0.95, 0.05])
        yield np.ascontiguousarray(x, np.uint8)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_fuzz_both_paths(archon, oracle, seed):
    rng = np.random.default_rng(seed)
    for x in _cases(rng, 40):
        P, B, b0 = oracle.forward(x)
        for path in ("0", "1"):
            os.environ["ARCHON_FORCE_PATH"] = path
            try:
                sa, bwt, base = archon.forward(x)
            finally:
                del os.environ["ARCHON_FORCE_PATH"]
            assert (sa == P).all() and (bwt == B).all() and base == b0, (path, x.size, x[:16])
        assert (archon.inverse(B, b0) == x).all()
        assert archon.validate(x, P)


def _natural_cases(rng, count):
    """bigger blocks, structures that steer the un-forced route: probe / recode, skip flag, run shortcut, doubling"""
    for _ in range(count):
        n = int(rng.choice([3000, 70000, 300000, 1 << 20, (1 << 21) + 5]))
        n += int(rng.integers(-50, 51))
        kind = int(rng.integers(0, 7))
        if kind == 0:        # uniform bytes
            x = rng.integers(0, 256, size=n, dtype=np.uint8)
        elif kind == 1:      # small alphabet, maybe with a rare extra symbol (defeats the 4-symbol probe late in the block)
            k = int(rng.choice([2, 3, 4, 5, 12, 16, 17, 40]))
            x = rng.choice(np.sort(rng.choice(256, size=k, replace=False)).astype(np.uint8), size=n)
            if rng.integers(0, 2):
                x[int(rng.integers(n // 2, n))] = np.uint8(rng.integers(0, 256))
        elif kind == 2:      # a motif repeated, with a few point defects
            m = int(rng.choice([1, 2, 3, 7, 100, 1000, 4099]))
            x = np.tile(rng.integers(0, 256, size=m, dtype=np.uint8), n // m + 1)[:n].copy()
            for _ in range(int(rng.integers(0, 4))):
                x[int(rng.integers(0, n))] ^= np.uint8(1 + rng.integers(0, 255))
        elif kind == 3:      # random text with a long duplicated chunk (deep ties next to shallow ones)
            x = rng.integers(97, 123, size=n, dtype=np.uint8)
            L = int(rng.integers(10, max(11, n // 3)))
            a, b = int(rng.integers(0, n - L)), int(rng.integers(0, n - L))
            x[b:b + L] = x[a:a + L].copy()
        elif kind == 4:      # runs of random length
            vals = rng.integers(0, 256, size=n // 20 + 2, dtype=np.uint8)
            x = np.repeat(vals, rng.integers(1, 40, size=vals.size))[:n]
            if x.size < n:
                x = np.concatenate([x, rng.integers(0, 256, size=n - x.size, dtype=np.uint8)])
        elif kind == 5:      # two halves of different character
            h = n // 2
            x = np.concatenate([rng.choice(np.frombuffer(b"ACGT", np.uint8), size=h), rng.integers(0, 256, size=n - h, dtype=np.uint8)])
        else:                # 0xFF-heavy with structure
            x = np.where(rng.random(n) < 0.9, 255, rng.integers(250, 256, size=n)).astype(np.uint8)
        yield np.ascontiguousarray(x, np.uint8)


@pytest.mark.small_block_default
@pytest.mark.parametrize("seed", [11, 12, 13])
def test_fuzz_natural_route(archon, oracle, seed):
    """the un-forced route, INCLUDING the product's own choice for small blocks (byte count + LSB passes below 8 MiB)"""
    rng = np.random.default_rng(seed)
    for x in _natural_cases(rng, 12):
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (x.size, x[:16], archon.stats())
        assert (archon.inverse(B, b0) == x).all()


@pytest.mark.parametrize("seed", [21, 22])
def test_fuzz_natural_route_streaming_machinery(archon, oracle, seed, monkeypatch):
    """the same generator through the graded machinery (small-block rule off: the default of this suite), with bucket mode and its
    range-relative records allowed from 64 KiB on"""
    monkeypatch.setenv("ARCHON_ALIGNED_MIN", "65536")
    monkeypatch.setenv("ARCHON_REL_MIN_SEG", "1")
    rng = np.random.default_rng(seed)
    for x in _natural_cases(rng, 12):
        P, B, b0 = oracle.forward(x)
        sa, bwt, base = archon.forward(x)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (x.size, x[:16], archon.stats())


def defect_case(seed):
    """a motif of a random period with defects: points, twin defects at one phase, a burst, the first and the last period,
    a second motif of the same period behind the first, random bytes in front (tools/fuzz_defects.py runs many more seeds)"""
    rng = np.random.default_rng(1000 + seed)
    p = int(rng.choice([1, 2, 3, 5, 7, 8, 16, 31, 64, 100, 255, 256, 257, 1000, 1024, 4099, 30000]))
    n = int(rng.integers(max(4 * p, 5000), 1500000))
    sigma = int(rng.choice([2, 3, 4, 16, 256]))
    motif = rng.integers(0, sigma, size=p, dtype=np.uint8)
    x = np.tile(motif, n // p + 2)[:n].copy()
    kind = int(rng.integers(0, 6))
    pos = rng.integers(0, n, size=int(rng.integers(1, 9)))
    if kind == 1:
        q = int(rng.integers(0, n // 2))
        pos = np.array([q, q + p * int(rng.integers(1, max(2, (n - q) // p)))])
        pos = pos[pos < n]
    elif kind == 2:
        q = int(rng.integers(0, n - 10))
        pos = np.arange(q, q + int(rng.integers(2, 9)))
    elif kind == 3:
        pos = np.array([int(rng.integers(0, min(p, n))), n - 1 - int(rng.integers(0, min(p, n)))])
    for q in pos:
        x[q] = (int(x[q]) + 1 + int(rng.integers(0, max(1, sigma - 1)))) % sigma
    if kind == 4:
        h = int(rng.integers(n // 4, 3 * n // 4))
        x[h:] = np.tile(rng.integers(0, sigma, size=p, dtype=np.uint8), (n - h) // p + 2)[:n - h]
    if kind == 5:
        a = int(rng.integers(0, n // 3))
        x[:a] = rng.integers(0, 256, size=a, dtype=np.uint8)
    return np.ascontiguousarray(x, np.uint8)


@pytest.mark.parametrize("seed", range(16))
def test_fuzz_period_defects(archon, oracle, seed):
    """the period-defect rounds (break_key / cont_key / k_zone_certify) on random defect patterns, both first stages"""
    x = defect_case(seed)
    P, B, b0 = oracle.forward(x)
    for path in (None, "0", "1"):
        if path is not None:
            os.environ["ARCHON_FORCE_PATH"] = path
        try:
            sa, bwt, base = archon.forward(x)
        finally:
            os.environ.pop("ARCHON_FORCE_PATH", None)
        assert (sa == P).all() and (bwt == B).all() and base == b0, (seed, path, x.size)
