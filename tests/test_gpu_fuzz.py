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
