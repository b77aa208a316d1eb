"""Synthetic block generators for the BASELINE.json configs.

All generators are pure integer arithmetic on the splitmix64 stream so that a C
restatement is trivial (see `splitmix64_words`); seeds are
``SEED_BASE + config_id`` (+ block index where a config has several blocks).

    cfg 1  text     one stream word per byte -> Zipf-ranked 96-symbol table
    cfg 2  random   successive little-endian bytes of the stream
    cfg 3  repeats  'a'*N, 'ab'*N/2, a 1000-byte random motif repeated
    cfg 4  dna      "ACGT"[w >> 62], one stream word per symbol
"""
import numpy as np

SEED_BASE = 0x20261003
_GOLDEN = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)
_CHUNK = 1 << 22  # words per chunk: bounds temporary memory to ~100 MB


def splitmix64_words(seed, start, count):
    """Words start..start+count-1 of splitmix64(seed): word i mixes seed+(i+1)*GOLDEN."""
    with np.errstate(over="ignore"):
        idx = np.arange(start + 1, start + count + 1, dtype=np.uint64)
        z = np.uint64(seed & 0xFFFFFFFFFFFFFFFF) + idx * _GOLDEN
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def gen_random(n, seed=SEED_BASE + 2):
    """cfg 2: uniform random bytes = successive LE bytes of the stream."""
    out = np.empty(n, dtype=np.uint8)
    nwords = (n + 7) // 8
    for w0 in range(0, nwords, _CHUNK):
        cnt = min(_CHUNK, nwords - w0)
        b = splitmix64_words(seed, w0, cnt).astype("<u8").view(np.uint8)
        lo = w0 * 8
        hi = min(n, lo + cnt * 8)
        out[lo:hi] = b[: hi - lo]
    return out


def gen_dna(n, seed=SEED_BASE + 4):
    """cfg 4: "ACGT"[w >> 62], one word per symbol."""
    table = np.frombuffer(b"ACGT", dtype=np.uint8)
    out = np.empty(n, dtype=np.uint8)
    for w0 in range(0, n, _CHUNK):
        cnt = min(_CHUNK, n - w0)
        w = splitmix64_words(seed, w0, cnt)
        out[w0 : w0 + cnt] = table[(w >> np.uint64(62)).astype(np.intp)]
    return out


# rank order of the 96-symbol "enwik-style" table: weight(rank r) = 360360 // (r + 1)
_TEXT_RANKED = (
    b" etaoinshrdlcumwfgypbvkjxqz\n,.ETAOINSHRDLCUMWFGYPBVKJXQZ0123456789"
)
_TEXT_RANKED += bytes(c for c in range(33, 127) if c not in _TEXT_RANKED)
assert len(_TEXT_RANKED) == 96 and len(set(_TEXT_RANKED)) == 96
_TEXT_SYMS = np.frombuffer(_TEXT_RANKED, dtype=np.uint8)
_TEXT_CUM = np.cumsum(np.array([360360 // (r + 1) for r in range(96)], dtype=np.uint64))


def gen_text(n, seed=SEED_BASE + 1):
    """cfg 1: order-0 Zipf-ranked printable text; symbol = first rank r with
    cum[r] > (u * total) >> 16 where u = top 16 bits of the stream word."""
    out = np.empty(n, dtype=np.uint8)
    total = _TEXT_CUM[-1]
    for w0 in range(0, n, _CHUNK):
        cnt = min(_CHUNK, n - w0)
        u = splitmix64_words(seed, w0, cnt) >> np.uint64(48)
        v = (u * total) >> np.uint64(16)
        r = np.searchsorted(_TEXT_CUM, v, side="right")
        out[w0 : w0 + cnt] = _TEXT_SYMS[r]
    return out


def gen_repeat(n, motif):
    """cfg 3: `motif` (bytes) repeated and truncated to n bytes."""
    m = np.frombuffer(bytes(motif), dtype=np.uint8)
    reps = (n + len(m) - 1) // len(m)
    return np.tile(m, reps)[:n].copy()


def gen_motif(n, motif_len=1000, seed=SEED_BASE + 3):
    """cfg 3c: a `motif_len`-byte random motif repeated."""
    return gen_repeat(n, gen_random(motif_len, seed).tobytes())


SHAPES = ("text", "random", "dna", "a", "ab", "motif")


def gen_shape(shape, n, block=0):
    """One block of a named shape; `block` offsets the seed (cfg 4: seed + b)."""
    if shape == "text":
        return gen_text(n, SEED_BASE + 1 + block)
    if shape == "random":
        return gen_random(n, SEED_BASE + 2 + block)
    if shape == "dna":
        return gen_dna(n, SEED_BASE + 4 + block)
    if shape == "a":
        return gen_repeat(n, b"a")
    if shape == "ab":
        return gen_repeat(n, b"ab")
    if shape == "motif":
        return gen_motif(n, 1000, SEED_BASE + 3 + block)
    raise ValueError(shape)
