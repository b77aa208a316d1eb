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


# ---- "prose": natural-text-like block in the deep-LCP regime ----------------------------
# Sentences drawn from a Zipf-ranked set of templates over a Zipf-ranked vocabulary (most slots keep the
# template's word, some take one of four alternatives); one paragraph in eight is one of 256 fixed boilerplate
# paragraphs (licence headers); then a number of long passages are copied to other places (vendored copies).  Measured on 64 MiB (tools/lcp_stats.c): see DESIGN.md 3.2.  Pure integer
# arithmetic on splitmix64 words, vectorised per chunk of sentences.
_PROSE_V = 4096          # vocabulary
_PROSE_T = 2048          # sentence templates
_PROSE_SLOTS = 16
_LETTERS = np.frombuffer(b"etaoinshrdlcumwfgypbvkjxq", dtype=np.uint8)      # 25 letters, index (p*q)//39 with p,q < 32
_prose_tables = {}


def _prose_build(seed):
    if seed in _prose_tables:
        return _prose_tables[seed]
    u = np.uint64
    # vocabulary: word w = 2..9 letters + ' '; row _PROSE_V is the sentence end ".\n"
    r = splitmix64_words(seed ^ 0x5EED0001, 0, 2 * _PROSE_V).reshape(_PROSE_V, 2)
    wl = (2 + (r[:, 0] & u(7))).astype(np.int64)
    words = np.full((_PROSE_V + 1, 10), ord(" "), dtype=np.uint8)
    for k in range(9):
        p = (r[:, 1] >> u(10 * (k % 6))) & u(31)
        q = ((r[:, 1] >> u(10 * (k % 6) + 5)) ^ (r[:, 0] >> u(8 + 3 * k))) & u(31)
        words[:_PROSE_V, k] = _LETTERS[((p * q) // u(39)).astype(np.intp)]
    cols = np.arange(10)[None, :]
    words[:_PROSE_V][cols >= wl[:, None]] = ord(" ")
    wl = np.concatenate([wl + 1, [2]])                      # + the trailing space; ".\n"
    words[_PROSE_V, 0] = ord(".")
    words[_PROSE_V, 1] = ord("\n")
    # templates: slot count 6..16, per slot a base word and four alternatives, ids skewed to the small (common) ones
    r = splitmix64_words(seed ^ 0x5EED0002, 0, _PROSE_T * _PROSE_SLOTS * 3).reshape(_PROSE_T, _PROSE_SLOTS, 3)

    def wid(a, sh):
        return ((((a >> u(sh)) & u(4095)) * ((a >> u(sh + 12)) & u(4095))) >> u(12)).astype(np.int64)

    base = wid(r[:, :, 0], 0)
    alt = np.stack([wid(r[:, :, 1], 0), wid(r[:, :, 1], 24), wid(r[:, :, 2], 0), wid(r[:, :, 2], 24)], axis=2)
    nslots = (6 + (r[:, 0, 0] >> u(48)) % u(11)).astype(np.int64)
    _prose_tables[seed] = (words, wl, base, alt, nslots)
    return _prose_tables[seed]


def gen_prose(n, seed=SEED_BASE + 6):
    """Deep-LCP natural-text-like block: template sentences + copied passages (see the comment above)."""
    words, wl, base, alt, nslots = _prose_build(seed)
    u = np.uint64
    out = np.empty(n + (1 << 21), dtype=np.uint8)
    fill = 0
    sent0 = 0
    per = 1 << 15                                           # sentences per chunk
    kk = np.arange(_PROSE_SLOTS + 1, dtype=np.int64)[None, :]
    while fill < n:
        r = splitmix64_words(seed, 2 * sent0, 2 * per).reshape(per, 2)
        # boilerplate: one paragraph (8 sentences) in eight repeats one of 256 fixed paragraphs word for word
        pr = splitmix64_words(seed ^ 0x5EED0004, sent0 // 8, per // 8)
        bid = ((((pr >> u(8)) & u(0xFFF)) * ((pr >> u(20)) & u(0xFFF))) >> u(16)).astype(np.int64)      # 0..255, skewed
        fixed = splitmix64_words(seed ^ 0x5EED0005, 0, 256 * 16).reshape(256, 8, 2)[bid].reshape(per, 2)
        r = np.where(np.repeat((pr & u(7)) == 0, 8)[:, None], fixed, r)
        sent0 += per
        t = ((((r[:, 0] & u(0xFFFF)) * ((r[:, 0] >> u(16)) & u(0xFFFF))) >> u(21))).astype(np.int64)
        ks = np.arange(_PROSE_SLOTS, dtype=np.uint64)[None, :]
        q = (r[:, 1][:, None] >> (u(2) * ks)) & u(3)
        a = ((r[:, 0][:, None] >> (u(32) + u(2) * ks)) & u(3)).astype(np.int64)
        w = np.where(q != 0, base[t], np.take_along_axis(alt[t], a[:, :, None], axis=2)[:, :, 0])
        ns = nslots[t][:, None]
        w = np.concatenate([w, np.full((per, 1), _PROSE_V, dtype=np.int64)], axis=1)
        w = np.where(kk == ns, _PROSE_V, w)                 # the sentence end right after its last slot
        ids = w[kk <= ns]                                   # row-major: sentence after sentence
        ln = wl[ids]
        end = np.cumsum(ln)
        total = int(end[-1])
        first = np.repeat(end - ln, ln)
        col = np.arange(total, dtype=np.int64) - first
        chunk = words[np.repeat(ids, ln), col]
        take = min(total, out.size - fill)
        out[fill:fill + take] = chunk[:take]
        fill += take
    x = out[:n].copy()
    del out
    # copied passages: lengths log-uniform in [256, 2 Mi), at most n/8; applied in order (later copies may overlap earlier ones)
    ncopies = max(2, n >> 21) if n >= (1 << 16) else 0
    r = splitmix64_words(seed ^ 0x5EED0003, 0, 3 * ncopies + 3)
    for j in range(ncopies):
        e = 8 + int(r[3 * j] % u(13))
        ln = (1 << e) + int((r[3 * j] >> u(8)) & u((1 << e) - 1))
        ln = min(ln, n // 8)
        src = int(r[3 * j + 1] % u(n - ln))
        dst = int(r[3 * j + 2] % u(n - ln))
        x[dst:dst + ln] = x[src:src + ln].copy()
    return x


def gen_motif_defects(n, motif_len=4099, defects=5, seed=SEED_BASE + 7):
    """cfg 3 with defects ("Gauntlet-style with glitches"): a `motif_len`-byte random motif repeated, `defects` single bytes
    changed at seeded positions -- every tied group straddles the defects, none is one clean run of the period."""
    x = gen_repeat(n, gen_random(motif_len, seed).tobytes()).copy()
    w = splitmix64_words(seed ^ 0x5DEFEC75, 0, 2 * defects)
    for k in range(defects):
        q = int(w[2 * k] % np.uint64(n))
        x[q] ^= np.uint8(1 + int(w[2 * k + 1] % np.uint64(255)))
    return x


def gen_random_copy(n, seed=SEED_BASE + 8):
    """a long non-periodic duplicate: uniform random bytes with the first n/8 bytes copied to the middle of the block -- every
    item inside the copy is tied with its twin far beyond any first stage (tools/dup_region.py; the pair chains' regime)"""
    x = gen_random(n, seed).copy()
    L = n // 8
    x[n // 2:n // 2 + L] = x[:L]
    return x


SHAPES = ("text", "random", "dna", "a", "ab", "motif", "prose", "motif_defects", "random_copy")


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
    if shape == "prose":
        return gen_prose(n, SEED_BASE + 6 + block)
    if shape == "motif_defects":
        return gen_motif_defects(n, 4099, 5, SEED_BASE + 7 + block)
    if shape == "random_copy":
        return gen_random_copy(n, SEED_BASE + 8 + block)
    raise ValueError(shape)
