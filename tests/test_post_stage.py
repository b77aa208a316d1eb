"""CPU: the optional MTF + zero-run + order-0 Huffman stage (dark-archon_amd/host/archon_post.cpp).
PARITY UNPINNED -- the reference has no such stage (README.md:2 only promises one): round trips and stream hygiene only."""
import ctypes
import os

import numpy as np
import pytest

import archon_synth as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def post():
    lib = ctypes.CDLL(os.path.join(ROOT, "dark-archon_amd", "libarchon.so"))
    lib.archon_post_bound.restype = ctypes.c_size_t
    lib.archon_post_bound.argtypes = [ctypes.c_size_t]
    lib.archon_post_encode.restype = ctypes.c_size_t
    lib.archon_post_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.archon_post_decode.restype = ctypes.c_int
    lib.archon_post_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    return lib


def _round_trip(post, data):
    data = np.ascontiguousarray(data, np.uint8)
    out = np.empty(post.archon_post_bound(data.size), np.uint8)
    m = post.archon_post_encode(data.ctypes.data, data.size, out.ctypes.data)
    assert m <= out.size
    back = np.empty(max(1, data.size), np.uint8)
    assert post.archon_post_decode(out.ctypes.data, m, back.ctypes.data, data.size) == 0
    assert (back[:data.size] == data).all()
    return m


def test_round_trips(post, oracle):
    rng = np.random.default_rng(3)
    cases = [np.zeros(0, np.uint8), np.zeros(1, np.uint8), np.full(100000, 7, np.uint8), S.gen_random(70001),
             rng.integers(0, 2, 50000).astype(np.uint8), np.arange(256, dtype=np.uint8).repeat(3)]
    for shape in ("text", "dna", "ab", "motif"):
        _, bwt, _ = oracle.forward(S.gen_shape(shape, 200000))       # what the stage really sees: BWT output
        cases.append(bwt)
    for c in cases:
        _round_trip(post, c)


def test_it_compresses(post, oracle):
    _, bwt, _ = oracle.forward(S.gen_text(1 << 20))
    assert _round_trip(post, bwt) < 0.85 * bwt.size   # memoryless Zipf text: no context for the BWT to expose (MTF ranks ~6 bits)
    _, bwt, _ = oracle.forward(S.gen_motif(1 << 20))
    assert _round_trip(post, bwt) < 0.01 * bwt.size   # a repeated motif: long runs after the BWT
    assert _round_trip(post, np.full(1 << 20, 65, np.uint8)) < 1024


def test_rejects_malformed(post):
    data = S.gen_text(10000)
    out = np.empty(post.archon_post_bound(data.size), np.uint8)
    m = post.archon_post_encode(data.ctypes.data, data.size, out.ctypes.data)
    back = np.empty(data.size, np.uint8)
    assert post.archon_post_decode(out.ctypes.data, m // 2, back.ctypes.data, data.size) != 0      # truncated
    assert post.archon_post_decode(out.ctypes.data, m, back.ctypes.data, data.size - 1) != 0        # wrong length
    bad = out.copy()
    bad[4:262] = 31                                                                                  # absurd code lengths
    assert post.archon_post_decode(bad.ctypes.data, m, back.ctypes.data, data.size) != 0
