"""GPU parity: inverse BWT (lf_build + lf_walk) vs the CPU oracle, and round trips."""
import numpy as np
import pytest

import archon_synth as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("shape", S.SHAPES)
@pytest.mark.parametrize("n", [1, 2, 9, 1000, 65536, 1 << 20])
def test_inverse_vs_oracle(archon, oracle, shape, n):
    x = S.gen_shape(shape, n)
    _, B, base = oracle.forward(x)
    out = archon.inverse(B, base)
    assert (out == x).all()


def test_round_trip_gpu_only(archon):
    """encode -> decode on the GPU: a size-independent property (no oracle involved)."""
    for shape in S.SHAPES:
        x = S.gen_shape(shape, (1 << 21) + 7)
        _, bwt, base = archon.forward(x, want_sa=False)
        assert (archon.inverse(bwt, base) == x).all(), shape


def test_inverse_rejects_non_bwt(archon, oracle):
    """a byte string that is not a BWT: the LF walk does not close over all rows (the oracle's walk says so too)"""
    bad = np.frombuffer(b"abab", np.uint8)   # LF permutation of "abab" with base 0 has two cycles
    import ctypes
    assert oracle.inverse(bad, 0)[0] != 0
    out = np.empty(4, np.uint8)
    rc = archon.lib().archon_hip_inverse(ctypes.c_void_p(bad.ctypes.data), 4, 0, ctypes.c_void_p(out.ctypes.data), 0)
    assert rc == archon.E_CORRUPT
    rc = archon.lib().archon_hip_inverse(ctypes.c_void_p(bad.ctypes.data), 4, 7, ctypes.c_void_p(out.ctypes.data), 0)
    assert rc == archon.E_ARG


@pytest.mark.parametrize("slab", ["16", "64"])
def test_inverse_long_chain_route(archon, oracle, slab, monkeypatch):
    """the single walk stores each sub-chain in a slab; chains that outgrow it are walked again (k_walk_emit).
    Tiny slabs make that route carry most of the block."""
    monkeypatch.setenv("ARCHON_INV_SLAB", slab)
    for shape, n in (("random", 300001), ("text", 1 << 20), ("a", 70000), ("dna", 123457)):
        x = S.gen_shape(shape, n)
        _, B, base = oracle.forward(x)
        assert (archon.inverse(B, base) == x).all(), (shape, n)


@pytest.mark.parametrize("rows", ["0", "1", "2"])
def test_inverse_walk_variants(archon, oracle, rows, monkeypatch):
    """the walk writes its slabs by quads through LDS rows (k_walk_rows: 128-byte rows = 1, the product's choice above 128 MiB; 64-byte rows = 2)
    or lane by lane (k_walk_queue = 0)"""
    monkeypatch.setenv("ARCHON_INV_ROWS", rows)
    for slab in ("0", "128", "192"):          # (192: a multiple of 64 only -- the 128-byte rows fall back to 64-byte ones)
        monkeypatch.setenv("ARCHON_INV_SLAB", slab)
        for shape, n in (("random", 300001), ("text", (1 << 21) + 5), ("dna", 400000), ("a", 140000)):
            x = S.gen_shape(shape, n)
            _, B, base = oracle.forward(x)
            assert (archon.inverse(B, base) == x).all(), (shape, n, slab)


def test_inverse_unaligned_device_buffers(archon, oracle):
    """chain copies start at any byte offset; so may the caller's buffers"""
    import torch
    x = S.gen_shape("text", 500003)
    _, B, base = oracle.forward(x)
    buf = torch.zeros(x.size + 64, dtype=torch.uint8, device="cuda")
    out = torch.zeros(x.size + 64, dtype=torch.uint8, device="cuda")
    for off_in, off_out in ((0, 0), (1, 3), (7, 2), (16, 5)):
        buf[off_in:off_in + x.size] = torch.from_numpy(B).cuda()
        archon.inverse_dev(buf[off_in:off_in + x.size], base, out[off_out:off_out + x.size])
        assert (out[off_out:off_out + x.size].cpu().numpy() == x).all(), (off_in, off_out)
