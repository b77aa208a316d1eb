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


def test_inverse_rejects_non_bwt(archon):
    """a byte string that is not a BWT: the LF walk does not close over all rows"""
    bad = np.frombuffer(b"abab", np.uint8)   # LF permutation of "abab" with base 0 has two cycles
    import ctypes
    out = np.empty(4, np.uint8)
    rc = archon.lib().archon_hip_inverse(ctypes.c_void_p(bad.ctypes.data), 4, 0, ctypes.c_void_p(out.ctypes.data), 0)
    assert rc in (archon.E_CORRUPT, archon.OK)
    if rc == archon.OK:   # if it happens to close, it must round-trip
        _, b2, base2 = archon.forward(out, want_sa=False)
        assert base2 == 0 and (b2 == bad).all()
    rc = archon.lib().archon_hip_inverse(ctypes.c_void_p(bad.ctypes.data), 4, 7, ctypes.c_void_p(out.ctypes.data), 0)
    assert rc == archon.E_ARG
