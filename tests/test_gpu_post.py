"""GPU: SURVEY.md 8(f) N4 on the device (csrc/post.hiph) -- MTF + zero-run digits + order-0 canonical Huffman per 32 KiB piece.
PARITY UNPINNED: the reference has no such stage (README.md:2 only promises one).  The host stage
(dark-archon_amd/host/archon_post.cpp) states the format; the device stage must produce the same bytes, and the host decoder
must read them back."""
import ctypes
import os
import struct
import subprocess

import numpy as np
import pytest

import archon_synth as S

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PIECE = 32 << 10


@pytest.fixture(scope="module")
def host():
    lib = ctypes.CDLL(os.path.join(ROOT, "dark-archon_amd", "libarchon.so"))
    lib.archon_post_bound.restype = ctypes.c_size_t
    lib.archon_post_bound.argtypes = [ctypes.c_size_t]
    lib.archon_post_encode.restype = ctypes.c_size_t
    lib.archon_post_encode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p]
    lib.archon_post_decode.restype = ctypes.c_int
    lib.archon_post_decode.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_size_t]
    return lib


def host_stream(host, bwt):
    """the block stream as the format states it: u32 pieces | u32 bytes of each piece | the pieces (host/archon_post.cpp)"""
    parts = []
    for o in range(0, bwt.size, PIECE):
        piece = np.ascontiguousarray(bwt[o:o + PIECE])
        out = np.empty(host.archon_post_bound(piece.size), np.uint8)
        m = host.archon_post_encode(piece.ctypes.data, piece.size, out.ctypes.data)
        parts.append(out[:m].tobytes())
    return struct.pack("<I", len(parts)) + b"".join(struct.pack("<I", len(p)) for p in parts) + b"".join(parts)


def device_stream(archon, bwt):
    import torch
    L = archon.lib()
    L.archon_hip_post_bound.restype = ctypes.c_size_t
    L.archon_hip_post_bound.argtypes = [ctypes.c_uint32]
    L.archon_hip_post_encode_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    cap = L.archon_hip_post_bound(bwt.size)
    d_in = torch.from_numpy(np.ascontiguousarray(bwt)).cuda() if bwt.size else torch.empty(1, dtype=torch.uint8, device="cuda")
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    got = ctypes.c_size_t(0)
    torch.cuda.synchronize()
    rc = L.archon_hip_post_encode_dev(d_in.data_ptr(), bwt.size, d_out.data_ptr(), cap, ctypes.byref(got), 0, None)
    assert rc == 0, L.archon_hip_last_error()
    return d_out[:got.value].cpu().numpy().tobytes()


def _cases(oracle):
    rng = np.random.default_rng(258)
    out = {
        "empty": np.zeros(0, np.uint8), "one": np.array([9], np.uint8), "one_zero": np.zeros(1, np.uint8),
        "run": np.full(100000, 7, np.uint8), "zeros_exact_piece": np.zeros(PIECE, np.uint8), "zeros_piece_plus_1": np.zeros(PIECE + 1, np.uint8),
        "random": S.gen_random(70001), "bits": rng.integers(0, 2, 50000).astype(np.uint8),
        "all_symbols": np.arange(256, dtype=np.uint8).repeat(3), "all_symbols_rev": np.arange(255, -1, -1, dtype=np.uint8).repeat(200),
        "chunk_edges": np.concatenate([np.full(4095, 1, np.uint8), np.array([2], np.uint8), np.full(4097, 1, np.uint8), rng.integers(0, 256, 9000).astype(np.uint8)]),
        "runs_across_chunks": np.repeat(rng.integers(0, 4, 60).astype(np.uint8), rng.integers(1, 9000, 60)),
        # code lengths beyond 20 bits before the weights are halved: Fibonacci-like symbol counts
        "fibonacci_counts": np.concatenate([np.full(c, v, np.uint8) for v, c in enumerate([1, 1, 2, 3, 5, 8, 13, 21, 34, 55, 89, 144, 233, 377, 610, 987, 1597, 2584, 4181, 6765, 10946])])[rng.permutation(28656)][:PIECE],
    }
    for shape in ("text", "dna", "ab", "motif", "prose"):
        _, bwt, _ = oracle.forward(S.gen_shape(shape, 300000))     # what the stage really sees: BWT output
        out["bwt_" + shape] = bwt
    return out


_NAMES = ["empty", "one", "one_zero", "run", "zeros_exact_piece", "zeros_piece_plus_1", "random", "bits", "all_symbols",
          "all_symbols_rev", "chunk_edges", "runs_across_chunks", "fibonacci_counts", "bwt_text", "bwt_dna", "bwt_ab",
          "bwt_motif", "bwt_prose"]


@pytest.mark.parametrize("name", _NAMES)
def test_device_decoder_reads_host_stream(archon, oracle, host, name):
    """the way back on the device (k_post_offsets, k_post_decode): the stream the HOST stage writes -- the statement of the
    format -- decodes to the bytes it was made from; so does the device stage's own stream (equal by the test below)"""
    import torch
    data = np.ascontiguousarray(_cases(oracle)[name], np.uint8)
    stream = np.frombuffer(host_stream(host, data), np.uint8)
    d_in = torch.from_numpy(stream.copy()).cuda()
    d_out = torch.full((data.size + 64,), 0xEE, dtype=torch.uint8, device="cuda")
    n = archon.post_decode_dev(d_in, stream.size, d_out[:max(1, data.size)] if data.size else d_out[:1])
    assert n == data.size
    got = d_out.cpu().numpy()
    assert (got[:n] == data).all()
    assert (got[max(n, 1):] == 0xEE).all()                   # nothing written behind the block


def test_device_decoder_rejects_malformed_streams(archon, oracle, host):
    import torch
    data = S.gen_text(100000)
    good = np.frombuffer(host_stream(host, data), np.uint8).copy()
    d_out = torch.empty(data.size, dtype=torch.uint8, device="cuda")

    def run(buf, nbytes=None):
        d_in = torch.from_numpy(np.ascontiguousarray(buf)).cuda()
        return archon.post_decode_dev(d_in, buf.size if nbytes is None else nbytes, d_out)
    assert run(good) == data.size
    for what in ("truncated", "piece_count", "piece_size", "piece_length", "code_length", "bits"):
        bad = good.copy()
        nbytes = None
        if what == "truncated":
            nbytes = good.size - 1000
        elif what == "piece_count":
            bad[0:4] = np.frombuffer(struct.pack("<I", 1 << 20), np.uint8)
        elif what == "piece_size":
            bad[4:8] = np.frombuffer(struct.pack("<I", 3), np.uint8)
        elif what == "piece_length":
            off = 4 + 4 * struct.unpack_from("<I", good.tobytes(), 0)[0]
            bad[off:off + 4] = np.frombuffer(struct.pack("<I", 12345), np.uint8)
        elif what == "code_length":
            off = 4 + 4 * struct.unpack_from("<I", good.tobytes(), 0)[0]
            bad[off + 4 + 40] = 33
        else:
            off = 4 + 4 * struct.unpack_from("<I", good.tobytes(), 0)[0]
            bad[off + 4 + 258: off + 4 + 258 + 2000] ^= 0x5A          # garbage bits: wrong length or an unknown code
        with pytest.raises(archon.ArchonError):
            run(bad, nbytes)
    # a buffer too small for the block
    small = torch.empty(1000, dtype=torch.uint8, device="cuda")
    with pytest.raises(archon.ArchonError):
        archon.post_decode_dev(torch.from_numpy(good).cuda(), good.size, small)


def test_inverse_post_round_trip(archon, oracle):
    """host block -> archon_hip_forward_post -> archon_hip_inverse_post: only packed streams cross the link, both ways"""
    L = archon.lib()
    L.archon_hip_post_bound.restype = ctypes.c_size_t
    L.archon_hip_post_bound.argtypes = [ctypes.c_uint32]
    for shape, n in (("text", 1 << 20), ("dna", 300001), ("random", 70000), ("a", 50000)):
        x = S.gen_shape(shape, n)
        cap = L.archon_hip_post_bound(n)
        out = np.empty(cap, np.uint8)
        got, base = ctypes.c_size_t(0), ctypes.c_uint32(0)
        rc = L.archon_hip_forward_post(x.ctypes.data, n, out.ctypes.data, cap, ctypes.byref(got), ctypes.cast(ctypes.byref(base), ctypes.c_void_p), 0)
        assert rc == 0, L.archon_hip_last_error()
        back = archon.inverse_post(out[:got.value], base.value, n)
        assert back.size == n and (back == x).all(), shape


@pytest.mark.parametrize("name", _NAMES)
def test_device_stream_equals_host_stream(archon, oracle, host, name):
    data = np.ascontiguousarray(_cases(oracle)[name], np.uint8)
    want = host_stream(host, data)
    got = device_stream(archon, data)
    assert len(got) == len(want) and got == want, (name, len(got), len(want))
    # and the host decoder reads the device's pieces back
    np_, = struct.unpack_from("<I", got, 0)
    off = 4 + 4 * np_
    back = []
    for k in range(np_):
        sz, = struct.unpack_from("<I", got, 4 + 4 * k)
        n_k = min(PIECE, data.size - k * PIECE)
        buf = np.frombuffer(got, np.uint8, sz, off).copy()
        out = np.empty(max(1, n_k), np.uint8)
        assert host.archon_post_decode(buf.ctypes.data, sz, out.ctypes.data, n_k) == 0
        back.append(out[:n_k])
        off += sz
    assert off == len(got)
    assert (np.concatenate(back) == data).all() if back else data.size == 0


def test_full_block_and_unaligned_input(archon, oracle, host):
    """a 64 MiB BWT (2048 pieces) and an input pointer that is not word-aligned"""
    import torch
    x = S.gen_text(16 << 20)
    _, bwt, _ = archon.forward(x)
    big = np.tile(bwt, 4)
    assert device_stream(archon, big) == host_stream(host, big)
    L = archon.lib()
    d = torch.from_numpy(np.concatenate([np.zeros(3, np.uint8), bwt[:100003]])).cuda()
    cap = L.archon_hip_post_bound(100003)
    d_out = torch.empty(cap, dtype=torch.uint8, device="cuda")
    got = ctypes.c_size_t(0)
    torch.cuda.synchronize()
    assert L.archon_hip_post_encode_dev(d.data_ptr() + 3, 100003, d_out.data_ptr(), cap, ctypes.byref(got), 0, None) == 0
    assert d_out[:got.value].cpu().numpy().tobytes() == host_stream(host, bwt[:100003])


def test_rejects_short_buffer(archon):
    import torch
    L = archon.lib()
    L.archon_hip_post_encode_dev.argtypes = [ctypes.c_void_p, ctypes.c_uint32, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
    d = torch.zeros(1000, dtype=torch.uint8, device="cuda")
    got = ctypes.c_size_t(0)
    assert L.archon_hip_post_encode_dev(d.data_ptr(), 1000, d.data_ptr(), 100, ctypes.byref(got), 0, None) < 0


def test_cli_post_container_round_trip(tmp_path):
    """`archon e -m -b<size>`: transform + post stage on the GPU, `archon d -b` reads it back"""
    exe = os.path.join(ROOT, "bin", "archon")
    src = tmp_path / "in.bin"
    data = np.concatenate([S.gen_text(3 << 20), S.gen_random(100001), np.zeros(70000, np.uint8), S.gen_shape("dna", 1 << 20)])
    data.tofile(src)
    enc, dec = tmp_path / "out.rm", tmp_path / "back.bin"
    r = subprocess.run([exe, "e", "-m", "-b1048576", str(src), str(enc)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert enc.stat().st_size < 0.8 * data.size
    r = subprocess.run([exe, "d", "-b1048576", str(enc), str(dec)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (np.fromfile(dec, np.uint8) == data).all()
