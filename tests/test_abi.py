"""CPU: the C-ABI shared libraries load and export every symbol their headers declare;
without a GPU the compute entry points fail loudly (no CPU fallback)."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(archon_[a-z0-9_]+)\s*\(", src)))


def test_hip_library_exports_header():
    import pyarchon
    lib = pyarchon.lib()
    names = declared_functions("archon_hip.h")
    assert len(names) >= 18
    for name in names:
        assert hasattr(lib, name), name
    assert set(pyarchon.SYMBOLS) <= set(names)
    for name in declared_functions("archon_hip_test.h"):          # the test-only routing entry point
        assert hasattr(lib, name), name


def test_library_reads_no_routing_from_the_environment():
    """routes are chosen through archon_hip_test_route (include/archon_hip_test.h), never through the
    environment of whoever links the library -- no getenv of a route name is compiled into the product sources."""
    import glob
    import pyarchon
    src = ""
    for path in glob.glob(os.path.join(ROOT, "dark-archon_amd", "csrc", "*")):
        text = open(path).read()
        text = re.sub(r"#ifdef ARCHON_EXPERIMENTS.*?#endif", "", text, flags=re.S)     # the experiments library (tools/) may
        src += text
    assert "getenv" not in src
    lib = pyarchon.lib()
    assert lib.archon_hip_test_route(b"NO_CHAINS", 1) == 0 and lib.archon_hip_test_route(b"RESET", 0) == 0
    assert lib.archon_hip_test_route(b"NO_SUCH_ROUTE", 1) < 0 and lib.archon_hip_test_route(b"PASS_RANGES", 5000) < 0


def test_host_library_exports_header():
    path = os.path.join(ROOT, "dark-archon_amd", "libarchon.so")
    if not os.path.exists(path):
        subprocess.run(["make", "-C", ROOT, "host"], check=True, capture_output=True)
    lib = ctypes.CDLL(path)
    for name in declared_functions("archon.h"):
        assert hasattr(lib, name), name


def test_stats_struct_size(tmp_path):
    """the ctypes mirror of archon_hip_stats has the size and the field offsets the C header gives it"""
    import subprocess
    import pyarchon
    names = [k for k, _ in pyarchon.Stats._fields_ if not k.startswith("_")]
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "archon_hip.h"\nint main(void){printf("%zu", sizeof(archon_hip_stats));'
                   + "".join('printf(" %%zu", offsetof(archon_hip_stats, %s));' % k for k in names) + "return 0;}\n")
    exe = tmp_path / "sz"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = [int(t) for t in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()]
    assert got[0] == ctypes.sizeof(pyarchon.Stats)
    assert got[1:] == [getattr(pyarchon.Stats, k).offset for k in names]


def test_no_cpu_fallback():
    """On a box without a GPU every compute call returns ARCHON_E_NODEVICE; on a GPU box
    this test checks argument validation instead."""
    import pyarchon
    L = pyarchon.lib()
    x = np.frombuffer(b"banana", np.uint8).copy()
    bwt = np.zeros(6, np.uint8)
    base = ctypes.c_uint32()
    rc = L.archon_hip_forward(ctypes.c_void_p(x.ctypes.data), 6, None, ctypes.c_void_p(bwt.ctypes.data),
                              ctypes.cast(ctypes.byref(base), ctypes.c_void_p), 0)
    if pyarchon.device_count() == 0:
        assert rc == pyarchon.E_NODEVICE
        assert b"no CPU fallback" in L.archon_hip_last_error()
        with pytest.raises(pyarchon.ArchonError):
            pyarchon.forward(x)
        with pytest.raises(pyarchon.ArchonError):
            pyarchon.inverse(x, 0)
        with pytest.raises(pyarchon.ArchonError):
            pyarchon.hist256(x)
    else:
        assert rc == 0 and bwt.tobytes() == b"nnbaaa" and base.value == 2
    assert L.archon_hip_forward(None, 6, None, None, None, 0) == pyarchon.E_ARG


def test_cli_usage_codes(tmp_path):
    """a7 main.cpp return codes: -1 usage, -2 cannot open input, -3 empty input"""
    exe = os.path.join(ROOT, "bin", "archon")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", ROOT, "cli"], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and "Usage: archon [e|d] <in> <out>" in r.stdout
    r = subprocess.run([exe, "x", "a", "b"], capture_output=True, text=True)
    assert r.returncode == 255
    r = subprocess.run([exe, "e", str(tmp_path / "missing"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 254
    (tmp_path / "empty").write_bytes(b"")
    r = subprocess.run([exe, "e", str(tmp_path / "empty"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 253
    (tmp_path / "short").write_bytes(b"abc")
    r = subprocess.run([exe, "d", str(tmp_path / "short"), str(tmp_path / "o")], capture_output=True, text=True)
    assert r.returncode == 254


def test_synth_generators_are_deterministic():
    import archon_synth as S
    import hashlib
    a = S.gen_random(1000)
    assert (a == S.gen_random(2000)[:1000]).all()
    assert hashlib.sha256(S.gen_random(4096).tobytes()).hexdigest() == hashlib.sha256(S.gen_random(4096).tobytes()).hexdigest()
    d = S.gen_dna(5000)
    assert set(np.unique(d)) <= set(b"ACGT")
    t = S.gen_text(20000)
    assert t.min() >= 10 and t.max() < 127 and (t == 32).mean() > 0.1
    assert S.gen_repeat(5, b"ab").tobytes() == b"ababa"
    assert (S.gen_motif(3000)[:1000] == S.gen_motif(3000)[1000:2000]).all()
    # splitmix64 known value: first output for seed 0 is 0xE220A8397B1DCDAF
    assert int(S.splitmix64_words(0, 0, 1)[0]) == 0xE220A8397B1DCDAF


REF_MAIN = "/root/reference/bwt/a7/src/main.cpp"


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference tree not present (GPU box)")
def test_reference_main_builds_against_the_shim(tmp_path):
    """INTEGRATION.md option A, exactly as written: the reference's own caller (bwt/a7/src/main.cpp, unmodified, fed
    through stdin so that its `#include "archon.h"` resolves to host/ref_shim/archon.h) compiles against the
    MI355X-backed class Archon (reference interface: bwt/a7/src/archon.h:8-29) and links with libarchon_hip.so."""
    pkg = os.path.join(ROOT, "dark-archon_amd")
    obj, exe = str(tmp_path / "main_hip.o"), str(tmp_path / "main_hip")
    with open(REF_MAIN, "rb") as src:
        subprocess.run(["g++", "-O3", "-DNDEBUG", "-x", "c++", "-I" + os.path.join(pkg, "host", "ref_shim"), "-c", "-o", obj, "-"],
                       stdin=src, check=True, capture_output=True)
    subprocess.run(["g++", "-o", exe, obj, os.path.join(pkg, "host", "archon_host.cpp"), "-I" + os.path.join(ROOT, "include"),
                    "-L" + pkg, "-larchon_hip", "-Wl,-rpath," + pkg], check=True, capture_output=True)
    r = subprocess.run([exe], capture_output=True, text=True)
    assert r.returncode == 255 and "Usage: archon [e|d] <in> <out>" in r.stdout      # main.cpp:13-16: usage, return -1
    # the reference's messages, then our compute entry point failing loudly without a GPU (no CPU fallback)
    inp = tmp_path / "in.txt"
    inp.write_bytes(b"abracadabra" * 10)
    r = subprocess.run([exe, "e", str(inp), str(tmp_path / "out.bwt")], capture_output=True, text=True)
    assert "Encoding SA..." in r.stdout


def test_options_and_context_binding_need_no_gpu():
    """archon_hip_set_option / archon_hip_bind_context (include/archon_hip.h): host-side state, per device.
    ADVICE r3: the context binding is per (thread, device) -- the two workers of one GPU must land on its two contexts on
    a node with an even number of GPUs, whatever order the threads start in."""
    import threading
    import pyarchon
    L = pyarchon.lib()
    assert pyarchon.get_option("pass_ranges", 3) == 0 and pyarchon.get_option("pass_b_buckets", 3) == 1
    pyarchon.set_option("pass_ranges", 1024, 3)
    assert pyarchon.get_option("pass_ranges", 3) == 1024 and pyarchon.get_option("pass_ranges", 2) == 0       # per device
    pyarchon.set_option("pass_ranges", 0, 3)
    with pytest.raises(pyarchon.ArchonError):
        pyarchon.set_option("pass_ranges", 5000, 3)
    with pytest.raises(pyarchon.ArchonError):
        pyarchon.set_option("no_such_option", 1, 3)
    # eight threads, two "GPUs" (devices 40 and 41: no HIP call is made), started in block order as the container's workers are:
    # worker w drives device 40 + w % 2.  Each device must see both of its contexts in use.
    ndev, seen = 2, {}
    order = threading.Semaphore(1)

    def worker(w):
        with order:
            seen[w] = L.archon_hip_context_of_thread(40 + w % ndev)
    ts = [threading.Thread(target=worker, args=(w,)) for w in range(4)]
    for t in ts:
        t.start()
        t.join()                       # strictly in worker order: the case that broke the process-wide counter
    for d in range(ndev):
        assert sorted(seen[w] for w in range(4) if w % ndev == d) == [0, 1], seen
    # an explicit binding wins and is per device
    res = {}

    def bound():
        L.archon_hip_bind_context(42, 1)
        res["a"] = (L.archon_hip_context_of_thread(42), L.archon_hip_context_of_thread(43))
    t = threading.Thread(target=bound)
    t.start()
    t.join()
    assert res["a"][0] == 1 and res["a"][1] == 0
    assert L.archon_hip_bind_context(42, 8) < 0 and L.archon_hip_bind_context(-1, 0) < 0 and L.archon_hip_bind_context(42, 7) == 0


def test_bench_uses_the_product_api_only():
    """what bench.py sets for N > 1 (pass ranges, pass B by ranges) goes through archon_hip_set_option -- the
    bench names neither the test header's call nor a route variable."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert "archon_hip_test_route" not in src and "ARCHON_PASS_RANGES" not in src and "ARCHON_NO_" not in src
    assert 'set_option("pass_ranges"' in src
    shard = open(os.path.join(ROOT, "dark-archon_amd", "archon_shard.py")).read()
    assert "test_route" not in shard and "ARCHON_" not in shard
