"""Six block-coder objects on six threads (tests/test_gpu_cli.py::test_block_coder_objects_keep_their_own_blocks) through libarchon.so,
many rounds, with a diagnosis when a resident validation fails: is the suffix array right, does the validation fail again?"""
import ctypes, os, sys, threading, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
torch.cuda.init()
import pyarchon, oracle_binding, archon_synth as S
pyarchon.lib()
if os.environ.get("STRESS_SYNC_ROUTES"):
    pyarchon.forward(np.arange(100, dtype=np.uint8))     # hands ARCHON_* route variables to the library
orc = oracle_binding.Oracle()
L = ctypes.CDLL(os.path.join(ROOT, "dark-archon_amd", "libarchon.so"))
L.archon_create.restype = ctypes.c_void_p; L.archon_create.argtypes = [ctypes.c_uint32]
L.archon_sa.restype = ctypes.POINTER(ctypes.c_uint32)
for fn in ("archon_destroy", "archon_validate", "archon_en_compute", "archon_sa", "archon_base_id", "archon_length"):
    getattr(L, fn).argtypes = [ctypes.c_void_p]
L.archon_en_read.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint32]
libc = ctypes.CDLL(None); libc.fopen.restype = ctypes.c_void_p; libc.fopen.argtypes = [ctypes.c_char_p, ctypes.c_char_p]; libc.fclose.argtypes = [ctypes.c_void_p]
n, T, reps = 150001, 6, int(sys.argv[1]) if len(sys.argv) > 1 else 30
shapes = ["random", "text", "dna", "random", "text", "dna"]
blocks = [S.gen_shape(sh, n, block=i) for i, sh in enumerate(shapes)]
want = [orc.forward(x) for x in blocks]
tmp = tempfile.mkdtemp()
for i, x in enumerate(blocks): x.tofile(os.path.join(tmp, "x%d.raw" % i))
barrier = threading.Barrier(T)
bad = []
def work(i):
    a = L.archon_create(n)
    fx = libc.fopen(os.path.join(tmp, "x%d.raw" % i).encode(), b"rb"); L.archon_en_read(a, fx, n); libc.fclose(fx)
    for rep in range(reps):
        rc = L.archon_en_compute(a)
        try: barrier.wait(60)
        except threading.BrokenBarrierError: return
        v1 = L.archon_validate(a)
        P = np.ctypeslib.as_array(L.archon_sa(a), shape=(n,)).copy()
        if rc != 0 or v1 != 1 or not (P == want[i][0]).all():
            v2 = L.archon_validate(a)
            bad.append((i, shapes[i], rep, "rc", rc, "validate", v1, "again", v2, "sa ok", bool((P == want[i][0]).all()), "base ok", L.archon_base_id(a) == want[i][2],
                        pyarchon.lib().archon_hip_last_error().decode()))
        try: barrier.wait(60)
        except threading.BrokenBarrierError: return
ts = [threading.Thread(target=work, args=(i,)) for i in range(T)]
[t.start() for t in ts]; [t.join(600) for t in ts]
for r in bad[:12]: print(r)
print("rounds", reps, "failures", len(bad))
