"""Phase stamps of the two LSB passes (experiments library, `make exp`): cycles of wave 0 of workgroup 0 per phase."""
import ctypes, os, sys
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("ARCHON_HIP_LIB", os.path.join(root, "dark-archon_amd", "libarchon_hip_exp.so"))
sys.path.insert(0, os.path.join(root, "dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 256 << 20
shape = sys.argv[2] if len(sys.argv) > 2 else "random"
x = torch.from_numpy(S.gen_shape(shape, n)).cuda()
sa = torch.empty(n, dtype=torch.int32, device="cuda"); bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
names = ["top/keys", "rank", "rank-barrier", "layout", "staging", "load-issue", "carry-in+barrier", "emit reads+stores", "tails+barrier", "advance+barrier", "prefetch-wait", "-", "  layout: read counts", "  layout: wave scan", "  layout: barrier 1", "  layout: tables", "  layout: quad table", "  layout: barrier 2"]
for rep in range(3):
    pyarchon.forward_dev(x, sa, bwt, base)
    st = pyarchon.stats()
buf = (ctypes.c_ulonglong * 64)()
assert pyarchon.lib().archon_hip_exp_stamps(buf) == 0
for which, tag in ((0, "pass A"), (1, "pass B")):
    v = [buf[which * 32 + i] for i in range(18)]
    tot = sum(v[:11]) or 1
    print("%s (%.3f ms): total %d cycles of wave 0 / workgroup 0" % (tag, st["ms_pass_text" if which == 0 else "ms_pass_rec"], tot))
    for nme, c in zip(names, v):
        print("   %-20s %9d  %5.1f %%" % (nme, c, 100.0 * c / tot))

ls = (ctypes.c_ulonglong * 16)()
assert pyarchon.lib().archon_hip_exp_ls_stamps(ls) == 0
lsn = ["loads issued", "load wait+barrier", "count atomics", "barrier", "scan", "scatter rem", "barrier", "rank loops", "barrier", "write IC", "barrier", "output stores"]
tot = sum(ls[i] for i in range(12)) or 1
print("local sort (%.3f ms), bucket 30000: %d cycles" % (st["ms_local_sort"], tot))
for i in range(12):
    print("   %-20s %9d  %5.1f %%" % (lsn[i], ls[i], 100.0 * ls[i] / tot))
