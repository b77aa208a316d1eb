"""A periodic block with point defects (the case DESIGN.md 3.2 names as the limit of the run shortcut): a 4099-byte random
motif over <MiB> MiB with <k> flipped bytes.  Usage: python tools/defect_motif.py [MiB=80] [flips=3]"""
import sys, os, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch, pyarchon
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 80) * 1000000
flips = int(sys.argv[2]) if len(sys.argv) > 2 else 3
rng = np.random.default_rng(5)
x = np.tile(rng.integers(0, 256, size=4099, dtype=np.uint8), n // 4099 + 1)[:n].copy()
for _ in range(flips):
    x[int(rng.integers(0, n))] ^= np.uint8(1 + rng.integers(0, 255))
x_t = torch.from_numpy(x).cuda()
sa = torch.empty(n, dtype=torch.int32, device="cuda"); bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
for r in range(2):
    pyarchon.forward_dev(x_t, sa, bwt, base)
    st = pyarchon.stats()
ok = pyarchon.validate_dev(x_t, sa)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
pyarchon.inverse_dev(bwt, int(base.item()), out)
keys = ("path", "radix_passes", "text_rounds", "doubling_rounds", "unresolved_initial", "unresolved_total", "seg_big_items", "period", "chain_items", "chain_pairs", "ms_total", "ms_doubling")
print(json.dumps({"n": n, "flips": flips, **{k: st[k] for k in keys}, "sa_lf_consistent": bool(ok), "round_trip": bool(torch.equal(out, x_t))}))
