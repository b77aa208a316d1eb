"""A block with a duplicated region: uniform random bytes with x[a:a+L] copied to x[b:b+L] -- every item inside the copy is
tied with its twin far beyond the streaming stage's five key bytes (tie list, then refinement rounds).
Usage: python tools/dup_region.py [MiB] [L MiB]"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
L = (int(sys.argv[2]) if len(sys.argv) > 2 else 32) << 20
x = S.gen_random(n).copy()
x[n // 2:n // 2 + L] = x[:L]
x_t = torch.from_numpy(x).cuda()
sa = torch.empty(n, dtype=torch.int32, device="cuda"); bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
for r in range(3):
    pyarchon.forward_dev(x_t, sa, bwt, base)
    st = pyarchon.stats()
print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v}))
assert pyarchon.validate_dev(x_t, sa)
