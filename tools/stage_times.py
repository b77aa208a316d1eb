"""Per-stage device times of the forward pipeline (HIP events inside the library)."""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n = int(sys.argv[1]) << 20 if len(sys.argv) > 1 else 256 << 20
shape = sys.argv[2] if len(sys.argv) > 2 else "random"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
x = torch.from_numpy(S.gen_shape(shape, n)).cuda()
sa = torch.empty(n, dtype=torch.int32, device="cuda"); bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
for r in range(reps):
    pyarchon.forward_dev(x, sa, bwt, base)
    st = pyarchon.stats()
    if r: print(json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v}))

if len(sys.argv) > 4 and sys.argv[4] == "inv":
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    for r in range(3):
        pyarchon.inverse_dev(bwt, int(base.item()), out)
        st = pyarchon.stats()
        print("inverse", json.dumps({k: (round(v, 3) if isinstance(v, float) else v) for k, v in st.items() if v}))
    assert torch.equal(out, x)
