"""(experiments library) the LF walk with parts switched off: ARCHON_EXP_WALK=1 no slab stores, 2 no symbol look-up, 3 neither, 4 no symbol rows in LDS either, 8 rows read back by the quads but not stored.
Results are invalid with a flag set; only ms_lf_walk is of interest."""
import os, sys, json
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
os.environ.setdefault("ARCHON_HIP_LIB", os.path.join(root, "dark-archon_amd", "libarchon_hip_exp.so"))
sys.path.insert(0, os.path.join(root, "dark-archon_amd"))
import numpy as np, torch, pyarchon, archon_synth as S
n = 256 << 20
x = torch.from_numpy(S.gen_shape("random", n)).cuda()
sa = torch.empty(n, dtype=torch.int32, device="cuda"); bwt = torch.empty(n, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
pyarchon.forward_dev(x, sa, bwt, base)
out = torch.empty(n, dtype=torch.uint8, device="cuda")
for flags in ("0", "1", "2", "4", "8"):
    os.environ["ARCHON_EXP_WALK"] = flags
    for r in range(2):
        try:
            pyarchon.inverse_dev(bwt, int(base.item()), out)
        except Exception as e:
            pass
        st = pyarchon.stats()
    print("flags", flags, "walk ms", round(st["ms_lf_walk"], 3))
