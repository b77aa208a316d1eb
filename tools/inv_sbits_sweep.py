"""Inverse of small and medium blocks against the rows per chain head (2^sbits): device time per block.
Usage: python tools/inv_sbits_sweep.py"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon
for mib, sweep in ((1, (3, 4, 5, 6)), (4, (3, 4, 5, 6, 7)), (16, (4, 5, 6, 7, 8)), (64, (6, 7, 8, 9)), (128, (7, 8, 9)), (256, (7, 8, 9))):
    n = mib << 20
    x = torch.from_numpy(S.gen_random(n)).cuda()
    bwt = torch.empty_like(x); base = torch.zeros(1, dtype=torch.int32, device="cuda"); out = torch.empty_like(x)
    pyarchon.forward_dev(x, None, bwt, base)
    b = int(base.item())
    row = {"block_MiB": mib}
    for sb in (-1,) + sweep:
        if sb < 0: os.environ.pop("ARCHON_INV_SBITS", None)
        else: os.environ["ARCHON_INV_SBITS"] = str(sb)
        ts = []
        for _ in range(6):
            pyarchon.inverse_dev(bwt, b, out)
            st = pyarchon.stats()
            ts.append((st["ms_total"], st["ms_lf_build"], st["walk_chains"], st["kernel_launches"]))
        assert torch.equal(out, x)
        t = sorted(ts)[len(ts) // 2]
        row["default" if sb < 0 else "sbits_%d" % sb] = {"ms": round(t[0], 3), "lf_build": round(t[1], 3), "chains": t[2], "launches": t[3]}
    print(json.dumps(row), flush=True)
