import sys, os
sys.path.insert(0, "dark-archon_amd"); sys.path.insert(0, "tests")
import numpy as np, pyarchon, archon_synth as S, oracle_binding as OB
o = OB.Oracle()
for shape, n in (("ab", 1000), ("a", 1000), ("ab", 70000), ("motif", 3000), ("text", 65536)):
    x = S.gen_shape(shape, n)
    sa, bwt, base = pyarchon.forward(x)
    st = pyarchon.stats()
    P, B, b0 = o.forward(x)
    bad = np.nonzero(bwt != B)[0]
    print(shape, n, "sa ok", (sa == P).all(), "bwt bad rows", bad.size, bad[:10], "sa at bad", sa[bad[:10]], "base", base, b0,
          {k: st[k] for k in ("path", "radix_passes", "text_rounds", "doubling_rounds", "unresolved_initial", "period", "chain_items", "tie_groups", "tie_items")})
