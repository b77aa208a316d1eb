"""7-pass route: device time of a 256 MiB block against the number of key bytes the LSB passes sort on (ARCHON_KEY_BYTES; 0 = the
library's own choice), every suffix array compared with the default run's.  Usage: python tools/key_bytes_sweep.py [shape ...] (real =
source text found on the machine, as tools/real_text.py collects it)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np, torch
import archon_synth as S, pyarchon

def real_text(cap):
    exts = (".py", ".h", ".hpp", ".txt", ".md", ".rst", ".c", ".cpp", ".json", ".html", ".js", ".cmake")
    buf, seen = bytearray(), set()
    for top in ("/usr/lib/python3", "/usr/lib/python3.10", "/usr/local/lib/python3.10/dist-packages", "/opt/rocm/include", "/usr/include", "/usr/share"):
        for dp, dn, fn in os.walk(top):
            for f in sorted(fn):
                if not f.endswith(exts):
                    continue
                p = os.path.join(dp, f)
                try:
                    st = os.stat(p)
                    key = (st.st_size, f)
                    if key in seen or st.st_size > (8 << 20) or os.path.islink(p):
                        continue
                    seen.add(key)
                    with open(p, "rb") as fh:
                        buf += fh.read()
                except OSError:
                    continue
                if len(buf) >= cap:
                    return np.frombuffer(bytes(buf[:cap]), np.uint8)
    return np.frombuffer(bytes(buf), np.uint8)

n = 256 << 20
for shape in (sys.argv[1:] or ["prose", "text", "real"]):
    x = torch.from_numpy(real_text(n) if shape == "real" else S.gen_shape(shape, n)).cuda()
    m = x.numel()
    sa0 = torch.empty(m, dtype=torch.int32, device="cuda"); sa = torch.empty_like(sa0)
    bwt = torch.empty(m, dtype=torch.uint8, device="cuda"); base = torch.zeros(1, dtype=torch.int32, device="cuda")
    row = {"shape": shape, "n": m}
    for kb in (0, 6, 5, 4, 3):
        if kb: os.environ["ARCHON_KEY_BYTES"] = str(kb)
        else: os.environ.pop("ARCHON_KEY_BYTES", None)
        ts = []
        for r in range(3):
            pyarchon.forward_dev(x, sa0 if kb == 0 else sa, bwt, base)
            st = pyarchon.stats()
            ts.append(st["ms_total"])
        same = True if kb == 0 else bool(torch.equal(sa, sa0))
        row["key_bytes_%d" % kb if kb else "default"] = {"ms": round(min(ts[1:]), 3), "passes": st["radix_passes"], "rounds": st["doubling_rounds"], "text_rounds": st["text_rounds"],
                                                         "ms_sort": round(st["ms_sort"], 3), "ms_rounds": round(st["ms_doubling"], 3), "same_sa": same}
    print(json.dumps(row), flush=True)
