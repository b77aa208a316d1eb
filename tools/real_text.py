"""Forward/inverse on REAL text found on the machine (Python sources, headers, docs under /usr and /opt): how the
pipeline behaves on natural, repetitive data rather than the synthetic shapes.  Usage: python tools/real_text.py [MiB]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "dark-archon_amd"))
import numpy as np
import pyarchon

cap = (int(sys.argv[1]) if len(sys.argv) > 1 else 256) << 20
exts = (".py", ".h", ".hpp", ".txt", ".md", ".rst", ".c", ".cpp", ".json", ".html", ".js", ".cmake")
buf = bytearray()
seen = set()
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
                    continue           # skip exact duplicates by (size, name): vendored copies would be one long repeat
                seen.add(key)
                with open(p, "rb") as fh:
                    buf += fh.read()
            except OSError:
                continue
            if len(buf) >= cap:
                break
        if len(buf) >= cap:
            break
    if len(buf) >= cap:
        break
x = np.frombuffer(bytes(buf[:cap]), np.uint8)
del buf
print("corpus bytes", x.size, "distinct bytes", int(np.unique(x).size), flush=True)
for rep in range(2):
    sa, bwt, base = pyarchon.forward(x)
    st = pyarchon.stats()
ok = pyarchon.validate(x, sa)
back = pyarchon.inverse(bwt, base)
si = pyarchon.stats()
keys = ("path", "radix_passes", "text_rounds", "doubling_rounds", "unresolved_initial", "unresolved_total", "seg_big_items", "ms_total", "ms_hist", "ms_sort", "ms_doubling", "period", "chain_items")
print(json.dumps({"n": int(x.size), **{k: st[k] for k in keys}, "sa_lf_consistent": bool(ok), "round_trip": bool((back == x).all()),
                  "inverse_ms": round(si["ms_total"], 3), "forward_MBps": round(x.size / 1e6 / (st["ms_total"] * 1e-3), 1)}))
