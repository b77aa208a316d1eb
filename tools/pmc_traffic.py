"""Build <tag dir>/pmc_traffic.json from the rocprofv3 --pmc passes tools/evidence.sh collects (bench.py reads the newest
profiles/rNN_final/pmc_traffic.json for roofline.traffic).

usage: tools/pmc_traffic.py <pmc dir (gpurun_out/<tag>/pmc)> <source label> <output json> [n] [shape]

Per kernel of the streaming path: mean FETCH_SIZE and WRITE_SIZE per dispatch (KiB, separate passes) ->
HBM bytes per launch = 2 x FETCH_SIZE (gfx950 tallies 128-byte read requests at 64 bytes, MI355X_MICROARCH.md
HBM section) + WRITE_SIZE.  bench.py reads the file for roofline.traffic."""
import collections
import csv
import glob
import json
import os
import sys

root, source, dest = sys.argv[1], sys.argv[2], sys.argv[3]
n = int(sys.argv[4]) if len(sys.argv) > 4 else 268435456
shape = sys.argv[5] if len(sys.argv) > 5 else "random"
ALG = {"bs::k_pass_text": 5, "bs::k_pass_rec": 9, "bs::k_local_sort": 13, "bs::k_hist16": 1}
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for cc in glob.glob(root + "/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(cc)):
        name = r["Kernel_Name"].split("(")[0].split("<")[0].replace("archon::", "").replace("void ", "").strip()
        if name in ALG and r["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
            vals[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
try:
    prev = json.load(open(dest))          # entries of other shapes / sizes / routes stay (each carries its own `source`)
except Exception:
    prev = {}
out = dict(prev)
out["_comment"] = ( "HBM traffic per launch of the streaming kernels, MI355X, rocprofv3 separate --pmc passes (tools/pmc.sh via "
                   "tools/evidence.sh): 2 x FETCH_SIZE (gfx950 under-count, MI355X_MICROARCH.md HBM section) + WRITE_SIZE, KiB -> bytes. "
                   "each entry names its source")
ent = {}
for name, c in vals.items():
    if "FETCH_SIZE" not in c or "WRITE_SIZE" not in c:
        continue
    # (both record formats of a pass are queued and the one that does not match the block returns at once: dispatches that moved
    #  less than a tenth of the kernel's largest are those empty twins, not samples)
    live = lambda v: [a for a in v if a >= 0.1 * max(v)] if max(v) > 0 else v
    f = sum(live(c["FETCH_SIZE"])) / len(live(c["FETCH_SIZE"]))
    w = sum(live(c["WRITE_SIZE"])) / len(live(c["WRITE_SIZE"]))
    ent[name] = {"fetch_size_kib": round(f, 1), "write_size_kib": round(w, 1), "hbm_bytes_per_launch": int((2 * f + w) * 1024),
                 "algorithmic_bytes_per_launch": ALG[name] * n, "source": source}
out["path1_%s_%d" % (shape, n)] = ent
json.dump(out, open(dest, "w"), indent=1)
print(json.dumps(ent, indent=1))
