"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch and mean duration."""
import csv, glob, sys, collections
root = sys.argv[1]
for cc in sorted(glob.glob(root + "/*/*/*_counter_collection.csv")):
    setname = cc.split("/")[-3]
    rows = list(csv.DictReader(open(cc)))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    seen = set()
    for r in rows:
        name = r["Kernel_Name"].split("(")[0].replace("archon::", "")
        agg[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
        key = (r["Dispatch_Id"])
        if key not in seen and "Start_Timestamp" in r:
            seen.add(key)
            dur[name].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print("==", setname)
    for name in agg:
        parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in agg[name].items()]
        d = dur.get(name)
        print("  %-56s n=%d %s %s" % (name[:56], len(next(iter(agg[name].values()))), ("ms=%.3f" % (sum(d) / len(d))) if d else "", " ".join(parts)))
