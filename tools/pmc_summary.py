"""Average the counters of a rocprofv3 --pmc counter_collection.csv per kernel name (plus mean dispatch duration).
usage: pmc_summary.py <counter_collection.csv> [substring]"""
import csv, sys
from collections import OrderedDict, defaultdict
path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_")
disp = OrderedDict()
for r in csv.DictReader(open(path)):
    if sub not in r["Kernel_Name"]:
        continue
    d = disp.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"].replace("void ", "").split("(")[0],
                                           "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = d.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
agg = defaultdict(lambda: defaultdict(float))
cnt = defaultdict(int)
for d in disp.values():
    cnt[d["k"]] += 1
    for k, v in d.items():
        if k != "k":
            agg[d["k"]][k] += v
names = sorted({k for a in agg.values() for k in a if k != "ns"})
print("kernel launches mean_us " + " ".join(names))
for k, a in agg.items():
    n = cnt[k]
    print(k, n, f"{a['ns'] / n / 1e3:.1f}", " ".join(f"{a[c] / n:.4g}" for c in names))
