"""Pivot a rocprofv3 counter_collection.csv: one row per dispatch of kernels matching a substring."""
import csv, sys
from collections import OrderedDict
path, sub = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "k_conv")
rows = OrderedDict()
for r in csv.DictReader(open(path)):
    if sub not in r["Kernel_Name"]:
        continue
    d = rows.setdefault(r["Dispatch_Id"], {"k": r["Kernel_Name"][:40], "grid": r["Grid_Size"],
                                           "ns": int(r["End_Timestamp"]) - int(r["Start_Timestamp"])})
    d[r["Counter_Name"]] = float(r["Counter_Value"])
names = []
for d in rows.values():
    for k in d:
        if k not in ("k", "grid", "ns") and k not in names:
            names.append(k)
print("disp kernel grid ns " + " ".join(names))
for i, d in rows.items():
    print(i, d["k"], d["grid"], d["ns"], " ".join(f"{d.get(n, 0):.4g}" for n in names))
