"""Durations and gaps of the last N kernels of a rocprofv3 kernel_trace.csv.  usage: trace_gaps.py <csv> [N]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-int(sys.argv[2]) if len(sys.argv) > 2 else -38:]
t0 = int(last[0]["Start_Timestamp"]); prev = None; busy = 0
for r in last:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev else 0.0
    busy += e - s
    print(f"{(s - t0) / 1e3:8.1f} us  dur {(e - s) / 1e3:6.1f}  gap {gap:6.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8s} {r['Kernel_Name'][:50]}")
    prev = e
print(f"span {(prev - t0) / 1e3:.1f} us, kernels busy {busy / 1e3:.1f} us")
