"""Per-kernel timeline of the last complete kernel chain in a rocprofv3 kernel-trace csv; argv[2] = substring of the chain's LAST kernel."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
fr = rows[idx[-2] + 1: idx[-1] + 1]
t0 = int(fr[0]['Start_Timestamp'])
tot = 0
for r in fr:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    tot += e - s
    print(f"{(s - t0) / 1e3:8.1f} {(e - s) / 1e3:7.1f} grid={r['Grid_Size_X']:>6}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']:<4} {r['Kernel_Name'][:64]}")
print(len(fr), 'kernels; busy', tot / 1e3, 'us; span', (int(fr[-1]['End_Timestamp']) - t0) / 1e3)
print('chain period', (int(rows[idx[-1]]['Start_Timestamp']) - int(rows[idx[-2]]['Start_Timestamp'])) / 1e3)
