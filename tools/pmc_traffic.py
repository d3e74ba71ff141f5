"""Parse rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-launch HBM traffic per kernel.

Corrections as MI355X_MICROARCH.md §HBM prescribes: counters are in KiB; on gfx950
FETCH_SIZE reports 1/2 of the bytes of a wide coalesced read stream -> doubled; WRITE_SIZE
is taken as is.  (Both are memory-side counters and include Infinity-Cache hits.)
usage: pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>
"""
import csv, hashlib, json, os, sys
from collections import defaultdict


def per_kernel(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    return acc


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(f) | set(w)):
    if not k.startswith(("void k_", "k_")):
        continue
    fk, wk = f.get(k, [0, 1]), w.get(k, [0, 1])
    rd = 2.0 * fk[0] / max(fk[1], 1) * 1024
    wr = wk[0] / max(wk[1], 1) * 1024
    out[k] = {"launches": fk[1], "read_bytes_per_launch": rd, "write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr}
_root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_h = hashlib.sha256()
for _f in ("og_kernels.hpp", "og_api.hip"):   # the device code these numbers were measured on (bench.py compares it with what is running)
    _h.update(open(os.path.join(_root, "openglottal_amd", "csrc", _f), "rb").read())
out["_kernel_source_sha"] = _h.hexdigest()[:16]
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if not isinstance(v, dict):
        continue
    print(f"{k[:60]:60s} n={v['launches']:4d} rd={v['read_bytes_per_launch']/1e6:9.2f} MB wr={v['write_bytes_per_launch']/1e6:9.2f} MB")
