"""Per-LAYER HBM traffic of the kernel chain from the two rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, one counter per run,
tools/profile_r03.sh): the dispatches of one chain come in a fixed order (k_conv_first, the 21 conv / transposed-conv launches,
k_sum_counts), so dispatch position = layer; averaged over all chains of the run.  Counters in KiB, FETCH_SIZE doubled as
MI355X_MICROARCH.md prescribes for gfx950 (tools/pmc_traffic.py).  Joined with the HIP-event table of tools/layer_profile.py.
usage: pmc_per_layer.py <fetch.csv> <write.csv> <layer_profile.txt> > table"""
import csv, sys


def chains(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out, cur = [], None
    for r in rows:
        n = r["Kernel_Name"]
        if n.startswith("void k_conv_first"):
            cur = []
            out.append(cur)
        if cur is not None and n.startswith(("void k_conv", "k_sum_counts")):
            cur.append((n, float(r["Counter_Value"])))
            if n.startswith("k_sum_counts"):
                cur = None
    return [c for c in out if c and c[-1][0].startswith("k_sum_counts")]


f, w = chains(sys.argv[1], "FETCH_SIZE"), chains(sys.argv[2], "WRITE_SIZE")
L = max(len(c) for c in f)
f = [c for c in f if len(c) == L]
w = [c for c in w if len(c) == L]
prof = [l.rstrip("\n") for l in open(sys.argv[3]) if l.strip()]
head, rows = prof[0], prof[2:]
print(head)
print(f"HBM-side bytes per launch: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) / WRITE_SIZE, averaged over {len(f)} chains of the same command")
print(f"{'layer':36s} {'kernel':20s} {'ms':>8s} {'executed frac':>13s} {'read MB':>9s} {'write MB':>9s} {'HBM MB':>9s} {'GB/s':>8s}")
for i in range(L - 1):   # the last dispatch is k_sum_counts (not in the event table)
    rd = 2.0 * sum(c[i][1] for c in f) / len(f) * 1024 / 1e6
    wr = sum(c[i][1] for c in w) / len(w) * 1024 / 1e6
    p = rows[i].split()
    layer, kernel, ms, frac = p[0], p[1], float(p[2]), float(p[-1])
    assert f[0][i][0].replace("void ", "").split("(")[0].replace(", ", ",").startswith(kernel.split("<")[0]), (f[0][i][0], kernel)
    print(f"{layer:36s} {kernel:20s} {ms:8.4f} {frac:13.3f} {rd:9.1f} {wr:9.1f} {rd + wr:9.1f} {(rd + wr) / ms:8.0f}")
