"""Analyse tools/ubench/occ_timeline's dump: where a workgroup's lifetime goes and how often a SIMD has nobody in
its MFMA phase.  usage: occ_timeline.py <dump.bin> <mfma_cycles_per_wave>"""
import sys
import numpy as np

st = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
mfma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
t0, t1, t2, t3 = (st[:, i].astype(np.int64) for i in range(4))
hw, xcc = st[:, 4].astype(np.int64), st[:, 5].astype(np.int64) & 15
# s_memtime is a 100 MHz-independent shader-clock counter per XCD; compare only within an XCD
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
sh = (hw >> 12) & 1
simd = (hw >> 4) & 3
slot = hw & 15
key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
pro, main, epi, life = t1 - t0, t2 - t1, t3 - t2, t3 - t0
print(f"workgroups {len(st)}, distinct CUs {len(np.unique(key))}")
t6, t7 = st[:, 6].astype(np.int64), st[:, 7].astype(np.int64)
for name, v in (("  setup (entry -> first DMA issue)", t6 - t0), ("  DMA issue + wave 0's vmcnt(0)", t7 - t6), ("  barrier (other waves)", t1 - t7)):
    print(f"{name:36s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
for name, v in (("prologue", pro), ("main loop", main), ("epilogue+stores", epi), ("lifetime", life)):
    print(f"{name:16s} mean {v.mean():9.0f}  p10 {np.percentile(v, 10):9.0f}  p50 {np.percentile(v, 50):9.0f}  p90 {np.percentile(v, 90):9.0f} cycles")
if mfma:
    print(f"own MFMA issue time per wave {mfma:.0f} cycles = {100 * mfma / life.mean():.1f}% of the mean lifetime")
# per CU: coverage of the timeline by main-loop phases
idle_frac, conc = [], []
for k in np.unique(key):
    m = key == k
    a, b = t1[m], t2[m]
    lo, hi = t0[m].min(), t3[m].max()
    ev = np.concatenate([np.stack([a, np.ones_like(a)], 1), np.stack([b, -np.ones_like(b)], 1)])
    ev = ev[np.argsort(ev[:, 0], kind="stable")]
    cur, last, idle, wsum = 0, lo, 0, 0
    for t, d in ev:
        if cur == 0:
            idle += t - last
        wsum += cur * (t - last)
        cur += d
        last = t
    idle += hi - last
    idle_frac.append(idle / (hi - lo))
    conc.append(wsum / (hi - lo))
print(f"per CU: fraction of the kernel with NO workgroup in its main loop: mean {np.mean(idle_frac):.3f} (p10 {np.percentile(idle_frac,10):.3f}, p90 {np.percentile(idle_frac,90):.3f});"
      f" mean workgroups in main loop {np.mean(conc):.2f}")
# resident workgroups per CU over time
res = []
for k in np.unique(key)[:64]:
    m = key == k
    lo, hi = t0[m].min(), t3[m].max()
    res.append((t3[m] - t0[m]).sum() / (hi - lo))
print(f"mean resident workgroups per CU {np.mean(res):.2f}; slots seen {sorted(np.unique(slot).tolist())}")
# gap between a slot being freed and the next workgroup starting on the same (CU, SIMD, slot)
gaps = []
for k in np.unique(key)[:64]:
    m = (key == k)
    for sl in np.unique(slot[m]):
        for sd in np.unique(simd[m]):
            mm = m & (slot == sl) & (simd == sd)
            if mm.sum() < 2:
                continue
            o = np.argsort(t0[mm])
            gaps.extend((t0[mm][o][1:] - t3[mm][o][:-1]).tolist())
if gaps:
    g = np.array(gaps)
    print(f"slot turnaround (previous stores done -> next entry): p10 {np.percentile(g,10):.0f} p50 {np.percentile(g,50):.0f} p90 {np.percentile(g,90):.0f} cycles")
