"""In-kernel shader clock per conv launch (diagnostic build path of k_conv_mfma_p)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(B)
for kv in sys.argv[2:]:
    k, v = kv.split("="); m.set_option(k, int(v))
F = 256
frames = torch.from_numpy(synth.bulk_gray_frames(F)).cuda()
area = torch.zeros(F, dtype=torch.int32, device="cuda")
for _ in range(6):
    m.segment_dev(frames, F, 256, 256, area)
m.sync()
prof = m.profile(frames, B, 256, 256, reps=3)
for rep in range(2):
    for _ in range(3):
        m.segment_dev(frames, F, 256, 256, area)
    mhz = m.clock_probe(frames, B, 256, 256)
    print("rep", rep)
    for p, c in zip(prof, mhz):
        tf = p["flops"] / (p["ms"] * 1e-3) / 1e12
        peak_at_clock = 157.3 * c / 2400.0 if c > 0 else 0
        print(f"{p['layer']:26s} {p['kernel']:24s} {c:8.1f} MHz  {tf:6.1f} TF/s = {100*tf/157.3:5.1f}% of 2.4GHz peak, "
              f"{(100*tf/peak_at_clock if c > 0 else 0):5.1f}% of peak at measured clock")

import numpy as np
from openglottal_amd._lib import lib, ptr, check
print("\nper-launch workgroup timeline (us, from s_memrealtime @100MHz)")
for i, p in enumerate(prof):
    raw = np.zeros((1024, 4), np.uint64)
    check(lib().og_unet_clock_probe_raw(m._h, i, ptr(raw)), "raw")
    M40 = np.uint64(0xFFFFFFFFFF)
    ok = (raw[:, 3] & M40) > (raw[:, 1] & M40)
    if not ok.any():
        continue
    r0, r1 = (raw[ok, 1] & M40).astype(np.float64) / 100.0, (raw[ok, 3] & M40).astype(np.float64) / 100.0
    cu = (raw[ok, 1] >> np.uint64(40)).astype(np.int64)   # (xcc<<16) | hw_id[23:8]
    if p["layer"] in ("ups.1.net.0.weight", "downs.1.net.3.weight"):
        xcc = (cu >> 16) & 0xF
        busy_ = r1 - r0
        print("  per-XCC mean busy us:", " ".join(f"x{x}:{busy_[xcc == x].mean():.0f}({(xcc == x).sum()})" for x in sorted(set(xcc.tolist()))))
        hw = cu & 0xFFFF   # HW_ID[23:8]: cu_id[3:0] sh_id[4] se_id[7:5] ...
        key = (xcc << 8) | (hw & 0xFF)
        per = {}
        for k_, b_ in zip(key.tolist(), busy_.tolist()):
            per.setdefault(k_, []).append(b_)
        sizes = sorted(len(v) for v in per.values())
        spread = [max(v) - min(v) for v in per.values() if len(v) > 1]
        print(f"  (xcc,se,sh,cu) groups: {len(per)} sizes {sizes[0]}..{sizes[-1]}; within-group spread mean {np.mean(spread):.1f} max {np.max(spread):.1f}; "
              f"between-group std {np.std([np.mean(v) for v in per.values()]):.1f}")
    span = r1.max() - r0.min()
    busy = (r1 - r0)
    print(f"{p['layer']:26s} wgs={ok.sum():4d} span={span:7.1f}us event_ms={p['ms']*1e3:7.1f}us start_skew(max)={r0.max()-r0.min():6.1f} "
          f"end_skew: p50={np.median(r1.max()-r1):6.1f} max={(r1.max()-r1).max():6.1f}  wg_busy mean={busy.mean():7.1f} min={busy.min():7.1f}")
