"""In-kernel timeline of the position-split (k_conv_wino_ps) and wave-split (k_conv_wino_w) Winograd launches of a one-frame chain: per launch, medians over the workgroups of
entry -> loop start, loop, loop end -> exit (non-reducing workgroups) / -> exit of the reducing workgroup, in microseconds of
shader clock (s_memtime / measured MHz)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(B)).cuda()
prof = m.profile(fr, B, 256, 256, reps=2)
m.clock_probe(fr, B, 256, 256)
mhz = 2400.0
buf = np.zeros((1024, 4), np.uint64)
for kv in sys.argv[2:]:
    k_, v_ = kv.split("="); m.set_option(k_, int(v_))
prof = m.profile(fr, B, 256, 256, reps=2)
m.clock_probe(fr, B, 256, 256)
for i, p in enumerate(prof[:-0 or None]):
    if not p["kernel"].startswith(("k_conv_wino_ps", "k_conv_wino_w")):
        continue
    check(lib().og_unet_clock_probe_raw(m._h, i, ptr(buf)), "raw")
    v = buf[buf[:, 0] > 0]
    red = (v[:, 3] >> np.uint64(63)).astype(bool)
    t = (v & np.uint64((1 << 63) - 1)).astype(np.float64) / mhz
    t0 = t[:, 0].min()
    pro, loop, tail = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]
    print(f"{p['layer']:28s} {p['kernel']:20s} wgs {len(v):4d} | prologue {np.median(pro):5.2f} | loop {np.median(loop):6.2f} "
          f"| tail non-reducer {np.median(tail[~red]) if (~red).any() else 0:5.2f} reducer {np.median(tail[red]) if red.any() else 0:5.2f} | first entry -> last exit {t[:,3].max()-t0:6.2f} us | event {p['ms']*1e3:6.1f} us")
