"""Fixed vs per-frame cost of the streamed entry point (og_unet_stream_u8: pinned host BGR in, areas out) against the resident
loop: frames/s by video length, so that the per-call fill / drain of the ring shows as the intercept."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth

feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
m.set_chunk(64)
for a in sys.argv[1:]:
    k, v = a.split("=")
    m.set_option(k, int(v))
N = 2048
bgr = torch.from_numpy(np.stack([synth.bench_frame_bgr(i % 64) for i in range(N)])).pin_memory().numpy()
gray = torch.from_numpy(synth.bulk_gray_frames(64)).cuda().repeat(N // 64, 1, 1).contiguous()
area = torch.zeros(N, dtype=torch.int32, device="cuda:0")
m.segment_stream(bgr[:128]); m.segment_dev(gray, 128, 256, 256, area); m.sync()
rows = []
for F in (64, 128, 256, 512, 1024, 2048):
    reps = max(2, 4096 // F)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); m.segment_stream(bgr[:F]); ts.append(time.perf_counter() - t0)
    tr = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); m.segment_dev(gray, F, 256, 256, area); m.sync(); tr.append(time.perf_counter() - t0)
    s, r = float(np.median(ts)), float(np.median(tr))
    rows.append((F, s, r))
    print(f"F {F:5d}: streamed {s * 1e3:8.3f} ms ({F / s:7.0f} frames/s)   resident {r * 1e3:8.3f} ms ({F / r:7.0f} frames/s)   difference {(s - r) * 1e3:6.3f} ms", flush=True)
F_, s_, r_ = map(np.array, zip(*rows))
ps = np.polyfit(F_, s_, 1); pr = np.polyfit(F_, r_, 1)
print(f"streamed: {ps[1] * 1e3:.3f} ms + {ps[0] * 1e6:.2f} us/frame   resident: {pr[1] * 1e3:.3f} ms + {pr[0] * 1e6:.2f} us/frame")
