"""Detector at one frame per call (the reference's TemporalDetector.detect pattern): wall time per call, split into the
python wrapper, the C-ABI host entry point, and the enqueue-only / device-only rates of the kernel chain."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd._lib import lib, ptr, check
from openglottal_amd.yolo import YoloV8Detector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
d = YoloV8Detector(synth.make_yolov8_state_dict(seed=7), device="cuda:0")
for a in sys.argv[2:]:
    k, v = a.split("=")
    d.set_option(k, int(v))
td = og.TemporalDetector(d, conf=0.25)
fr = np.random.RandomState(0).randint(0, 256, (n, 256, 256, 3), dtype=np.uint8)
for f in fr[:20]: td.detect(f)
t0 = time.perf_counter()
for f in fr: td.detect(f)
t_det = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for i in range(n): d.detect_batch(fr[i:i + 1], 0.25)
t_abi = (time.perf_counter() - t0) / n
dev = torch.from_numpy(fr).to("cuda:0")
best = torch.empty((n, 5), dtype=torch.float32, device="cuda:0")
torch.cuda.synchronize()
L = lib()
t0 = time.perf_counter()
for i in range(n):
    check(L.og_yolo_detect_u8_dev(d._h, ptr(dev[i]), 1, 256, 256, 0.25, ptr(best[i]), None), "detect_u8_dev")
t_enq = (time.perf_counter() - t0) / n
check(L.og_yolo_sync(d._h), "sync")
t_all = (time.perf_counter() - t0) / n
print(json.dumps({"detect_ms_per_frame": round(t_det * 1e3, 3), "c_abi_host_entry_ms": round(t_abi * 1e3, 3),
                  "enqueue_only_ms": round(t_enq * 1e3, 3), "back_to_back_chain_ms": round(t_all * 1e3, 3)}))
