"""BASELINE config C5 on one GPU: the BAGLS evaluation loop (scripts/eval_bagls.py) over a 3 500-frame stand-in of mixed
frame sizes, entirely on the device (evaluate.evaluate_device): canvas letterbox, BGR->gray, stateless YOLO pass, full-frame
U-Net, box-gated row, crop -> 256x256 -> project back, confusion counts.  Prints one JSON line and writes the per-frame
metrics in the shape of the reference's results/bagls_eval.json.  usage: python tools/bench_bagls.py [n] [out.json]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import openglottal_amd as og
from openglottal_amd import evaluate as E, synth
from openglottal_amd.yolo import YoloV8Detector
n = int(sys.argv[1]) if len(sys.argv) > 1 else 3500
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)
m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(64)
cm = og.UNet(1, 1, feats); cm.load_state_dict(synth.make_unet_state_dict(feats, seed=11, head_scale=3.0, head_bias=-2.5)); cm.to("cuda:0").eval(); cm.set_chunk(64)
det = og.TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device="cuda:0"), conf=0.25)
t0 = time.perf_counter(); frames, gts = synth.bagls_standin(n); t_gen = time.perf_counter() - t0
E.evaluate_device(frames[:256], gts[:256], m, det, cm)          # warm-up: arenas, graphs
t0 = time.perf_counter()
agg, st = E.evaluate_device(frames, gts, m, det, cm)
el = time.perf_counter() - t0
m.set_option("precision", 1); cm.set_option("precision", 1)     # secondary: the same loop on the split-precision kernels
E.evaluate_device(frames[:256], gts[:256], m, det, cm)
t0 = time.perf_counter()
agg2, st2 = E.evaluate_device(frames, gts, m, det, cm)
el2 = time.perf_counter() - t0
dd = {p: float(np.abs(np.array(agg[p]["dice"]) - np.array(agg2[p]["dice"])).max()) for p in agg}
out = sys.argv[2] if len(sys.argv) > 2 else os.path.join("gpurun_out", "bagls_eval_standin.json")
E.dump_json(out, agg, st, meta={"bagls_dir": "synthetic stand-in (synth.bagls_standin), random-init weights", "frames": n})
print(json.dumps({"pipeline": "C5 BAGLS evaluation loop, 3 pipelines, mixed frame sizes, all on the device", "frames": n,
                  "frames_per_s": round(n / el, 1), "seconds": round(el, 2), "generate_s": round(t_gen, 1),
                  "summary": E.summarize(agg), "det_stats": st, "bytes_back_per_frame": 40,
                  "split_precision": {"frames_per_s": round(n / el2, 1), "max_abs_per_frame_dice_difference_vs_f32": dd, "det_stats": st2}}))
