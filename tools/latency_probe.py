"""Where does batch-1 ("latency mode": one frame per kernel chain) spend its time: host enqueue vs device?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
fr = torch.from_numpy(synth.bulk_gray_frames(256)).cuda(); area = torch.zeros(256, dtype=torch.int32, device="cuda")
for prec in (0, 1):
    m.set_option("precision", prec)
    for graphs in (1, 0):
        m.set_graphs(bool(graphs))
        for lanes in (1, 2, 3):
            m.set_option("lanes", lanes); m.set_chunk(1)
            m.segment_dev(fr, 256, 256, 256, area); m.sync()
            t0 = time.perf_counter(); m.segment_dev(fr, 256, 256, 256, area); t1 = time.perf_counter(); m.sync(); t2 = time.perf_counter()
            print(f"precision {prec} graphs {graphs} lanes {lanes}: enqueue {1e6*(t1-t0)/256:7.1f} us/frame, total {1e6*(t2-t0)/256:7.1f} us/frame -> {256/(t2-t0):7.0f} fps", flush=True)
