import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, numpy as np
import openglottal_amd as og
from openglottal_amd import synth
from openglottal_amd.yolo import YoloV8Detector
feats = (32, 64, 128, 256)
sd = synth.make_unet_state_dict(feats, seed=1)
ysd = synth.make_yolov8_state_dict(seed=7)
fr = synth.random_gray_frames(40, seed=3)
bgr = np.random.RandomState(0).randint(0, 256, (8, 256, 256, 3), dtype=np.uint8)
torch.cuda.init()
free0 = None
for i in range(12):
    m = og.UNet(1, 1, feats); m.load_state_dict(sd); m.to("cuda:0").eval(); m.set_chunk(8)
    m.segment(fr, want_mask=False)
    d = YoloV8Detector(ysd, device="cuda:0"); d.detect_batch(bgr)
    del m, d
    import gc; gc.collect()
    free, tot = torch.cuda.mem_get_info()
    if i == 1: free0 = free
    print(i, round((tot - free) / 2**20), "MiB used", flush=True)
print("leak per cycle MiB:", round((free0 - free) / 2**20 / 10, 2))
