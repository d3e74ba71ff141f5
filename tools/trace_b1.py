import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import openglottal_amd as og
from openglottal_amd import synth
feats = (32, 64, 128, 256)
m = og.UNet(1, 1, feats); m.load_state_dict(synth.make_unet_state_dict(feats, seed=1, head_scale=3.47, head_bias=-2.89)); m.to("cuda:0").eval()
m.set_chunk(1); m.set_option("dual", int(sys.argv[1]) if len(sys.argv) > 1 else 0)
fr = torch.from_numpy(synth.bulk_gray_frames(64)).cuda(); area = torch.zeros(64, dtype=torch.int32, device="cuda")
for _ in range(3):
    m.segment_dev(fr, 64, 256, 256, area); m.sync()
