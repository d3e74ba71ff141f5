"""``python -m openglottal_amd.cli run <video> --pipeline {unet,unet-only}``.

Counterpart of the two U-Net branches of `openglottal/cli.py:46-103` (`_cmd_run`): same flags,
same `features.json` payload (kinematic features without the private ``_area`` key, plus
``pipeline``/``video``).  ``<video>`` may be a ``.npy``/``.npz`` frame stack (or an AVI when OpenCV
is importable); weights are a torch ``state_dict`` file (U-Net, `weights_only=True`) and a flat
``.npz`` export (YOLO, see yolo.py).
"""
from __future__ import annotations

import argparse
import json
import os
import sys


def main(argv=None) -> int:
    ap = argparse.ArgumentParser(prog="openglottal_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    r = sub.add_parser("run")
    r.add_argument("video")
    r.add_argument("--pipeline", choices=["unet", "unet-only"], default="unet-only")
    r.add_argument("--unet-weights", required=True)
    r.add_argument("--yolo-weights", default=None)
    r.add_argument("--device", default="cuda")
    r.add_argument("-o", "--output", default="output")
    a = ap.parse_args(argv)

    import torch

    from . import TemporalDetector, UNet, extract_features_unet

    if a.pipeline == "unet" and not a.yolo_weights:
        ap.error("--yolo-weights is required for --pipeline unet")
    model = UNet(1, 1, (32, 64, 128, 256)).to(a.device)
    model.load_state_dict(torch.load(a.unet_weights, map_location="cpu", weights_only=True))
    model.eval()
    detector = TemporalDetector(a.yolo_weights) if a.pipeline == "unet" else None
    feats = extract_features_unet(a.video, detector, model, a.device)
    if feats is None:
        print("No features extracted (empty video or silent waveform).", file=sys.stderr)
        return 1
    os.makedirs(a.output, exist_ok=True)
    out = {k: (None if v is None else float(v)) for k, v in feats.items() if not k.startswith("_")}
    out.update(pipeline=a.pipeline, video=str(a.video))
    path = os.path.join(a.output, "features.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=2)
    print(f"Features saved to {path}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
