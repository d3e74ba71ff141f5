"""``python -m openglottal_amd.cli run <video> --pipeline {unet,unet-only}``.

Counterpart of the two U-Net branches of `openglottal/cli.py:46-103` (`_cmd_run`): same flags, same
`features.json` payload (every key of the feature dict incl. the ``_area`` waveform as a list, `cli.py:97`), same
messages; ``--annotate`` additionally records ``pipeline``/``video`` in the file.  ``<video>`` may be a ``.npy``/``.npz`` frame stack (or an AVI when OpenCV
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
    r.add_argument("--annotate", action="store_true", help="also write the pipeline and video names into features.json")
    r.add_argument("--precision", choices=["f32", "split"], default="f32",
                   help="f32: exact fp32 kernels (default, the parity reference); split: opt-in f16 hi/lo split precision "
                        "(2.5x faster, same reference fixtures and tolerance; fails loudly if an activation leaves the f16 range)")
    a = ap.parse_args(argv)

    import torch

    from . import TemporalDetector, UNet, extract_features_unet

    if a.pipeline == "unet" and not a.yolo_weights:
        ap.error("--yolo-weights is required for --pipeline unet")
    model = UNet(1, 1, (32, 64, 128, 256)).to(a.device)
    model.load_state_dict(torch.load(a.unet_weights, map_location="cpu", weights_only=True))
    model.eval()
    if a.precision == "split":
        model.set_option("precision", 1)
    detector = TemporalDetector(a.yolo_weights) if a.pipeline == "unet" else None
    feats = extract_features_unet(a.video, detector, model, a.device)
    if feats is None:
        print("No glottis detected — check your weights or input video.")
        return 1
    os.makedirs(a.output, exist_ok=True)
    path = os.path.join(a.output, "features.json")
    save = {k: v.tolist() if hasattr(v, "tolist") else v for k, v in feats.items()}   # as cli.py:97: all keys, _area as a list
    if a.annotate:
        save.update(pipeline=a.pipeline, video=str(a.video))
    with open(path, "w") as f:
        json.dump(save, f, indent=2)
    print(f"Features saved to {path}")
    for k, v in feats.items():
        if not k.startswith("_"):
            print(f"  {k}: {v:.4f}" if isinstance(v, float) else f"  {k}: {v}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
