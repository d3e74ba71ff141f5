#!/usr/bin/env python3
"""Counterpart of the reference's `scripts/benchmark_video_speed.py` with the SAME command line.

It times two things over the same frames (seeded `RandomState(1234+i)` instead of the reference's unseeded randint, or a
`.npy` / `.npz` / image-directory video):

  1. the reference's loop VERBATIM (benchmark_video_speed.py:89-101) with only the imports swapped — one
     `cvtColor` + `unet_segment_frame(gray, model, device)` (+ `detector.detect(frame)`) per frame, host arrays in and out;
  2. the same work as ONE call of `area_waveform` (what `extract_features_unet` does in this package): frames streamed to the
     device in micro-batches, BGR→gray on the device, detector network batched, areas back.

Weights: a torch state_dict file (`--unet-weights`, loaded with `weights_only=True`) and a flat `.npz` export (`--yolo-weights`,
see openglottal_amd/yolo.py); when a path does not exist the seeded synthetic weights of the test-suite are used and said so
(the reference's weight files are not part of its repository snapshot).

  python scripts/benchmark_video_speed.py --frames 502 --device cuda
  python scripts/benchmark_video_speed.py --frames 502 --device cuda --yolo-weights weights/openglottal_yolo.npz
"""
from __future__ import annotations

import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main() -> None:
    p = argparse.ArgumentParser(description="Benchmark U-Net video processing speed (MI355X path).")
    p.add_argument("--frames", type=int, default=502, help="Number of frames to simulate (default: 502, GIRAFE median).")
    p.add_argument("--device", type=str, default="cuda", help="Device: cuda or cuda:N (there is no CPU / MPS path).")
    p.add_argument("--unet-weights", type=str, default="weights/openglottal_unet.pt", help="Path to U-Net weights.")
    p.add_argument("--yolo-weights", type=str, default=None, help="If set, run YOLO+UNet (slower); else U-Net only.")
    p.add_argument("--warmup", type=int, default=20, help="Warmup frames before timing.")
    p.add_argument("--video", type=str, default=None, help="Optional: .npy/.npz/frame directory to use real frames and include load time.")
    args = p.parse_args()

    import torch

    from openglottal_amd import TemporalDetector, UNet, synth
    from openglottal_amd.features import area_waveform, load_frames_bgr
    from openglottal_amd.utils import bgr_to_gray, unet_segment_frame
    from openglottal_amd.yolo import YoloV8Detector

    device = args.device
    feats = (32, 64, 128, 256)
    model = UNet(1, 1, feats).to(device)
    if os.path.exists(args.unet_weights):
        model.load_state_dict(torch.load(args.unet_weights, map_location="cpu", weights_only=True))
    else:
        print(f"Note: {args.unet_weights} not found — seeded synthetic U-Net weights (tests/golden calibration)")
        model.load_state_dict(synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506))
    model.eval()

    detector = None
    if args.yolo_weights:
        if os.path.exists(args.yolo_weights):
            detector = TemporalDetector(args.yolo_weights)
        else:
            print(f"Note: {args.yolo_weights} not found — random-init YOLOv8n weights")
            detector = TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device=device))
        print("Pipeline: YOLO+UNet (detection-gated)")
    else:
        print("Pipeline: U-Net only")

    if args.video:
        t0 = time.perf_counter()
        frames_bgr = load_frames_bgr(args.video)
        load_s = time.perf_counter() - t0
        n_frames = min(len(frames_bgr), args.frames)
        frames_bgr = frames_bgr[:n_frames]
        print(f"Loaded {n_frames} frames from {args.video} in {load_s:.2f} s")
    else:
        n_frames = args.frames
        frames_bgr = [synth.bench_frame_bgr(i) for i in range(n_frames)]
        load_s = 0.0
        print(f"Using {n_frames} synthetic 256×256 frames, RandomState(1234+i) (no load time)")

    if detector:
        detector.reset()
    for frm in frames_bgr[: args.warmup]:
        unet_segment_frame(bgr_to_gray(frm), model, device)
        if detector is not None:
            detector.detect(frm)
    torch.cuda.synchronize()

    # 1. timed run, the reference's loop body with the imports swapped (benchmark_video_speed.py:89-101)
    areas_loop = []
    t0 = time.perf_counter()
    for frm_bgr in frames_bgr:
        gray_full = bgr_to_gray(frm_bgr)
        mask_full = unet_segment_frame(gray_full, model, device)
        if detector is None:
            areas_loop.append(float(np.sum(mask_full > 0)))
        else:
            box = detector.detect(frm_bgr)
            if box is None:
                areas_loop.append(0.0)
            else:
                x1, y1, x2, y2 = box
                areas_loop.append(float(np.sum(mask_full[y1:y2, x1:x2] > 0)))
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    fps = n_frames / elapsed

    # 2. the same frames through the batched frame loop (extract_features_unet's path in this package)
    area_waveform(frames_bgr[: max(args.warmup, 64)], detector, model, device)
    t0 = time.perf_counter()
    wave = area_waveform(frames_bgr, detector, model, device)
    elapsed_b = time.perf_counter() - t0
    d = np.abs(np.asarray(areas_loop) - wave)

    print(f"\nResults ({n_frames} frames, device={device}):")
    print(f"  Per-frame loop (reference call pattern): {elapsed:.3f} s  →  {fps:.1f} frames/s")
    print(f"  Batched frame loop (area_waveform):      {elapsed_b:.3f} s  →  {n_frames / elapsed_b:.1f} frames/s")
    print(f"  Areas of the two runs: identical on {int((d == 0).sum())} of {n_frames} frames, max |difference| {int(d.max())} px "
          "(one frame per launch splits K across workgroups: sums re-associate in the last bits, which moves pixels whose logit is ~0)")
    if load_s > 0:
        print(f"  Video load time: {load_s:.2f} s")
    print("\nPaper claim: 502 frames in ~11 s (~47 frames/s) on MPS.")
    print(f"  Per-frame loop: {fps:.1f} fps  →  502 frames in {502 / fps:.2f} s  {'✓ within claim' if fps >= 502 / 11.0 else '✗'}")


if __name__ == "__main__":
    main()
