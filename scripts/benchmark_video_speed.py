#!/usr/bin/env python3
"""Frames-per-second of the glottal-area frame loop on MI355X, two ways over the same frames.

Command line compatible with the reference harness of the same name (same flag names), implementation
written against THIS package:

  per-frame   one `unet_segment_frame` call (and, with --yolo-weights, one `TemporalDetector.detect` call) per frame with
              host arrays in and out -- the call pattern of the reference's frame loop (features.py:234-245)
  streamed    the whole video in one `area_waveform` call: frames go to the device in micro-batches, BGR->gray, U-Net,
              threshold and area count happen there, the detector network runs batched, only the areas come back

Both passes use the same arithmetic (the canonical form of DESIGN.md §4), so their area waveforms must be identical; the
script checks that and reports it.  Frames are the seeded stream `RandomState(1234 + i)` of SURVEY §8(d), or a
`.npy` / `.npz` / image-directory video given with --video.  Weight files that do not exist are replaced by the seeded
synthetic weights of the test-suite, and the output says so (the reference's snapshot ships no weights).

  python scripts/benchmark_video_speed.py --frames 502
  python scripts/benchmark_video_speed.py --frames 502 --yolo-weights weights/openglottal_yolo.npz
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

WIDTHS = (32, 64, 128, 256)
PAPER_FPS = 502 / 11.0   # the reference's published claim: a 502-frame video in about 11 s on Apple MPS


def parse_args() -> argparse.Namespace:
    ap = argparse.ArgumentParser(description="Time the U-Net (optionally YOLO-gated) area-waveform loop on an MI355X.")
    ap.add_argument("--frames", type=int, default=502, help="how many frames to process (502 = a median GIRAFE recording)")
    ap.add_argument("--device", default="cuda", help="'cuda' or 'cuda:N'; this package has no CPU or MPS path")
    ap.add_argument("--unet-weights", default="weights/openglottal_unet.pt", help="torch state_dict of the U-Net (weights_only load)")
    ap.add_argument("--yolo-weights", default=None, help="flat .npz export of the YOLOv8n detector; enables the detection-gated pipeline")
    ap.add_argument("--warmup", type=int, default=20, help="frames pushed through both passes before the clock starts")
    ap.add_argument("--video", default=None, help=".npy / .npz / directory of images to read frames from (read time reported separately)")
    ap.add_argument("--json", action="store_true", help="also print the numbers as one JSON line")
    return ap.parse_args()


def build_unet(path: str, device: str):
    import torch

    from openglottal_amd import UNet, synth

    net = UNet(1, 1, WIDTHS).to(device)
    if os.path.exists(path):
        net.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
        origin = path
    else:
        net.load_state_dict(synth.make_unet_state_dict(WIDTHS, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506))
        origin = f"seeded synthetic weights ({path} does not exist)"
    return net.eval(), origin


def build_detector(path: str | None, device: str):
    if not path:
        return None, None
    from openglottal_amd import TemporalDetector, synth
    from openglottal_amd.yolo import YoloV8Detector

    if os.path.exists(path):
        return TemporalDetector(path), path
    backend = YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device=device)
    return TemporalDetector(backend), f"random-init YOLOv8n ({path} does not exist)"


def frame_source(video: str | None, count: int):
    """(list of BGR frames, seconds spent reading them, description)."""
    from openglottal_amd import synth
    from openglottal_amd.features import load_frames_bgr

    if video is None:
        return [synth.bench_frame_bgr(i) for i in range(count)], 0.0, f"{count} seeded 256x256 BGR frames, RandomState(1234+i)"
    t = time.perf_counter()
    frames = load_frames_bgr(video)[:count]
    return frames, time.perf_counter() - t, f"{len(frames)} frames of {video}"


def per_frame_pass(frames, net, detector, device) -> np.ndarray:
    """One device round trip per frame, as the reference does it."""
    from openglottal_amd.utils import bgr_to_gray, unet_segment_frame

    wave = np.zeros(len(frames))
    if detector is not None:
        detector.reset()
    for i, frame in enumerate(frames):
        mask = unet_segment_frame(bgr_to_gray(frame), net, device)
        if detector is None:
            wave[i] = np.count_nonzero(mask)
            continue
        hit = detector.detect(frame)
        if hit is not None:
            left, top, right, bottom = hit
            wave[i] = np.count_nonzero(mask[top:bottom, left:right])
    return wave


def per_frame_overlapped_pass(frames, net, detector, device) -> np.ndarray:
    """Still one frame at a time, but the detector's network is started BEFORE the U-Net call of the same frame and collected after it
    (`TemporalDetector.submit` / `result`): the mask does not depend on the box, so the two chains share the GPU."""
    from openglottal_amd.utils import bgr_to_gray, unet_segment_frame

    wave = np.zeros(len(frames))
    detector.reset()
    for i, frame in enumerate(frames):
        detector.submit(frame)
        mask = unet_segment_frame(bgr_to_gray(frame), net, device)
        hit = detector.result()
        if hit is not None:
            left, top, right, bottom = hit
            wave[i] = np.count_nonzero(mask[top:bottom, left:right])
    return wave


def streamed_pass(frames, net, detector, device) -> np.ndarray:
    from openglottal_amd.features import area_waveform

    return area_waveform(frames, detector, net, device)


def timed(fn, *a):
    import torch

    torch.cuda.synchronize()
    t = time.perf_counter()
    out = fn(*a)
    torch.cuda.synchronize()
    return out, time.perf_counter() - t


def main() -> None:
    args = parse_args()
    net, unet_origin = build_unet(args.unet_weights, args.device)
    detector, det_origin = build_detector(args.yolo_weights, args.device)
    frames, read_s, what = frame_source(args.video, args.frames)
    n = len(frames)
    if n == 0:
        raise SystemExit("no frames to process")
    print(f"pipeline : {'YOLO + U-Net, area counted inside the tracked box' if detector else 'U-Net only'}")
    print(f"U-Net    : {unet_origin}")
    if detector:
        print(f"detector : {det_origin}")
    print(f"frames   : {what}" + (f" (read in {read_s:.2f} s)" if read_s else ""))

    head = frames[:max(1, min(n, args.warmup))]
    per_frame_pass(head, net, detector, args.device)
    streamed_pass(frames[:max(len(head), min(n, 128))], net, detector, args.device)   # also sizes the pinned ring and captures the graphs

    loop_wave, loop_s = timed(per_frame_pass, frames, net, detector, args.device)
    stream_wave, stream_s = timed(streamed_pass, frames, net, detector, args.device)
    same = int(np.count_nonzero(loop_wave == stream_wave))

    res = {"frames": n, "pipeline": "yolo+unet" if detector else "unet-only",
           "per_frame_fps": round(n / loop_s, 1), "per_frame_ms_per_frame": round(1e3 * loop_s / n, 3),
           "streamed_fps": round(n / stream_s, 1), "identical_areas": same,
           "max_area_difference_px": int(np.abs(loop_wave - stream_wave).max()), "read_s": round(read_s, 3)}
    print(f"\nper-frame calls : {loop_s:8.3f} s  = {res['per_frame_fps']:9.1f} frames/s  ({res['per_frame_ms_per_frame']} ms per frame)")
    if detector is not None:
        per_frame_overlapped_pass(head, net, detector, args.device)
        ov_wave, ov_s = timed(per_frame_overlapped_pass, frames, net, detector, args.device)
        res["per_frame_overlapped_fps"] = round(n / ov_s, 1)
        res["per_frame_overlapped_identical_areas"] = int(np.count_nonzero(ov_wave == loop_wave))
        print(f"  detector under the U-Net call of the same frame (submit / result): {ov_s:8.3f} s  = {res['per_frame_overlapped_fps']:9.1f} frames/s, "
              f"{res['per_frame_overlapped_identical_areas']} of {n} areas identical")
    print(f"streamed video  : {stream_s:8.3f} s  = {res['streamed_fps']:9.1f} frames/s")
    print(f"area waveforms  : {same} of {n} frames identical, largest difference {res['max_area_difference_px']} px")
    verdict = "faster than" if res["per_frame_fps"] >= PAPER_FPS else "SLOWER than"
    print(f"published figure: {PAPER_FPS:.1f} frames/s (502 frames in ~11 s, Apple MPS); the per-frame pass here is {verdict} it "
          f"({502 / res['per_frame_fps']:.2f} s for 502 frames)")
    if args.json:
        print(json.dumps(res))


if __name__ == "__main__":
    main()
