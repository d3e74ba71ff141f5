#!/usr/bin/env python3
"""Throughput bench of the U-Net-only glottal segmentation path on MI355X.

Counterpart of the reference's scripts/benchmark_video_speed.py:69-109.  Frames are the reference's synthetic frames,
seeded as SURVEY 8(d) asks: frame i = RandomState(1234+i).randint(0,256,(256,256,3),uint8) BGR.  A "step" is one pass of
the frame loop (BGR->gray, u8 -> /255 -> U-Net -> sigmoid -> >0.5 -> per-frame area) over this rank's frames, followed by
the area-waveform all-gather (RCCL) when N > 1.  ONE JSON line:

  value           frames/s with the (gray) frames already RESIDENT IN HBM when the timed region starts and the int32 areas
                  left on the device (the contract's headline).
  host_inclusive  the reference's timed region (benchmark_video_speed.py:83-109 / SURVEY 8(d)): pinned host BGR u8 frames
                  -> H2D -> BGR->gray on the device -> U-Net -> int32 areas back on the host, through the streaming engine
                  (og_unet_stream_u8); N = 1 only.
  roofline        dominant kernel against the f32 MFMA peak, HIP events around every launch, live in this run.  The 64-column
                  3x3 layers run in Winograd F(2x2,3x3) form (all f32): `achieved` counts the direct form's FLOPs,
                  `mfma_executed` the 16/36 of them the matrix pipe issues.
  direct_form     the same chain on the direct 3x3 kernels (option wino=0): frames/s, chain fraction and dominant-kernel roofline.
  cpu_baseline    the oracle's torch-CPU restatement of the same loop body on the host cores, 1 thread and all cores.

  python bench.py                                   # 1 GPU, 512 frames per step
  python bench.py --gpus 8                          # starts its own 8 ranks (one process per GPU, RCCL), relays rank 0's line
  python bench.py --gpus 8 --total-frames 10000     # config C4: ONE 10 000-frame video sharded over the ranks (strong scaling)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...
                                                    # the same ranks started by an outer launcher (RANK / WORLD_SIZE set): used as they are
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense, = f32 vector peak
PEAK_HBM_BYTES = 8.0e12
LAYER_BOUNDARY_BYTES_PER_FRAME = 183.5e6  # every layer reads its inputs once and writes its output once, fp32
PEAK_F16_MFMA_TFLOPS = 2500.0       # MI355X_MICROARCH.md: dense f16/bf16 MFMA; the split-precision mode spends 3 f16 MFMA FLOPs per f32 FLOP


def host_cores() -> int:
    """CPUs this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0))
    try:
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            n = min(n, max(1, int(int(q) / int(p))))
    except Exception:
        pass
    return n


def cpu_model() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def kernel_source_sha() -> str:
    """Identifies the device code a committed PMC measurement belongs to."""
    h = hashlib.sha256()
    for f in ("og_kernels.hpp", "og_api.hip"):
        h.update(open(os.path.join(ROOT, "openglottal_amd", "csrc", f), "rb").read())
    return h.hexdigest()[:16]


def pmc_traffic(kernel: str, frames_per_launch: int):
    """HBM bytes per launch of `kernel` from the newest committed rocprofv3 PMC passes (profiles/*pmc_traffic_chunkN.json,
    tools/pmc_traffic.py: separate FETCH_SIZE / WRITE_SIZE runs of this command).  Counters cannot be collected inside this
    process; the entry therefore names its source file and says whether that file was measured on the device code that is
    running now (`same_kernels`: sha of the kernel sources stored next to the numbers)."""
    import glob

    paths = sorted(glob.glob(os.path.join(ROOT, "profiles", f"*pmc_traffic_chunk{frames_per_launch}.json")))
    if not paths:
        return None
    d = json.load(open(paths[-1]))
    want = kernel.replace(",", ", ").replace(">", "")  # rocprof prints all template args, e.g. "<2, 0, 16, 2>"
    for name, v in d.items():
        if isinstance(v, dict) and want in name:
            return {"hbm_bytes_per_launch": round(v["hbm_bytes_per_launch"]), "read": round(v["read_bytes_per_launch"]),
                    "write": round(v["write_bytes_per_launch"]), "source": os.path.basename(paths[-1]),
                    "same_kernels": d.get("_kernel_source_sha") == kernel_source_sha()}
    return None


def cpu_baseline(sd, budget_s: float = 10.0, max_frames: int = 256):
    """Reference loop semantics (one frame per call, batch 1, BGR->gray included) on the host cores, timed on the oracle's
    torch-CPU restatement (the same oneDNN kernels the reference runs), at 1 thread and at all cores."""
    import torch

    from oracle import unet_oracle as O

    from openglottal_amd import synth
    from openglottal_amd.utils import bgr_to_gray

    sd_t = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in sd.items()}
    frames = [synth.bench_frame_bgr(i) for i in range(8)]

    def one(f):
        g = bgr_to_gray(f)
        x = torch.from_numpy(g.astype("float32") / 255.0)[None, None]
        with torch.no_grad():
            prob = torch.sigmoid(O.forward_torch(sd_t, x)).squeeze().numpy()
        return float(np.sum(((prob > 0.5).astype(np.uint8) * 255) > 0))

    def run(threads, budget):
        torch.set_num_threads(threads)
        for i in range(2):
            one(frames[i])
        n, t0 = 0, time.perf_counter()
        while n < max_frames and time.perf_counter() - t0 < budget:
            one(frames[n % 8])
            n += 1
        return n, time.perf_counter() - t0

    cores = host_cores()
    n1, e1 = run(1, budget_s * 0.5)
    na, ea = run(cores, budget_s)
    return {"value": round(na / ea, 2), "unit": "frames/s", "cores": int(cores), "kind": "port",
            "value_1_thread": round(n1 / e1, 2), "cpu_model": cpu_model(),
            "sample": f"{na} frames at {cores} threads in {ea:.1f} s, {n1} frames at 1 thread in {e1:.1f} s; seeded 256x256 BGR frames, "
                      "one frame per call (BGR->gray, /255, oracle.forward_torch fp32, sigmoid, >0.5, sum), as features.py:234-238"}


def pipeline_legs(model, dev, chunk: int):
    """Secondary keys (N = 1): the pipelines either side of the U-Net-only headline, on this run's kernels.
    gated_c3         BASELINE configs[2], YOLO+UNet detection-gated (features.py:234-245 with a detector): `area_waveform` on a BGR
                     video -- batched YOLOv8n pass of block k + 1 on a worker thread under the U-Net pass of block k, the temporal state
                     machine (detector.py:61-96) per frame on the host, areas counted inside the boxes by the fused head -- on 502
                     frames (the reference harness's length) and 2 000; plus the reference's literal per-frame loop on 200 frames.
    crop_c5_standin  BASELINE configs[4] on one GPU: the BAGLS evaluation loop (scripts/eval_bagls.py:120-232: U-Net-only, YOLO+UNet,
                     YOLO-Crop+UNet rows; two U-Net passes per frame) over 3 500 mixed-size frames, everything on the device.
    Random-init detector weights (the reference's are absent): speed, not detection quality; detector parity is unpinned."""
    import openglottal_amd as og
    from openglottal_amd import evaluate as E, synth
    from openglottal_amd.features import area_waveform
    from openglottal_amd.utils import bgr_to_gray, unet_segment_frame
    from openglottal_amd.yolo import YoloV8Detector

    det = og.TemporalDetector(YoloV8Detector(synth.make_yolov8_state_dict(seed=7, cls_bias=1.0), device=dev), conf=0.25)
    bgr = np.stack([synth.bench_frame_bgr(i) for i in range(2000)])
    area_waveform(bgr[:256], det, model)      # warm-up: detector arena, ring, graphs
    gated = {"unit": "frames/s", "region": "host BGR u8 video -> areas on the host (area_waveform: streamed, detector pass of block k+1 under the U-Net pass of block k)"}
    waves = {}
    for n in (502, 2000):
        t0 = time.perf_counter()
        waves[n] = area_waveform(bgr[:n], det, model)
        gated[f"frames_{n}"] = round(n / (time.perf_counter() - t0), 1)
    assert np.array_equal(waves[502], waves[2000][:502])
    gated["value"] = gated["frames_2000"]
    gated["frames_with_a_box"] = int((waves[2000] > 0).sum())
    det.reset()
    n_pf = 200
    for f in bgr[:16]:
        det.detect(f); unet_segment_frame(bgr_to_gray(f), model, dev)
    det.reset()
    t0 = time.perf_counter()
    pf = np.zeros(n_pf)
    for i, f in enumerate(bgr[:n_pf]):            # the reference's loop body, one call each per frame (features.py:235-245)
        mask = unet_segment_frame(bgr_to_gray(f), model, dev)
        box = det.detect(f)
        if box is not None:
            x1, y1, x2, y2 = box
            pf[i] = float(np.sum(mask[y1:y2, x1:x2] > 0))
    gated["per_frame_loop"] = {"value": round(n_pf / (time.perf_counter() - t0), 1), "unit": "frames/s", "frames": n_pf,
                               "same_areas_as_streamed": bool(np.array_equal(pf, waves[502][:n_pf]))}
    det.reset()
    t0 = time.perf_counter()
    po = np.zeros(n_pf)
    for i, f in enumerate(bgr[:n_pf]):            # the same loop with the detector's chain started before / collected after the U-Net call
        det.submit(f)
        mask = unet_segment_frame(bgr_to_gray(f), model, dev)
        box = det.result()
        if box is not None:
            x1, y1, x2, y2 = box
            po[i] = float(np.sum(mask[y1:y2, x1:x2] > 0))
    gated["per_frame_loop"]["detector_overlapped"] = {"value": round(n_pf / (time.perf_counter() - t0), 1), "unit": "frames/s",
                                                      "same_areas": bool(np.array_equal(po, pf)),
                                                      "note": "TemporalDetector.submit(frame) ... unet_segment_frame(gray) ... result(): the mask does not depend on the box"}
    t0 = time.perf_counter()
    frames, gts = synth.bagls_standin(3500)
    t_gen = time.perf_counter() - t0
    feats = (32, 64, 128, 256)
    cm = og.UNet(1, 1, feats)
    cm.load_state_dict(synth.make_unet_state_dict(feats, seed=11, head_scale=3.0, head_bias=-2.5))
    cm.to(dev).eval()
    cm.set_chunk(chunk)
    E.evaluate_device(frames[:256], gts[:256], model, det, cm)
    t0 = time.perf_counter()
    agg, st = E.evaluate_device(frames, gts, model, det, cm)
    el = time.perf_counter() - t0
    crop = {"value": round(3500 / el, 1), "unit": "frames/s", "frames": 3500, "seconds": round(el, 3), "generate_s_outside_region": round(t_gen, 1),
            "region": "host originals (mixed sizes) -> packed H2D -> canvas letterbox, BGR->gray, stateless YOLO pass, full-frame U-Net, gated row, "
                      "crop -> 256x256 -> project back, confusion counts on the device -> 40 bytes per frame back (evaluate_device)",
            "det_stats": st, "mean_dice": {k: round(float(np.mean(v["dice"])), 4) for k, v in agg.items()}}
    return gated, crop


def launch_command(argv: list[str], n: int, port: int) -> list[str]:
    """The command `--gpus N` expands to when no launcher has set RANK / WORLD_SIZE: one process per GPU under
    torch.distributed.run on this node, rendezvous on 127.0.0.1 (the container hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def pick_json_line(text: str) -> str | None:
    """Rank 0's result line among whatever else the children wrote to stdout."""
    for line in reversed(text.splitlines()):
        line = line.strip()
        if line.startswith("{") and line.endswith("}") and '"metric"' in line:
            return line
    return None


def self_launch(argv: list[str], n: int) -> int:
    """Parent of `python bench.py --gpus N` (N > 1, no RANK in the environment).  It never touches the GPU (no torch import, no
    HIP call): it starts N fresh ranks as CHILD processes, lets their stderr through, relays rank 0's JSON line and returns the
    launcher's exit code (non-zero if any rank failed)."""
    import socket
    import subprocess

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes needs it on this driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    p = subprocess.run(launch_command(argv, n, port), stdout=subprocess.PIPE, env=env, text=True)
    line = pick_json_line(p.stdout or "")
    if line is not None:
        print(line, flush=True)
    elif p.stdout:
        sys.stderr.write(p.stdout)
    if p.returncode != 0:
        sys.stderr.write(f"bench.py: a rank failed (launcher exit code {p.returncode})\n")
        return p.returncode
    if line is None:
        sys.stderr.write("bench.py: the ranks printed no result line\n")
        return 1
    return 0


def plumbing_selftest(args) -> int:
    """What a rank does around the device work, without the device work (tests/test_bench_launcher.py)."""
    import torch
    import torch.distributed as dist

    from openglottal_amd.dist import all_gather_areas, env_rank_world, shard_range

    rank, _, world = env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if world > 1:
        dist.init_process_group("gloo")
    if rank == args.plumbing_fail_rank:
        return 3
    n_total = args.total_frames if args.total_frames > 0 else args.frames * world
    lo, hi = shard_range(n_total, rank, world) if args.total_frames > 0 else (rank * args.frames, (rank + 1) * args.frames)
    local = torch.arange(lo, hi, dtype=torch.int32) % 1000
    wave = all_gather_areas(local, n_total) if world > 1 else local
    ok = bool(torch.equal(wave.to(torch.int32), torch.arange(n_total, dtype=torch.int32) % 1000))
    c4 = None
    if world > 1 and args.total_frames <= 0 and not args.no_c4_strong:   # the extra strong-scaling leg of an N > 1 run: 10 000 frames, ragged last rank
        lo4, hi4 = shard_range(10000, rank, world)
        w4 = all_gather_areas(torch.arange(lo4, hi4, dtype=torch.int32) % 1000, 10000)
        sizes = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
        dist.all_gather(sizes, torch.tensor([hi4 - lo4], dtype=torch.int64))
        c4 = {"frames": 10000, "per_rank_frames": [int(v.item()) for v in sizes],
              "waveform_ok": bool(torch.equal(w4.to(torch.int32), torch.arange(10000, dtype=torch.int32) % 1000))}
        ok = ok and c4["waveform_ok"]
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "plumbing selftest (no device work; not a measurement)", "world": world, "n_gpus": world,
                          "frames_per_step_all_gpus": n_total, "waveform_ok": ok, "scaling": "strong" if args.total_frames > 0 else "weak",
                          "c4_strong": c4}), flush=True)
    return 0 if ok else 4


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--frames", type=int, default=512, help="frames per GPU per step (weak scaling; the default mode)")
    ap.add_argument("--total-frames", type=int, default=0,
                    help="> 0: ONE video of this many frames sharded over the ranks with shard_range (strong scaling; "
                         "10000 = BASELINE config C4), last rank ragged")
    ap.add_argument("--chunk", type=int, default=64, help="frames per kernel chain (micro-batch of the frame loop)")
    ap.add_argument("--no-graphs", action="store_true")
    ap.add_argument("--lanes", type=int, default=2, choices=(1, 2),
                    help="2 = odd micro-batches on the twin handle's stream (default); 1 = single stream "
                         "(per-kernel durations under rocprofv3 are then not inflated by the other lane)")
    ap.add_argument("--no-latency-mode", action="store_true", help="skip the 1-frame-per-launch leg")
    ap.add_argument("--no-direct-form", action="store_true", help="skip the leg that times the same chain on the direct 3x3 kernels (wino=0)")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="og_unet_set_option knob for A/B measurements (results are bit-identical across them), repeatable")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-host-inclusive", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the Dice-delta leg on the reference fixture")
    ap.add_argument("--no-pipelines", action="store_true",
                    help="skip the secondary pipeline legs (N = 1): gated_c3 = YOLO+UNet streamed (BASELINE configs[2]), "
                         "crop_c5_standin = the BAGLS loop on 3 500 mixed-size frames (configs[4] on one GPU)")
    ap.add_argument("--no-c4-strong", action="store_true",
                    help="N > 1 without --total-frames: skip the extra c4_strong leg (ONE 10 000-frame video sharded over the ranks)")
    ap.add_argument("--no-split-precision", action="store_true", help="skip the exploratory split-precision leg")
    ap.add_argument("--plumbing-selftest", action="store_true",
                    help="CPU-only check of the N > 1 plumbing (self-launch, rendezvous, shard_range, ragged all-gather over gloo, "
                         "JSON relay, exit codes) with NO device work: the line it prints says so and is not a measurement")
    ap.add_argument("--plumbing-fail-rank", type=int, default=-1, help=argparse.SUPPRESS)
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(sys.argv[1:], args.gpus))   # one process per GPU; this parent has not touched the GPU

    if args.plumbing_selftest:
        raise SystemExit(plumbing_selftest(args))

    # stdout carries ONE JSON line: libraries that chat on file descriptor 1 (RCCL prints a version banner at communicator
    # creation, gloo its connection report) are sent to stderr until the line is printed
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    import openglottal_amd as og
    from openglottal_amd import synth
    from openglottal_amd.dist import all_gather_areas, env_rank_world, shard_range

    rank, local_rank, world = env_rank_world()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    # OG_BENCH_BACKEND=gloo: rehearsal of the N > 1 logic (sharding, ragged last rank, gather, max-over-ranks timing) with several
    # ranks SHARING one GPU (RCCL refuses two ranks on one device): ranks map onto the visible devices round-robin and the tiny
    # collectives go through host memory.  The product path and the driver's runs use nccl (= RCCL over xGMI).
    backend = os.environ.get("OG_BENCH_BACKEND", "nccl")
    n_dev = max(1, torch.cuda.device_count())
    dev_index = local_rank if backend == "nccl" else local_rank % n_dev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    force_dist = os.environ.get("OG_BENCH_FORCE_DIST") == "1"  # rehearse the RCCL path on one GPU (world 1)
    if world > 1 or force_dist:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)
    coll_dev = dev if backend == "nccl" else torch.device("cpu")

    feats = (32, 64, 128, 256)
    sd = synth.make_unet_state_dict(feats, seed=20260227, head_scale=3.4732823371887207, head_bias=-2.890756130218506)
    model = og.UNet(1, 1, feats)
    model.load_state_dict(sd)
    model.to(dev).eval()
    model.set_chunk(args.chunk)
    model.set_graphs(not args.no_graphs)
    model.set_option("dual", 1 if args.lanes == 2 else 0)
    for kv in args.option:
        k, v = kv.split("=")
        model.set_option(k, int(v))

    strong = args.total_frames > 0
    if strong:
        n_total = args.total_frames
        lo, hi = shard_range(n_total, rank, world)
    else:
        n_total = args.frames * world
        lo, hi = rank * args.frames, (rank + 1) * args.frames
    F = hi - lo
    # this rank's frames of the seeded stream (frame i = RandomState(1234+i), benchmark_video_speed.py:69 seeded), pinned
    bgr_host = torch.empty((max(F, 1), 256, 256, 3), dtype=torch.uint8).pin_memory()
    bh = bgr_host.numpy()
    for j in range(F):
        bh[j] = synth.bench_frame_bgr(lo + j)
    bgr_dev = bgr_host[:F].to(dev)
    frames = torch.empty((max(F, 1), 256, 256), dtype=torch.uint8, device=dev)   # gray, resident in HBM before timing
    if F:
        model.bgr2gray_dev(bgr_dev, F, 256, 256, frames)
        model.sync()
    del bgr_dev
    area = torch.zeros(max(F, 1), dtype=torch.int32, device=dev)

    busy = [0.0]   # this rank's own time in the frame loop (launch -> its last area counted), without the waits for other ranks

    def step():
        t_in = time.perf_counter()
        if F:
            model.segment_dev(frames, F, 256, 256, area)
        model.sync()  # kernels run on the handle's stream; the collective on torch's
        busy[0] += time.perf_counter() - t_in
        return all_gather_areas(area[:F].to(coll_dev), n_total, force=force_dist) if (world > 1 or force_dist) else area[:F]

    def fence():
        if world > 1 or force_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        wave = step()
    fence()
    busy[0] = 0.0
    t0 = time.perf_counter()
    for _ in range(args.steps):
        wave = step()
    fence()
    el = time.perf_counter() - t0
    per_rank = [[F, busy[0]]]
    if world > 1:
        t = torch.tensor([el], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
        mine = torch.tensor([float(F), busy[0]], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = [[int(v[0].item()), float(v[1].item())] for v in allr]
    assert wave.numel() == n_total and int(wave.min()) >= 0
    wave = wave.clone()   # `area` is reused by the legs below

    out = None
    if rank == 0:
        fps = args.steps * n_total / el
        out = {
            "metric": "frames/sec 256×256 U-Net-only (frames resident in HBM)", "value": round(fps, 1), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3),
            "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "U-Net-only 256x256 grayscale synthetic video (BASELINE configs[1]; configs[3] with --total-frames 10000), "
                                   "features (32,64,128,256), frames = RandomState(1234+i) BGR -> gray, frame loop u8->/255->UNet->sigmoid->>0.5->area, "
                                   "gray frames resident in HBM, int32 areas left on the device",
                       "frames_per_step_all_gpus": n_total, "frames_this_rank": F, "frames_per_launch": args.chunk,
                       "hip_graphs": not args.no_graphs, "lanes": args.lanes,
                       "conv_form": ("direct 3x3 convs (--option wino=0)" if any(o == "wino=0" for o in args.option) else
                                     "3x3 convs in Winograd F(2x2,3x3) form: f32 transforms, f32 MFMA, 16/36 of the direct form's multiplies; "
                                     "same reference fixtures and tolerance as the direct kernels (--option wino=0)"),
                       "sharding": (f"{'one video' if strong else 'frames'} x{world} (shard_range), all_gather(int32 area) per step over {backend}" if world > 1 else "none"),
                       "flop_per_frame": model.flops_per_frame(256, 256)},
            "tflops": round(fps * model.flops_per_frame(256, 256) / 1e12, 2),
            # what the collective layer itself reports, and every rank's own rate (frames of its shard / its own time in the
            # frame loop, waits for the other ranks excluded): the scaling curve's raw material
            "world": world,
            "collective": ({"backend": dist.get_backend(), "ranks": dist.get_world_size(),
                            "rccl_version": ".".join(str(v) for v in torch.cuda.nccl.version()) if backend == "nccl" else None}
                           if (world > 1 or force_dist) else None),
            "per_rank_fps": [round(args.steps * f / b, 1) if b > 0 else 0.0 for f, b in per_rank],
            "per_rank_frames": [int(f) for f, _ in per_rank],
        }
        # whole-chain fractions (wall clock): binding roof = f32 MFMA; HBM with SURVEY 8(d)'s layer-boundary model
        out["chain_algorithmic_over_mfma_peak"] = round(out["tflops"] / PEAK_F32_MFMA_TFLOPS / world, 4)   # algorithmic FLOPs (the Winograd layers issue 16/36 of theirs): not a roofline fraction
        out["chain_frac_hbm_layer_boundary_model"] = round(fps * LAYER_BOUNDARY_BYTES_PER_FRAME / PEAK_HBM_BYTES / world, 4)
    if world == 1 and not args.no_parity:
        # BASELINE metric, second half ("Dice delta vs CPU ref"): the 128-frame full-width fixture that the reference's own
        # unet_segment_frame produced on the CPU (tests/golden/unet_full128.npz), run with exactly this configuration
        gpath = os.path.join(ROOT, "tests", "golden", "unet_full128.npz")
        if os.path.exists(gpath):
            g = np.load(gpath)
            gfr, ggt = synth.full128_frames()
            gd = torch.from_numpy(gfr).to(dev)
            ga = torch.zeros(128, dtype=torch.int32, device=dev)
            gm = torch.zeros((128, 256, 256), dtype=torch.uint8, device=dev)
            model.segment_dev(gd, 128, 256, 256, ga, mask_dev=gm)
            model.sync()
            mk, ar = gm.cpu().numpy(), ga.cpu().numpy()
            ref = np.unpackbits(g["masks_packed"], axis=1)[:, :65536].reshape(128, 256, 256)
            flips = int(((mk > 0) != (ref > 0)).sum())
            dd = max(abs(og.dice(mk[i], ggt[i]) - float(g["dice_vs_gt"][i])) for i in range(80))
            band = float(np.load(os.path.join(ROOT, "tests", "golden", "unet_full128_self_noise.npz"))["band"])
            out["parity"] = {"fixture": "tests/golden/unet_full128.npz (reference unet_segment_frame on CPU, 80 structured + 48 stream frames)",
                             "dice_delta_vs_cpu_ref_max": round(dd, 8), "flipped_mask_pixels": flips, "of_pixels": 128 * 65536,
                             "frames_with_area_difference": int((ar.astype(np.int64) != g["areas"]).sum()),
                             "band": band,
                             "note": "every flipped pixel sits where the reference's own logit is within `band` of zero: the reference's own run-to-run "
                                     "difference on these frames, tests/golden/unet_full128_self_noise.npz (tests/test_gpu_bench_config.py)"}
            # the HARD fixture (tests/golden/unet_trained_hard.npz: the trained net de-tuned to full f32 mantissas and near-zero logits
            # over whole regions, evaluated by the reference): flips only inside the band, frames with margins above it exact
            hpath, wpath = (os.path.join(ROOT, "tests", "golden", f) for f in ("unet_trained_hard.npz", "unet_trained_full.npz"))
            if os.path.exists(hpath) and os.path.exists(wpath):
                gh, g9 = np.load(hpath), np.load(wpath)
                hm = og.UNet(1, 1, feats)
                hm.load_state_dict(synth.detuned_weights({k[2:]: g9[k] for k in g9.files if k.startswith("W:")}))
                hm.to(dev).eval()
                hm.set_chunk(args.chunk); hm.set_graphs(not args.no_graphs); hm.set_option("dual", 1 if args.lanes == 2 else 0)
                hfr = np.concatenate([synth.glottis_frames(4, 20, seed=99)[0], synth.degraded_glottis_frames()[0]])
                nh = len(hfr)
                ha = torch.zeros(nh, dtype=torch.int32, device=dev)
                hk = torch.zeros((nh, 256, 256), dtype=torch.uint8, device=dev)
                hm.segment_dev(torch.from_numpy(hfr).to(dev), nh, 256, 256, ha, mask_dev=hk)
                hm.sync()
                hmk, har = hk.cpu().numpy(), ha.cpu().numpy().astype(np.int64)
                href = np.unpackbits(gh["masks_packed"], axis=1)[:, :65536].reshape(nh, 256, 256)
                fl = np.argwhere(((hmk > 0) != (href > 0)).reshape(nh, -1))
                nz = {(int(f), int(p)): float(v) for f, p, v in zip(gh["near_zero_frame"], gh["near_zero_pixel"], gh["near_zero_logit"])}
                safe = gh["abs_logit_min"] > band
                out["parity"]["hard_fixture"] = {
                    "fixture": "tests/golden/unet_trained_hard.npz (reference unet_segment_frame on the de-tuned trained net: full f32 mantissas; "
                               "104 frames)",
                    "reference_pixels_abs_logit_lt_1e-2": int(json.load(open(os.path.join(ROOT, "tests", "golden", "meta.json")))["trained_hard"]["n_abs_logit_lt_1e2"]),
                    "reference_pixels_abs_logit_lt_1e-3": int(len(gh["near_zero_logit"])),
                    "reference_pixels_inside_band": int((np.abs(gh["near_zero_logit"]) <= band).sum()),
                    "flipped_mask_pixels": int(len(fl)), "of_pixels": nh * 65536,
                    "flips_outside_band": int(sum(abs(nz.get((int(f), int(p)), 1.0)) > band for f, p in fl)),
                    "frames_with_area_difference": int((har != gh["areas"]).sum()),
                    "frames_with_margin_above_band": int(safe.sum()),
                    "of_those_exact": int(sum(bool(np.array_equal(hmk[i] > 0, href[i] > 0)) and har[i] == gh["areas"][i] for i in np.flatnonzero(safe)))}
                del hm
    if not args.no_host_inclusive:
        # SURVEY 8(d) / benchmark_video_speed.py:83-109: first H2D enqueue -> last area on the host, BGR->gray inside; at N > 1
        # every rank streams ITS shard from its own pinned host memory and the areas are all-gathered inside the region
        hf = bgr_host[:F]
        if F:
            model.segment_stream(hf)   # warm-up: one whole pass (the ring's lanes + 2 pinned slots and every lane's graphs are created on first use)
        reps = max(1, min(args.steps, 5))
        fence(); t1 = time.perf_counter()
        for _ in range(reps):
            a_host = model.segment_stream(hf)[1] if F else np.zeros(0, np.int32)
            if world > 1 or force_dist:
                w_host = all_gather_areas(torch.from_numpy(a_host).to(coll_dev), n_total, force=force_dist)
        fence()
        eh = time.perf_counter() - t1
        if world > 1:
            t = torch.tensor([eh], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            eh = float(t.item())
            assert torch.equal(w_host.cpu().to(torch.int32), wave.cpu().to(torch.int32))
        else:
            assert np.array_equal(a_host, wave.cpu().numpy())       # same integers as the resident leg
        if rank == 0:
            out["host_inclusive"] = {"value": round(reps * n_total / eh, 1), "unit": "frames/s", "frames": n_total, "passes": reps,
                                     "region": "pinned host BGR u8 [F,256,256,3] -> H2D -> BGR->gray (device) -> U-Net -> int32 areas on the host "
                                               "(og_unet_stream_u8: ring of pinned micro-batches, copies under the kernel chains)"
                                               + (", + all_gather of the areas; max over ranks" if world > 1 else ""),
                                     "pcie_bytes_per_frame": 256 * 256 * 3 + 4}

    def one_frame_per_chain(lanes=0):
        n1 = min(F, 256)
        model.set_chunk(1)
        model.set_option("lanes", lanes)
        model.segment_dev(frames, n1, 256, 256, area); model.sync()
        e1 = 1e9
        for _ in range(3):
            fence(); t1 = time.perf_counter()
            model.segment_dev(frames, n1, 256, 256, area); model.sync()
            fence(); e1 = min(e1, time.perf_counter() - t1)
        model.set_option("lanes", 0)
        model.set_chunk(args.chunk)
        return {"frames_per_launch": 1, "value": round(n1 / e1, 1), "unit": "frames/s", "frames": n1}, area[:n1].clone()

    if world == 1 and F and not args.no_latency_mode:
        # BASELINE configs[1] wording "batch=1": one frame per kernel chain, frames still resident in HBM.  Same (canonical)
        # arithmetic as the headline, so the same integers
        out["latency_mode"], a1 = one_frame_per_chain()
        assert torch.equal(a1, wave[:a1.numel()].to(a1.device))
        out["latency_mode"]["note"] = ("canonical form (identical areas to the headline leg, asserted); three lanes; one frame per chain takes the "
                                       "wave-split Winograd kernels (k_conv_wino_w / _wp) and k_convt_w: the same sums as the 64-frame launches")
        l1, a1b = one_frame_per_chain(1)
        assert torch.equal(a1b, a1)
        out["latency_mode"]["one_lane"] = {"value": l1["value"], "unit": "frames/s", "us_per_frame": round(1e6 / l1["value"], 1)}
        # the reference's call pattern itself (utils.py:218-241): one host frame in, one mask out, one synchronisation per call
        from openglottal_amd.utils import unet_segment_frame
        gh_ = frames[:64].cpu().numpy()
        for g_ in gh_[:16]:
            unet_segment_frame(g_, model, dev)
        t1 = time.perf_counter()
        for g_ in gh_:
            unet_segment_frame(g_, model, dev)
        out["latency_mode"]["unet_segment_frame_ms_per_call"] = round(1e3 * (time.perf_counter() - t1) / len(gh_), 4)
        if not any(o.startswith(("wino=", "splitk=", "precision=")) for o in args.option):
            # opt-in, non-canonical: direct kernels with K split over workgroups (round 2's latency path); a frame's logits
            # then depend on the size of its launch, which is why it is not the default
            model.set_option("wino", 0); model.set_option("splitk", 1)
            out["latency_mode_optin_splitk"], a1s = one_frame_per_chain()
            out["latency_mode_optin_splitk"]["frames_whose_area_differs_from_canonical"] = int((a1s != a1).sum())
            model.set_option("wino", 1); model.set_option("splitk", 0)
    def roofline(kernel_prefix, peak_equiv, note=None):
        """Dominant kernel (largest share of chain time among kernels starting with `kernel_prefix`) from HIP events around
        every launch of one eager chain.  `achieved` / `frac` = MFMA FLOPs the kernel EXECUTES / its time, against the dense peak
        of the instruction it issues: a fraction of a roof, <= 1 by construction.  A Winograd launch executes 16/36 of the
        direct form's (= SURVEY 8(d)'s algorithmic) multiplies: `algorithmic_tflops` / `algorithmic_over_peak` carry that count."""
        B = min(args.chunk, F)
        prof = model.profile(frames, B, 256, 256, reps=max(3, min(20, args.steps)))
        ex = lambda p: p["flops"] / (2.25 if p["kernel"].startswith("k_conv_wino") else 1.0)
        per = {}
        for p in prof:
            per[p["kernel"]] = per.get(p["kernel"], 0.0) + p["ms"]
        name = max((k for k in per if k.startswith(kernel_prefix)), key=lambda k: per[k])
        dom = [p for p in prof if p["kernel"] == name]
        fl, fx, ms = sum(p["flops"] for p in dom), sum(ex(p) for p in dom), sum(p["ms"] for p in dom)
        tot_ms, tot_fl, tot_fx = sum(p["ms"] for p in prof), sum(p["flops"] for p in prof), sum(ex(p) for p in prof)
        ach, alg = fx / (ms * 1e-3) / 1e12, fl / (ms * 1e-3) / 1e12
        tr = pmc_traffic(name, B)   # HBM bytes per launch from the committed rocprofv3 PMC passes (or None)
        r = {"bound": "mfma", "kernel": name, "achieved": round(ach, 2), "peak": peak_equiv,
             "unit": "TFLOP/s", "frac": round(ach / peak_equiv, 4),
             "algorithmic_tflops": round(alg, 2), "algorithmic_over_peak": round(alg / peak_equiv, 4),
             "traffic": (tr or {}).get("hbm_bytes_per_launch") if (tr or {}).get("same_kernels") else None,
             "traffic_detail": tr, "launches_per_chain": len(dom), "avg_launch_ms": round(ms / len(dom), 4),
             "share_of_chain_time": round(ms / tot_ms, 3), "frames_per_launch": B, "chain_ms": round(tot_ms, 3),
             "chain": {"executed_tflops": round(tot_fx / (tot_ms * 1e-3) / 1e12, 2), "frac": round(tot_fx / (tot_ms * 1e-3) / 1e12 / peak_equiv, 4),
                       "algorithmic_tflops": round(tot_fl / (tot_ms * 1e-3) / 1e12, 2),
                       "note": "all launches of one chain on one lane, HIP events per launch"},
             "kernel_source_sha": kernel_source_sha()}
        if name.startswith("k_conv_wino"):
            note = (note + "; " if note else "") + ("Winograd F(2x2,3x3) in f32: achieved / frac count the MFMA FLOPs issued (16/36 of the direct form's); "
                                                    "algorithmic_* count SURVEY 8(d)'s direct-form FLOPs")
        if note:
            r["note"] = note
        return r, {k: round(v, 4) for k, v in per.items()}

    if world == 1 and F and not args.no_roofline:
        out["roofline"], out["per_kernel_ms"] = roofline("k_conv_", PEAK_F32_MFMA_TFLOPS if not any(o.startswith("precision=1") for o in args.option) else round(PEAK_F16_MFMA_TFLOPS / 3, 1))
    if world == 1 and F and not args.no_direct_form and not any(o.startswith(("wino=", "precision=")) for o in args.option):
        # The same chain with the DIRECT 3x3 kernels (round 1's arithmetic, 36 multiplies per output window instead of 16): what
        # every launch that does not fill the chip takes, and the continuity figure against the f32 MFMA peak
        model.set_option("wino", 0)
        for _ in range(args.warmup):
            step()
        fence(); t1 = time.perf_counter()
        for _ in range(args.steps):
            wd = step()
        fence(); ed = time.perf_counter() - t1
        fpsd = args.steps * n_total / ed
        wd = wd.clone()
        rld, perd = roofline("k_conv_mfma_o", PEAK_F32_MFMA_TFLOPS)
        out["direct_form"] = {"value": round(fpsd, 1), "unit": "frames/s", "ms_per_step": round(1e3 * ed / args.steps, 3),
                              "tflops": round(fpsd * model.flops_per_frame(256, 256) / 1e12, 2),
                              "chain_frac_mfma": round(fpsd * model.flops_per_frame(256, 256) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),   # direct form: executed = algorithmic
                              "frames_whose_area_differs_from_canonical_form": int((wd != wave).sum()),
                              "max_area_difference_px": int((wd.to(torch.int64) - wave.to(torch.int64)).abs().max()),
                              "note": "another arithmetic form (option wino=0), compared ACROSS forms: both inside the reference's own noise band",
                              "roofline": rld, "per_kernel_ms": perd}
        model.set_option("wino", 1)
    if world == 1 and F and not args.no_split_precision and not any(o.startswith("precision=") for o in args.option):
        # Exploratory secondary mode, NEVER the headline: f16 hi/lo operand pairs, 3 x v_mfma_f32_32x32x16_f16 per f32 product,
        # f32 accumulation; passes the same reference fixtures at the same tolerance (tests/test_gpu_split_precision.py)
        model.set_option("precision", 1)
        for _ in range(args.warmup):
            step()
        fence(); t1 = time.perf_counter()
        for _ in range(args.steps):
            w2 = step()
        fence(); e2 = time.perf_counter() - t1
        fps2 = args.steps * n_total / e2
        w2 = w2.clone()   # `area` is reused by the legs below
        flips = int((w2 != wave).sum())
        max_da = int((w2.to(torch.int64) - wave.to(torch.int64)).abs().max())
        rl, per = roofline("k_conv_mfma_h", round(PEAK_F16_MFMA_TFLOPS / 3, 1),
                           "achieved = algorithmic f32-equivalent FLOP/s; peak = dense f16 MFMA peak / 3 (three f16 MFMAs per f32 product); "
                           "the bare 3-MFMA loop on random data sustains 454-526 of it (tools/ubench/mfma_f16_split)")
        lat2 = None
        if not args.no_latency_mode:
            n1 = min(F, 256)
            model.set_chunk(1)
            model.segment_dev(frames, n1, 256, 256, area); model.sync()
            fence(); t1 = time.perf_counter()
            model.segment_dev(frames, n1, 256, 256, area); model.sync()
            fence(); e1 = time.perf_counter() - t1
            model.set_chunk(args.chunk)
            lat2 = {"frames_per_launch": 1, "value": round(n1 / e1, 1), "unit": "frames/s", "frames": n1}
        hi2 = None
        if not args.no_host_inclusive:
            hf = bgr_host[:F]
            model.segment_stream(hf)   # warm-up: one whole pass, as in the f32 leg
            reps = max(1, min(args.steps, 5))
            fence(); t1 = time.perf_counter()
            for _ in range(reps):
                _, a_host2 = model.segment_stream(hf)
            eh2 = time.perf_counter() - t1
            assert np.array_equal(a_host2, w2.cpu().numpy())
            hi2 = {"value": round(reps * F / eh2, 1), "unit": "frames/s", "frames": F, "passes": reps}
        out["split_precision"] = {"value": round(fps2, 1), "host_inclusive": hi2, "latency_mode": lat2, "unit": "frames/s", "vs_f32_path": round(fps2 / fps, 3), "dtype": "f16 hi/lo x3 MFMA, f32 accumulate",
                                  "ms_per_step": round(1e3 * e2 / args.steps, 3), "tflops_f32_equivalent": round(fps2 * model.flops_per_frame(256, 256) / 1e12, 2),
                                  "frames_whose_area_differs_from_f32_path": flips, "max_area_difference_px": max_da,
                                  "area_note": "both precisions sit inside the logit band of the reference fixture (the reference's own noise); they decide pixels whose "
                                               "logit is ~0 differently (seeded random weights on noise frames have many such pixels; the trained fixture none)",
                                  "chain_frac_hbm_layer_boundary_model": round(fps2 * LAYER_BOUNDARY_BYTES_PER_FRAME / PEAK_HBM_BYTES, 4),
                                  "roofline": rl, "per_kernel_ms": per}
        model.set_option("precision", 0)
    if world == 1 and F and not args.no_pipelines:
        out["gated_c3"], out["crop_c5_standin"] = pipeline_legs(model, dev, args.chunk)
    if world > 1 and not strong and not args.no_c4_strong:
        # BASELINE configs[3] as written, in the SAME line as the weak-scaling loop the driver's command measures: ONE 10 000-frame
        # video, contiguous shards (shard_range: the last rank is ragged), every rank streams ITS shard from its own pinned host memory
        # through the frame loop, and the int32 areas are all-gathered INSIDE the timed region (features.py:234-245 sharded)
        n4 = 10000
        lo4, hi4 = shard_range(n4, rank, world)
        F4 = hi4 - lo4
        h4 = torch.empty((max(F4, 1), 256, 256, 3), dtype=torch.uint8).pin_memory()
        h4n = h4.numpy()
        for j in range(F4):
            h4n[j] = synth.bench_frame_bgr(lo4 + j)
        model.segment_stream(h4[:min(F4, 6 * args.chunk)])   # warm-up: every ring slot and lane used once
        busy4 = 0.0
        reps4 = 3
        fence(); t4 = time.perf_counter()
        for _ in range(reps4):
            tb = time.perf_counter()
            a4 = model.segment_stream(h4[:F4])[1] if F4 else np.zeros(0, np.int32)
            busy4 += time.perf_counter() - tb
            w4 = all_gather_areas(torch.from_numpy(a4).to(coll_dev), n4)
        fence()
        e4 = time.perf_counter() - t4
        t = torch.tensor([e4], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        e4 = float(t.item())
        mine = torch.tensor([float(F4), busy4], dtype=torch.float64, device=coll_dev)
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        assert w4.numel() == n4 and int(w4.min()) >= 0
        if rank == 0:
            # the first frames of the video are the weak leg's rank-0 frames: the same integers, whatever the sharding
            k4 = min(n4, args.frames)
            assert torch.equal(w4[:k4].cpu().to(torch.int32), wave[:k4].cpu().to(torch.int32))
            out["c4_strong"] = {"value": round(reps4 * n4 / e4, 1), "unit": "frames/s", "scaling": "strong", "frames": n4, "passes": reps4,
                                "region": "one 10 000-frame BGR video in pinned host memory, shard_range over the ranks (last rank ragged): H2D -> BGR->gray "
                                          "-> U-Net -> areas -> all_gather(int32) inside the region; max over ranks",
                                "collective": {"backend": dist.get_backend(), "ranks": dist.get_world_size()},
                                "per_rank_frames": [int(v[0].item()) for v in allr],
                                "per_rank_fps": [round(reps4 * v[0].item() / v[1].item(), 1) if v[1].item() > 0 else 0.0 for v in allr]}
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(sd)
    if world > 1 or force_dist:
        dist.destroy_process_group()
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    if rank == 0:
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
