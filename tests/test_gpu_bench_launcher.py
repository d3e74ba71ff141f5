"""-m gpu: `python bench.py --gpus N` end to end on the box.  The box has ONE GPU, so the two ranks the command starts share it and
the area-waveform all-gather goes over gloo (`OG_BENCH_BACKEND=gloo`; on a node the same command runs one rank per GPU over RCCL):
self-launch from a parent that never touches the GPU, sharding, ragged strong-scaling video, the gathered waveform checked inside
bench.py against the per-rank results, per-rank rates, the host-inclusive leg with its all-gather, ONE JSON line on stdout."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra):
    env = dict(os.environ, OG_BENCH_BACKEND="gloo")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline"] + extra,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_starts_its_own_two_ranks_weak_scaling():
    out = _bench(["--gpus", "2", "--frames", "128"])
    assert out["n_gpus"] == 2 and out["world"] == 2 and out["scaling"] == "weak" and out["unit"] == "frames/s"
    assert out["collective"]["ranks"] == 2 and out["collective"]["backend"] == "gloo"
    assert out["config"]["frames_per_step_all_gpus"] == 256 and out["per_rank_frames"] == [128, 128]
    assert len(out["per_rank_fps"]) == 2 and all(v > 100 for v in out["per_rank_fps"])
    assert out["value"] > 1000 and out["host_inclusive"]["value"] > 1000 and out["host_inclusive"]["frames"] == 256
    assert "roofline" not in out and "cpu_baseline" not in out and "gated_c3" not in out          # N = 1 legs only
    # config C4 in the same line: one 10 000-frame video, contiguous shards, the all-gather inside the region, per-rank rates
    c4 = out["c4_strong"]
    assert c4["frames"] == 10000 and c4["per_rank_frames"] == [5000, 5000] and c4["scaling"] == "strong" and c4["collective"]["ranks"] == 2
    assert c4["value"] > 1000 and len(c4["per_rank_fps"]) == 2 and all(v > 100 for v in c4["per_rank_fps"])


def test_bench_strong_scaling_ragged_video_on_two_ranks():
    out = _bench(["--gpus", "2", "--total-frames", "301"])           # ceil(301 / 2) = 151 + 150
    assert out["world"] == 2 and out["scaling"] == "strong"
    assert out["config"]["frames_per_step_all_gpus"] == 301 and out["per_rank_frames"] == [151, 150]
    assert out["value"] > 1000
