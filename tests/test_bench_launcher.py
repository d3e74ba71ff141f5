"""CPU: `python bench.py --gpus N` starts its own ranks: argument plumbing, self-launch through
torch.distributed.run on 127.0.0.1, shard_range + ragged all-gather over gloo, relay of rank 0's JSON line, exit codes.
No device work happens here (`--plumbing-selftest`); the GPU legs of the same launcher run on the box."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_launch_command_and_relay_helpers():
    import bench

    cmd = bench.launch_command(["--gpus", "8", "--total-frames", "10000", "--steps", "3"], 8, 29511)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=8" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--total-frames", "10000", "--steps", "3"]     # the ranks get the caller's flags verbatim
    noise = 'NCCL version 2.x\n{"not": "ours"}\n{"metric": "m", "value": 1}\ntrailing chatter\n'
    assert json.loads(bench.pick_json_line(noise)) == {"metric": "m", "value": 1}
    assert bench.pick_json_line("nothing here\n") is None


def _run(extra, timeout=300):
    env = dict(os.environ)
    env.pop("RANK", None)
    env.pop("WORLD_SIZE", None)
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--plumbing-selftest"] + extra, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, text=True, timeout=timeout, env=env)


def test_self_launch_two_ranks_ragged_video():
    p = _run(["--gpus", "2", "--total-frames", "10003"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, p.stdout                     # ONE JSON line on stdout, whatever the children chatted
    out = json.loads(lines[0])
    assert out["world"] == 2 and out["n_gpus"] == 2 and out["frames_per_step_all_gpus"] == 10003 and out["waveform_ok"] and out["scaling"] == "strong"


def test_self_launch_weak_scaling_and_single_rank():
    p = _run(["--gpus", "3", "--frames", "17"])
    assert p.returncode == 0, p.stderr[-2000:]
    out = json.loads(p.stdout.strip().splitlines()[-1])
    assert out["world"] == 3 and out["frames_per_step_all_gpus"] == 51 and out["scaling"] == "weak"
    # an N > 1 run without --total-frames also carries config C4 (ONE 10 000-frame video, strong scaling, ragged last rank)
    assert out["c4_strong"]["frames"] == 10000 and out["c4_strong"]["per_rank_frames"] == [3334, 3334, 3332] and out["c4_strong"]["waveform_ok"]
    p = _run(["--gpus", "2", "--frames", "9", "--no-c4-strong"])
    assert p.returncode == 0 and json.loads(p.stdout.strip().splitlines()[-1])["c4_strong"] is None
    p = _run(["--gpus", "1", "--frames", "5"])
    assert p.returncode == 0 and json.loads(p.stdout.strip())["world"] == 1


def test_a_failing_rank_fails_the_launcher():
    p = _run(["--gpus", "2", "--plumbing-fail-rank", "1"])
    assert p.returncode != 0
    assert "rank failed" in p.stderr
