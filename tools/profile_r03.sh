#!/bin/bash
# Round-3 evidence run on the GPU box.  Writes under gpurun_out/prof_r03/; the summaries are then copied to profiles/r03_*.
#   rocprofv3 kernel stats of bench.py on ONE lane (per-kernel durations are then not those of two kernels sharing the chip),
#   the two PMC passes behind roofline.traffic (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains beside them),
#   per-layer tables at 64 frames and at one frame per chain, the one-frame latency table, the in-kernel timeline of the
#   position-split launches, the self-launched N > 1 rehearsals (ranks sharing the one GPU over gloo), the harness script.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/prof_r03
mkdir -p $O
COMMON="--lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-latency-mode --no-host-inclusive --no-split-precision --no-parity"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $COMMON > $O/bench_under_rocprof_lanes1.json 2> $O/stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python3 bench.py $COMMON --no-roofline --no-direct-form > /dev/null 2> $O/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python3 bench.py $COMMON --no-roofline --no-direct-form > /dev/null 2> $O/write.err
echo "write rc=$?"
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_chunk64.json > $O/pmc_traffic_chunk64.txt; echo "pmc rc=$?"
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats_chunk64_lanes1.csv
python3 tools/layer_profile.py 64 2>&1 | grep -v amdgpu > $O/layer_profile_wino_chunk64.txt
python3 tools/layer_profile.py 1 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain.txt
python3 tools/layer_profile.py 1 wino_ps=0 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain_without_position_split.txt
python3 tools/layer_profile.py 1 wino=0 splitk=1 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain_optin_splitk.txt
python3 tools/latency_r03.py 2>&1 | grep -v amdgpu > $O/latency_one_frame_per_chain.txt
python3 tools/ps_timeline.py 1 2>&1 | grep -v amdgpu | cut -c1-60,104-200 > $O/ps_timeline_one_frame.txt
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 bench.py --total-frames 10000 --no-split-precision --no-direct-form > $O/bench_c4_total_frames_10000_n1.json 2> /dev/null; echo "c4 rc=$?"
OG_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --no-cpu-baseline > $O/bench_selflaunch_gloo_shared_gpu_n2.json 2> $O/bench_selflaunch_gloo_shared_gpu_n2.err; echo "n2 rc=$?"
OG_BENCH_BACKEND=gloo python3 bench.py --gpus 3 --steps 3 --total-frames 10000 --no-cpu-baseline > $O/bench_selflaunch_gloo_shared_gpu_c4_n3.json 2> /dev/null; echo "n3 rc=$?"
OG_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --no-cpu-baseline --no-split-precision --no-direct-form > $O/bench_rccl_world1_forced_dist.json 2> $O/bench_rccl_world1_forced_dist.err; echo "rccl rc=$?"
python3 scripts/benchmark_video_speed.py --frames 502 --json 2>&1 | grep -v amdgpu > $O/benchmark_video_speed_unet_only.txt
python3 scripts/benchmark_video_speed.py --frames 502 --json --yolo-weights weights/none.npz 2>&1 | grep -v amdgpu > $O/benchmark_video_speed_gated.txt
ls -la $O
