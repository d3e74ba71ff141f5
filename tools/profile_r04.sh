#!/bin/bash
# Round-4 evidence run on the GPU box.  Writes under gpurun_out/prof_r04/ (whole logs, not tails); the summaries are then copied to
# profiles/r04_*.
#   rocprofv3 kernel stats of bench.py on ONE lane (per-kernel durations are then not those of two kernels sharing the chip) and of
#   the ONE-FRAME chain, the two PMC passes behind roofline.traffic (FETCH_SIZE / WRITE_SIZE, separate runs, no trace domains beside
#   them), per-layer tables at 64 frames and at one frame per chain (round 4's kernels, round 3's, k_conv_wino only), the one-frame
#   latency table, the in-kernel timelines of the wave-split / position-row-split launches, the chunk-mix microbenchmark of the
#   32-column Winograd kernel, the soak of the new kernels, the self-launched N > 1 rehearsals (ranks sharing the one GPU over gloo:
#   weak loop + c4_strong leg), the harness script, the detector's call latency.
set -uo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/prof_r04
mkdir -p $O
COMMON="--lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-latency-mode --no-host-inclusive --no-split-precision --no-parity --no-pipelines"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $COMMON > $O/bench_under_rocprof_lanes1.json 2> $O/stats.err
echo "stats rc=$?"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python3 bench.py $COMMON --no-roofline --no-direct-form > /dev/null 2> $O/fetch.err
echo "fetch rc=$?"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python3 bench.py $COMMON --no-roofline --no-direct-form > /dev/null 2> $O/write.err
echo "write rc=$?"
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_chunk64.json > $O/pmc_traffic_chunk64.txt; echo "pmc rc=$?"
cp $O/pmc_traffic_chunk64.json profiles/r04_pmc_traffic_chunk64.json   # bench.py reads roofline.traffic from the newest profiles/*pmc_traffic_chunk64.json whose source sha matches
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); cp "$S" $O/kernel_stats_chunk64_lanes1.csv
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o one -- python3 tools/one_frame_chains.py 2000 > $O/one_frame_chains_under_rocprof.txt 2> $O/stats1.err
echo "stats1 rc=$?"
S1=$(find $O/stats1 -name "*kernel_stats.csv" | head -1); cp "$S1" $O/kernel_stats_one_frame_per_chain_lanes1.csv
python3 tools/layer_profile.py 64 2>&1 | grep -v amdgpu > $O/layer_profile_wino_chunk64.txt
python3 tools/layer_profile.py 1 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain.txt
python3 tools/layer_profile.py 1 wino_w=0 convt_w=0 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain_round3_kernels.txt
python3 tools/layer_profile.py 1 wino_w=2 2>&1 | grep -v amdgpu > $O/layer_profile_one_frame_per_chain_wave_split_everywhere.txt
python3 tools/latency_r04.py 2>&1 | grep -v amdgpu > $O/latency_one_frame_per_chain.txt
python3 tools/ps_timeline.py 1 2>&1 | grep -v amdgpu > $O/timeline_one_frame_per_chain.txt
python3 tools/ps_timeline.py 1 wino_w=2 2>&1 | grep -v amdgpu > $O/timeline_one_frame_per_chain_wave_split_everywhere.txt
python3 tools/ps_timeline.py 1 wino_w=4 2>&1 | grep -v amdgpu > $O/timeline_one_frame_per_chain_row_split_everywhere.txt
tools/ubench/wino1_chunk_mix > $O/ubench_wino1_chunk_mix.txt 2>&1; echo "ubench rc=$?"
tools/ubench/wino1_two_waves_per_simd > $O/ubench_wino1_two_waves_per_simd.txt 2>&1
tools/ubench/gather_policy > $O/ubench_gather_policy.txt 2>&1
tools/ubench/mfma_16x16x4_order > $O/ubench_mfma_16x16x4_order.txt 2>&1
python3 tools/soak_w.py 8 2>&1 | grep -v amdgpu > $O/soak_wave_split_determinism.txt; echo "soak rc=$?"
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err; echo "bench rc=$?"
python3 bench.py --total-frames 10000 --no-split-precision --no-direct-form --no-pipelines > $O/bench_c4_total_frames_10000_n1.json 2> /dev/null; echo "c4 rc=$?"
OG_BENCH_BACKEND=gloo python3 bench.py --gpus 2 --steps 5 --no-cpu-baseline > $O/bench_selflaunch_gloo_shared_gpu_n2.json 2> $O/bench_selflaunch_gloo_shared_gpu_n2.err; echo "n2 rc=$?"
OG_BENCH_BACKEND=gloo python3 bench.py --gpus 3 --steps 3 --no-cpu-baseline > $O/bench_selflaunch_gloo_shared_gpu_n3.json 2> /dev/null; echo "n3 rc=$?"
OG_BENCH_FORCE_DIST=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --steps 5 --no-cpu-baseline --no-split-precision --no-direct-form --no-pipelines > $O/bench_rccl_world1_forced_dist.json 2> $O/bench_rccl_world1_forced_dist.err; echo "rccl rc=$?"
python3 scripts/benchmark_video_speed.py --frames 502 --json 2>&1 | grep -v amdgpu > $O/benchmark_video_speed_unet_only.txt
python3 scripts/benchmark_video_speed.py --frames 502 --json --yolo-weights weights/none.npz 2>&1 | grep -v amdgpu > $O/benchmark_video_speed_gated.txt
python3 tools/bench_detect_latency.py 2>&1 | grep -v amdgpu > $O/detector_and_unet_call_latency.txt
python3 tools/bench_yolo.py 2048 256 2>&1 | grep -v amdgpu > $O/detector_batched_throughput.txt
ls -la $O
