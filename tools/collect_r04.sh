#!/bin/bash
# Copies the summaries of tools/profile_r04.sh's run (gpurun_out/prof_r04/, scratch) into profiles/r04_* (tracked).
set -euo pipefail
cd "$(dirname "$0")/.."
S=gpurun_out/prof_r04
for f in bench_default.json bench_under_rocprof_lanes1.json bench_c4_total_frames_10000_n1.json bench_selflaunch_gloo_shared_gpu_n2.json \
         bench_selflaunch_gloo_shared_gpu_n3.json bench_rccl_world1_forced_dist.json kernel_stats_chunk64_lanes1.csv \
         kernel_stats_one_frame_per_chain_lanes1.csv pmc_traffic_chunk64.json pmc_traffic_chunk64.txt layer_profile_wino_chunk64.txt \
         layer_profile_one_frame_per_chain.txt layer_profile_one_frame_per_chain_round3_kernels.txt \
         layer_profile_one_frame_per_chain_wave_split_everywhere.txt latency_one_frame_per_chain.txt timeline_one_frame_per_chain.txt \
         timeline_one_frame_per_chain_wave_split_everywhere.txt timeline_one_frame_per_chain_row_split_everywhere.txt \
         ubench_wino1_chunk_mix.txt ubench_wino1_two_waves_per_simd.txt ubench_gather_policy.txt ubench_mfma_16x16x4_order.txt soak_wave_split_determinism.txt one_frame_chains_under_rocprof.txt \
         benchmark_video_speed_unet_only.txt benchmark_video_speed_gated.txt detector_and_unet_call_latency.txt \
         detector_batched_throughput.txt; do
    cp "$S/$f" "profiles/r04_$f"
done
cp "$S/bench_rccl_world1_forced_dist.err" profiles/r04_bench_rccl_world1_forced_dist.log
cp "$S/fetch/bench_counter_collection.csv" profiles/r04_pmc_fetch_size_counter_collection.csv
cp "$S/write/bench_counter_collection.csv" profiles/r04_pmc_write_size_counter_collection.csv
cp gpurun_out/prof_r04_run.log profiles/r04_profile_run.log
ls profiles/r04_* | wc -l
