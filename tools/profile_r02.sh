#!/bin/bash
# Evidence run on the GPU box: rocprofv3 kernel stats of bench.py (one lane: per-kernel durations are then not those of two
# kernels sharing the chip) and the two PMC passes behind roofline.traffic.  Writes under gpurun_out/prof_r02/.
set -euo pipefail
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
O=gpurun_out/prof_r02
mkdir -p $O
COMMON="--lanes 1 --steps 4 --warmup 1 --no-cpu-baseline --no-latency-mode --no-host-inclusive"   # (the direct_form and split_precision legs stay: their kernels get stats and PMC traffic too)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python3 bench.py $COMMON > $O/bench_under_rocprof.json 2> $O/stats.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o bench -- python3 bench.py $COMMON --no-roofline > /dev/null 2> $O/fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o bench -- python3 bench.py $COMMON --no-roofline > /dev/null 2> $O/write.err
find $O -name "*.csv" | head -20
F=$(find $O/fetch -name "*counter_collection.csv" | head -1); W=$(find $O/write -name "*counter_collection.csv" | head -1)
python3 tools/pmc_traffic.py "$F" "$W" $O/pmc_traffic_chunk64.json > $O/pmc_traffic.txt
cat $O/pmc_traffic.txt
S=$(find $O/stats -name "*kernel_stats.csv" | head -1); head -30 "$S"
