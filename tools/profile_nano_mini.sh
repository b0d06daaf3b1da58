#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the nano-mini bench (tools/bench_nano_mini.py).
# Usage: bash tools/profile_nano_mini.sh <tag> [batch]      -> gpurun_out/prof/<tag>_nano_mini_*
set -e
TAG=${1:-r02}; B=${2:-512}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
python3 tools/bench_nano_mini.py --batch $B --steps 8 --warmup 3 --split --cpu > gpurun_out/${TAG}_nano_mini_bench.json 2> gpurun_out/${TAG}_nano_mini_bench.err
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_nano_mini --output-format csv -- python3 tools/bench_nano_mini.py --batch $B --steps 4 --warmup 2 > gpurun_out/prof_${TAG}_nano_mini.log 2>&1
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc -o ${TAG}_nano_mini_$C --output-format csv -- python3 tools/bench_nano_mini.py --batch $B --steps 2 --warmup 1 --no-decode > gpurun_out/pmc_${TAG}_nano_mini_$C.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc/${TAG}_nano_mini_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/${TAG}_nano_mini_WRITE_SIZE_counter_collection.csv gpurun_out/pmc_traffic_nano_mini_b$B.json > gpurun_out/pmc_${TAG}_nano_mini_summary.txt
head -14 gpurun_out/pmc_${TAG}_nano_mini_summary.txt
head -25 gpurun_out/prof/${TAG}_nano_mini_kernel_stats.csv | cut -c1-200
cat gpurun_out/${TAG}_nano_mini_bench.json
