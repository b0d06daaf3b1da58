#!/bin/bash
# Run on the GPU box (via gpurun): kernel-trace stats of the default bench + the two PMC passes that feed roofline.traffic.
# Usage: bash tools/profile_round.sh <tag> [batch]      -> gpurun_out/prof/<tag>_*  gpurun_out/pmc/<tag>_*
set -e
TAG=${1:-r01}; B=${2:-512}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_train --output-format csv -- python3 bench.py --batch $B --steps 5 --warmup 2 --no-decode --no-cpu-baseline > gpurun_out/prof_${TAG}_train.log 2>&1
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_full --output-format csv -- python3 bench.py --batch $B --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/prof_${TAG}_full.log 2>&1
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc -o ${TAG}_$C --output-format csv -- python3 bench.py --batch $B --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-kernel-timing > gpurun_out/pmc_${TAG}_$C.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc/${TAG}_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/${TAG}_WRITE_SIZE_counter_collection.csv gpurun_out/pmc_traffic_b$B.json > gpurun_out/pmc_${TAG}_summary.txt
cat gpurun_out/pmc_${TAG}_summary.txt
grep metric gpurun_out/prof_${TAG}_train.log | cut -c1-200
