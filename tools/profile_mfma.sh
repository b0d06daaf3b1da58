#!/bin/bash
# GPU box: one PMC pass for matrix-pipe occupancy per kernel family.  Usage: bash tools/profile_mfma.sh <tag> [batch]
set -e
TAG=${1:-r01}; B=${2:-2048}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc -o ${TAG}_mfma --output-format csv -- python3 bench.py --batch $B --steps 2 --warmup 1 --no-decode --no-cpu-baseline --no-kernel-timing > gpurun_out/pmc_${TAG}_mfma.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc/${TAG}_mfma_counter_collection.csv gpurun_out/pmc_mfma_b$B.json | tee gpurun_out/pmc_${TAG}_mfma_summary.txt
