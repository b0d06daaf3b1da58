#!/bin/bash
# GPU box: the cross-attention layer benchmark (tools/bench_cross_attention.py) under rocprofv3: kernel stats + one PMC pass for
# matrix-pipe occupancy + the two HBM-traffic passes.  Usage: bash tools/profile_xattn.sh <tag> [B]
set -e
TAG=${1:-r02}; B=${2:-2048}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof gpurun_out/pmc
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG}_xattn --output-format csv -- python3 tools/bench_cross_attention.py $B > gpurun_out/${TAG}_xattn_bench.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE -d gpurun_out/pmc -o ${TAG}_xattn_mfma --output-format csv -- python3 tools/bench_cross_attention.py $B > gpurun_out/pmc_${TAG}_xattn_mfma.log 2>&1
python3 tools/pmc_mfma.py gpurun_out/pmc/${TAG}_xattn_mfma_counter_collection.csv gpurun_out/${TAG}_xattn_pmc_mfma.json | tee gpurun_out/${TAG}_xattn_pmc_mfma_summary.txt
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc -o ${TAG}_xattn_$C --output-format csv -- python3 tools/bench_cross_attention.py $B > gpurun_out/pmc_${TAG}_xattn_$C.log 2>&1
done
python3 tools/pmc_traffic.py gpurun_out/pmc/${TAG}_xattn_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/${TAG}_xattn_WRITE_SIZE_counter_collection.csv gpurun_out/${TAG}_xattn_pmc_traffic.json > gpurun_out/${TAG}_xattn_pmc_traffic_summary.txt
cat gpurun_out/${TAG}_xattn_bench.txt gpurun_out/${TAG}_xattn_pmc_traffic_summary.txt
