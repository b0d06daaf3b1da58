#!/bin/bash
# GPU box: per-shape HBM traffic of the step's GEMMs (two PMC passes, one counter each) -> gpurun_out/gemm_traffic.txt
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
for C in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $C -d gpurun_out/pmc -o gt_$C --output-format csv -- python3 tools/gemm_traffic_probe.py > gpurun_out/pmc_gt_$C.log 2>&1
done
python3 tools/gemm_traffic_fold.py gpurun_out/gemm_probe_order.json gpurun_out/pmc/gt_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/gt_WRITE_SIZE_counter_collection.csv gpurun_out/gemm_traffic.txt
