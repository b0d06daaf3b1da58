#!/bin/bash
# One PMC pass over an arbitrary python tool: bash tools/pmc_run.sh <tag> "<counters>" <script.py> [args...]   -> gpurun_out/pmc/<tag>_counter_collection.csv
set -e
TAG=$1; CTRS=$2; shift 2
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc $CTRS -d gpurun_out/pmc -o ${TAG} --output-format csv -- python3 "$@" > gpurun_out/pmc_${TAG}.log 2>&1
