#!/bin/bash
# Kernel-trace stats of a short training-only bench run (GPU box).  Usage: bash tools/quick_profile.sh <tag> [bench args...]
# 6 steps are launched (2 warm-up + 4 timed); the summary is per step.
set -e
TAG=${1:-q}; shift || true
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG} --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/prof_${TAG}.log 2>&1
python3 tools/prof_summary.py gpurun_out/prof/${TAG}_kernel_stats.csv 6 60 | tee gpurun_out/prof_${TAG}_summary.txt
