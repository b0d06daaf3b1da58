#!/bin/bash
# Run on the GPU box (via gpurun): rocprofv3 kernel-trace stats of a Hugging Face decoder's train step (tools/bench_hf_decoder.py).
# (absolute output directory: the bench tool changes into a scratch directory for its checkpoint)
# Usage: bash tools/profile_hf_decoder.sh <tag> <size> <batch> [extra bench flags...]      -> gpurun_out/prof/<tag>_<size>_kernel_stats.csv
set -e
TAG=${1:-r02}; SIZE=${2:-qwen2-1.5b}; B=${3:-256}; shift 3 || true
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o ${TAG}_${SIZE} --output-format csv -- python3 tools/bench_hf_decoder.py --size $SIZE --batch $B --steps 4 --warmup 1 --no-decode "$@" > gpurun_out/prof_${TAG}_${SIZE}.log 2>&1
head -30 gpurun_out/prof/${TAG}_${SIZE}_kernel_stats.csv | cut -c1-160
