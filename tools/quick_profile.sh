#!/bin/bash
# Kernel-trace stats of a short training-only bench run (GPU box).  Usage: bash tools/quick_profile.sh <tag> [bench args...]
set -e
TAG=${1:-q}; shift || true
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
rocprofv3 --kernel-trace --stats -d gpurun_out/prof -o ${TAG} --output-format csv -- python3 bench.py --steps 4 --warmup 2 --no-decode --no-cpu-baseline --no-kernel-timing "$@" > gpurun_out/prof_${TAG}.log 2>&1
python3 - <<PY
import csv
rows=list(csv.DictReader(open('gpurun_out/prof/${TAG}_kernel_stats.csv')))
for r in rows[:22]:
    n=r['Name'].replace('(anonymous namespace)::','')
    print(f"{n[:84]:84s} {r['Calls']:>6s} {float(r['TotalDurationNs'])/1e6:9.2f} ms {float(r['AverageNs'])/1e3:9.1f} us")
PY
