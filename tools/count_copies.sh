#!/bin/bash
# Run on the GPU box: how many device-to-device copy kernels (and which torch elementwise kernels) one train step of bench.py launches
# -- kernel-trace of runs with 2 and 6 timed steps; the difference / 4 is the per-step count.  Usage: bash tools/count_copies.sh
set -e
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/prof
for K in 2 6; do
  rocprofv3 --kernel-trace --stats -d "$GRAFT_REPO_ROOT/gpurun_out/prof" -o copies_$K --output-format csv -- python3 bench.py --steps $K --warmup 1 --no-decode --no-cpu-baseline --no-kernel-timing > gpurun_out/copies_$K.log 2>&1
done
python3 - <<'P'
import csv
def load(k):
    return {r['Name']: (int(r['Calls']), float(r['TotalDurationNs'])) for r in csv.DictReader(open(f'gpurun_out/prof/copies_{k}_kernel_stats.csv'))}
a, b = load(2), load(6)
rows = []
for n in b:
    c0, t0 = a.get(n, (0, 0.0)); c1, t1 = b[n]
    if c1 != c0:
        rows.append(((t1 - t0) / 4e6, (c1 - c0) / 4, n[:110]))
tot = sum(r[0] for r in rows)
print(f'per-step kernel time {tot:.1f} ms')
for t, c, n in sorted(rows, reverse=True):
    if 'gemm' in n or 'attn' in n or 'conv' in n or 'ln_' in n: continue
    print(f'{t:8.3f} ms  {c:7.1f} calls  {n}')
P
