#!/bin/bash
# GPU box: wave-state and LDS counters of the encoder attention kernels alone (tools/bench_attention.py).  Usage: bash tools/pmc_attention.sh <tag> [B]
set -e
TAG=${1:-att}; B=${2:-512}
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE -d gpurun_out/pmc -o ${TAG}_a --output-format csv -- python3 tools/bench_attention.py $B > gpurun_out/pmc_${TAG}_a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_SALU GRBM_GUI_ACTIVE -d gpurun_out/pmc -o ${TAG}_b --output-format csv -- python3 tools/bench_attention.py $B > gpurun_out/pmc_${TAG}_b.log 2>&1
python3 - <<PY
import csv, collections
for part in 'ab':
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open('gpurun_out/pmc/${TAG}_%s_counter_collection.csv' % part)):
        n = r['Kernel_Name']
        if 'attn_' not in n: continue
        k = n.split('(')[0].replace('void ', '').replace('(anonymous namespace)::', '')
        acc[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE': acc[k]['launches'] += 1
    for k, d in acc.items():
        print(k, {c: (round(v / d['launches'] / 1e6, 3)) for c, v in d.items() if c != 'launches'}, 'launches', int(d['launches']))
PY
