#!/usr/bin/env python3
"""Per-kernel wave-state / LDS / matrix-pipe ratios from ONE rocprofv3 --pmc pass (csv): python tools/pmc_kernels.py <counter_collection.csv> [substr]
Counters used when present: SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE."""
import csv
import sys
acc = {}
sub = sys.argv[2] if len(sys.argv) > 2 else ''
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Kernel_Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    if sub and sub not in n:
        continue
    n = n[:70]
    d = acc.setdefault(n, {})
    d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
    d['_n'] = d.get('_n', 0) + 1
for n, d in sorted(acc.items(), key=lambda kv: -kv[1].get('GRBM_GUI_ACTIVE', 0)):
    wc = max(d.get('SQ_WAVE_CYCLES', 0.0), 1.0)
    cyc = d.get('GRBM_GUI_ACTIVE', 0.0) / 8.0          # shader cycles of the kernel (sum over launches)
    parts = [f'{n:70s}']
    if cyc > 0 and 'SQ_VALU_MFMA_BUSY_CYCLES' in d:
        parts.append(f"mfma {100 * d['SQ_VALU_MFMA_BUSY_CYCLES'] / (cyc * 1024):5.1f}%")
    for key, lab in (('SQ_WAIT_ANY', 'parked'), ('SQ_WAIT_INST_ANY', 'stall'), ('SQ_ACTIVE_INST_ANY', 'issue'), ('SQ_WAIT_INST_LDS', 'lds-stall'),
                     ('SQ_ACTIVE_INST_VALU', 'valu'), ('SQ_ACTIVE_INST_LDS', 'lds-inst')):
        if key in d:
            parts.append(f'{lab} {100 * d[key] / wc:5.1f}%')
    if cyc > 0 and 'SQ_LDS_IDX_ACTIVE' in d:
        parts.append(f"lds-busy {100 * d['SQ_LDS_IDX_ACTIVE'] / (cyc * 256):5.1f}% conflict {100 * d.get('SQ_LDS_BANK_CONFLICT', 0) / max(d['SQ_LDS_IDX_ACTIVE'], 1):4.1f}%")
    if cyc > 0:
        parts.append(f"waves/simd {wc * 4 / (cyc * 1024):4.2f}")
    print(' '.join(parts))
