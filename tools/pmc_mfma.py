#!/usr/bin/env python3
"""Fold one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY
SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE, with --kernel-trace) into per-kernel-family matrix-pipe occupancy.

    python tools/pmc_mfma.py <counter_collection.csv> [out.json]

MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs): the share of SIMD-cycles of the kernel's own
duration in which the matrix pipe was executing.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs (checked against the
kernel-trace durations of the same pass: 13-19 counts per ns = 8 x 1.7-2.4 GHz), hence the / 8 (MI355X_MICROARCH.md: the counter counts cycles, 4 per pass of a 16x16x32
bf16 MFMA).  The wave-state counters are in quad-cycles and only used as ratios of SQ_WAVE_CYCLES.
"""
import csv
import json
import sys

FAMILIES = (('gemm256_kernel<false, false, 8>', 'fused cross-attention (K/V projection + attention)'), ('gemm3_kernel', 'gemm3 (256x128, overlapped epilogue)'),
            ('gemm256_kernel', 'gemm256 (persistent 256^2)'), ('gemm_bf16_kernel', 'gemm 128^2'), ('gemm_skinny', 'gemm skinny'),
            ('attn_fwd2', 'attention fwd (LDS-resident, encoder)'), ('attn_bwd3', 'attention bwd (LDS-resident, one softmax pass, encoder)'), ('attn_bwd2', 'attention bwd (LDS-resident, two-phase, encoder)'),
            ('attn_fwd', 'attention fwd'), ('attn_bwd_dq', 'attention bwd dQ'), ('attn_bwd_dkv', 'attention bwd dK/dV'),
            ('conv_mfma_bwd_weight', 'conv bwd-weight'), ('conv_mfma_kernel', 'conv fwd / bwd-data'), ('ln_bwd', 'layernorm bwd'))
N_SIMD = 256 * 4


def main():
    acc = {}
    for r in csv.DictReader(open(sys.argv[1])):
        name = r['Kernel_Name']
        fam = next((lab for key, lab in FAMILIES if key in name), None)
        if fam is None:
            continue
        d = acc.setdefault(fam, {})
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            d['launches'] = d.get('launches', 0) + 1
    out = {}
    print(f'{"kernel family":56s} {"launches":>8s} {"MFMA busy":>10s} {"waves parked":>13s} {"issue stall":>12s} {"issuing":>8s}')
    for fam, d in acc.items():
        cyc = d.get('GRBM_GUI_ACTIVE', 0.0)
        if cyc <= 0:
            continue
        wc = max(d.get('SQ_WAVE_CYCLES', 0.0), 1.0)
        o = {'launches': d.get('launches', 0), 'mfma_busy_frac': d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (cyc / 8.0 * N_SIMD),
             'wait_any_frac_of_wave_cycles': d.get('SQ_WAIT_ANY', 0.0) / wc,
             'wait_inst_any_frac_of_wave_cycles': d.get('SQ_WAIT_INST_ANY', 0.0) / wc,
             'active_inst_any_frac_of_wave_cycles': d.get('SQ_ACTIVE_INST_ANY', 0.0) / wc}
        out[fam] = o
        print(f'{fam:56s} {o["launches"]:8d} {o["mfma_busy_frac"] * 100:9.1f}% {o["wait_any_frac_of_wave_cycles"] * 100:12.1f}% '
              f'{o["wait_inst_any_frac_of_wave_cycles"] * 100:11.1f}% {o["active_inst_any_frac_of_wave_cycles"] * 100:7.1f}%')
    if len(sys.argv) > 2:
        json.dump(out, open(sys.argv[2], 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
