#!/usr/bin/env python3
"""Per-step anatomy from a rocprofv3 --kernel-trace --stats CSV:  python tools/prof_summary.py <kernel_stats.csv> <steps> [top]"""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2])
top = int(sys.argv[3]) if len(sys.argv) > 3 else 45
CATS = [('gemm', r'gemm256_kernel|gemm_bf16_kernel|gemm3_kernel|gemm_skinny|splitk_reduce'), ('attention', r'attn_|attention'),
        ('conv+transpose', r'conv_|nchw_to_nhwc'), ('layernorm', r'\bln_|lnnd_'), ('cross-entropy', r'\bce_|scale_bf16'),
        ('colsum', r'colsum'), ('grad-normaliser', r'scale_by_norm|sumsq'), ('dropout_apply', r'dropout_apply'),
        ('copies', r'copyBuffer|fillBuffer|FillFunctor|elementwise_kernel')]
tot = {}
all_ms = 0.0
for r in rows:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    ms = float(r['TotalDurationNs']) / 1e6 / steps
    all_ms += ms
    for c, pat in CATS:
        if re.search(pat, n):
            tot[c] = tot.get(c, 0.0) + ms
            break
    else:
        tot['other'] = tot.get('other', 0.0) + ms
for r in rows[:top]:
    n = r['Name'].replace('(anonymous namespace)::', '').replace('void ', '')
    print(f"{n[:96]:96s} {float(r['Calls']) / steps:7.1f}/step {float(r['TotalDurationNs']) / 1e6 / steps:8.3f} ms/step {float(r['AverageNs']) / 1e3:9.1f} us")
print('---- per step (ms):', ', '.join(f'{k} {v:.2f}' for k, v in sorted(tot.items(), key=lambda kv: -kv[1])), f'| all kernels {all_ms:.2f}')
