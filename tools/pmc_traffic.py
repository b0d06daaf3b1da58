#!/usr/bin/env python3
"""Fold two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md prescribes) into HBM
bytes per launch of the dominant kernel.  gfx950 corrections (same guide, section HBM): both counters are in KiB;
FETCH_SIZE reports exactly 1/2 of the bytes of wide coalesced streaming reads -> doubled; WRITE_SIZE is exact.

    python tools/pmc_traffic.py gpurun_out/pmc/r01_FETCH_SIZE_counter_collection.csv gpurun_out/pmc/r01_WRITE_SIZE_counter_collection.csv
"""
import csv
import json
import sys


def per_kernel(path, counter, detail=False):
    tot, n = {}, {}
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name']
        k = k.replace('(anonymous namespace)::', '')
        if detail:
            k = k.split('(')[0][-70:]
        else:
            k = 'gemm_all' if ('gemm_bf16_kernel' in k or 'gemm256_kernel' in k or 'gemm3_kernel' in k) else k.split('(')[0].split('<')[0][-60:]
        tot[k] = tot.get(k, 0.0) + float(r['Counter_Value'])
        n[k] = n.get(k, 0) + 1
    return tot, n


def main():
    f_tot, f_n = per_kernel(sys.argv[1], 'FETCH_SIZE')
    w_tot, w_n = per_kernel(sys.argv[2], 'WRITE_SIZE')
    out = {}
    for k in f_tot:
        rd = 2.0 * f_tot[k] * 1024.0
        wr = w_tot.get(k, 0.0) * 1024.0
        out[k] = {'launches': f_n[k], 'read_bytes_per_launch': rd / f_n[k], 'write_bytes_per_launch': wr / max(1, w_n.get(k, 1)),
                  'hbm_bytes_per_launch': rd / f_n[k] + wr / max(1, w_n.get(k, 1)), 'total_GB': (rd + wr) / 1e9}
    # second view: every template instantiation on its own (the fused cross-attention kernel is gemm256_kernel<false, false, 8>)
    fd, fdn = per_kernel(sys.argv[1], 'FETCH_SIZE', True)
    wd, wdn = per_kernel(sys.argv[2], 'WRITE_SIZE', True)
    out['by_kernel'] = {k: {'launches': fdn[k], 'hbm_bytes_per_launch': 2.0 * fd[k] * 1024.0 / fdn[k] + wd.get(k, 0.0) * 1024.0 / max(1, wdn.get(k, 1))}
                        for k in fd if 'gemm' in k or 'attn' in k}
    top = sorted(((k, v) for k, v in out.items() if k != 'by_kernel'), key=lambda kv: -kv[1]['total_GB'])[:12]
    for k, v in top:
        print(f"{v['total_GB']:9.2f} GB  launches {v['launches']:6d}  {v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch  {k}")
    json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else 'profiles/pmc_traffic.json', 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
