#!/usr/bin/env python3
"""Join tools/gemm_traffic_probe.py's shape list with its FETCH_SIZE and WRITE_SIZE passes (rocprofv3 --pmc counter_collection csv):
python tools/gemm_traffic_fold.py <order.json> <fetch.csv> <write.csv> [out.txt].  gfx950: both counters in KiB, FETCH_SIZE counts half of
the bytes of wide streaming reads (MI355X_MICROARCH.md, HBM section) -> doubled; WRITE_SIZE exact."""
import csv
import json
import sys


def groups(path, counter):
    """per marker-delimited group: sum of the counter over the GEMM dispatches in it"""
    rows = [r for r in csv.DictReader(open(path)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    out, cur, seen_marker = [], None, False
    for r in rows:
        n = r['Kernel_Name']
        if 'FillFunctor' in n:
            if cur is not None:
                out.append(cur)
            cur, seen_marker = [0.0, 0], True
        elif seen_marker and ('gemm' in n or 'splitk' in n):
            cur[0] += float(r['Counter_Value'])
            cur[1] += 1
    return [g for g in out if g[1] > 0]


def main():
    order = json.load(open(sys.argv[1]))
    f, w = groups(sys.argv[2], 'FETCH_SIZE'), groups(sys.argv[3], 'WRITE_SIZE')
    assert len(f) == len(order) == len(w), (len(f), len(w), len(order))
    lines = [f"{'M':>7} {'N':>6} {'K':>7} {'form':18} {'read MB':>9} {'A+B+epi MB':>11} {'ratio':>6} {'A MB':>8} {'B MB':>8} {'epi MB':>8} {'write MB':>9} {'C MB':>8} {'ratio':>6} {'kernels/call':>5}"]
    for o, (fv, fn), (wv, wn) in zip(order, f, w):
        rd = 2.0 * fv * 1024 / o['reps'] / 1e6
        wr = wv * 1024 / o['reps'] / 1e6
        a, b, c, e = (o[k] / 1e6 for k in ('bytes_A', 'bytes_B', 'bytes_C', 'bytes_epilogue_operand'))
        e_rd = e if o['epilogue'] in ('res', 'acc', 'mul_aux') else 0.0
        c_wr = c + (e if o['epilogue'] == 'gelu_dout' else 0.0)
        form = ('A^T' if o['a_kmajor'] else 'A') + ('.B' if o['b_kmajor'] else '.B^T') + ' ' + o['out'] + ('+' + o['epilogue'] if o['epilogue'] else '')
        lines.append(f"{o['M']:7d} {o['N']:6d} {o['K']:7d} {form:18} {rd:9.1f} {a + b + e_rd:11.1f} {rd / (a + b + e_rd):6.2f} {a:8.1f} {b:8.1f} {e_rd:8.1f} "
                     f"{wr:9.1f} {c_wr:8.1f} {wr / c_wr:6.2f} {fn / o['reps']:5.1f}")
    txt = '\n'.join(lines)
    print(txt)
    if len(sys.argv) > 4:
        open(sys.argv[4], 'w').write(txt + '\n')


if __name__ == '__main__':
    main()
