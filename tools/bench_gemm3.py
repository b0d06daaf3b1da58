#!/usr/bin/env python3
"""gemm3 (256 x 128 tiles, two accumulator sets, epilogue inside the next tile's K loop) against the 256^2 kernel on the
model's class-1 shapes (bf16 C + bias): correctness (bit-equal expected: same K order per output) and time.
    python tools/bench_gemm3.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device('cuda:0')


def main():
    shapes = [('enc qkv  B=1024', 266240, 1536, 512), ('kv-proj  B=2048', 131072, 1536, 768), ('q-proj   B=2048', 74232, 768, 768),
              ('dec qkv  B=2048', 74232, 2304, 768), ('K=2048', 133120, 512, 2048), ('ragged', 70001, 1160, 512)]
    for name, M, N, K in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev)
        outs, line = {}, f'{name:18s} M={M} N={N} K={K}:'
        for mode in ('0', '1', '2'):
            os.environ['I2T_GEMM3'] = mode
            c = torch.zeros(M, N, dtype=BF16, device=dev)
            ops.gemm(a, w, c, M, N, K, bias=bias)
            torch.cuda.synchronize()
            outs[mode] = c
            t = timeit(lambda: ops.gemm(a, w, c, M, N, K, bias=bias), reps=10)
            line += f'  [{mode}] {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF'
        ref = a.float()[:2048] @ w.float().t() + bias
        e0 = float((outs['0'][:2048].float() - ref).abs().max())
        d1 = float((outs['1'].float() - outs['0'].float()).abs().max())
        d2 = float((outs['2'].float() - outs['0'].float()).abs().max())
        print(line + f'   err0 {e0:.3g}  |g3-old| {d1:.3g} {d2:.3g}', flush=True)
    # class 2: GELU with the pre-activation as a second output (the MLP's first GEMM)
    for name, M, N, K in [('enc fc   B=1024', 266240, 2048, 512), ('dec fc   B=2048', 74232, 3072, 768)]:
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        bias = torch.randn(N, device=dev)
        outs, line = {}, f'{name:18s} M={M} N={N} K={K} gelu:'
        for mode in ('0', '1', '2'):
            os.environ['I2T_GEMM3'] = mode
            h = torch.zeros(M, N, dtype=BF16, device=dev)
            pre = torch.zeros(M, N, dtype=BF16, device=dev)
            ops.gemm(a, w, h, M, N, K, bias=bias, act=1, aux_out=pre)
            torch.cuda.synchronize()
            outs[mode] = (h, pre)
            t = timeit(lambda: ops.gemm(a, w, h, M, N, K, bias=bias, act=1, aux_out=pre), reps=10)
            line += f'  [{mode}] {t * 1e6:7.1f} us {2.0 * M * N * K / t / 1e12:6.0f} TF'
        d = [max(float((outs[m][0].float() - outs['0'][0].float()).abs().max()), float((outs[m][1].float() - outs['0'][1].float()).abs().max())) for m in ('1', '2')]
        print(line + f'   |g3-old| {d[0]:.3g} {d[1]:.3g}', flush=True)
    os.environ['I2T_GEMM3'] = '0'


if __name__ == '__main__':
    main()
