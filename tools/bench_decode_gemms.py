#!/usr/bin/env python3
"""The GEMM shapes of one greedy-decode step of the benchmark model (M = captions per batch; d = 768, 12 layers) on the large-tile
persistent kernel vs the 128^2 kernel, alone on the GPU:   python tools/bench_decode_gemms.py [M]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
    d = 768
    shapes = [('qkv          bf16+bias', 3 * d, d, dict()), ('attn out     f32+res', d, d, dict(res=True)), ('cross q      bf16+bias', d, d, dict()),
              ('mlp c_fc     bf16+gelu', 4 * d, d, dict(act=1)), ('mlp c_proj   f32+res', d, 4 * d, dict(res=True)),
              ('lm_head      f32', 50257, d, dict(f32=True))]
    for name, N, K, kw in shapes:
        a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.03).to(BF16)
        bias = torch.randn(N, device=dev) * 0.1
        ldc = (N + 7) // 8 * 8
        out = torch.zeros(M, ldc, dtype=F32 if (kw.get('res') or kw.get('f32')) else BF16, device=dev)
        args = dict(bias=None if kw.get('f32') else bias, act=kw.get('act', 0), residual=out if kw.get('res') else None)
        res = []
        for mt in ('40', '100000'):
            os.environ['I2T_G256_MIN_TILES'] = mt
            t = timeit(lambda: ops.gemm(a, w, out, M, N, K, **args), reps=30)
            res.append(t)
        fl = 2.0 * M * N * K
        print(f'{name:24s} M={M} N={N:5d} K={K:4d}  256^2 persistent {res[0] * 1e6:7.1f} us {fl / res[0] / 1e12:6.1f} TF   128^2 {res[1] * 1e6:7.1f} us {fl / res[1] / 1e12:6.1f} TF')
    os.environ.pop('I2T_G256_MIN_TILES', None)


if __name__ == '__main__':
    main()
