#!/usr/bin/env python3
"""Decode-step GEMMs at mid-sized caption batches (64 < M <= 2048): the router's choice vs the weight-streaming (skinny) kernel run on
64-row chunks vs the deterministic split-K form (i2t_gemm_bf16_ws); weight bytes / time against the HBM peak.   python tools/probe_mid_gemm.py [d ff nq]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def main():
    d, ff, nq = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1536, 8960, 2048)
    shapes = [('qkv  bf16+bias', nq, d, dict()), ('o    f32+res', d, d, dict(res=True)), ('g|u  bf16', 2 * ff, d, dict(nobias=True)),
              ('down f32+res', d, ff, dict(res=True))]
    ws = torch.empty(32 << 20, dtype=F32, device=dev)
    for M in (64, 128, 256, 512, 1024, 2048, 4096):
        for name, N, K, kw in shapes:
            a = (torch.randn(M, K, device=dev) * 0.5).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.03).to(BF16)
            bias = None if (kw.get('nobias') or kw.get('res')) else torch.randn(N, device=dev) * 0.1
            out = torch.zeros(M, N, dtype=F32 if kw.get('res') else BF16, device=dev)
            args = dict(bias=bias, residual=out if kw.get('res') else None)
            t0 = timeit(lambda: ops.gemm(a, w, out, M, N, K, **args), reps=20)

            def chunks():
                for m0 in range(0, M, 64):
                    o = out[m0:m0 + 64]
                    ops.gemm(a[m0:m0 + 64], w, o, min(64, M - m0), N, K, bias=bias, residual=o if kw.get('res') else None)
            t1 = timeit(chunks, reps=20) if M <= 512 else float('nan')
            t2 = timeit(lambda: ops.gemm(a, w, out, M, N, K, workspace=ws, **args), reps=20)
            wb = N * K * 2
            print(f'M={M:5d} {name:15s} N={N:5d} K={K:5d}  router {t0 * 1e6:7.1f} us ({wb / t0 / 1e9:6.0f} GB/s of weights)   '
                  f'skinny x{(M + 63) // 64:2d} {t1 * 1e6:7.1f} us ({wb / t1 / 1e9:6.0f} GB/s)   split-K+ws {t2 * 1e6:7.1f} us ({wb / t2 / 1e9:6.0f} GB/s)')


if __name__ == '__main__':
    main()
