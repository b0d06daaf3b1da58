#!/usr/bin/env python3
"""fp8 (e4m3, block-scaled MFMA) GEMM throughput next to the bf16 GEMM at the same shapes (frozen-decoder projections).
    python tools/bench_gemm_fp8.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def main():
    for M, N, K in ((12800, 12288, 4096), (12800, 4096, 4096), (12800, 22016, 4096), (12800, 4096, 11008), (65536, 2304, 768), (16384, 8960, 1536)):
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF16)
        out = torch.empty(M, N, dtype=BF16, device=dev)
        x8, sx = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
        w8, sw = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
        ops.quant_rows_fp8(w, w8, sw, N, K)
        tq = timeit(lambda: ops.quant_rows_fp8(x, x8, sx, M, K), reps=10)
        t8 = timeit(lambda: ops.gemm_fp8(x8, sx, w8, sw, out, M, N, K), reps=10)
        t16 = timeit(lambda: ops.gemm(x, w, out, M, N, K), reps=10)
        fl = 2.0 * M * N * K
        print(f'M={M} N={N} K={K}: fp8 {t8 * 1e6:8.1f} us {fl / t8 / 1e12:7.1f} TF ({fl / t8 / 5e15 * 100:4.1f} % of 5 PF) + quantise x {tq * 1e6:6.1f} us'
              f'   | bf16 {t16 * 1e6:8.1f} us {fl / t16 / 1e12:7.1f} TF   -> {t16 / (t8 + tq):.2f}x incl. quantisation', flush=True)


if __name__ == '__main__':
    main()
