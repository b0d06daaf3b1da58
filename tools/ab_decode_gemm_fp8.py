#!/usr/bin/env python3
"""Decode-step GEMMs of a 7-B decoder at small caption batches (M rows): bf16 through the deterministic split-K form the decode step uses
(ops.gemm(workspace=...)) against e4m3 weights through i2t_gemm_fp8 -- the step is a read of the weights, and fp8 halves their bytes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = torch.device('cuda:0')
BF16, F32 = torch.bfloat16, torch.float32


def main():
    ws = torch.empty(16 << 20, dtype=F32, device=dev)
    for M in (256, 1024):
        tot16 = tot8 = 0.0
        for (N, K) in ((12288, 4096), (4096, 4096), (22016, 4096), (4096, 11008)):
            x = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF16)
            out = torch.empty(M, N, dtype=BF16, device=dev)
            x8, sx = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
            w8, sw = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
            ops.quant_rows_fp8(w, w8, sw, N, K)
            t16 = timeit(lambda: ops.gemm(x, w, out, M, N, K, workspace=ws), reps=20)

            def f8():
                ops.quant_rows_fp8(x, x8, sx, M, K)
                ops.gemm_fp8(x8, sx, w8, sw, out, M, N, K)
            t8 = timeit(f8, reps=20)
            tot16 += t16; tot8 += t8
            print(f'M={M} N={N} K={K}: bf16 {t16 * 1e6:7.1f} us ({N * K * 2 / t16 / 1e12:4.2f} TB/s of weights)   fp8 {t8 * 1e6:7.1f} us ({N * K / t8 / 1e12:4.2f} TB/s)', flush=True)
        print(f'M={M}: one block bf16 {tot16 * 1e6:.0f} us, fp8 {tot8 * 1e6:.0f} us -> x{tot16 / tot8:.2f}', flush=True)


if __name__ == '__main__':
    main()
