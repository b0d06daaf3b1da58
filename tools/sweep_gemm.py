#!/usr/bin/env python3
"""K sweep of one GEMM shape family: time per launch and TFLOP/s (fit: time = rounds * (c + s * K))."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def main():
    M = int(sys.argv[1]) if len(sys.argv) > 1 else 133120
    for N in (512, 1536, 2048):
        for K in (128, 256, 512, 1024, 2048, 4096):
            x = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
            y = torch.empty(M, N, device=dev, dtype=BF16)
            y32 = torch.empty(M, N, device=dev, dtype=F32)
            res = torch.randn(M, N, device=dev)
            t = timeit(lambda: ops.gemm(x, w, y, M, N, K), reps=10)
            t32 = timeit(lambda: ops.gemm(x, w, y32, M, N, K, residual=res), reps=10)
            fl = 2.0 * M * N * K
            print(f'M={M} N={N:5d} K={K:5d}  bf16-out {t * 1e6:8.1f} us {fl / t / 1e12:7.1f} TF   f32+res {t32 * 1e6:8.1f} us {fl / t32 / 1e12:7.1f} TF', flush=True)


if __name__ == '__main__':
    main()
