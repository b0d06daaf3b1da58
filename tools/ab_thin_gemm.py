#!/usr/bin/env python3
"""The LoRA adapters' thin GEMMs (N = 128 or K = 128) at Llama-2-7B sizes: time per shape (one process per routing, I2T_G256_NARROW=0|1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = torch.device('cuda:0')
BF16, F32 = torch.bfloat16, torch.float32


def main():
    M = 12820
    for K in (4096, 11008, 12288, 22016):
        x = torch.randn(M, K, device=dev).to(BF16)
        a = (torch.randn(128, K, device=dev) * 0.02).to(BF16)        # u = x A^T   (A.B^T)
        bt = (torch.randn(K, 128, device=dev) * 0.02).to(BF16)       # du = dY (sB) (A.B, B k-major)
        u = torch.empty(M, 128, dtype=BF16, device=dev)
        t1 = timeit(lambda: ops.gemm(x, a, u, M, 128, K), reps=20)
        t2 = timeit(lambda: ops.gemm(x, bt, u, M, 128, K, b_kmajor=True), reps=20)
        u32 = torch.zeros(M, 128, dtype=F32, device=dev)

        def splitk(b, **kw):
            u32.zero_()
            ops.gemm(x, b, u32, M, 128, K, accumulate=True, **kw)
            ops.cast_f32_bf16(u32, u)
        t3 = timeit(lambda: splitk(a), reps=20)
        t4 = timeit(lambda: splitk(bt, b_kmajor=True), reps=20)
        by = M * K * 2
        print(f'narrow256={os.environ.get("I2T_G256_NARROW", "0")} M={M} N=128 K={K}: A.B^T {t1 * 1e6:7.1f} us ({by / t1 / 1e12:4.2f} TB/s)   A.B {t2 * 1e6:7.1f} us ({by / t2 / 1e12:4.2f} TB/s)   split-K f32+cast: A.B^T {t3 * 1e6:7.1f} us  A.B {t4 * 1e6:7.1f} us', flush=True)


if __name__ == '__main__':
    main()
