#!/usr/bin/env python3
"""dW = dY^T . X timing for the small-output shapes (split-K + atomics)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402
dev = torch.device('cuda:0')
for (M, N, K) in [(512, 512, 266240), (1536, 512, 266240), (512, 2048, 266240), (768, 768, 37236), (2304, 768, 37236), (768, 3072, 37236)]:
    Kp = (K + 7) // 8 * 8
    dy = torch.randn(Kp, M, device=dev).to(torch.bfloat16)
    x = torch.randn(Kp, N, device=dev).to(torch.bfloat16)
    dw = torch.zeros(M, N, device=dev)
    t = timeit(lambda: ops.gemm(dy, x, dw, M, N, K, a_kmajor=True, b_kmajor=True, accumulate=True), reps=20)
    print(f'M={M} N={N} K={K}: {t * 1e6:8.1f} us {2.0 * M * N * K / t / 1e12:7.1f} TF', flush=True)
