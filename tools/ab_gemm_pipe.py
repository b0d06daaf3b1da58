#!/usr/bin/env python3
"""The epilogue classes that LOAD per element (GELU' input, fp32 residual, accumulate) at the benchmark's encoder / decoder shapes."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402
dev = torch.device('cuda:0')
BF16 = torch.bfloat16
for (M, d, ff) in [(199680, 512, 2048), (110336, 768, 3072)]:
    g = torch.Generator().manual_seed(1)
    dy = torch.randn(M, d, generator=g).to(dev).to(BF16)
    w2 = (torch.randn(d, ff, generator=g) * 0.05).to(dev).to(BF16)          # c_proj weight [d, ff]
    pre = torch.randn(M, ff, generator=g).to(dev).to(BF16)
    dpre = torch.empty(M, ff, dtype=BF16, device=dev)
    h = torch.randn(M, ff, generator=g).to(dev).to(BF16)
    x = torch.randn(M, d, generator=g).to(dev)
    out = torch.empty(M, d, device=dev)
    bias = torch.randn(d, generator=g).to(dev)
    t4 = timeit(lambda: ops.gemm(dy, w2, dpre, M, ff, d, b_kmajor=True, act=2, aux_in=pre), reps=20)
    t3 = timeit(lambda: ops.gemm(h, w2, out, M, d, ff, bias=bias, residual=x), reps=20)
    t5 = timeit(lambda: ops.gemm(dy, w2[:, :d].contiguous(), out, M, d, d, b_kmajor=True, accumulate=True), reps=20)
    print(f'M={M} d={d} ff={ff}: dgelu (K={d}) {2.0 * M * ff * d / t4 / 1e12:7.1f} TF   f32+bias+residual (K={ff}) {2.0 * M * d * ff / t3 / 1e12:7.1f} TF   '
          f'f32 accumulate (K={d}) {2.0 * M * d * d / t5 / 1e12:7.1f} TF', flush=True)
