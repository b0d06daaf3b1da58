#!/usr/bin/env python3
"""GEMM micro-benchmark over the greedy-decode shapes (M = captions per batch, one token per caption per step).

    python tools/bench_gemm_decode.py [M ...]         (I2T_GEMM=v1 / I2T_G256_MIN_TILES=n select the kernel family)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def timeit(fn, reps=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    Ms = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384]
    # name, N, K, per-step count, kind
    shapes = [('qkv', 2304, 768, 12, 'bf16'), ('proj+res', 768, 768, 18, 'res'), ('cross q', 768, 768, 6, 'bf16'),
              ('fc+gelu', 3072, 768, 12, 'gelu'), ('fc2+res', 768, 3072, 12, 'res'), ('lm_head', 50257, 768, 1, 'f32')]
    for M in Ms:
        tot_t = tot_f = 0.0
        for name, N, K, cnt, kind in shapes:
            x = torch.randn(M, K, device=dev).to(BF16)
            w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
            bias = torch.randn((N + 3) // 4 * 4, device=dev)
            ldc = (N + 63) // 64 * 64
            if kind == 'res':
                out = torch.zeros(M, N, device=dev, dtype=F32)
                fn = lambda: ops.gemm(x, w, out, M, N, K, bias=bias, residual=out)
            elif kind == 'gelu':
                out = torch.zeros(M, N, device=dev, dtype=BF16)
                fn = lambda: ops.gemm(x, w, out, M, N, K, bias=bias, act=1)
            elif kind == 'f32':
                out = torch.zeros(M, ldc, device=dev, dtype=F32)
                fn = lambda: ops.gemm(x, w, out, M, N, K)
            else:
                out = torch.zeros(M, N, device=dev, dtype=BF16)
                fn = lambda: ops.gemm(x, w, out, M, N, K, bias=bias)
            t = timeit(fn)
            fl = 2.0 * M * N * K
            tot_t += t * cnt
            tot_f += fl * cnt
            print(f'M={M:6d} {name:9s} N={N:6d} K={K:5d}  {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s  x{cnt}', flush=True)
        print(f'M={M:6d} per token step: {tot_t * 1e3:.3f} ms  {tot_f / tot_t / 1e12:.1f} TFLOP/s', flush=True)


if __name__ == '__main__':
    main()
