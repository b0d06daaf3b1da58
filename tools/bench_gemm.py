#!/usr/bin/env python3
"""GEMM micro-benchmark over the nano-224 shapes (B = 128): TFLOP/s per shape and layout, HIP-event timed."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def timeit(fn, reps=20):
    for _ in range(3):
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
    shapes = [('enc qkv', 33280, 1536, 512), ('enc proj', 33280, 512, 512), ('enc fc', 33280, 2048, 512), ('enc fc2', 33280, 512, 2048),
              ('projector', 25088, 512, 8192), ('dec qkv', 8192, 2304, 768), ('dec proj', 8192, 768, 768), ('dec fc', 8192, 3072, 768),
              ('dec fc2', 8192, 768, 3072), ('lm_head', 8192, 50257, 768)]
    tot = {'fwd': [0, 0], 'dX': [0, 0], 'dW': [0, 0]}
    for name, M, N, K in shapes:
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        ldy = (N + 7) // 8 * 8
        dy = torch.zeros(M, ldy, device=dev, dtype=BF16)
        dy[:, :N] = torch.randn(M, N, device=dev).to(BF16)
        y = torch.zeros(M, ldy, device=dev, dtype=BF16)
        dx = torch.empty(M, K, device=dev, dtype=BF16)
        dw = torch.zeros(N, K, device=dev, dtype=F32)
        fl = 2.0 * M * N * K
        t_f = timeit(lambda: ops.gemm(x, w, y, M, N, K))
        t_x = timeit(lambda: ops.gemm(dy, w, dx, M, K, N, b_kmajor=True))
        t_w = timeit(lambda: ops.gemm(dy, x, dw, N, K, M, a_kmajor=True, b_kmajor=True, accumulate=True))
        for k, t in (('fwd', t_f), ('dX', t_x), ('dW', t_w)):
            tot[k][0] += fl
            tot[k][1] += t
        print(f'{name:10s} M={M:6d} N={N:6d} K={K:5d}  fwd {fl / t_f / 1e12:7.1f}  dX {fl / t_x / 1e12:7.1f}  dW {fl / t_w / 1e12:7.1f} TFLOP/s')
    for k, (f, t) in tot.items():
        print(f'total {k}: {f / t / 1e12:.1f} TFLOP/s  ({t * 1e3:.2f} ms)')


if __name__ == '__main__':
    main()
