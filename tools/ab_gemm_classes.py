#!/usr/bin/env python3
"""A/B of the bf16 epilogue classes of the persistent GEMM under an environment knob read at the first call (one process per setting,
e.g. I2T_GEMM3=0|2): prints TF per shape and a checksum of every output (bit-equal kernels print the same checksums).  Within one
process the class measured FIRST on a shape runs on colder clocks (up to -10 %): compare the same column across processes, never two
columns of one.  Round 3 used it for (a) gemm3 vs the 256^2 kernel (843-850 vs 775-798 TF at K = 512: gemm3's default rule went off),
(b) a v_permlane16_swap store path without the LDS transpose (16 rows x 64 B per store instead of 4 rows x 128 B: bit-equal, 2-4 %
SLOWER on every shape -- dropped; the LDS round trip is not what the epilogue waits for, the store footprint is)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

dev = torch.device('cuda:0')
BF16 = torch.bfloat16


def csum(t):
    return int(t.view(torch.int16).to(torch.int64).sum().item()) & 0xFFFFFFFF


def main():
    tag = os.environ.get('I2T_GEMM3', 'default')
    for (M, N, K) in [(99840, 2048, 512), (99840, 1536, 512), (110336, 3072, 768), (110336, 2304, 768), (110336, 768, 768), (65536, 4096, 4096)]:
        g = torch.Generator(device='cpu').manual_seed(M + N + K)
        x = torch.randn(M, K, generator=g).to(dev).to(BF16)
        w = (torch.randn(N, K, generator=g) * 0.05).to(dev).to(BF16)
        bias = torch.randn(N, generator=g).to(dev)
        y, pre = torch.empty(M, N, device=dev, dtype=BF16), torch.empty(M, N, device=dev, dtype=BF16)
        fl = 2.0 * M * N * K
        t1 = timeit(lambda: ops.gemm(x, w, y, M, N, K, bias=bias), reps=20)
        c1 = csum(y)
        t2 = timeit(lambda: ops.gemm(x, w, y, M, N, K, bias=bias, act=1, aux_out=pre), reps=20)
        c2, c3 = csum(y), csum(pre)
        t3 = timeit(lambda: ops.gemm(x, w, y, M, N, K, bias=bias, drop=(2, 777, 429496729, 1 / 0.9)), reps=20) if N % 3 == 0 else 0.0
        c4 = csum(y)
        print(f'gemm3={tag} {M}x{N}x{K}: bias {fl / t1 / 1e12:7.1f} TF  gelu+pre {fl / t2 / 1e12:7.1f} TF  drop2 {fl / t3 / 1e12 if t3 else 0:7.1f} TF   '
              f'sums {c1:08x} {c2:08x} {c3:08x} {c4:08x}', flush=True)


if __name__ == '__main__':
    main()
