#!/usr/bin/env python3
"""Per-tile fixed cost of the persistent GEMM: time = rounds * (c + s * K) for one shape family, with the epilogue intact,
without its global stores (I2T_G256_DBG=1) and without any epilogue (I2T_G256_DBG=2).  One process per setting:

    for d in 0 1 2; do I2T_G256_DBG=$d python tools/dissect_epilogue.py; done
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')


def main():
    M, N = 65536, 2048                      # 256 x 8 = 2048 tiles = 8 rounds on 256 CUs
    rounds = (M // 256) * (N // 256) / 256
    pts = []
    for K in (256, 512, 1024, 2048):
        x = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        y = torch.empty(M, N, device=dev, dtype=BF16)
        bias = torch.randn(N, device=dev)
        t = timeit(lambda: ops.gemm(x, w, y, M, N, K), reps=20)
        tg = timeit(lambda: ops.gemm(x, w, y, M, N, K, bias=bias, act=1), reps=20)
        pts.append((K, t / rounds * 1e6, tg / rounds * 1e6))
        print(f'dbg={os.environ.get("I2T_G256_DBG", "0")} K={K:5d}  plain {t / rounds * 1e6:7.2f} us/tile   gelu {tg / rounds * 1e6:7.2f} us/tile', flush=True)
    (k0, a0, g0), (k1, a1, g1) = pts[0], pts[-1]
    s = (a1 - a0) / (k1 - k0) * 64
    print(f'   plain: {s:.3f} us per K-tile, fixed {a0 - s * k0 / 64:.2f} us/tile;  gelu fixed {g0 - (g1 - g0) / (k1 - k0) * k0:.2f} us/tile')


if __name__ == '__main__':
    main()
