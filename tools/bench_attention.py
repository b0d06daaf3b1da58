#!/usr/bin/env python3
"""Attention kernels at the bench's shapes: encoder self-attention (T = 260, 8 heads, dense), decoder self-attention (packed
ragged rows <= 64, 12 heads, causal), with and without probability dropout.  Time, TFLOP/s (4 Tq Tk d per head forward, x2.5
backward) per kernel.    python tools/bench_attention.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops, rng  # noqa: E402
from tools.bench_gemm import timeit  # noqa: E402

BF16 = torch.bfloat16
dev = torch.device('cuda:0')


def run(tag, B, H, T, causal, drop, cu=None, total=0, lens=None):
    d = 64 * H
    shape = (total, 3 * d) if cu is not None else (B, T, 3 * d)
    qkv = (torch.randn(*shape, device=dev) * 0.5).to(BF16)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o = torch.empty(*shape[:-1], d, dtype=BF16, device=dev)
    do = torch.randn_like(o)
    n_stat = H * total if cu is not None else B * H * T
    lse, ws = torch.empty(n_stat, device=dev), torch.empty(n_stat, device=dev)
    dqkv = torch.empty_like(qkv)
    dr = (1, rng.site_key(1, 2), rng.threshold(0.1), rng.scale(rng.threshold(0.1))) if drop else None
    kw = dict(cu_q=cu, cu_k=cu, total_q=total) if cu is not None else {}
    fl = 4.0 * H * 64 * (sum(l * l for l in lens) if lens else B * T * T) * (0.5 if causal else 1.0)
    tf = timeit(lambda: ops.attention_fwd(q, k, v, o, lse, B, H, T, T, causal, drop=dr, **kw), reps=10)
    tb = timeit(lambda: ops.attention_bwd(q, k, v, o, do, lse, ws, dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:], B, H, T, T, causal,
                                          drop=dr, **kw), reps=10)
    print(f'{tag:34s} fwd {tf * 1e6:8.1f} us {fl / tf / 1e12:6.1f} TF   bwd {tb * 1e6:8.1f} us {2.5 * fl / tb / 1e12:6.1f} TF', flush=True)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
    for drop in (False, True):
        run(f'encoder T=260 H=8 B={B} drop={int(drop)}', B, 8, 260, False, drop)
        run(f'encoder T=256 H=8 B={B} drop={int(drop)}', B, 8, 256, False, drop)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(8, 65, (B,), generator=g)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    for drop in (False, True):
        run(f'decoder packed<=64 H=12 B={B} drop={int(drop)}', B, 12, 64, True, drop, cu.to(dev), int(cu[-1]), [int(x) for x in lens])


if __name__ == '__main__':
    main()
