#!/usr/bin/env python3
"""One decoder cross-attention layer forward at the bench shapes (packed caption rows x 64 memory rows per image, 12 heads x 64):
q projection, k/v projection, packed-row attention, output projection + residual.  Time and TFLOP/s per part and for the block.

    python tools/bench_cross_attention.py [B] [d]      (also usable under rocprofv3 --pmc, see tools/profile_mfma.sh)
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
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    d = int(sys.argv[2]) if len(sys.argv) > 2 else 768          # 1280: the decoder of the reference's gpu/nano.yaml (20 heads of 64)
    H, S = d // 64, 64
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(8, 65, (B,), generator=g)
    cu = torch.zeros(B + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    M = int(cu[-1])
    cu = cu.to(dev)
    ln3 = torch.randn(M, d, device=dev).to(BF16)
    mem = torch.randn(B * S, d, device=dev).to(BF16)
    win = (torch.randn(3 * d, d, device=dev) * 0.03).to(BF16)
    wout = (torch.randn(d, d, device=dev) * 0.03).to(BF16)
    bin_, bout = torch.randn(3 * d, device=dev) * 0.1, torch.randn(d, device=dev) * 0.1
    x1 = torch.randn(M, d, device=dev)
    q = torch.empty(M, d, dtype=BF16, device=dev)
    kv = torch.empty(B, S, 2 * d, dtype=BF16, device=dev)
    co = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, device=dev)
    x2 = torch.empty(M, d, device=dev)
    parts = [
        ('q projection', lambda: ops.gemm(ln3, win[:d], q, M, d, d, bias=bin_[:d]), 2.0 * M * d * d),
        ('k/v projection', lambda: ops.gemm(mem, win[d:], kv.view(B * S, 2 * d), B * S, 2 * d, d, bias=bin_[d:]), 2.0 * B * S * 2 * d * d),
        ('attention', lambda: ops.attention_fwd(q, kv[..., :d], kv[..., d:], co, lse, B, H, 64, S, False, cu_q=cu, total_q=M), 4.0 * M * S * d),
        ('out projection + res', lambda: ops.gemm(co, wout, x2, M, d, d, bias=bout, residual=x1), 2.0 * M * d * d),
    ]
    fused = ('k/v projection + attention (fused)', lambda: ops.xattn_kv_fused(mem, win[d:], bin_[d:], q, kv, co, lse, B, S, H, 64, cu_q=cu, total_q=M),
             2.0 * B * S * 2 * d * d + 4.0 * M * S * d)
    tot_t = tot_f = 0.0
    ts = []
    for name, fn, fl in parts:
        t = timeit(fn, reps=20)
        ts.append(t)
        tot_t += t
        tot_f += fl
        print(f'{name:22s} {t * 1e6:8.1f} us  {fl / t / 1e12:7.1f} TFLOP/s', flush=True)
    tf = timeit(fused[1], reps=20)
    print(f'{fused[0]:36s} {tf * 1e6:8.1f} us  {fused[2] / tf / 1e12:7.1f} TFLOP/s   (unfused: {(ts[1] + ts[2]) * 1e6:.1f} us)', flush=True)
    ft = ts[0] + tf + ts[3]
    print(f'cross-attention block, fused  {ft * 1e6:8.1f} us  {tot_f / ft / 1e12:7.1f} TFLOP/s = {tot_f / ft / 2.5e15 * 100:.1f} % of 2.5 PF')
    print(f'cross-attention block   {tot_t * 1e6:8.1f} us  {tot_f / tot_t / 1e12:7.1f} TFLOP/s = {tot_f / tot_t / 2.5e15 * 100:.1f} % of 2.5 PF   '
          f'(B = {B}: {M} packed query rows, {B * S} memory rows, {tot_f / 1e9:.0f} GFLOP)')


if __name__ == '__main__':
    main()
