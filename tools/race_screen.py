#!/usr/bin/env python3
"""Race screen for the persistent GEMM kernels: the non-atomic forms have no data-dependent path, so repeated launches must be
bit-identical -- also while another stream keeps the memory system busy (a DMA that lands late would show as a changed tile)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault('I2T_G256_MIN_TILES', '1')
from image2text_amd import ops  # noqa: E402

dev = torch.device('cuda:0')
BF16, F32 = torch.bfloat16, torch.float32
torch.manual_seed(0)
side = torch.cuda.Stream()
noise_a = torch.randn(64 << 20, device=dev)
noise_b = torch.empty_like(noise_a)
bad = 0
shapes = [(266240 // 8, 512, 512), (33280, 1536, 512), (18617, 768, 768), (18617, 3072, 768), (4096, 768, 3072), (2500, 2440, 128),
          (700, 264, 256), (33280, 512, 2048)]
for (M, N, K) in shapes:
    for form in ('fwd_bf16', 'fwd_f32res', 'dx'):
        a = torch.randn(M, K, device=dev).to(BF16)
        w = (torch.randn(N, K, device=dev) * 0.05).to(BF16)
        res = torch.randn(M, N, device=dev)
        wt = w.t().contiguous()

        def run():
            if form == 'fwd_bf16':
                o = torch.empty(M, N, device=dev, dtype=BF16)
                ops.gemm(a, w, o, M, N, K)
            elif form == 'fwd_f32res':
                o = torch.empty(M, N, device=dev, dtype=F32)
                ops.gemm(a, w, o, M, N, K, residual=res)
            else:
                o = torch.empty(M, N, device=dev, dtype=BF16)
                ops.gemm(a, wt, o, M, N, K, b_kmajor=True)
            return o
        ref = run()
        torch.cuda.synchronize()
        for it in range(40):
            if it % 2:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a)          # concurrent HBM traffic
            out = run()
            if not torch.equal(out, ref):
                bad += 1
                d = (out.float() - ref.float()).abs()
                print(f'MISMATCH {form} M={M} N={N} K={K} iter {it}: {int((d > 0).sum())} elements, max {float(d.max()):.4g}', flush=True)
        torch.cuda.synchronize()
    print(f'M={M} N={N} K={K}: done', flush=True)
# the same screen for the persistent kernel on fp8 operands (class 9: its own fragment-refill schedule and fence budgets)
for (M, N, K) in [(12800, 4096, 4096), (16384, 2304, 768), (3000, 3460, 256), (8192, 4096, 11008)]:
    x = torch.randn(M, K, device=dev).to(BF16)
    w = (torch.randn(N, K, device=dev) / K ** 0.5).to(BF16)
    x8, sx = torch.empty(M, K, dtype=torch.uint8, device=dev), torch.empty(M, device=dev)
    w8, sw = torch.empty(N, K, dtype=torch.uint8, device=dev), torch.empty(N, device=dev)
    ops.quant_rows_fp8(x, x8, sx, M, K)
    ops.quant_rows_fp8(w, w8, sw, N, K)
    res = torch.randn(M, N, device=dev)
    for form in ('bf16', 'f32res'):
        def run8():
            o = torch.empty(M, N, device=dev, dtype=BF16 if form == 'bf16' else F32)
            ops.gemm_fp8(x8, sx, w8, sw, o, M, N, K, residual=res if form == 'f32res' else None)
            return o
        ref = run8()
        torch.cuda.synchronize()
        for it in range(40):
            if it % 2:
                with torch.cuda.stream(side):
                    noise_b.copy_(noise_a)
            out = run8()
            if not torch.equal(out, ref):
                bad += 1
                d = (out.float() - ref.float()).abs()
                print(f'MISMATCH fp8 {form} M={M} N={N} K={K} iter {it}: {int((d > 0).sum())} elements, max {float(d.max()):.4g}', flush=True)
        torch.cuda.synchronize()
    print(f'fp8 M={M} N={N} K={K}: done', flush=True)
print('mismatches:', bad)
sys.exit(1 if bad else 0)
