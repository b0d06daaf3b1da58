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
print('mismatches:', bad)
sys.exit(1 if bad else 0)
