#!/usr/bin/env python3
"""HBM traffic of the training step's GEMM shapes, one shape at a time (GPU box; run under rocprofv3 --pmc, see tools/gemm_traffic.sh).

Every shape runs REPS times between two marker launches (a one-element torch fill), so that the dispatches of a --pmc pass can be
grouped by shape without knowing how many kernels a call launches.  gpurun_out/gemm_probe_order.json lists the shapes with their
algorithmic bytes (A, B, C and the epilogue operands separately); tools/gemm_traffic_fold.py joins it with the two counter passes."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
REPS = 3
# (M, N, K, a_kmajor, b_kmajor, out f32, epilogue) -- the step's heaviest shapes (bench.py --gemm-breakdown, B = 3072)
SHAPES = [
    (798720, 2048, 512, 0, 0, 0, 'gelu_dout'), (798720, 2048, 512, 0, 1, 0, 'mul_aux'), (798720, 512, 2048, 0, 0, 1, 'res'),
    (798720, 1536, 512, 0, 0, 0, ''), (798720, 512, 2048, 0, 1, 0, ''), (798720, 512, 1536, 0, 1, 0, ''), (798720, 512, 512, 0, 0, 1, 'res'),
    (512, 2048, 798720, 1, 1, 1, 'acc'), (2048, 512, 798720, 1, 1, 1, 'acc'), (1536, 512, 798720, 1, 1, 1, 'acc'), (512, 512, 798720, 1, 1, 1, 'acc'),
    (110265, 3072, 768, 0, 0, 0, 'gelu_dout'), (110265, 3072, 768, 0, 1, 0, 'mul_aux'), (110265, 768, 3072, 0, 0, 1, 'res'), (110265, 768, 3072, 0, 1, 0, ''),
    (110265, 2304, 768, 0, 0, 0, ''), (110265, 768, 2304, 0, 1, 0, ''), (110265, 768, 768, 0, 0, 1, 'res'), (110265, 768, 768, 0, 1, 0, ''),
    (768, 3072, 110265, 1, 1, 1, 'acc'), (3072, 768, 110265, 1, 1, 1, 'acc'), (2304, 768, 110265, 1, 1, 1, 'acc'), (768, 768, 110265, 1, 1, 1, 'acc'),
    (110265, 50257, 768, 0, 0, 0, ''), (110265, 768, 50257, 0, 1, 1, ''), (50257, 768, 110265, 1, 1, 1, 'acc'),
    (602112, 8192, 512, 0, 1, 0, ''), (602112, 512, 8192, 0, 0, 1, ''), (512, 8192, 602112, 1, 1, 1, 'acc'),
]


def main():
    dev = torch.device('cuda:0')
    g = torch.Generator(device='cuda').manual_seed(0)
    marker = torch.zeros(1, device=dev)
    order = []
    for (M, N, K, ak, bk, f32, epi) in SHAPES:
        PAD = int(os.environ.get('I2T_PROBE_PAD', '64'))      # leading dimensions: 64 elements = a 128-byte line (the engine's VOCAB_PAD); 8 = the ABI's minimum

        def mat(r, c, scale):
            return (torch.randn(r, (c + PAD - 1) // PAD * PAD, device=dev, generator=g) * scale).to(BF16)[:, :c]
        a = mat(K, M, 0.5) if ak else mat(M, K, 0.5)
        b = mat(K, N, 0.05) if bk else mat(N, K, 0.05)
        Np = (N + PAD - 1) // PAD * PAD
        out = torch.zeros(M, Np, dtype=F32 if f32 else BF16, device=dev)
        kw = dict(a_kmajor=bool(ak), b_kmajor=bool(bk))
        extra = 0
        if epi == 'res':
            kw['residual'] = torch.randn(M, Np, device=dev, generator=g)
            extra = 4 * M * N
        elif epi == 'acc':
            kw['accumulate'] = True
            extra = 4 * M * N
        elif epi == 'gelu_dout':
            kw.update(act=ops.ACT_GELU_DOUT, aux_out=torch.empty(M, Np, dtype=BF16, device=dev), bias=torch.zeros(N, device=dev))
            extra = 2 * M * N
        elif epi == 'mul_aux':
            kw.update(act=ops.ACT_MUL_AUX, aux_in=(torch.rand(M, Np, device=dev, generator=g)).to(BF16))
            extra = 2 * M * N
        torch.cuda.synchronize()
        marker.zero_()
        for _ in range(REPS):
            ops.gemm(a, b, out, M, N, K, **kw)
        torch.cuda.synchronize()
        order.append(dict(M=M, N=N, K=K, a_kmajor=ak, b_kmajor=bk, out='f32' if f32 else 'bf16', epilogue=epi, reps=REPS,
                          bytes_A=2 * M * K, bytes_B=2 * N * K, bytes_C=(4 if f32 else 2) * M * N, bytes_epilogue_operand=extra))
        del a, b, out, kw
        torch.cuda.empty_cache()
    marker.zero_()
    torch.cuda.synchronize()
    os.makedirs('gpurun_out', exist_ok=True)
    json.dump(order, open('gpurun_out/gemm_probe_order.json', 'w'), indent=1)


if __name__ == '__main__':
    main()
