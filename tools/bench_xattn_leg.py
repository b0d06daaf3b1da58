#!/usr/bin/env python3
"""The fused cross-attention launch alone at the benchmark's decoder width (d = 768, B = 3072) and at the gpu/nano.yaml width (d = 1280,
B = 2048), with and without probability dropout: avg us and fraction of the 2.5 PF bf16 peak (bench.py's xattn_width_leg)."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
from bench import xattn_width_leg  # noqa: E402
dev = torch.device('cuda:0')
for images, d in ((3072, 768), (2048, 1280)):
    r = xattn_width_leg(ops, dev, images=images, d=d, dropout=0.1, reps=30)
    print(f"d={d} B={images}: dropout {r['dropout']['avg_us']:8.1f} us {r['dropout']['tflops'] / 25:6.2f} %   no dropout {r['no_dropout']['avg_us']:8.1f} us "
          f"{r['no_dropout']['tflops'] / 25:6.2f} %", flush=True)
