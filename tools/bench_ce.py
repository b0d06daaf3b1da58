#!/usr/bin/env python3
"""Cross-entropy over the benchmark's logits (110 k packed rows x 50 257): two-kernel form against the one-pass kernel."""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402
dev = torch.device('cuda:0')
M, V, ld = int(sys.argv[1]) if len(sys.argv) > 1 else 110265, 50257, 50264
logits = torch.empty(M, ld, dtype=torch.bfloat16, device=dev)
logits.normal_(0, 1.0)
logits[:, V:] = 0
labels = torch.randint(0, V, (M,), device=dev)
w = torch.rand(M, device=dev)
lse, loss, one = torch.empty(M, device=dev), torch.zeros(1, device=dev), torch.ones(1, device=dev)


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def two():
    ops.ce_fwd(logits, ld, labels, w, 1.0, -100, lse, loss, M, V)
    ops.ce_bwd(logits, ld, labels, w, 1.0, -100, lse, one, M, V)


def one_pass():
    ops.ce_fwd_bwd(logits, ld, labels, w, 1.0, -100, lse, loss, M, V)


for name, fn in (('ce_fwd + ce_bwd', two), ('ce_fwd_bwd', one_pass), ('ce_fwd + ce_bwd', two), ('ce_fwd_bwd', one_pass)):
    logits.normal_(0, 1.0)
    ms = timeit(fn)
    print(f'{name:18s} {ms:7.3f} ms   {2.0 * M * ld * 2 / ms / 1e9:6.2f} TB/s of (read + write once)', flush=True)
