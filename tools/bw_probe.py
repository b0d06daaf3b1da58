#!/usr/bin/env python3
"""HBM bandwidth probe with torch kernels: write-only (fill), read-only (sum), read+write (copy, add) on 1 GiB tensors."""
import torch
dev = torch.device('cuda:0')
n = 1 << 28          # 1 GiB of fp32
x = torch.randn(n, device=dev)
y = torch.empty(n, device=dev)
xb = torch.randn(n, device=dev).to(torch.bfloat16)
yb = torch.empty(n, device=dev, dtype=torch.bfloat16)


def t(fn, reps=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


gb = n * 4 / 1e9
print(f'fill f32   (write {gb:.2f} GB): {gb / t(lambda: y.fill_(1.0)):.0f} GB/s')
print(f'fill bf16  (write {gb / 2:.2f} GB): {gb / 2 / t(lambda: yb.fill_(1.0)):.0f} GB/s')
print(f'sum f32    (read  {gb:.2f} GB): {gb / t(lambda: x.sum()):.0f} GB/s')
print(f'copy f32   (r+w {2 * gb:.2f} GB): {2 * gb / t(lambda: y.copy_(x)):.0f} GB/s')
print(f'add_ f32   (2r+w {3 * gb:.2f} GB): {3 * gb / t(lambda: torch.add(x, y, out=y)):.0f} GB/s')
print(f'copy bf16  (r+w {gb:.2f} GB): {gb / t(lambda: yb.copy_(xb)):.0f} GB/s')
