#!/usr/bin/env python3
"""Per-kernel timing of the 6x6 conv stack (3 -> 8 -> 16 -> 32 channels, 224 x 224) forward / backward-data / backward-weight.

    python tools/bench_conv.py [B] [only]      (only = substring filter on the kernel label, e.g. "conv3 fwd")
Usable under ``rocprofv3 --pmc ...`` (one process, a handful of launches per kernel).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from image2text_amd import ops  # noqa: E402

BF16, F32 = torch.bfloat16, torch.float32
dev = torch.device('cuda:0')
H = W = 224


def timeit(fn, reps=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    only = sys.argv[2] if len(sys.argv) > 2 else ''
    chans = [3, 8, 16, 32]
    img = torch.randn(B, 3, H, W, device=dev)
    ws = torch.empty(32 * 36 * 32, dtype=BF16, device=dev)
    scratch = torch.empty(32 * 36 * 16, dtype=F32, device=dev)
    wts = [torch.randn(chans[i + 1], chans[i], 6, 6, device=dev) * 0.05 for i in range(3)]
    bias = [torch.randn(chans[i + 1], device=dev) * 0.1 for i in range(3)]
    acts = [torch.empty(B, H, W, chans[i + 1], dtype=BF16, device=dev) for i in range(2)]
    acts.append(torch.empty(B, chans[3], H, W, dtype=BF16, device=dev))
    res = []

    def run(label, fn, flops, nbytes):
        if only and only not in label:
            return
        t = timeit(fn)
        res.append((label, t))
        print(f'{label:22s} {t * 1e6 / B:8.3f} us/img  {flops * B / t / 1e12:7.1f} TFLOP/s  {nbytes * B / t / 1e12:6.2f} TB/s (algorithmic)', flush=True)

    px = H * W
    cur, lay = img, ops.LAYOUT_NCHW_F32
    for i in range(3):
        cin, cout, last = chans[i], chans[i + 1], i == 2
        x, y, xl = cur, acts[i], lay
        run(f'conv{i + 1} fwd', lambda: ops.conv6_fwd(x, xl, i > 0, wts[i], bias[i], y, last, ws, B, cin, cout, H, W),
            2.0 * px * 36 * cin * cout, px * (cin * (4 if i == 0 else 2) + cout * 2))
        if last and (not only or 'nhwc-out' in only):
            y2 = torch.empty(B, H, W, cout, dtype=BF16, device=dev)
            run(f'conv{i + 1} fwd nhwc-out', lambda: ops.conv6_fwd(x, xl, i > 0, wts[i], bias[i], y2, False, ws, B, cin, cout, H, W),
                2.0 * px * 36 * cin * cout, px * (cin * 2 + cout * 2))
        cur, lay = y, ops.LAYOUT_NHWC_BF16
    dy_nchw = torch.randn(B, 32, H, W, device=dev).to(BF16)
    dy = torch.empty(B, H, W, 32, dtype=BF16, device=dev)
    run('nchw->nhwc', lambda: ops.nchw_to_nhwc(dy_nchw, dy, B, 32, H, W), 0.0, px * 32 * 4)
    ops.nchw_to_nhwc(dy_nchw, dy, B, 32, H, W)
    for i in (2, 1, 0):
        cin, cout = chans[i], chans[i + 1]
        xin = img if i == 0 else acts[i - 1]
        xl = ops.LAYOUT_NCHW_F32 if i == 0 else ops.LAYOUT_NHWC_BF16
        dw, db = torch.zeros_like(wts[i]), torch.zeros_like(bias[i])
        d = dy
        run(f'conv{i + 1} bwd-weight', lambda: ops.conv6_bwd_weight(d, ops.LAYOUT_NHWC_BF16, xin, xl, i > 0, dw, db, scratch, B, cin, cout, H, W),
            2.0 * px * 36 * cin * cout, px * (cin * (4 if i == 0 else 2) + cout * 2))
        run(f'conv{i + 1} bwd-weight nodb', lambda: ops.conv6_bwd_weight(d, ops.LAYOUT_NHWC_BF16, xin, xl, i > 0, dw, None, scratch, B, cin, cout, H, W),
            2.0 * px * 36 * cin * cout, px * (cin * (4 if i == 0 else 2) + cout * 2))
        if i > 0:
            dx = torch.empty(B, H, W, cin, dtype=BF16, device=dev)
            run(f'conv{i + 1} bwd-data', lambda: ops.conv6_bwd_data(d, ops.LAYOUT_NHWC_BF16, wts[i], acts[i - 1], dx, ws, B, cin, cout, H, W),
                2.0 * px * 36 * cin * cout, px * (cout * 2 + cin * 4))
            dy = dx
    print(f'total {sum(t for _, t in res) * 1e6 / B:.2f} us/img')


if __name__ == '__main__':
    main()
