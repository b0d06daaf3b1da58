"""Fused AdamW over the flat parameter arena (one HIP launch per step; also refreshes the bf16 weight shadow).

Drop-in ``torch.optim.Optimizer`` with ``torch.optim.AdamW`` semantics (decoupled weight decay, bias correction).  The
reference builds ``torch.optim.AdamW(param_groups)`` (trainer.py:145-172); that keeps working on the arena views (the
engine re-casts the bf16 shadow when it sees parameter versions change) -- this class is the MI355X-native equivalent:
per-group lr / weight-decay become a per-segment table read by the kernel.
"""
import torch

from .. import ops


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._model = model          # the VisionEncoderDecoder that owns the arena
        self._step = 0
        self._tables = None
        self.grad_scale = grad_scale
        b = {tuple(g['betas']) for g in self.param_groups}
        e = {g['eps'] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise ValueError('FusedAdamW needs the same betas/eps in every param group (lr / weight_decay may differ)')

    def _build(self, arena):
        by_ptr = {}
        for g in self.param_groups:
            for p in g['params']:
                by_ptr[p.data_ptr()] = (g['lr'], g['weight_decay'])
        ends, lrs, wds = [], [], []
        items = sorted(arena.entries.items(), key=lambda kv: kv[1][0])
        for i, (name, (off, n, _)) in enumerate(items):
            end = items[i + 1][1][0] if i + 1 < len(items) else arena.total
            lr, wd = by_ptr.get(arena.p32.data_ptr() + 4 * off, (0.0, 0.0))      # params outside every group stay frozen
            ends.append(end); lrs.append(lr); wds.append(wd)
        dev = arena.device
        self._tables = (torch.tensor(ends, dtype=torch.long, device=dev), torch.tensor(lrs, dtype=torch.float32, device=dev),
                        torch.tensor(wds, dtype=torch.float32, device=dev), len(ends),
                        tuple(g['lr'] for g in self.param_groups))
        self._m = torch.zeros_like(arena.p32)
        self._v = torch.zeros_like(arena.p32)
        self._arena = arena

    @torch.no_grad()
    def step(self, closure=None):
        arena = self._model._engine.arena
        if arena is None:
            raise RuntimeError('FusedAdamW.step before any forward/backward: the parameter arena does not exist yet')
        if self._tables is None or self._arena is not arena or self._tables[4] != tuple(g['lr'] for g in self.param_groups):
            m, v = getattr(self, '_m', None), getattr(self, '_v', None)
            same = self._tables is not None and self._arena is arena
            self._build(arena)
            if same:
                self._m, self._v = m, v
        self._step += 1
        ends, lrs, wds, nseg, _ = self._tables
        g0 = self.param_groups[0]
        ops.adamw_step(arena.p32, arena.g32, self._m, self._v, arena.pbf, arena.total, ends, lrs, wds, nseg,
                       g0['betas'][0], g0['betas'][1], g0['eps'], self._step, self.grad_scale)
        return None

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=True)
