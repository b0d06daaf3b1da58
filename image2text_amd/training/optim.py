"""Fused optimizers over the flat parameter arena: ONE HIP launch per step, which also refreshes the bf16 weight shadow.

``FusedAdamW`` has ``torch.optim.AdamW`` semantics (decoupled weight decay, bias correction); ``SNRAdam`` (exported from
``models/optimizer.py``, where the reference keeps it) has the reference's SNRAdam semantics.  The reference builds
``optim_clazz(param_groups)`` (trainer.py:145-172); ``torch.optim.AdamW`` itself keeps working on the arena views (the
engine re-casts the bf16 shadow when it sees parameter versions change) -- these classes are the MI355X-native
equivalents: per-group lr / weight-decay become a per-segment table read by the kernel, moments live in two flat arenas.

The arena is found through the parameters themselves (``engine.arena_of``): it exists once the model has run one
forward on the GPU (``HotPath.prepare``), which is always the case by the first ``step()``.
"""
import torch

from .. import ops
from ..engine import arena_of


class _ArenaOptimizer(torch.optim.Optimizer):
    _kernel = None                   # staticmethod: ops.adamw_step / ops.snradam_step

    def __init__(self, params, defaults, model=None, grad_scale=1.0):
        super().__init__(params, defaults)
        self._model = model          # optional: the VisionEncoderDecoder that owns the arena (else found via the params)
        self._step = 0
        self._tables = None
        self._arena = None
        self.grad_scale = grad_scale
        b = {tuple(g['betas']) for g in self.param_groups}
        e = {g['eps'] for g in self.param_groups}
        if len(b) != 1 or len(e) != 1:
            raise NotImplementedError(f'{type(self).__name__} takes one (betas, eps) for all param groups (lr / weight_decay may '
                                      'differ per group): the bias corrections are launch-wide scalars')

    def _find_arena(self):
        if self._model is not None:
            arena = self._model._engine.arena
        else:
            arena = None
            for g in self.param_groups:
                for p in g['params']:
                    arena = arena_of(p)
                    break
                if arena is not None or g['params']:
                    break
        if arena is None:
            raise RuntimeError(f'{type(self).__name__}.step: the parameters are not views of a live parameter arena -- run one '
                               'forward/backward of the VisionEncoderDecoder on the GPU first (there is no per-tensor CPU path)')
        return arena

    def _frozen_signature(self):
        return tuple(p.requires_grad for g in self.param_groups for p in g['params'])

    def _refresh_hyper(self, arena, hyper):
        lrs, wds = self._seg_host
        for (lr, wd), g in zip(hyper, self.param_groups):
            for p in g['params']:
                i = self._seg_of.get(p.data_ptr())
                if i is not None and lrs[i] >= 0.0:
                    lrs[i], wds[i] = lr, wd
        ends, dl, dw, n, _ = self._tables
        dl.copy_(torch.tensor(lrs, dtype=torch.float32), non_blocking=True)
        dw.copy_(torch.tensor(wds, dtype=torch.float32), non_blocking=True)
        self._tables = (ends, dl, dw, n, hyper)

    def _has_state(self, arena, name: str) -> bool:
        """An arena entry the kernels update (lr >= 0): a trainable parameter that some backward writes.  Frozen / ``skip_grad``
        segments (LoRA's base weights, a frozen backbone or decoder) never get moments: none are saved, none are expected."""
        p = arena.params.get(name)
        return p is not None and p.requires_grad and name not in getattr(arena, 'skip_grad', ())

    def state_dict(self):
        """torch's layout plus the flat moments: ``state`` maps an arena entry name to its (exp_avg, exp_avg_sq) slices, ``step`` is
        the launch-wide step count -- so ``accelerator.save_state`` / a resumed run keep the moments and the bias correction.
        Only segments the optimizer actually updates are emitted, as HOST copies (no second device copy of the moments at save time:
        a frozen 7-B decoder would otherwise add ~54 GB of zeros to the device and to every checkpoint)."""
        sd = super().state_dict()
        sd['i2t_step'] = self._step
        if self._arena is not None:
            a = self._arena
            sd['i2t_moments'] = {name: (self._m[o:o + n].detach().to('cpu', copy=True), self._v[o:o + n].detach().to('cpu', copy=True))
                                 for name, (o, n, _) in a.entries.items() if self._has_state(a, name)}
        return sd

    def load_state_dict(self, state_dict):
        state_dict = dict(state_dict)
        self._step = int(state_dict.pop('i2t_step', 0))
        self._pending_moments = state_dict.pop('i2t_moments', None)        # applied once the arena exists (first step)
        super().load_state_dict(state_dict)
        self._tables = None
        if self._arena is not None:
            self._apply_pending_moments(self._arena)

    def _apply_pending_moments(self, arena):
        pm = getattr(self, '_pending_moments', None)
        if pm is None:
            return
        # names the checkpoint does not carry (frozen at save time, or new) start from zero moments, as a fresh optimizer would
        self._m.zero_()
        self._v.zero_()
        for name, (m, v) in pm.items():
            e = arena.entries.get(name)
            if e is not None and e[1] == m.numel():
                self._m[e[0]:e[0] + e[1]].copy_(m)
                self._v[e[0]:e[0] + e[1]].copy_(v)
        self._pending_moments = None

    def _build(self, arena):
        by_ptr = {}
        # parameters no backward ever writes (a backbone run under no_grad: their .grad stays None and torch's optimizers skip them)
        skipped = {arena.p32.data_ptr() + 4 * arena.entries[n][0] for n in getattr(arena, 'skip_grad', ())}
        self._skipped = skipped
        for g in self.param_groups:
            for p in g['params']:
                if arena_of(p) is not arena:
                    raise RuntimeError(f'{type(self).__name__}: a parameter of shape {tuple(p.shape)} lives outside the arena')
                # (a frozen parameter -- LoRA's base weights -- keeps its value whatever group it was handed over in)
                by_ptr[p.data_ptr()] = (g['lr'], g['weight_decay'], p) if (p.requires_grad and p.data_ptr() not in skipped) \
                    else (-1.0, 0.0, p)      # lr < 0: skipped by the kernels
        ends, lrs, wds, self._params = [], [], [], []
        items = sorted(arena.entries.items(), key=lambda kv: kv[1][0])
        for i, (name, (off, n, _)) in enumerate(items):
            end = items[i + 1][1][0] if i + 1 < len(items) else arena.total
            lr, wd, p = by_ptr.get(arena.p32.data_ptr() + 4 * off, (-1.0, 0.0, None))     # params outside every group stay frozen (lr < 0: skipped)
            ends.append(end); lrs.append(lr); wds.append(wd)
            if p is not None:
                self._params.append((name, p))
        dev = arena.device
        self._tables = (torch.tensor(ends, dtype=torch.long, device=dev), torch.tensor(lrs, dtype=torch.float32, device=dev),
                        torch.tensor(wds, dtype=torch.float32, device=dev), len(ends),
                        tuple((g['lr'], g['weight_decay']) for g in self.param_groups))
        if self._arena is not arena:
            self._m = torch.zeros_like(arena.p32)
            self._v = torch.zeros_like(arena.p32)
        self._arena = arena
        self._seg_host = (list(lrs), list(wds))
        self._seg_of = {arena.p32.data_ptr() + 4 * off: i for i, (name, (off, n, _)) in enumerate(items)}
        self._frozen_sig = self._frozen_signature()
        self._apply_pending_moments(arena)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        arena = self._find_arena()
        hyper = tuple((g['lr'], g['weight_decay']) for g in self.param_groups)
        if self._tables is None or self._arena is not arena or self._frozen_sig != self._frozen_signature():
            self._build(arena)
        elif self._tables[4] != hyper:                                   # an lr scheduler edited the groups: same segments, new values --
            self._refresh_hyper(arena, hyper)                            # two small in-place uploads instead of rebuilding three tables
        missing = [n for n, p in self._params if p.requires_grad and p.grad is None and p.data_ptr() not in self._skipped]
        if missing:
            raise RuntimeError(f'{type(self).__name__}.step: {len(missing)} parameters have no gradient (first: {missing[0]}); the '
                               'fused step updates the whole arena at once')
        self._step += 1
        ends, lrs, wds, nseg, _ = self._tables
        g0 = self.param_groups[0]
        type(self)._kernel(arena.p32, arena.g32, self._m, self._v, arena.pbf, arena.total, ends, lrs, wds, nseg,
                           g0['betas'][0], g0['betas'][1], g0['eps'], self._step, self.grad_scale)
        arena.generation += 1           # values changed behind torch's version counters (e.g. merged LoRA weights must be rebuilt)
        return loss

    def zero_grad(self, set_to_none: bool = True):
        super().zero_grad(set_to_none=True)


class FusedAdamW(_ArenaOptimizer):
    _kernel = staticmethod(ops.adamw_step)

    def __init__(self, params, model=None, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, grad_scale=1.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay), model=model, grad_scale=grad_scale)


class SNRAdam(_ArenaOptimizer):
    """Reference models/optimizer.py:6-113 (same constructor; ``trainer.py:169`` picks it with ``use_snr_optim``): the
    denominator is the bias-corrected running standard deviation of the gradient around its running mean, so a parameter
    whose gradient is consistent over time takes larger steps."""
    _kernel = staticmethod(ops.snradam_step)

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), weight_decay: float = 0.0, eps: float = 1e-8):
        if lr <= 0.0:
            raise ValueError('Invalid learning rate: {}'.format(lr))
        if eps < 0.0:
            raise ValueError('Invalid epsilon value: {}'.format(eps))
        for i in (0, 1):
            if not 0.0 <= betas[i] < 1.0:
                raise ValueError('Invalid beta parameter at index {}: {}'.format(i, betas[i]))
        if weight_decay < 0:
            raise ValueError('Invalid weight_decay value: {}'.format(weight_decay))
        super().__init__(params, dict(lr=lr, betas=betas, weight_decay=weight_decay, eps=eps))
