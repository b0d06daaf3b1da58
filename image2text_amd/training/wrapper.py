"""ModelTrainerWrapper: label shift, loss weights and the weighted-CE train/val step (reference training/wrapper.py).

The step runs as ONE autograd node: encoder + text segment of the decoder + tied lm_head in bf16 + fused cross-entropy, with the
hand-written HIP backward; fp32 logits are never materialised (bf16 logits are overwritten in place by their gradient).
Trainer options (configs/trainer.py:7-15):
  * ``mask_fraction`` / ``random_mask_fraction`` (MLM corruption of the decoder inputs, wrapper.py:161-182): one HIP launch builds
    the inputs from the labels -- BOS shift, ignored -> EOS and the corruption, draws from a counter hash of (step seed, element);
  * ``moco_momentum`` / ``moco_alpha`` (momentum distillation, wrapper.py:30-33,46-59,134-144): a second VisionEncoderDecoder
    (``model_m``, its own flat arenas) is run on the same packed rows without gradient, its bf16 logits are the soft targets of the
    fused distillation cross-entropy, and after every train step it is moved towards the model by one EMA launch over the arenas;
  * ``add_contrastive_loss`` is refused (NotImplementedError): it differentiates the PROMPT rows of ``hidden_state``.
"""
from typing import Tuple

import torch
import torch.nn as nn

from ..configs.models import VisionEncoderDecoderConfig
from ..configs.trainer import TrainerWrapperConfig
from ..engine import F32, HotPath
from ..models.vision_encoder_decoder import VisionEncoderDecoder
from .. import ops


class _LMLossFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, hook, wrapper, images, ids, labels, weights, save, distill):
        model = wrapper.model
        eng: HotPath = model._engine
        a = eng.prepare(model.training and save)
        cfg = model.config
        B, L = ids.shape
        enc_out, enc_ctx = eng.encode(images, save)
        ncls = enc_out.shape[1]
        mem = eng._mem_bf16(enc_out) if cfg.use_cross_attn else None
        off = ncls if cfg.use_soft_prompting else 0
        T = min(L, eng.dec.block - off)
        vl = wrapper._pack_rows(labels[:, :T], B, T) if wrapper.pack_rows else None
        if vl is not None:      # rows past each caption's last label are dead (causal + zero loss weight): not computed
            M = vl.total
            ids_p = ids[:, :T][vl.mask]
            lab, w = labels[:, :T][vl.mask].contiguous(), weights[:, :T][vl.mask].contiguous()
        else:
            M = B * T
            ids_p = ids[:, :T]
            lab = labels[:, :T].contiguous().view(M)
            w = weights[:, :T].contiguous().view(M)
        _, hb, dctx = eng.decode_segment(B, T, mem, ncls, save, ids=ids_p, pos_offset=off, vl=vl)
        logits = eng.logits_bf16(hb, M, capacity=B * T)
        lse = torch.empty(M, dtype=F32, device=a.device)
        loss = torch.zeros(1, dtype=F32, device=a.device)
        inv_t = 1.0 / wrapper.temperature
        teacher = lse_t = None
        if distill:
            # the momentum twin on the same (packed) rows, never differentiated; training-mode dropout as in the reference, whose
            # forward_m runs under no_grad but with the module in train() (wrapper.py:68-71)
            em: HotPath = wrapper.model_m._engine
            em.prepare(model.training)
            enc_m, _ = em.encode(images, False)
            mem_m = em._mem_bf16(enc_m) if cfg.use_cross_attn else None
            _, hb_m, _ = em.decode_segment(B, T, mem_m, ncls, False, ids=ids_p, pos_offset=off, vl=vl, dropout_without_save=model.training)
            teacher = em.logits_bf16(hb_m, M, capacity=B * T)
            lse_t = torch.empty(M, dtype=F32, device=a.device)
            ops.ce_distill_fwd(logits, eng.dec.Vp, teacher, em.dec.Vp, wrapper.alpha, lab, w, inv_t, wrapper.ignore_index, lse, lse_t, loss,
                               M, eng.dec.V)
        else:
            ops.ce_fwd(logits, eng.dec.Vp, lab, w, inv_t, wrapper.ignore_index, lse, loss, M, eng.dec.V)
        ctx.pack = (wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls, teacher, lse_t) if save else None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls, teacher, lse_t = ctx.pack
        eng: HotPath = wrapper.model._engine
        a = eng.arena
        eng.notify_grads_ready('begin')          # e.g. the DP exchange drains whatever is still in flight on the arena
        a.begin_backward()
        gscale = g.reshape(1).to(F32).contiguous()              # stays on the device: no host sync
        if teacher is not None:
            ops.ce_distill_bwd(logits, eng.dec.Vp, teacher, teacher.stride(0), wrapper.alpha, lab, w, inv_t, wrapper.ignore_index, lse, lse_t,
                               gscale, M, eng.dec.V)
        else:
            ops.ce_bwd(logits, eng.dec.Vp, lab, w, inv_t, wrapper.ignore_index, lse, gscale, M, eng.dec.V)
        dmem = torch.zeros(B * ncls, eng.dec.d, dtype=F32, device=a.device)
        eng.decode_backward(dctx, logits, None, dmem)
        eng.notify_grads_ready('decoder')
        eng.encode_backward(enc_ctx, dmem)
        eng.notify_grads_ready('encoder')
        a.attach_grads()
        ctx.pack = None
        return None, None, None, None, None, None, None, None


class ModelTrainerWrapper(nn.Module):
    """Same constructor and ``train_step`` / ``val_step`` contract as the reference (wrapper.py:13-78)."""

    def __init__(self, model_config: VisionEncoderDecoderConfig, tokenizer, trainer_config: TrainerWrapperConfig,
                 ignore_index: int = -100):
        super().__init__()
        tc = trainer_config
        if tc.add_contrastive_loss:
            raise NotImplementedError('contrastive loss is outside the HIP hot path (it differentiates the prompt rows of hidden_state)')
        self.model = VisionEncoderDecoder(config=model_config)
        self.is_momentum = tc.moco_momentum is not None and tc.moco_alpha is not None
        self.model_m = VisionEncoderDecoder(config=model_config) if self.is_momentum else None
        self.tokenizer = tokenizer
        self.ignore_index = ignore_index
        self.temperature = tc.training_temperature
        self.weight_fn = tc.weight_fn
        self.eos_token_weight = tc.eos_token_weight
        self.mask_fraction = tc.mask_fraction
        self.random_mask_fraction = tc.random_mask_fraction
        self.momentum = tc.moco_momentum
        self.alpha = tc.moco_alpha
        if self.mask_fraction > 0 and getattr(tokenizer, 'mask_token_id', None) is None:
            raise ValueError('mask_fraction > 0 needs a tokenizer with a mask_token_id (trainer.py:124-125 adds <MSK>)')
        self.pack_rows = True      # skip the dead caption rows past the last label (result-preserving; see _pack_rows)
        self._corrupt_step = 0
        self.copy_momentum_params()

    def _pack_rows(self, labels, B: int, T: int):
        """Row packing for the decoder: with the causal mask a text row only sees earlier rows, and rows past a caption's last
        non-ignored label carry zero loss weight, so nothing they compute can reach the loss or any gradient.  Returns the
        packed-row description (cumulative lengths, positions, row mask) or None when every row is live.  Needs the lengths on
        the host: one small device->host copy at the very start of the step."""
        valid = labels != self.ignore_index
        idx = torch.arange(1, T + 1, device=labels.device)
        lens = (valid * idx).amax(dim=1)                       # last live position + 1 (0: no label at all)
        lens_host = lens.tolist()
        total = int(sum(lens_host))
        if total == B * T or total == 0:
            return None
        cu = torch.zeros(B + 1, dtype=torch.int32, device=labels.device)
        cu[1:] = torch.cumsum(lens, 0)
        mask = idx[None, :] <= lens[:, None]
        pos = (idx - 1).to(torch.int32).expand(B, T)[mask].contiguous()
        from types import SimpleNamespace
        return SimpleNamespace(cu=cu, pos=pos, total=total, mask=mask, lens_host=lens_host)

    @torch.no_grad()
    def copy_momentum_params(self):
        """model_m <- model (wrapper.py:46-50; also what reset_moco_after_k_epochs triggers)."""
        if not self.is_momentum:
            return
        self.model_m.load_state_dict(self.model.state_dict())

    @torch.no_grad()
    def _momentum_update(self):
        """param_m <- param_m * momentum + param * (1 - momentum) for every parameter (wrapper.py:52-59): ONE launch over the two
        flat arenas (same module tree -> same layout), which also refreshes the twin's bf16 shadow."""
        if not self.is_momentum:
            return
        am, a = self.model_m._engine.arena, self.model._engine.arena
        if am is None or a is None or am.total != a.total or list(am.entries) != list(a.entries):
            raise RuntimeError('momentum update before both models ran on the GPU, or their parameter layouts differ')
        am.refresh_shadow()                                   # (a torch-side edit of model_m since the last cast is honoured first)
        ops.ema_update(am.p32, a.p32, am.pbf, a.total, self.momentum)

    def forward(self, images, input_ids, attn_msk=None) -> Tuple[torch.Tensor, torch.Tensor]:
        out = self.model(images=images, ids=input_ids, attn_msk=attn_msk)
        return out.logits, out.hidden_state

    def train_step(self, images, labels):
        return self._step(images, labels, True)

    def val_step(self, images, labels):
        return self._step(images, labels, False)

    def get_weights(self, labels):
        """wrapper.py:80-96: per-token weights, normalised per sequence (1e-3 in the denominator), divided by the batch."""
        if self.weight_fn == 'constant':
            w = torch.ones_like(labels, dtype=torch.float)
        elif self.weight_fn == 'inverse_sqrt_position':
            n = labels.size(1)
            w = (1.0 / torch.sqrt(torch.arange(1, n + 1, dtype=torch.float, device=labels.device))).expand(labels.size(0), -1).clone()
        else:
            raise ValueError(f'unknown weight_fn: {self.weight_fn}')
        if self.eos_token_weight is not None:
            w[labels == self.tokenizer.eos_token_id] = self.eos_token_weight
        w[labels == self.ignore_index] = 0.0
        return (w / (1e-3 + w.sum(dim=-1, keepdim=True))) / w.size(0)

    def _corruption_seed(self) -> int:
        """A fresh 64-bit seed per corrupted step, derived from torch's seed (torch.manual_seed reproduces a run) and the rank."""
        from ..engine import _dp_rank
        self._corrupt_step += 1
        x = (torch.initial_seed() * 0x9E3779B97F4A7C15 + self._corrupt_step * 0xD1B54A32D192ED03 + _dp_rank() * 0x8CB92BA72F3D8DD7) & (2 ** 64 - 1)
        return x ^ (x >> 29)

    def _step(self, images, labels, is_train: bool):
        dev = next(self.model.parameters()).device
        labels = labels.to(dev).contiguous()
        eos, bos = self.tokenizer.eos_token_id, self.tokenizer.bos_token_id
        bs, sl = labels.shape
        # decoder inputs: [BOS, labels[:-1]], ignored -> EOS, MLM corruption on training steps only (validation data is never masked)
        corrupt = is_train and self.mask_fraction > 0
        ids = torch.empty_like(labels)
        ops.lm_inputs(labels, ids, bs, sl, bos, eos, self.tokenizer.mask_token_id if corrupt else None, self.tokenizer.vocab_size,
                      self.ignore_index, self.mask_fraction if corrupt else 0.0, self.random_mask_fraction if corrupt else 0.0,
                      self._corruption_seed() if corrupt else 0)
        # the reference truncates the labels to the logits' length BEFORE weighting them (wrapper.py:122-133): the per-sequence
        # normaliser only covers the positions that are kept
        eng = self.model._engine
        T = min(sl, eng.dec.block - (eng.enc.ncls if self.model.config.use_soft_prompting else 0))
        weights = self.get_weights(labels[:, :T])
        save = torch.is_grad_enabled()
        distill = self.is_momentum and is_train
        loss = _LMLossFunction.apply(self.model._grad_hook(dev), self, images, ids, labels, weights, save, distill)
        step = 'train' if is_train else 'val'
        if is_train:
            self._momentum_update()
        return loss, {f'{step}_loss_lm': loss.detach()}
