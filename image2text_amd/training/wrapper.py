"""ModelTrainerWrapper: label shift, loss weights and the weighted-CE train/val step (reference training/wrapper.py).

Default ``trainer: {}`` path only (causal LM loss); momentum distillation, MLM corruption and the contrastive loss are
default-off in the reference (configs/trainer.py:7-15) and refused loudly here.  The step runs as ONE autograd node:
encoder + text segment of the decoder + tied lm_head in bf16 + fused cross-entropy, with the hand-written HIP backward;
fp32 logits are never materialised (bf16 logits are overwritten in place by their gradient).
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
    def forward(ctx, hook, wrapper, images, ids, labels, weights, save):
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
            _, hb, dctx = eng.decode_segment(B, T, mem, ncls, save, ids=ids[:, :T][vl.mask], pos_offset=off, vl=vl)
            lab, w = labels[:, :T][vl.mask].contiguous(), weights[:, :T][vl.mask].contiguous()
        else:
            M = B * T
            _, hb, dctx = eng.decode_segment(B, T, mem, ncls, save, ids=ids[:, :T], pos_offset=off)
            lab = labels[:, :T].contiguous().view(M)
            w = weights[:, :T].contiguous().view(M)
        logits = eng.logits_bf16(hb, M, capacity=B * T)
        lse = torch.empty(M, dtype=F32, device=a.device)
        loss = torch.zeros(1, dtype=F32, device=a.device)
        inv_t = 1.0 / wrapper.temperature
        ops.ce_fwd(logits, eng.dec.Vp, lab, w, inv_t, wrapper.ignore_index, lse, loss, M, eng.dec.V)
        ctx.pack = (wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls) if save else None
        return loss[0]

    @staticmethod
    def backward(ctx, g):
        wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls = ctx.pack
        eng: HotPath = wrapper.model._engine
        a = eng.arena
        eng.notify_grads_ready('begin')          # e.g. the DP exchange drains whatever is still in flight on the arena
        a.begin_backward()
        gscale = g.reshape(1).to(F32).contiguous()              # stays on the device: no host sync
        ops.ce_bwd(logits, eng.dec.Vp, lab, w, inv_t, wrapper.ignore_index, lse, gscale, M, eng.dec.V)
        dmem = torch.zeros(B * ncls, eng.dec.d, dtype=F32, device=a.device)
        eng.decode_backward(dctx, logits, None, dmem)
        eng.notify_grads_ready('decoder')
        eng.encode_backward(enc_ctx, dmem)
        eng.notify_grads_ready('encoder')
        a.attach_grads()
        ctx.pack = None
        return None, None, None, None, None, None, None


class ModelTrainerWrapper(nn.Module):
    """Same constructor and ``train_step`` / ``val_step`` contract as the reference (wrapper.py:13-78)."""

    def __init__(self, model_config: VisionEncoderDecoderConfig, tokenizer, trainer_config: TrainerWrapperConfig,
                 ignore_index: int = -100):
        super().__init__()
        tc = trainer_config
        if tc.moco_momentum is not None and tc.moco_alpha is not None:
            raise NotImplementedError('momentum distillation (moco_*) is outside the HIP hot path (SURVEY.md 8(f) next #4)')
        if tc.mask_fraction > 0:
            raise NotImplementedError('MLM corruption (mask_fraction > 0) is outside the HIP hot path')
        if tc.add_contrastive_loss:
            raise NotImplementedError('contrastive loss is outside the HIP hot path')
        self.model = VisionEncoderDecoder(config=model_config)
        self.model_m = None
        self.is_momentum = False
        self.tokenizer = tokenizer
        self.ignore_index = ignore_index
        self.temperature = tc.training_temperature
        self.weight_fn = tc.weight_fn
        self.eos_token_weight = tc.eos_token_weight
        self.pack_rows = True      # skip the dead caption rows past the last label (result-preserving; see _pack_rows)

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
        return SimpleNamespace(cu=cu, pos=pos, total=total, mask=mask)

    def copy_momentum_params(self):
        return

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

    def _step(self, images, labels, is_train: bool):
        dev = next(self.model.parameters()).device
        labels = labels.to(dev)
        eos, bos = self.tokenizer.eos_token_id, self.tokenizer.bos_token_id
        ids = torch.where(labels != self.ignore_index, labels, torch.full_like(labels, eos))
        bs, sl = ids.shape
        ids = torch.cat((torch.full((bs, 1), bos, dtype=torch.long, device=dev), ids), dim=1)[:, :sl].contiguous()
        # the reference truncates the labels to the logits' length BEFORE weighting them (wrapper.py:122-133): the per-sequence
        # normaliser only covers the positions that are kept
        eng = self.model._engine
        T = min(sl, eng.dec.block - (eng.enc.ncls if self.model.config.use_soft_prompting else 0))
        weights = self.get_weights(labels[:, :T])
        save = torch.is_grad_enabled()
        loss = _LMLossFunction.apply(self.model._grad_hook(dev), self, images, ids, labels, weights, save)
        step = 'train' if is_train else 'val'
        return loss, {f'{step}_loss_lm': loss.detach()}
