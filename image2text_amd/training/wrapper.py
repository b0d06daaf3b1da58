"""ModelTrainerWrapper: label shift, loss weights and the weighted-CE train/val step (reference training/wrapper.py).

The step runs as ONE autograd node: encoder + text segment of the decoder + tied lm_head in bf16 + fused cross-entropy, with the
hand-written HIP backward; fp32 logits are never materialised (bf16 logits are overwritten in place by their gradient).
Trainer options (configs/trainer.py:7-15):
  * ``mask_fraction`` / ``random_mask_fraction`` (MLM corruption of the decoder inputs, wrapper.py:161-182): one HIP launch builds
    the inputs from the labels -- BOS shift, ignored -> EOS and the corruption, draws from a counter hash of (step seed, element);
  * ``moco_momentum`` / ``moco_alpha`` (momentum distillation, wrapper.py:30-33,46-59,134-144): a second VisionEncoderDecoder
    (``model_m``, its own flat arenas) is run on the same packed rows without gradient, its bf16 logits are the soft targets of the
    fused distillation cross-entropy, and after every train step it is moved towards the model by one EMA launch over the arenas;
  * ``add_contrastive_loss`` (wrapper.py:98-118): the hidden rows [prompt rows | text rows] against the target embeddings of the
    whole batch -- one GEMM whose -inf column bias masks the ignored positions, the same fused cross-entropy kernels on its
    diagonal labels, and a backward that runs the prompt segment of the decoder (the only consumer of its gradient) into the
    encoder output and scatter-adds the target-embedding gradient into wte.
"""
from typing import Tuple

import os

import torch
import torch.nn as nn

from ..configs.models import VisionEncoderDecoderConfig
from ..configs.trainer import TrainerWrapperConfig
from ..engine import BF16, F32, HotPath
from ..models.vision_encoder_decoder import VisionEncoderDecoder
from .. import ops


def _ce_forward(ctx, eng, logits, lab, w, inv_t, ignore_index, lse, loss, M, save) -> bool:
    """The weighted cross-entropy of a step.  A forward that will be differentiated (``save``) takes the one-pass kernel: every row is
    read once and overwritten with its gradient on the spot (the logits of a training step have no other reader) -- True tells the
    backward that only the upstream scalar is left to apply.  I2T_CE_ONE_PASS=0 / a vocabulary past the kernel's 65 536 columns: the
    two-kernel form (ce_fwd now, ce_bwd in the backward)."""
    if save and eng.dec.V <= ops.CE_ONE_PASS_MAX_V and os.environ.get('I2T_CE_ONE_PASS', '1') != '0':
        ops.ce_fwd_bwd(logits, eng.dec.Vp, lab, w, inv_t, ignore_index, lse, loss, M, eng.dec.V)
        return True
    ops.ce_fwd(logits, eng.dec.Vp, lab, w, inv_t, ignore_index, lse, loss, M, eng.dec.V)
    return False


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
        mem = eng._mem_bf16(enc_out) if eng.cross_inputs else None
        off = ncls if cfg.use_soft_prompting else 0
        T = min(L, eng.dec.block - off)
        if eng.dec.prefixed:       # Hugging Face decoder + soft prompt: one causal sequence [encoder outputs | text] (engine.decode_prefixed)
            tvl = wrapper._pack_rows(labels[:, :T], B, T) if wrapper.pack_rows else None
            if tvl is not None:     # rows past a caption's last label are dead (causal, zero loss weight): n_p + len_b rows per sequence
                M = tvl.total
                lab, w = labels[:, :T][tvl.mask].contiguous(), weights[:, :T][tvl.mask].contiguous()
            else:
                M = B * T
                lab, w = labels[:, :T].contiguous().view(M), weights[:, :T].contiguous().view(M)
            _, hb, dctx = eng.decode_prefixed(B, T, enc_out, mem, save, ids, text_mask=tvl.mask if tvl is not None else None)
            logits = eng.logits_bf16(hb, M, capacity=B * T)
            lse = torch.empty(M, dtype=F32, device=a.device)
            loss = torch.zeros(1, dtype=F32, device=a.device)
            inv_t = 1.0 / wrapper.temperature
            ctx.ce_done = _ce_forward(ctx, eng, logits, lab, w, inv_t, wrapper.ignore_index, lse, loss, M, save)
            ctx.pack = (wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls, None, None, None) if save else None
            return loss[0], torch.zeros((), dtype=F32, device=a.device)
        vl = wrapper._pack_rows(labels[:, :T], B, T) if (wrapper.pack_rows and eng.dec.causal) else None      # (dead rows are dead under a causal mask only)
        if vl is not None:      # rows past each caption's last label are dead (causal + zero loss weight): not computed
            M = vl.total
            ids_p = ids[:, :T][vl.mask]
            lab, w = labels[:, :T][vl.mask].contiguous(), weights[:, :T][vl.mask].contiguous()
        else:
            M = B * T
            ids_p = ids[:, :T]
            lab = labels[:, :T].contiguous().view(M)
            w = weights[:, :T].contiguous().view(M)
        hid, hb, dctx = eng.decode_segment(B, T, mem, ncls, save, ids=ids_p, pos_offset=off, vl=vl)
        logits = eng.logits_bf16(hb, M, capacity=B * T)
        lse = torch.empty(M, dtype=F32, device=a.device)
        loss = torch.zeros(1, dtype=F32, device=a.device)
        inv_t = 1.0 / wrapper.temperature
        teacher = lse_t = None
        ctx.ce_done = False
        if distill:
            # the momentum twin on the same (packed) rows, never differentiated; training-mode dropout as in the reference, whose
            # forward_m runs under no_grad but with the module in train() (wrapper.py:68-71)
            em: HotPath = wrapper.model_m._engine
            em.prepare(model.training)
            enc_m, _ = em.encode(images, False)
            mem_m = em._mem_bf16(enc_m) if em.cross_inputs else None
            _, hb_m, _ = em.decode_segment(B, T, mem_m, ncls, False, ids=ids_p, pos_offset=off, vl=vl, dropout_without_save=model.training)
            teacher = em.logits_bf16(hb_m, M, capacity=B * T)
            lse_t = torch.empty(M, dtype=F32, device=a.device)
            ops.ce_distill_fwd(logits, eng.dec.Vp, teacher, em.dec.Vp, wrapper.alpha, lab, w, inv_t, wrapper.ignore_index, lse, lse_t, loss,
                               M, eng.dec.V)
        else:
            ctx.ce_done = _ce_forward(ctx, eng, logits, lab, w, inv_t, wrapper.ignore_index, lse, loss, M, save)
        con = None
        loss_c = torch.zeros(1, dtype=F32, device=a.device)
        if wrapper.add_contrastive_loss:
            con = _contrastive_forward(wrapper, eng, enc_out, mem, hid, labels, B, T, ncls, vl, save, loss_c)
        ctx.pack = (wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls, teacher, lse_t, con) if save else None
        return loss[0], loss_c[0]

    @staticmethod
    def backward(ctx, g, g_c):
        wrapper, enc_ctx, dctx, logits, lab, w, lse, inv_t, B, M, ncls, teacher, lse_t, con = ctx.pack
        eng: HotPath = wrapper.model._engine
        a = eng.arena
        eng.notify_grads_ready('begin')          # e.g. the DP exchange drains whatever is still in flight on the arena
        a.begin_backward()
        if g is None:
            g = torch.zeros((), dtype=F32, device=a.device)
        gscale = g.reshape(1).to(F32).contiguous()              # stays on the device: no host sync
        if teacher is not None:
            ops.ce_distill_bwd(logits, eng.dec.Vp, teacher, teacher.stride(0), wrapper.alpha, lab, w, inv_t, wrapper.ignore_index, lse, lse_t,
                               gscale, M, eng.dec.V)
        elif getattr(ctx, 'ce_done', False):     # the forward's one-pass kernel already left w/T (softmax - onehot) in place: apply the upstream scalar
            ops.scale_bf16(logits, M * eng.dec.Vp, gscale)
        else:
            ops.ce_bwd(logits, eng.dec.Vp, lab, w, inv_t, wrapper.ignore_index, lse, gscale, M, eng.dec.V)
        dmem = torch.zeros(B * ncls, eng.dec.d, dtype=F32, device=a.device)
        if eng.dec.prefixed:
            eng.decode_prefixed_backward(dctx, logits, None, dmem)
        elif con is not None and g_c is not None:
            dhid, dph = _contrastive_backward(wrapper, eng, con, g_c, M)
            if con.pctx is not None:      # text rows and prompt rows are ONE sequence to the blocks' gradient normalisers: lock step
                _, dxp = eng.decode_backward_pair(dctx, logits, dhid, con.pctx, None, dph, dmem)
                dmem.view(B, ncls, -1)[:, :con.n_p].add_(dxp.view(B, con.n_p, -1))      # d/d(inputs_embeds) of the prompt rows
            else:
                eng.decode_backward(dctx, logits, dhid, dmem)
        else:
            eng.decode_backward(dctx, logits, None, dmem)
        eng.notify_grads_ready('decoder')
        eng.encode_backward(enc_ctx, dmem)
        eng.notify_grads_ready('encoder')
        a.attach_grads()
        ctx.pack = None
        return None, None, None, None, None, None, None, None


def _contrastive_rows(B: int, T: int, n_p: int, Lc: int, vl, dev):
    """Row maps of the contrastive term's hidden rows (wrapper.py:99-104): position c of sequence b is prompt row c (c < n_p) or text
    row c - n_p.  Returns (source rows in the text segment's row space, their destinations in [B * Lc], prompt sources, destinations)."""
    import numpy as np
    n_pc, Lt = min(n_p, Lc), max(Lc - n_p, 0)
    b = np.arange(B, dtype=np.int64)
    if vl is None:
        t = np.arange(min(Lt, T), dtype=np.int64)
        src = (b[:, None] * T + t[None]).ravel()
        dst = (b[:, None] * Lc + n_p + t[None]).ravel()
    else:
        full = np.asarray(vl.lens_host, dtype=np.int64)
        lens = np.minimum(full, Lt)
        cu = np.zeros(B + 1, dtype=np.int64)
        cu[1:] = np.cumsum(full)
        seq = np.repeat(b, lens)
        within = np.arange(int(lens.sum())) - np.repeat(np.cumsum(lens) - lens, lens)
        src, dst = cu[seq] + within, seq * Lc + n_p + within
    c = np.arange(n_pc, dtype=np.int64)
    psrc, pdst = (b[:, None] * n_p + c[None]).ravel(), (b[:, None] * Lc + c[None]).ravel()
    to = lambda x: torch.from_numpy(x.astype(np.int32)).to(dev)
    return to(src), to(dst), to(psrc), to(pdst)


def _contrastive_forward(wrapper, eng, enc_out, mem, hid, labels, B, T, ncls, vl, save, loss_out):
    """The contrastive term (reference training/wrapper.py:98-118) on the device: hidden rows [prompt rows | text rows] of every
    sequence against the target embeddings wte[label] of the whole batch; ignored positions are masked out as columns through a
    -inf GEMM bias and skipped as rows.  Adds the loss into loss_out; returns what backward needs."""
    from types import SimpleNamespace
    a, cfg = eng.arena, wrapper.model.config
    d, dev = eng.dec.d, a.device
    n_p = min(ncls, eng.dec.block) if cfg.use_soft_prompting else 0
    L = labels.shape[1]
    Lc = min(L, n_p + T)
    N = B * Lc
    pctx = ph = None
    if n_p:        # the prompt rows of hidden_state: their own causal segment (text never attends to them), differentiated here
        ph, _, pctx = eng.decode_segment(B, n_p, mem, ncls, save, embeds=enc_out[:, :n_p].reshape(B * n_p, -1), pos_offset=0,
                                         drop_plan=eng.dec_drop_prompt)
    src, dst, psrc, pdst = _contrastive_rows(B, T, n_p, Lc, vl, dev)
    Hc = torch.zeros(N, d, dtype=F32, device=dev)
    tmp = torch.empty(max(src.numel(), psrc.numel(), 1), d, dtype=F32, device=dev)
    if src.numel():
        ops.gather_rows(hid, src, src.numel(), d, out_f32=tmp)
        ops.scatter_rows(tmp, dst, Hc, src.numel(), d)
    if psrc.numel():
        ops.gather_rows(ph, psrc, psrc.numel(), d, out_f32=tmp)
        ops.scatter_rows(tmp, pdst, Hc, psrc.numel(), d)
    Hb = torch.empty(N, d, dtype=BF16, device=dev)
    ops.cast_f32_bf16(Hc, Hb)
    lab_c = labels[:, :Lc].contiguous().view(N)
    keep = lab_c != wrapper.ignore_index
    ids0 = torch.where(keep, lab_c, torch.zeros_like(lab_c))
    E = torch.empty(N, d, dtype=F32, device=dev)
    ops.embed_fwd(ids0, a.P(eng.n_wte), None, E, N, 1, d, 0, eng.dec.V)
    Eb = torch.empty(N, d, dtype=BF16, device=dev)
    ops.cast_f32_bf16(E, Eb)
    Np = (N + 7) // 8 * 8
    colbias = torch.zeros(Np, dtype=F32, device=dev)
    colbias[:N].masked_fill_(~keep, float('-inf'))
    P = torch.zeros(N, Np, dtype=BF16, device=dev)
    ops.gemm(Hb, Eb, P, N, N, d, bias=colbias)
    w_c = wrapper.get_weights(labels[:, :Lc]).contiguous().view(N)
    diag = torch.where(keep, torch.arange(N, device=dev), torch.full_like(lab_c, wrapper.ignore_index))
    lse = torch.empty(N, dtype=F32, device=dev)
    inv_t = 1.0 / wrapper.contrastive_temperature
    ops.ce_fwd(P, Np, diag, w_c, inv_t, wrapper.ignore_index, lse, loss_out, N, N)
    if not save:
        return None
    return SimpleNamespace(P=P, Np=Np, N=N, diag=diag, w=w_c, lse=lse, inv_t=inv_t, Hb=Hb, Eb=Eb, ids0=ids0, pctx=pctx, n_p=n_p, B=B,
                           rows=(src, dst, psrc, pdst))


def _contrastive_backward(wrapper, eng, con, g_c, M):
    """Gradient of the contrastive term onto the target embeddings (accumulated into wte's gradient) and onto the hidden rows:
    returns (d text hidden fp32 [M, d], d prompt hidden fp32 [B n_p, d] | None)."""
    a = eng.arena
    d, dev, N = eng.dec.d, a.device, con.N
    ops.ce_bwd(con.P, con.Np, con.diag, con.w, con.inv_t, wrapper.ignore_index, con.lse, g_c.reshape(1).to(F32).contiguous(), N, N)
    dH = torch.empty(N, d, dtype=F32, device=dev)
    ops.gemm(con.P, con.Eb, dH, N, d, N, b_kmajor=True)
    dE = torch.zeros(N, d, dtype=F32, device=dev)
    ops.gemm(con.P, con.Hb, dE, N, d, N, a_kmajor=True, b_kmajor=True, accumulate=True)
    ops.embed_bwd(con.ids0, dE, a.G(eng.n_wte), None, N, 1, d, 0, eng.dec.V)
    src, dst, psrc, pdst = con.rows
    tmp = torch.empty(max(src.numel(), psrc.numel(), 1), d, dtype=F32, device=dev)
    dhid = torch.zeros(M, d, dtype=F32, device=dev)
    if src.numel():
        ops.gather_rows(dH, dst, src.numel(), d, out_f32=tmp)
        ops.scatter_rows(tmp, src, dhid, src.numel(), d)
    dph = None
    if con.pctx is not None:
        dph = torch.zeros(con.B * con.n_p, d, dtype=F32, device=dev)
        ops.gather_rows(dH, pdst, psrc.numel(), d, out_f32=tmp)
        ops.scatter_rows(tmp, psrc, dph, psrc.numel(), d)
    return dhid, dph


class ModelTrainerWrapper(nn.Module):
    """Same constructor and ``train_step`` / ``val_step`` contract as the reference (wrapper.py:13-78)."""

    def __init__(self, model_config: VisionEncoderDecoderConfig, tokenizer, trainer_config: TrainerWrapperConfig,
                 ignore_index: int = -100):
        super().__init__()
        tc = trainer_config
        self.add_contrastive_loss = tc.add_contrastive_loss
        self.contrastive_temperature = tc.training_contrastive_temperature
        self.model = VisionEncoderDecoder(config=model_config)
        if self.model._engine.dec.prefixed and (tc.add_contrastive_loss or (tc.moco_momentum is not None and tc.moco_alpha is not None)):
            raise NotImplementedError('contrastive loss / momentum distillation with a Hugging Face decoder and a soft prompt are outside '
                                      'the HIP hot path (one causal sequence [encoder outputs | text]: engine.decode_prefixed)')
        if tc.add_contrastive_loss and model_config.use_soft_prompting and not self.model._engine.dec.causal:
            raise NotImplementedError('contrastive loss with a non-causal decoder and a soft prompt: the prompt rows then attend to the '
                                      'text rows, which the two-segment backward does not cover')
        self.is_momentum = tc.moco_momentum is not None and tc.moco_alpha is not None
        self.model_m = VisionEncoderDecoder(config=model_config) if self.is_momentum else None
        if self.model_m is not None:        # the teacher's dropout masks are independent of the student's (reference wrapper.py:68-71)
            self.model_m._engine.seed_salt = 0xA5A5F00DC0FFEE11
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
        am.generation += 1

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
        loss, loss_c = _LMLossFunction.apply(self.model._grad_hook(dev), self, images, ids, labels, weights, save, distill)
        step = 'train' if is_train else 'val'
        metrics = {f'{step}_loss_lm': loss.detach()}
        if self.add_contrastive_loss:                          # wrapper.py:206-209
            metrics[f'{step}_loss_contrastive'] = loss_c.detach()
            loss = loss + loss_c
        if is_train:
            self._momentum_update()
        return loss, metrics
