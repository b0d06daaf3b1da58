"""Host-side orchestration of the captioning hot path over the HIP kernels (train step + forward).

This is the MI355X replacement for what torch autograd + ATen do under the reference's
``VisionEncoderDecoder.forward`` (models/vision_encoder_decoder.py:51-134), ``VisionTransformerEncoder.forward``
(models/encoder.py:163-178), ``TransformerDecoder.forward`` (models/decoder.py:214-256) and
``TransformerBlock.forward`` (models/layers.py:565-608): an explicit forward and a hand-written backward that
launch the C-ABI kernels (``ops``) on the current HIP stream.

Data layout in HBM
  * parameters: one flat fp32 arena (the ``nn.Parameter``s are views into it, so ``state_dict`` /
    ``load_state_dict`` / any torch optimizer keep working), one flat fp32 gradient arena (``p.grad`` are views),
    one flat bf16 shadow of the parameters (the MFMA operands), refreshed by the fused AdamW kernel or by one cast
    launch whenever torch-side code changed a parameter (tracked through ``Tensor._version``);
  * residual stream fp32 [B*T, d]; every GEMM operand bf16; LayerNorm reads fp32 and writes bf16 straight into the
    next GEMM's A operand; GEMM epilogues fold bias, GELU(+pre-activation for backward) and the residual add;
  * no transposed copies of weights or activations exist: dX = dY.W and dW = dY^T.X use the k-major operand paths
    of the GEMM kernel (LDS transposed reads).

Reference quirks reproduced on purpose (each pinned by a golden fixture):
  * user attention masks are numerically inert (oracle/_mask_to_additive): self-attention is causal-only in the
    decoder and unmasked in the encoder;
  * with soft prompting text rows never see prompt columns and prompt-row logits are sliced off, so the prompt rows
    and the text rows are two independent causal segments; the training path runs only the text segment
    (positions offset by n_cls), ``forward`` also runs the prompt segment to return the full ``hidden_state`` -- differentiably:
    when a loss touches those rows (a custom loss on ``hidden_state``, the contrastive term) the two segments' backward passes run
    in lock step and share every block's gradient normaliser (``_lockstep``);
  * LayerNormND is applied twice with shared weights; "patches" are a flat chunking of the CHW conv output;
  * ``normalize_gradients`` (models/functions.py:19-24) rescales the residual-stream gradient to unit L2 norm at
    every block output, per replica.
"""
import os
import re
import weakref
from types import SimpleNamespace
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import ops, rng
from .engine_family import FamilyBlocks, family_spec
from .engine_llama import LlamaBlocks
from .engine_lora import LPAD, LoraAdapters
from .engine_vit import ViTEncoder
from .lib import I2TError

BF16, F32 = torch.bfloat16, torch.float32
# I2T_GELU_DOUT=0: the MLP's first GEMM keeps the pre-activation and the backward re-evaluates GELU' (A/B runs, bit-compatible with round 3)
GELU_KEEPS_DERIVATIVE = os.environ.get('I2T_GELU_DOUT', '1') != '0'
# I2T_FOLD_NORMALISER=0: every block runs its own grad_normalize pass over the incoming gradient (A/B runs)
NORMALISER_FOLDED = os.environ.get('I2T_FOLD_NORMALISER', '1') != '0'
# the embedding dropouts applied by the producers of the embedded rows (forward) and by the lowest block's last LayerNorm backward
# (backward) instead of by passes of their own (I2T_EMB_DROP_FUSED=0: the passes, for A/B runs)
EMB_DROP_FUSED = os.environ.get('I2T_EMB_DROP_FUSED', '1') not in ('', '0')


# leading dimension of every logits buffer: the vocabulary rounded up to this many columns.  64 bf16 columns = one 128-byte line, so a
# row of 50257 logits starts on a line boundary (at 8 columns -- 16-byte alignment, the ABI's minimum -- every 256-column tile segment of
# a row straddled two lines: the lm_head GEMM wrote 1.22x its bytes and the two gradient GEMMs fetched the logits' gradient 1.35x,
# profiles/r04_gemm_traffic_by_shape.txt).  I2T_VOCAB_PAD=8 restores the old layout for A/B runs.
VOCAB_PAD = max(8, int(os.environ.get('I2T_VOCAB_PAD', '64')) // 8 * 8)          # (a multiple of 8: the ABI's 16-byte row alignment)


# every parameter of 4096 elements or more starts on a multiple of this many elements of the flat arenas: 64 = a 128-byte line of the bf16 shadow and two of the
# fp32 parameter / gradient arenas (8 -- 16 bytes, the kernels' minimum -- left the gradient GEMMs' 64-byte atomic rows and the optimizer's
# segments straddling lines).  I2T_ARENA_ALIGN=8: the old layout, for A/B runs.
ARENA_ALIGN = max(8, int(os.environ.get('I2T_ARENA_ALIGN', '64')) // 8 * 8)


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def _ff_mult(rotator) -> float:
    return rotator.ff_mult if hasattr(rotator, 'ff_mult') else rotator.ff_mult_factor


def _dp_rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


class DropPlan:
    """Dropout sites of one training forward: a step seed + the tower's rates.  ``get`` returns the
    (mode, key, thr, scale) tuple the kernels take, or None when that site is off.  Sites (reference lines):
    emb = embedding dropout (decoder.py:243, encoder.py:171); qkv = per-token q/k/v multipliers (layers.py:454-461,
    rate attn_dropout); sdpa = dropout_p of the self-attention probabilities (layers.py:465); resid = after attn.c_proj
    (layers.py:469); xattn = nn.MultiheadAttention's attention-weight dropout (layers.py:537-542); mlp = after
    mlp.c_proj (layers.py:485); xresid = after the cross-attention output projection -- transformers' GPT-2 only (its
    crossattention ends in resid_dropout like its self-attention; nn.MultiheadAttention has no such site), ``xresid=True``;
    lora_<site> = the input dropout of a LoRA adapter (peft's lora_dropout, rate ``p_lora``).
    The same plan object is kept in the saved context and re-evaluated in backward."""
    KINDS = {'emb': 0, 'qkv': 1, 'sdpa': 2, 'resid': 3, 'xattn': 4, 'mlp': 5, 'xresid': 6, 'lora_attn_c_attn': 7, 'lora_xattn_c_attn': 8,
             'lora_mlp_c_fc': 9, 'lora_mlp_c_proj': 10, 'lora_qkv': 11, 'lora_o': 12, 'lora_gu': 13, 'lora_dn': 14}

    def __init__(self, seed: int, tower: int, p: float, p_attn: float, compact_layer: int = -1, live_rows: int = 0, xresid: bool = False,
                 p_lora: float = 0.0):
        """compact_layer / live_rows: in that layer only the first ``live_rows`` rows of every sequence are computed after
        the K/V projections (the encoder's last block: only its CLS rows are ever read), so its sdpa / resid / mlp masks are
        indexed over those rows alone; ``get`` then appends live_rows to the tuple so that a full-row consumer (the oracle)
        can place the mask (rows past live_rows are dead, any mask does)."""
        self.seed, self.tower, self.p, self.p_attn = seed, tower, p, p_attn
        self.compact_layer, self.live_rows, self.xresid, self.p_lora = compact_layer, live_rows, xresid, p_lora

    def get(self, layer: int, kind: str):
        p = self.p_attn if kind == 'qkv' else (self.p_lora if kind.startswith('lora_') else self.p)      # lora_*: the adapters' input dropout
        if p <= 0.0 or (kind == 'xresid' and not self.xresid):
            return None
        site = self.tower * 4096 + layer * 16 + self.KINDS[kind]
        e = (2 if kind == 'qkv' else 1, rng.site_key(self.seed, site), rng.threshold(p), rng.scale(rng.threshold(p)))
        if layer == self.compact_layer and kind in ('sdpa', 'resid', 'mlp'):
            e = e + (self.live_rows,)
        return e


_ARENAS = weakref.WeakSet()


def arena_of(param: torch.Tensor):
    """The live ParamArena whose fp32 buffer ``param`` is a view of, or None (optimizers find their arena through this)."""
    ptr = param.data_ptr()
    for a in _ARENAS:
        base = a.p32.data_ptr()
        if base <= ptr < base + 4 * a.total and a.p32.device == param.device:
            return a
    return None


_MOE_PARAM = re.compile(r'^(.*)\.experts\.(\d+)\.(l1|l2)\.(weight|bias)$')
_LORA_A = re.compile(r'^(.*\.)?lora_params\.h\d+_\w+_A$')
_QKV_PARAM = re.compile(r'^(.*\.self_attn)\.(q|k|v)_proj\.(weight|bias)$')


def _arena_order(named):
    """Arena order = ``named_parameters()`` order, except that the parameters of one MoELinear (reference layers.py:301-328) are
    regrouped so that the operands of its two GEMMs are single contiguous views (engine_family.py):

        [experts.*.l1.weight | gate layer 0 weight]   -> one [E P + G, in] matrix: every expert's l1 and the gate's first layer
        [experts.*.l1.bias   | gate layer 0 bias]     -> its bias vector (a zero pad entry stands in for a bias-free gate)
        [experts.*.l2.weight] [experts.*.l2.bias]     -> stacked [E, out, P] / [E, out] for the W2aug pack kernel
        gate layer 2 weight / bias (when the gate has a hidden layer)

    Returns (name, parameter | None, numel, shape) tuples; None marks a pad entry (zeros, never trained: no optimizer group holds
    it, so the fused optimizers keep it frozen)."""
    groups, order = {}, []
    qkv = {}
    for name, p in named:
        mq = _QKV_PARAM.match(name)
        if mq:      # transformers' Llama / Qwen2 attention: q | k | v weights adjacent, then q | k | v biases (Qwen2) -> one fused GEMM
            if mq.group(1) not in qkv:
                qkv[mq.group(1)] = {}
                order.append((mq.group(1) + '.<qkv>', 'qkv'))
            qkv[mq.group(1)][f'{mq.group(2)}_proj.{mq.group(3)}'] = p
            continue
        m = _MOE_PARAM.match(name)
        key = m.group(1) if m else (name.split('.expert_gates.')[0] if '.expert_gates.model.' in name else None)
        if key is None:
            order.append((name, p))
            continue
        if key not in groups:
            groups[key] = {}
            order.append((key, None))
        groups[key][name[len(key) + 1:]] = p
    out = []
    for name, p in order:
        if isinstance(p, str):
            g = qkv[name[:-len('.<qkv>')]]
            for k in [f'{x}_proj.{w}' for w in ('weight', 'bias') for x in 'qkv']:
                if k in g:
                    out.append((f"{name[:-len('.<qkv>')]}.{k}", g[k], g[k].numel(), g[k].shape))
            continue
        if p is not None:
            out.append((name, p, p.numel(), p.shape))
            if _LORA_A.match(name) and p.shape[0] < LPAD:    # LoRA rank -> LPAD zero-padded rows: lora_A is the [LPAD, in] operand of the adapter GEMMs
                pad = (LPAD - p.shape[0]) * p.shape[1]
                out.append((name + '.<pad>', None, pad, torch.Size([pad])))
            continue
        g = groups[name]
        E = 1 + max(int(k.split('.')[1]) for k in g if k.startswith('experts.'))
        gate0_w = g['expert_gates.model.0.weight']
        seq = [f'experts.{e}.l1.weight' for e in range(E)] + ['expert_gates.model.0.weight']
        seq += [f'experts.{e}.l1.bias' for e in range(E)] + ['expert_gates.model.0.bias']
        seq += [f'experts.{e}.l2.weight' for e in range(E)] + [f'experts.{e}.l2.bias' for e in range(E)]
        seq += [k for k in g if k.startswith('expert_gates.') and k not in seq]
        for k in seq:
            if k in g:
                out.append((f'{name}.{k}', g[k], g[k].numel(), g[k].shape))
            else:       # bias-free gate: zeros of the gate's first-layer width keep the bias vector of the fused GEMM contiguous
                out.append((f'{name}.{k}', None, gate0_w.shape[0], torch.Size([gate0_w.shape[0]])))
    return out


class ParamArena:
    """Flat fp32 parameters + fp32 gradients + bf16 shadow; parameters of ``module`` become views into it."""

    def __init__(self, module: torch.nn.Module, device: torch.device):
        _ARENAS.add(self)
        self.device = device
        self.entries: Dict[str, Tuple[int, int, torch.Size]] = {}
        self.params: Dict[str, torch.nn.Parameter] = {}
        off = 0
        for name, p, numel, shape in _arena_order(list(module.named_parameters())):          # (tied parameters appear once)
            if numel >= 4096:                      # matrices start on a line; the small entries between them (biases, norms, the MoE
                off = _round_up(off, ARENA_ALIGN)  # family's per-expert pieces, whose kernels rely on their packing) keep the 16-byte rule
            self.entries[name] = (off, numel, shape)
            if p is not None:
                self.params[name] = p
            off += _round_up(numel, 8)
        self.total = off
        self.p32 = torch.zeros(off, dtype=F32, device=device)
        self.g32 = torch.zeros(off, dtype=F32, device=device)
        self.pbf = torch.zeros(off, dtype=BF16, device=device)
        with torch.no_grad():
            for name, p in self.params.items():
                o, n, shape = self.entries[name]
                view = self.p32[o:o + n].view(shape)
                view.copy_(p.data.to(device=device, dtype=F32))
                p.data = view
        self.skip_grad = set()          # names no backward ever writes (a backbone the reference runs under no_grad): their .grad stays None
        self._versions = None
        self.generation = 0             # bumped whenever parameter VALUES change (torch-side writes seen by refresh_shadow, fused optimizer steps)
        self.grads_attached = False

    def valid(self) -> bool:
        for name, p in self.params.items():
            o, n, _ = self.entries[name]
            if p.data_ptr() != self.p32.data_ptr() + 4 * o or p.device != self.device:
                return False
        return True

    def refresh_shadow(self):
        """Re-cast fp32 -> bf16 if any parameter was modified by torch-side code since the last cast."""
        v = tuple(p._version for p in self.params.values())
        if v != self._versions:
            ops.cast_f32_bf16(self.p32, self.pbf, self.total)
            if ops.precise():          # I2T_PRECISE=1: the cast also wrote the parameters' low-order bf16 terms (ops.py)
                ops.register_shadow(self.pbf)
            self._versions = v
            self.generation += 1

    def P(self, name: str) -> Optional[torch.Tensor]:
        e = self.entries.get(name)
        return None if e is None else self.p32[e[0]:e[0] + e[1]].view(e[2])

    def W(self, name: str) -> torch.Tensor:
        o, n, shape = self.entries[name]
        return self.pbf[o:o + n].view(shape)

    def G(self, name: str) -> Optional[torch.Tensor]:
        e = self.entries.get(name)
        return None if e is None else self.g32[e[0]:e[0] + e[1]].view(e[2])

    def trainable(self, name: str) -> bool:
        """False for a frozen parameter (requires_grad off, e.g. LoRA's base weights): backward skips its gradient GEMM"""
        p = self.params.get(name)
        return (p is None or p.requires_grad) and name not in self.skip_grad

    def Gt(self, name: str) -> Optional[torch.Tensor]:
        """the gradient view of a TRAINABLE parameter, else None (kernels skip null gradient outputs)"""
        return self.G(name) if self.trainable(name) else None

    def span(self, kind: str, names, shape):
        """One view over several ADJACENT entries (e.g. q_proj | k_proj | v_proj weights = the fused projection's [N, K] matrix);
        kind: 'P' fp32 parameters, 'W' bf16 shadow, 'G' fp32 gradients."""
        o0, end = self.entries[names[0]][0], None
        for n in names:
            o, numel, _ = self.entries[n]
            if end is not None and o != end:
                raise I2TError(f'arena entries {names} are not adjacent (a member whose size is not a multiple of {ARENA_ALIGN} elements? I2T_ARENA_ALIGN=8 relaxes it)')
            if numel % 8 and n != names[-1]:
                raise I2TError(f'arena entry {n} ({numel} elements) breaks the 8-element alignment of a fused view')
            end = o + numel
        buf = {'P': self.p32, 'W': self.pbf, 'G': self.g32}[kind]
        return buf[o0:end].view(shape)

    def begin_backward(self):
        """Zero the gradient arena when this is a fresh accumulation window (every p.grad is None)."""
        fresh = all(p.grad is None for p in self.params.values())
        if fresh:
            self.g32.zero_()
        else:
            for name, p in self.params.items():
                if p.grad is not None and p.grad.data_ptr() != self.G(name).data_ptr():
                    raise I2TError(f'{name}.grad was replaced by a foreign tensor; call zero_grad(set_to_none=True)')

    def attach_grads(self):
        for name, p in self.params.items():
            if p.requires_grad and p.grad is None and name not in self.skip_grad:
                p.grad = self.G(name)


def decoder_hot_config(model):
    """The TransformerDecoderConfig the decoder's arithmetic follows: the model config's own, or the one a GPT2HuggingfaceDecoder
    derived from its checkpoint (models/decoder.py); None for the Llama-2 / Qwen2 decoders (their own spec, ``llama_spec``)."""
    if getattr(model.decoder, 'llama_spec', None) is not None:
        return None
    return getattr(model.decoder, 'hot_config', None) or model.config.decoder_config


class HotPath(FamilyBlocks, LlamaBlocks, LoraAdapters, ViTEncoder):
    """Forward/backward of one VisionEncoderDecoder over the HIP kernels."""

    def __init__(self, model: torch.nn.Module):
        self.model = model
        self.cfg = model.config
        self.arena: Optional[ParamArena] = None
        ecfg, dcfg = self.cfg.vision_encoder_config, decoder_hot_config(model)
        self.dcfg = dcfg
        # the encoder output reaches the decoder's cross-attention: the model asks for it and -- Hugging Face decoders only, which
        # drop the input otherwise (reference decoder.py:341-361) -- the decoder has the layers
        self.cross_inputs = bool(self.cfg.use_cross_attn) and getattr(model.decoder, 'use_cross_attn', True)
        self.has_bridge = model.has_bridge
        self.ep = 'encoder.0.' if self.has_bridge else 'encoder.'
        self.dp = 'decoder.'
        self.vit = not hasattr(ecfg, 'transformer_config')         # PretrainedViTConfig: torchvision backbone + head (engine_vit.ViTEncoder)
        if self.vit:
            self.vit_setup(model, ecfg)
        else:
            eac = ecfg.transformer_config.attn_config
            self.enc = SimpleNamespace(kind='scratch', d=eac.n_embd, H=eac.n_head, L=ecfg.n_layer, ncls=ecfg.n_cls,
                                       P2=ecfg.num_patches ** 2, causal=ecfg.transformer_config.is_causal,
                                       ff=int(_ff_mult(ecfg.transformer_config.rotator_config) * eac.n_embd),
                                       dropout=eac.dropout, attn_dropout=eac.attn_dropout,
                                       k=ecfg.feature_extractor_kernel_size[0])
            self.enc.fam = family_spec(ecfg.transformer_config, ecfg.n_layer)
        ls = getattr(model.decoder, 'llama_spec', None)
        if ls is not None:
            # Llama-2 / Qwen2 blocks (engine_llama.LlamaBlocks): RMSNorm, rotary embedding, grouped K/V heads, SwiGLU; no learned
            # positions, no dropout, no cross-attention, no gradient normaliser; the parameters keep transformers' names
            self.dec = SimpleNamespace(d=ls.d, H=ls.H, L=ls.L, V=ls.V, Vp=_round_up(ls.V, VOCAB_PAD), block=ls.block, causal=True, ff=ls.ff,
                                       dropout=0.0, attn_dropout=0.0, fam=None, grad_norm=False, advpos=False, llama=ls,
                                       prefixed=bool(self.cfg.use_soft_prompting))
            self.dcfg = dcfg = SimpleNamespace(skip_alternate_cross_attn=False, advanced_pos_emb_gate_sizes=None, n_layer=ls.L,
                                               transformer_config=SimpleNamespace(is_cross_attn=False))
            self.dec.lora = getattr(model.decoder, 'lora', None)              # LoRA adapters on the Llama / Qwen2 blocks (engine_llama._llama_lora)
            self.n_wte = self.dp + ls.wte                                     # (Falcon: transformer.word_embeddings)
            self.n_head = self.n_wte if ls.tied else f'{self.dp}backbone.lm_head.weight'
        else:
            dac = dcfg.transformer_config.attn_config
            self.dec = SimpleNamespace(d=dac.n_embd, H=dac.n_head, L=dcfg.n_layer, V=dcfg.vocab_size,
                                       Vp=_round_up(dcfg.vocab_size, VOCAB_PAD), block=dcfg.block_size,
                                       causal=dcfg.transformer_config.is_causal,
                                       ff=int(_ff_mult(dcfg.transformer_config.rotator_config) * dac.n_embd),
                                       dropout=dac.dropout, attn_dropout=dac.attn_dropout, llama=None)
            # the nano-mini block family (multi-query / MoE / sparse / head widths other than 64): engine_family.FamilyBlocks
            self.dec.fam = family_spec(dcfg.transformer_config, dcfg.n_layer, force=not dcfg.transformer_config.is_causal)
            self.dec.grad_norm = not hasattr(model.decoder, 'hot_config')     # layers.py:606-607; transformers' GPT-2 block has none
            self.dec.prefixed = hasattr(model.decoder, 'hot_config') and bool(self.cfg.use_soft_prompting)     # see decode_prefixed
            self.dec.advpos = bool(dcfg.use_advanced_pos_emb)       # decoder.wpe = one MLP per position (layers.py:617-638)
            self.n_wte = self.n_head = f'{self.dp}transformer.wte.weight'      # token embedding / lm_head weight (tied, decoder.py:189-204)
            self.dec.lora = getattr(model.decoder, 'lora', None)              # LoRA adapters on a GPT2HuggingfaceDecoder (engine._lora_*)
            if self.dec.lora is not None and self.dec.fam is not None:
                raise NotImplementedError('LoRA adapters run on the dense decoder blocks only')
        self._sparse_idx, self._sparse_versions, self.sparse_epoch = {'enc': None, 'dec': None}, None, 0
        self._refresh_sparse_sets()
        self._moe_cache, self._sub_cache = {}, {}
        self._lora_merge_list = []
        self.moe_trace = None           # tests set a dict: site -> [(gate values, routing weights), ...] of every MoELinear forward
        if not self.vit:
            gates = list(ecfg.feature_extractor_gate_sizes or [])
            chans = [ecfg.input.n_channels] + gates + [ecfg.n_channels]
            self.conv = [(f'{self.ep}feature_extractor.model.{2 * i}', chans[i], chans[i + 1]) for i in range(len(chans) - 1)]
            # MFMA (channels-last) convolutions when the kernel is 6x6 and every intermediate width is 8/16/32 channels
            self.conv_mfma = (ecfg.feature_extractor_kernel_size[0] == 6 and all(c in (8, 16, 32) for c in chans[1:-1])
                              and chans[0] <= 8 and chans[-1] <= 32 and all(c <= 16 for c in chans[:-1]))
            # non-causal encoder: its output reads only the CLS rows, so the last block runs on those rows (block_fwd_cls);
            # I2T_FULL_LAST_BLOCK=1 keeps the full-row form (A/B runs)
            self.cls_only_last = ((not self.enc.causal) and self.enc.L >= 1 and os.environ.get('I2T_FULL_LAST_BLOCK') != '1'
                                  and self.enc.fam is None)
            self.patch = (ecfg.input.width // ecfg.num_patches, ecfg.input.height // ecfg.num_patches)
            self.input_d = ecfg.n_channels * self.patch[0] * self.patch[1]
        self.dec_cross = [dcfg.transformer_config.is_cross_attn and not (dcfg.skip_alternate_cross_attn and l % 2)
                          for l in range(dcfg.n_layer)]
        # fused cross-attention forward (K/V projection + attention in one launch) when the shapes allow; I2T_XATTN_FUSED=0: A/B runs
        self.xattn_fused = os.environ.get('I2T_XATTN_FUSED') != '0'
        self._logits_cache: Dict[int, torch.Tensor] = {}
        self._ws = None
        self.grad_ready_hooks = []      # callables(which: 'begin' | 'decoder' | 'encoder'), e.g. the data-parallel exchange
        # mixed into every training forward's dropout seed: a second model stepped at the same cadence (the momentum twin,
        # reference wrapper.py:68-71,197-198: forward_m draws fresh torch RNG) must not repeat this model's masks
        self.seed_salt = 0
        # fp8 (e4m3) operands for the GEMMs of FROZEN decoder weights (engine_llama._lin; BASELINE.json configs[4]); off unless I2T_FP8=1
        self.fp8 = os.environ.get('I2T_FP8', '0') not in ('', '0') or bool(getattr(model.decoder, 'fp8_request', False))
        self.fp8_fuse = os.environ.get('I2T_FP8_FUSE', '1') != '0'      # producers emit the e4m3 operand themselves (engine_llama, csrc/fp8.hip)
        # a frozen PretrainedViT backbone on e4m3 operands is its OWN opt-in (I2T_FP8_VIT=1): `fp8` above covers decoder weights only, as the
        # reference's 4-bit loading does (models/decoder.py:292-299 touches the decoder); 12 backbone layers of e4m3 GEMMs cost feature
        # fidelity (tests/test_fp8_gpu.py::test_frozen_vit_backbone_on_fp8_operands bounds it against oracle/vit.py)
        self.fp8_vit = os.environ.get('I2T_FP8_VIT', '0') not in ('', '0')

    def _refresh_sparse_sets(self):
        """(Re-)read the sparse layers' position sets when their buffers changed (they are part of the state dict: a checkpoint
        with other draws may be loaded after construction); everything derived from them is dropped."""
        sparse = [(t, p, n) for t, p, n, f in (('enc', self.ep, self.enc.L, self.enc.fam), ('dec', self.dp, self.dec.L, self.dec.fam))
                  if f is not None and f.sparse]
        if not sparse:
            return
        bufs = dict(self.model.named_buffers())
        # (storage address + version counter: load_state_dict bumps the version in place, .to(device) replaces the tensor)
        versions = tuple((b.data_ptr(), b._version) for b in (bufs[f'{p}transformer.h.{l}.{k}'] for _, p, n in sparse for l in range(n)
                                                               for k in ('input_mask_idx', 'input_mask_not_idx')))
        if versions == self._sparse_versions:
            return
        self._sparse_versions = versions
        for t, p, n in sparse:
            self._sparse_idx[t] = self._sparse_sets(self.model, p, n)
        self._sub_cache = {k: v for k, v in getattr(self, '_sub_cache', {}).items() if isinstance(k, tuple) and k and k[0] == 'posplan'}
        self.sparse_epoch += 1

    @staticmethod
    def _sparse_sets(model, prefix: str, n_layer: int):
        """Host copies of every layer's kept / skipped position sets (the persistent buffers of layers.py:557-558)."""
        bufs = dict(model.named_buffers())
        return [(bufs[f'{prefix}transformer.h.{l}.input_mask_idx'].cpu().numpy().astype(np.int64),
                 bufs[f'{prefix}transformer.h.{l}.input_mask_not_idx'].cpu().numpy().astype(np.int64)) for l in range(n_layer)]

    def notify_grads_ready(self, which: str):
        for hook in self.grad_ready_hooks:
            hook(which)

    # ------------------------------------------------------------------------------------------------ plumbing
    def prepare(self, training: bool):
        dev = next(self.model.parameters()).device
        if dev.type != 'cuda':
            raise I2TError('the image2text_amd hot path runs on the MI355X only: move the model to cuda (no CPU path)')
        if self.arena is None or self.arena.device != dev or not self.arena.valid():
            self.arena = ParamArena(self.model, dev)
            if self.vit:
                self.arena.skip_grad = self.vit_skip_grad()
            self._ws = torch.zeros(4, dtype=F32, device=dev)
            self._conv_ws_pool = {}
            self._conv_scratch = torch.empty(32 * 36 * 16, dtype=F32, device=dev)
            self._logits_cache.clear()
            self._moe_cache.clear()
            self._sub_cache.clear()
            self._lora_merge_list = []
        if ops.precise():
            if training:
                raise I2TError('I2T_PRECISE=1 is the parity mode of the inference forward: unset it to train')
            if getattr(self.arena.pbf, '_i2t_lo', None) is None:       # shadow cast before the mode was switched on: cast again, both terms
                self.arena._versions = None
        self.arena.refresh_shadow()
        self._refresh_sparse_sets()
        self.enc_drop = self.dec_drop = self.dec_drop_prompt = None
        if training:
            # one fresh 64-bit seed per training forward, derived from torch's seed (torch.manual_seed reproduces a run, and
            # a later torch.manual_seed restarts the sequence) and from the data-parallel rank: replicas seeded identically
            # must still draw different dropout masks on their different shards
            base = torch.initial_seed()
            if getattr(self, '_seed_base', None) != base:
                self._seed_base = base
                self._seed_state = (base ^ (0x9E3779B97F4A7C15 * _dp_rank())) & (2 ** 64 - 1)      # rank 0: torch's seed itself
            self._seed_state = (self._seed_state * 6364136223846793005 + 1442695040888963407) & (2 ** 64 - 1)
            step_seed = self._seed_state ^ self.seed_salt
            if self.enc.dropout > 0 or self.enc.attn_dropout > 0:
                self.enc_drop = DropPlan(step_seed, 0, self.enc.dropout, self.enc.attn_dropout,
                                         compact_layer=self.enc.L - 1 if self.cls_only_last else -1, live_rows=self.enc.ncls)
            p_lora = self.dec.lora.p if getattr(self.dec, 'lora', None) is not None else 0.0
            if self.dec.dropout > 0 or self.dec.attn_dropout > 0 or p_lora > 0:
                hf_sites = hasattr(self.model.decoder, 'hot_config')          # transformers' GPT-2: resid_dropout after crossattention.c_proj
                self.dec_drop = DropPlan(step_seed, 1, self.dec.dropout, self.dec.attn_dropout, xresid=hf_sites, p_lora=p_lora)
                self.dec_drop_prompt = DropPlan(step_seed, 2, self.dec.dropout, self.dec.attn_dropout, xresid=hf_sites, p_lora=p_lora)
        return self.arena

    @property
    def _conv_ws(self):
        """Workspace the convolution launchers repack a layer's weights into (bf16 MFMA operand image) before the conv kernel
        reads it.  One per HIP stream: concurrent decode lanes run their encoders on separate streams, and nothing orders one
        lane's repack against another lane's conv kernel, so a shared buffer would be a cross-stream write/read race."""
        key = torch.cuda.current_stream(self.arena.device).cuda_stream
        ws = self._conv_ws_pool.get(key)
        if ws is None:
            n = max(32 * 36 * 32, max(self.arena.entries[f'{n}.weight'][1] for n, _, _ in self.conv))
            ws = self._conv_ws_pool[key] = torch.empty(n, dtype=F32, device=self.arena.device)
        return ws

    def _empty(self, *shape, dtype=F32):
        return torch.empty(*shape, dtype=dtype, device=self.arena.device)

    # ------------------------------------------------------------------------------------------------ block
    def block_fwd(self, pfx: str, x, B, T, d, H, ff, causal, mem_bf, S, save: bool, plan: Optional[DropPlan] = None, layer: int = 0,
                  vl=None):
        """vl (packed variable-length rows): namespace(cu=int32[B+1] device, total=rows, ...) -- then x is [total, d], T is the
        maximum length and attention walks each sequence's own rows."""
        a = self.arena
        M = vl.total if vl is not None else B * T
        cu = vl.cu if vl is not None else None
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))      # attention operand views
        dr = {k: (plan.get(layer, k) if plan is not None else None) for k in ('qkv', 'sdpa', 'resid', 'xattn', 'mlp', 'xresid')}
        sv = SimpleNamespace(x=x, cross=False, dr=dr)
        ln1, m1, r1 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x, a.P(f'{pfx}.ln_1.weight'), a.P(f'{pfx}.ln_1.bias'), ln1, m1, r1, M, d)
        qkv = self._empty(M, 3 * d, dtype=BF16)
        ldrop = (lambda site: plan.get(layer, f'lora_{site}') if plan is not None else None)
        lo = {site: self._lora_site(layer, site) for site in ('attn_c_attn', 'xattn_c_attn', 'mlp_c_fc', 'mlp_c_proj')} \
            if (pfx.startswith(self.dp) and getattr(self.dec, 'lora', None) is not None) else {}
        sv.lo, sv.lo_drop = {}, {site: ldrop(site) for site in lo}
        if lo.get('attn_c_attn') is not None:      # LoRA: the adapter rides in the GEMM's K panel (engine_lora.py)
            sv.lo['attn_c_attn'] = self._lora_gemm(lo['attn_c_attn'], ln1, a.W(f'{pfx}.attn.c_attn.weight'), qkv, M, ldrop('attn_c_attn'), save,
                                                   bias=a.P(f'{pfx}.attn.c_attn.bias'), drop=dr['qkv'])
        else:
            ops.gemm(ln1, a.W(f'{pfx}.attn.c_attn.weight'), qkv, M, 3 * d, d, bias=a.P(f'{pfx}.attn.c_attn.bias'), drop=dr['qkv'])
        ao, lse = self._empty(M, d, dtype=BF16), self._empty(H * M)
        q3 = v3(qkv, 3 * d)
        ops.attention_fwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], v3(ao, d), lse, B, H, T, T, causal, drop=dr['sdpa'],
                          cu_q=cu, cu_k=cu, total_q=M)
        x1 = self._empty(M, d)
        ops.gemm(ao, a.W(f'{pfx}.attn.c_proj.weight'), x1, M, d, d, bias=a.P(f'{pfx}.attn.c_proj.bias'), residual=x,
                 drop=dr['resid'])
        sv.ln1, sv.m1, sv.r1, sv.qkv, sv.ao, sv.lse, sv.x1 = ln1, m1, r1, qkv, ao, lse, x1
        x2 = x1
        if mem_bf is not None:
            if f'{pfx}.cross_attn.in_proj_weight' not in a.entries:
                raise ValueError('Model not configured for cross attn inputs!!!')         # reference layers.py:598-599
            win, bin_ = a.W(f'{pfx}.cross_attn.in_proj_weight'), a.P(f'{pfx}.cross_attn.in_proj_bias')
            ln3, m3, r3 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
            ops.layernorm_fwd(x1, a.P(f'{pfx}.ln_3.weight'), a.P(f'{pfx}.ln_3.bias'), ln3, m3, r3, M, d)
            q = self._empty(M, d, dtype=BF16)
            ops.gemm(ln3, win[:d], q, M, d, d, bias=bin_[:d])
            kv = self._empty(B, S, 2 * d, dtype=BF16)
            co, lse_c = self._empty(M, d, dtype=BF16), self._empty(H * M)
            if lo.get('xattn_c_attn') is not None:      # adapted K/V projection: the un-fused form
                sv.lo['xattn_c_attn'] = self._lora_gemm(lo['xattn_c_attn'], mem_bf, win[d:], kv.view(B * S, 2 * d), B * S, ldrop('xattn_c_attn'),
                                                        save, bias=bin_[d:])
                ops.attention_fwd(v3(q, d), kv[..., :d], kv[..., d:], v3(co, d), lse_c, B, H, T, S, False, drop=dr['xattn'],
                                  cu_q=cu, total_q=M)
            elif self.xattn_fused and S == 64 and H % 2 == 0 and d == 64 * H and not ops.precise():
                # ONE launch: K/V projection GEMM whose waves run the attention of their (image, head) out of the accumulators
                # (K and V are written once for the backward pass and never read back here)
                ops.xattn_kv_fused(mem_bf, win[d:], bin_[d:], v3(q, d), kv, v3(co, d), lse_c, B, S, H, T, drop=dr['xattn'],
                                   cu_q=cu, total_q=M)
            else:
                ops.gemm(mem_bf, win[d:], kv.view(B * S, 2 * d), B * S, 2 * d, d, bias=bin_[d:])
                ops.attention_fwd(v3(q, d), kv[..., :d], kv[..., d:], v3(co, d), lse_c, B, H, T, S, False, drop=dr['xattn'],
                                  cu_q=cu, total_q=M)
            x2 = self._empty(M, d)
            ops.gemm(co, a.W(f'{pfx}.cross_attn.out_proj.weight'), x2, M, d, d,
                     bias=a.P(f'{pfx}.cross_attn.out_proj.bias'), residual=x1, drop=dr['xresid'])
            sv.cross, sv.ln3, sv.m3, sv.r3, sv.q, sv.kv, sv.co, sv.lse_c, sv.mem = True, ln3, m3, r3, q, kv, co, lse_c, mem_bf
        ln2, m2, r2 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x2, a.P(f'{pfx}.ln_2.weight'), a.P(f'{pfx}.ln_2.bias'), ln2, m2, r2, M, d)
        h = self._empty(M, ff, dtype=BF16)
        pre = self._empty(M, ff, dtype=BF16) if save else None
        if lo.get('mlp_c_fc') is not None:
            sv.lo['mlp_c_fc'] = self._lora_gemm(lo['mlp_c_fc'], ln2, a.W(f'{pfx}.mlp.c_fc.weight'), h, M, ldrop('mlp_c_fc'), save,
                                                bias=a.P(f'{pfx}.mlp.c_fc.bias'), act=1, aux_out=pre)
        else:
            # the second output of a step that will be differentiated is GELU'(pre), not pre: the backward GEMM's epilogue is then one
            # multiply per element (ops.ACT_MUL_AUX) instead of re-evaluating exp + rcp; the LoRA forms keep the pre-activation
            dout = save and GELU_KEEPS_DERIVATIVE and lo.get('mlp_c_proj') is None
            ops.gemm(ln2, a.W(f'{pfx}.mlp.c_fc.weight'), h, M, ff, d, bias=a.P(f'{pfx}.mlp.c_fc.bias'),
                     act=ops.ACT_GELU_DOUT if dout else 1, aux_out=pre)
            sv.pre_is_grad = dout
        x3 = self._empty(M, d)
        if lo.get('mlp_c_proj') is not None:
            sv.lo['mlp_c_proj'] = self._lora_gemm(lo['mlp_c_proj'], h, a.W(f'{pfx}.mlp.c_proj.weight'), x3, M, ldrop('mlp_c_proj'), save,
                                                  bias=a.P(f'{pfx}.mlp.c_proj.bias'), residual=x2, drop=dr['mlp'])
        else:
            ops.gemm(h, a.W(f'{pfx}.mlp.c_proj.weight'), x3, M, d, ff, bias=a.P(f'{pfx}.mlp.c_proj.bias'), residual=x2,
                     drop=dr['mlp'])
        sv.x2, sv.ln2, sv.m2, sv.r2, sv.h, sv.pre = x2, ln2, m2, r2, h, pre
        sv.layer = layer
        return x3, (sv if save else None)

    def _linear_bwd(self, dyb, M, N, K, x_bf, wname: str, bname: Optional[str], dx_out=None, dy_sumsq=None, **dx_kw):
        """y = x W^T + b with y [M,N], x [M,K], W [N,K]: accumulates dW, db; returns/fills dX when requested.
        dy_sumsq (1-float device tensor): dyb is an UN-normalised gradient whose normaliser 1 / (sqrt(dy_sumsq) + 1e-6) the three
        consumers apply themselves (ops.gemm alpha_sumsq): no pass over dyb exists just to rescale it."""
        a = self.arena
        gb = a.Gt(bname) if bname else None
        if gb is not None:
            ops.colsum(dyb, gb, M, N, accumulate=True, alpha_sumsq=dy_sumsq)
        if a.trainable(wname):
            ops.gemm(dyb, x_bf, a.G(wname), N, K, M, a_kmajor=True, b_kmajor=True, accumulate=True, alpha_sumsq=dy_sumsq)
        if dx_out is not None:
            ops.gemm(dyb, a.W(wname), dx_out, M, K, N, b_kmajor=True, alpha_sumsq=dy_sumsq, **dx_kw)
        return dx_out

    def block_bwd(self, pfx: str, sv, dx, dxb, B, T, d, H, ff, causal, S, dmem, emit_last_bf16: bool, vl=None, sumsq_out=None,
                  dx_pre=None, dxb_sumsq=None, last_bf16_drop=None, dx_mask=None):
        """dx (fp32) / dxb (bf16 copy): gradient w.r.t. the block output; dxb already normalised, dx too unless dx_pre (1 float:
        sum(dx^2)) is given -- then the first LayerNorm backward that accumulates onto dx applies 1 / (||dx|| + 1e-6) on the fly
        (the normaliser's fp32 rescale pass is not run).  On return dx (and dxb when emit_last_bf16) hold the gradient w.r.t. the
        block input."""
        a = self.arena
        M = vl.total if vl is not None else B * T
        cu = vl.cu if vl is not None else None
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        ws = self._empty(H * M)
        dr = sv.dr
        # ---- MLP: x3 = x2 + drop(c_proj(gelu(c_fc(ln_2 x2)))).  dxb arrives already masked with the MLP dropout (the
        # producer of the bf16 copy applies it: the branch sees the masked gradient, dx -- the residual path -- does not)
        dpre = self._empty(M, ff, dtype=BF16)
        svlo = getattr(sv, 'lo', None) or {}
        lsite = (lambda site: self._lora_site(sv.layer, site))
        ldrop = (lambda site: sv.lo_drop.get(site))          # the adapter's input-dropout mask of the forward pass
        if svlo.get('mlp_c_proj') is not None:        # LoRA (engine_lora.py): fp32 dh = dY . W + dropout(du . A), then the GELU derivative
            dh32 = self._lora_bwd(lsite('mlp_c_proj'), svlo['mlp_c_proj'], dxb, sv.h, a.W(f'{pfx}.mlp.c_proj.weight'),
                                  a.Gt(f'{pfx}.mlp.c_proj.weight'), a.Gt(f'{pfx}.mlp.c_proj.bias'), M, ldrop('mlp_c_proj'))
            ops.dgelu_mul(dh32, sv.pre, dpre)
        else:
            # dxb_sumsq: dxb is the block-output gradient as the layer above left it -- masked, NOT normalised; mlp.c_proj's three
            # backward launches apply 1 / (||dx|| + 1e-6) in their epilogues (no grad_normalize pass ran for this block)
            self._linear_bwd(dxb, M, d, ff, sv.h, f'{pfx}.mlp.c_proj.weight', f'{pfx}.mlp.c_proj.bias' if a.G(f'{pfx}.mlp.c_proj.bias') is not None else None,
                             dx_out=dpre, act=ops.ACT_MUL_AUX if getattr(sv, 'pre_is_grad', False) else 2, aux_in=sv.pre, dy_sumsq=dxb_sumsq)
        dln = self._empty(M, d, dtype=BF16)
        dln2 = dln
        if svlo.get('mlp_c_fc') is not None:
            dln2 = self._lora_bwd(lsite('mlp_c_fc'), svlo['mlp_c_fc'], dpre, sv.ln2, a.W(f'{pfx}.mlp.c_fc.weight'),
                                  a.Gt(f'{pfx}.mlp.c_fc.weight'), a.Gt(f'{pfx}.mlp.c_fc.bias'), M, ldrop('mlp_c_fc'))
        else:
            self._linear_bwd(dpre, M, ff, d, sv.ln2, f'{pfx}.mlp.c_fc.weight', f'{pfx}.mlp.c_fc.bias' if a.G(f'{pfx}.mlp.c_fc.bias') is not None else None,
                             dx_out=dln)
        ops.layernorm_bwd(dln2, sv.x2, a.P(f'{pfx}.ln_2.weight'), sv.m2, sv.r2, dx, a.Gt(f'{pfx}.ln_2.weight'),
                          a.Gt(f'{pfx}.ln_2.bias'), M, d, dx_accumulate=True, dx_bf16=dxb,
                          bf16_drop=dr.get('xresid') if sv.cross else dr['resid'],      # next consumer of dxb: the (cross-)attention output projection's backward
                          dx_pre_sumsq=dx_pre)                              # first fp32 use of dx in the block
        # ---- cross attention: x2 = x1 + out_proj(attn(q(ln_3 x1), kv(mem)))
        if sv.cross:
            win = a.W(f'{pfx}.cross_attn.in_proj_weight')
            gin, gbin = a.G(f'{pfx}.cross_attn.in_proj_weight'), a.G(f'{pfx}.cross_attn.in_proj_bias')
            dco = self._empty(M, d, dtype=BF16)
            self._linear_bwd(dxb, M, d, d, sv.co, f'{pfx}.cross_attn.out_proj.weight', f'{pfx}.cross_attn.out_proj.bias',
                             dx_out=dco)
            dq, dkv = self._empty(M, d, dtype=BF16), self._empty(B, S, 2 * d, dtype=BF16)
            ops.attention_bwd(v3(sv.q, d), sv.kv[..., :d], sv.kv[..., d:], v3(sv.co, d), v3(dco, d), sv.lse_c, ws, v3(dq, d),
                              dkv[..., :d], dkv[..., d:], B, H, T, S, False, drop=dr['xattn'], cu_q=cu, total_q=M)
            dqf, dkvf = dq, dkv.view(B * S, 2 * d)
            train_in, train_inb = a.trainable(f'{pfx}.cross_attn.in_proj_weight'), a.trainable(f'{pfx}.cross_attn.in_proj_bias')
            if train_inb:
                ops.colsum(dqf, gbin[:d], M, d, accumulate=True)
            if train_in:
                ops.gemm(dqf, sv.ln3, gin[:d], d, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            ops.gemm(dqf, win[:d], dln, M, d, d, b_kmajor=True)
            if svlo.get('xattn_c_attn') is not None:
                dmem32 = self._lora_bwd(lsite('xattn_c_attn'), svlo['xattn_c_attn'], dkvf, sv.mem, win[d:], gin[d:] if train_in else None,
                                        gbin[d:] if train_inb else None, B * S, ldrop('xattn_c_attn'))
                ops.add_(dmem, dmem32)
            else:
                if train_inb:
                    ops.colsum(dkvf, gbin[d:], B * S, 2 * d, accumulate=True)
                if train_in:
                    ops.gemm(dkvf, sv.mem, gin[d:], 2 * d, d, B * S, a_kmajor=True, b_kmajor=True, accumulate=True)
                ops.gemm(dkvf, win[d:], dmem, B * S, d, 2 * d, b_kmajor=True, accumulate=True)
            ops.layernorm_bwd(dln, sv.x1, a.P(f'{pfx}.ln_3.weight'), sv.m3, sv.r3, dx, a.Gt(f'{pfx}.ln_3.weight'),
                              a.Gt(f'{pfx}.ln_3.bias'), M, d, dx_accumulate=True, dx_bf16=dxb, bf16_drop=dr['resid'])
        # ---- self attention: x1 = x + drop(c_proj(attn(mult * c_attn(ln_1 x)))); dxb carries the resid-dropout mask
        dao = self._empty(M, d, dtype=BF16)
        self._linear_bwd(dxb, M, d, d, sv.ao, f'{pfx}.attn.c_proj.weight',
                         f'{pfx}.attn.c_proj.bias' if a.G(f'{pfx}.attn.c_proj.bias') is not None else None, dx_out=dao)
        dqkv = self._empty(M, 3 * d, dtype=BF16)
        q3, g3 = v3(sv.qkv, 3 * d), v3(dqkv, 3 * d)
        ops.attention_bwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], v3(sv.ao, d), v3(dao, d), sv.lse, ws, g3[..., :d],
                          g3[..., d:2 * d], g3[..., 2 * d:], B, H, T, T, causal, drop=dr['sdpa'], cu_q=cu, cu_k=cu, total_q=M,
                          out_drop=dr['qkv'])                  # gradient w.r.t. the un-multiplied q/k/v
        dln1 = dln
        if svlo.get('attn_c_attn') is not None:
            dln1 = self._lora_bwd(lsite('attn_c_attn'), svlo['attn_c_attn'], dqkv, sv.ln1, a.W(f'{pfx}.attn.c_attn.weight'),
                                  a.Gt(f'{pfx}.attn.c_attn.weight'), a.Gt(f'{pfx}.attn.c_attn.bias'), M, ldrop('attn_c_attn'))
        else:
            self._linear_bwd(dqkv, M, 3 * d, d, sv.ln1, f'{pfx}.attn.c_attn.weight',
                             f'{pfx}.attn.c_attn.bias' if a.G(f'{pfx}.attn.c_attn.bias') is not None else None, dx_out=dln)
        # last writer of dx in this block: it also leaves sum(dx^2) for the next block's gradient normaliser
        # (emit_last_bf16 with last_bf16_drop: the bf16 copy is the NEXT block's incoming gradient, already masked for ITS mlp.c_proj)
        ops.layernorm_bwd(dln1, sv.x, a.P(f'{pfx}.ln_1.weight'), sv.m1, sv.r1, dx, a.Gt(f'{pfx}.ln_1.weight'),
                          a.Gt(f'{pfx}.ln_1.bias'), M, d, dx_accumulate=True, dx_bf16=dxb if emit_last_bf16 else None,
                          bf16_drop=last_bf16_drop if emit_last_bf16 else None, sumsq_out=sumsq_out,
                          dx_mask=dx_mask)      # (the tower's lowest block: the embedding dropout's backward, applied to the dx it stores)

    # ---- the encoder's LAST block, CLS rows only.  The encoder output is ln_f of the first ncls rows (encoder.py:172-173); the
    # patch rows of the last block feed nothing, forward or backward.  K and V still come from every row; the queries, the
    # attention output projection and the whole MLP run on B*ncls rows instead of B*T (ncls = 64 of T = 260 in nano-224).
    def block_fwd_cls(self, pfx: str, x, B, T, d, H, ff, ncls, save: bool, plan: Optional[DropPlan], layer: int):
        a = self.arena
        M, Mc = B * T, B * ncls
        dr = {k: (plan.get(layer, k) if plan is not None else None) for k in ('qkv', 'sdpa', 'resid', 'mlp')}
        ln1, m1, r1 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x, a.P(f'{pfx}.ln_1.weight'), a.P(f'{pfx}.ln_1.bias'), ln1, m1, r1, M, d)
        qkv = self._empty(M, 3 * d, dtype=BF16)
        ops.gemm(ln1, a.W(f'{pfx}.attn.c_attn.weight'), qkv, M, 3 * d, d, bias=a.P(f'{pfx}.attn.c_attn.bias'), drop=dr['qkv'])
        q3 = qkv.view(B, T, 3 * d)
        ao, lse = self._empty(B, ncls, d, dtype=BF16), self._empty(H * Mc)
        ops.attention_fwd(q3[:, :ncls, :d], q3[..., d:2 * d], q3[..., 2 * d:], ao, lse, B, H, ncls, T, False, drop=dr['sdpa'])
        xc = self._empty(Mc, d)
        ops.copy_rows(x, T * d, xc, ncls * d, B, ncls, d)
        x1 = self._empty(Mc, d)
        ops.gemm(ao.view(Mc, d), a.W(f'{pfx}.attn.c_proj.weight'), x1, Mc, d, d, bias=a.P(f'{pfx}.attn.c_proj.bias'), residual=xc,
                 drop=dr['resid'])
        ln2, m2, r2 = self._empty(Mc, d, dtype=BF16), self._empty(Mc), self._empty(Mc)
        ops.layernorm_fwd(x1, a.P(f'{pfx}.ln_2.weight'), a.P(f'{pfx}.ln_2.bias'), ln2, m2, r2, Mc, d)
        h = self._empty(Mc, ff, dtype=BF16)
        pre = self._empty(Mc, ff, dtype=BF16) if save else None
        dout = save and GELU_KEEPS_DERIVATIVE
        ops.gemm(ln2, a.W(f'{pfx}.mlp.c_fc.weight'), h, Mc, ff, d, bias=a.P(f'{pfx}.mlp.c_fc.bias'), act=ops.ACT_GELU_DOUT if dout else 1, aux_out=pre)
        x3 = self._empty(Mc, d)
        ops.gemm(h, a.W(f'{pfx}.mlp.c_proj.weight'), x3, Mc, d, ff, bias=a.P(f'{pfx}.mlp.c_proj.bias'), residual=x1, drop=dr['mlp'])
        sv = SimpleNamespace(x=x, dr=dr, ln1=ln1, m1=m1, r1=r1, qkv=qkv, ao=ao, lse=lse, x1=x1, ln2=ln2, m2=m2, r2=r2, h=h, pre=pre, pre_is_grad=dout)
        return x3, (sv if save else None)

    def block_bwd_cls(self, pfx: str, sv, dcls, dx_full, B, T, d, H, ff, ncls):
        """dcls fp32 [B*ncls, d]: gradient w.r.t. the block's CLS-row output.  dx_full fp32 [B, T, d] (its contents on entry do not matter): receives the
        gradient w.r.t. the block input (every row: K and V saw them all)."""
        a = self.arena
        M, Mc = B * T, B * ncls
        dr = sv.dr
        bias = lambda n: n if a.G(n) is not None else None
        dxb = self._empty(Mc, d, dtype=BF16)
        # (the patch rows' gradient is zero: same norm as over the full block output); ws[1] collects sum(dx_full^2) below
        ops.grad_normalize(dcls, self._ws[0:1], dxb, bf16_drop=dr['mlp'], clear_after=self._ws[1:2])
        dpre = self._empty(Mc, ff, dtype=BF16)
        self._linear_bwd(dxb, Mc, d, ff, sv.h, f'{pfx}.mlp.c_proj.weight', bias(f'{pfx}.mlp.c_proj.bias'), dx_out=dpre,
                         act=ops.ACT_MUL_AUX if sv.pre_is_grad else 2, aux_in=sv.pre)
        dln = self._empty(Mc, d, dtype=BF16)
        self._linear_bwd(dpre, Mc, ff, d, sv.ln2, f'{pfx}.mlp.c_fc.weight', bias(f'{pfx}.mlp.c_fc.bias'), dx_out=dln)
        ops.layernorm_bwd(dln, sv.x1, a.P(f'{pfx}.ln_2.weight'), sv.m2, sv.r2, dcls, a.G(f'{pfx}.ln_2.weight'),
                          a.G(f'{pfx}.ln_2.bias'), Mc, d, dx_accumulate=True, dx_bf16=dxb, bf16_drop=dr['resid'])
        dao = self._empty(B, ncls, d, dtype=BF16)
        self._linear_bwd(dxb, Mc, d, d, sv.ao.view(Mc, d), f'{pfx}.attn.c_proj.weight', bias(f'{pfx}.attn.c_proj.bias'),
                         dx_out=dao.view(Mc, d))
        # the forward GEMM's per-token q|k|v multipliers: applied by the attention backward on its way out where the resident-operand
        # kernel runs (it knows that the queries are the first ncls rows of T-row sequences), by a pass over dqkv elsewhere
        fused_mask = dr['qkv'] is not None and ops.attention_bwd_takes_q_seq(ncls, T, dr['sdpa'])
        dqkv = self._empty(B, T, 3 * d, dtype=BF16)
        dqkv[:, ncls:, :d].zero_()                                              # dq of the patch rows: zero (no query ran there); the rest is written below
        q3 = sv.qkv.view(B, T, 3 * d)
        ops.attention_bwd(q3[:, :ncls, :d], q3[..., d:2 * d], q3[..., 2 * d:], sv.ao, dao, sv.lse, self._empty(H * Mc),
                          dqkv[:, :ncls, :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:], B, H, ncls, T, False, drop=dr['sdpa'],
                          out_drop=dr['qkv'] if fused_mask else None, out_drop_q_seq=T)
        if dr['qkv'] is not None and not fused_mask:
            ops.dropout_apply(dqkv, M, 3 * d, dr['qkv'])                         # full-row index space, as in the forward GEMM
        dln1 = self._empty(M, d, dtype=BF16)
        self._linear_bwd(dqkv.view(M, 3 * d), M, 3 * d, d, sv.ln1, f'{pfx}.attn.c_attn.weight', bias(f'{pfx}.attn.c_attn.bias'),
                         dx_out=dln1)
        ops.copy_rows(dcls, ncls * d, dx_full, T * d, B, ncls, d)               # residual path of the CLS rows
        ops.layernorm_bwd(dln1, sv.x, a.P(f'{pfx}.ln_1.weight'), sv.m1, sv.r1, dx_full.view(M, d), a.G(f'{pfx}.ln_1.weight'),
                          a.G(f'{pfx}.ln_1.bias'), M, d, dx_accumulate=True, sumsq_out=self._ws[1:2],
                          acc_period=T, acc_rows=ncls)      # only the CLS rows of dx_full hold a value: the patch rows are written, not added onto

    def _blocks_bwd(self, prefix: str, saves: List, dx, B, T, d, H, ff, causal, S, dmem, vl=None, presummed_slot=None,
                    normalize: bool = True, lowest_dx_mask=None):
        """presummed_slot: index into self._ws that already holds sum(dx^2) of the incoming gradient (None: reduce it here).
        Inside the loop every block's last LayerNorm backward leaves that sum for the block below (two alternating floats).
        normalize=False (Hugging Face GPT-2 blocks have no normalize_gradients): the same launches with the 'sum' pinned to the
        value whose normaliser 1 / (sqrt(s) + 1e-6) is 1."""
        dxb = self._empty(dx.shape[0], d, dtype=BF16)
        slot, presummed = (presummed_slot, True) if presummed_slot is not None else (0, False)
        if not normalize:
            if getattr(self, '_unit_sq', None) is None or self._unit_sq.device != dx.device:
                self._unit_sq = torch.full((1,), (1.0 - 1e-6) ** 2, dtype=F32, device=dx.device)
            for l in reversed(range(len(saves))):
                ops.grad_normalize(dx, self._unit_sq, dxb, bf16_drop=saves[l].dr['mlp'], presummed=True, keep_f32=True)
                self.block_bwd(f'{prefix}transformer.h.{l}', saves[l], dx, dxb, B, T, d, H, ff, causal, S, dmem, emit_last_bf16=False,
                               vl=vl, dx_pre=self._unit_sq, dx_mask=lowest_dx_mask if l == 0 else None)
            return
        fold = NORMALISER_FOLDED and not any(getattr(sv, 'lo', None) for sv in saves)      # (LoRA's mlp.c_proj backward reads a normalised dxb)
        folded = False          # dxb already holds this block's incoming gradient (raw, masked): written by the block above
        for l in reversed(range(len(saves))):
            cur, nxt = self._ws[slot:slot + 1], self._ws[1 - slot:2 - slot]
            if folded:
                nxt.zero_()                                    # the accumulator this block's last LayerNorm backward adds into
            else:
                # normalize_gradients at the block output; the bf16 copy feeds mlp.c_proj's backward -> carries the MLP mask
                # (fp32 dx stays un-normalised here: the block's first LayerNorm backward rescales it while accumulating)
                ops.grad_normalize(dx, cur, dxb, bf16_drop=saves[l].dr['mlp'], presummed=presummed, clear_after=nxt, keep_f32=True)
            # every block but the lowest hands the one below its gradient as an UN-normalised bf16 copy (written by its last LayerNorm
            # backward, which also leaves sum(dx^2)): that block's mlp.c_proj backward applies the normaliser in its GEMM epilogues
            hand_down = fold and l > 0
            self.block_bwd(f'{prefix}transformer.h.{l}', saves[l], dx, dxb, B, T, d, H, ff, causal, S, dmem, emit_last_bf16=hand_down,
                           vl=vl, sumsq_out=nxt, dx_pre=cur, dxb_sumsq=cur if folded else None,
                           last_bf16_drop=saves[l - 1].dr['mlp'] if hand_down else None, dx_mask=lowest_dx_mask if l == 0 else None)
            slot, presummed, folded = 1 - slot, True, hand_down

    # ------------------------------------------------------------------------------------------------ encoder
    def encode(self, images: torch.Tensor, save: bool):
        if self.vit:
            return self.vit_encode(images, save)
        a, e = self.arena, self.enc
        images = images.to(device=a.device, dtype=F32).contiguous()
        B, C, Hh, Ww = images.shape
        acts, cur = [], images
        for i, (nm, cin, cout) in enumerate(self.conv):
            last = i == len(self.conv) - 1
            if self.conv_mfma:      # intermediates channels-last (NHWC); the last layer writes NCHW = the flat patches
                y = self._empty(*((B, cout, Hh, Ww) if last else (B, Hh, Ww, cout)), dtype=BF16)
                ops.conv6_fwd(cur, ops.LAYOUT_NCHW_F32 if i == 0 else ops.LAYOUT_NHWC_BF16, i > 0, a.P(f'{nm}.weight'),
                              a.P(f'{nm}.bias'), y, last, self._conv_ws, B, cin, cout, Hh, Ww)
            else:
                y = self._empty(B, cout, Hh, Ww, dtype=BF16)
                ops.conv_fwd(cur, i > 0, a.P(f'{nm}.weight'), a.P(f'{nm}.bias'), y, self._conv_ws, B, cin, cout, Hh, Ww, e.k)
            acts.append(y)
            cur = y
        M0, d, T = B * e.P2, e.d, e.ncls + e.P2
        flat = cur.view(M0, self.input_d)                       # encoder.py:166 flat reshape: same memory
        proj = self._empty(M0, d)
        ops.gemm(flat, a.W(f'{self.ep}projector.weight'), proj, M0, d, self.input_d, bias=a.P(f'{self.ep}projector.bias'))
        g, bta = a.P(f'{self.ep}ln_input.weight'), a.P(f'{self.ep}ln_input.bias')
        wpe = a.P(f'{self.ep}transformer.wpe.weight')
        y1, st1, st2 = self._empty(B, e.P2, d), self._empty(B, ops.LNND_STATS_STRIDE), self._empty(B, ops.LNND_STATS_STRIDE)
        ops.layernorm_nd_fwd(proj, None, g, bta, y1, e.P2 * d, st1, B, e.P2, d)
        x = self._empty(B, T, d)
        plan = self.enc_drop
        emb_drop = plan.get(0, 'emb') if plan is not None else None
        # the embedding dropout (encoder.py:170) is applied by the two producers of x while they write it (elementwise mask over [B, T, d])
        fuse_emb = emb_drop is not None and EMB_DROP_FUSED and int(emb_drop[0]) == 1 and B * T * d < 2 ** 32
        ops.layernorm_nd_fwd(y1, wpe, g, bta, x[:, e.ncls:], T * d, st2, B, e.P2, d, drop=emb_drop if fuse_emb else None, drop_base=e.ncls * d)
        ops.bcast_rows(a.P(f'{self.ep}cls_token'), x, T * d, B, e.ncls, d, drop=emb_drop if fuse_emb else None)
        if emb_drop is not None and not fuse_emb:
            ops.dropout_apply(x, B * T, d, emb_drop)
        saves, cur_x, perm = [], x.view(B * T, d), None
        for l in range(e.L - 1 if self.cls_only_last else e.L):
            if e.fam is not None:
                cur_x, sv, perm = self.fam_layer_fwd(f'{self.ep}transformer.h.{l}', e.fam, cur_x, perm, B, T, None, 0, save, plan, l, None,
                                                     self.sparse_subset('enc', l, B, T, 0, None))
            else:
                cur_x, sv = self.block_fwd(f'{self.ep}transformer.h.{l}', cur_x, B, T, d, e.H, e.ff, e.causal, None, 0, save, plan, l)
            saves.append(sv)
        Mc = B * e.ncls
        if self.cls_only_last:      # the last block's patch rows are never read: compute its CLS rows only
            cls, sv = self.block_fwd_cls(f'{self.ep}transformer.h.{e.L - 1}', cur_x, B, T, d, e.H, e.ff, e.ncls, save, plan, e.L - 1)
            saves.append(sv)
        elif perm is not None:      # the stream is stored as the last sparse layer's pieces: fetch the CLS rows through its index
            cls = self._empty(Mc, d)
            ops.gather_rows(cur_x, perm.view(B, T)[:, :e.ncls].reshape(-1).contiguous(), Mc, d, out_f32=cls)
        else:
            cls = self._empty(Mc, d)
            ops.copy_rows(cur_x, T * d, cls, e.ncls * d, B, e.ncls, d)
        mf, rf = self._empty(Mc), self._empty(Mc)
        gf, bf = a.P(f'{self.ep}transformer.ln_f.weight'), a.P(f'{self.ep}transformer.ln_f.bias')
        if self.has_bridge:
            lnf = self._empty(Mc, d, dtype=BF16)
            ops.layernorm_fwd(cls, gf, bf, lnf, mf, rf, Mc, d)
            enc_out = self._empty(Mc, self.dec.d)
            ops.gemm(lnf, a.W('encoder.1.weight'), enc_out, Mc, self.dec.d, d)
        else:
            lnf = None
            enc_out = self._empty(Mc, d)
            ops.layernorm_fwd(cls, gf, bf, enc_out, mf, rf, Mc, d)
        ctx = None
        if save:
            ctx = SimpleNamespace(images=images, acts=acts, flat=flat, proj=proj, y1=y1, st1=st1, st2=st2, saves=saves,
                                  cls=cls, mf=mf, rf=rf, lnf=lnf, B=B, Hh=Hh, Ww=Ww, emb_drop=emb_drop)
        return enc_out.view(B, e.ncls, -1), ctx

    def encode_backward(self, ctx, denc: torch.Tensor):
        """denc: fp32 [B*ncls, d_out] gradient w.r.t. the encoder output (bridge output when there is one)."""
        if self.vit:
            return self.vit_encode_backward(ctx, denc)
        a, e = self.arena, self.enc
        B, d, T, Mc = ctx.B, e.d, e.ncls + e.P2, ctx.B * e.ncls
        gf = a.P(f'{self.ep}transformer.ln_f.weight')
        dcls = self._empty(Mc, d)
        if self.has_bridge:
            dencb = self._empty(Mc, self.dec.d, dtype=BF16)
            ops.cast_f32_bf16(denc, dencb)
            dlnf = self._empty(Mc, d, dtype=BF16)
            self._linear_bwd(dencb, Mc, self.dec.d, d, ctx.lnf, 'encoder.1.weight', None, dx_out=dlnf)
            ops.layernorm_bwd(dlnf, ctx.cls, gf, ctx.mf, ctx.rf, dcls, a.G(f'{self.ep}transformer.ln_f.weight'),
                              a.G(f'{self.ep}transformer.ln_f.bias'), Mc, d)
        else:
            ops.layernorm_bwd(denc, ctx.cls, gf, ctx.mf, ctx.rf, dcls, a.G(f'{self.ep}transformer.ln_f.weight'),
                              a.G(f'{self.ep}transformer.ln_f.bias'), Mc, d)
        dx = torch.empty(B, T, d, dtype=F32, device=a.device) if self.cls_only_last else torch.zeros(B, T, d, dtype=F32, device=a.device)
        # the embedding dropout's backward: a mask on the gradient the lowest block hands down -- applied by that block's last LayerNorm
        # backward while it stores dx (dense nanoGPT blocks), by a pass over dx elsewhere
        emb_mask = ctx.emb_drop if (ctx.emb_drop is not None and EMB_DROP_FUSED and int(ctx.emb_drop[0]) == 1 and B * T * d < 2 ** 32) else None
        masked = False
        if self.cls_only_last:
            self.block_bwd_cls(f'{self.ep}transformer.h.{e.L - 1}', ctx.saves[-1], dcls, dx, B, T, d, e.H, e.ff, e.ncls)
            masked = emb_mask is not None and e.L > 1
            self._blocks_bwd(self.ep, ctx.saves[:-1], dx.view(B * T, d), B, T, d, e.H, e.ff, e.causal, 0, None, presummed_slot=1,
                             lowest_dx_mask=emb_mask if masked else None)
        elif e.fam is not None:
            ops.copy_rows(dcls, e.ncls * d, dx, T * d, B, e.ncls, d)
            dxf, dperm = dx.view(B * T, d), None
            for l in reversed(range(e.L)):
                dxf, dperm = self.fam_layer_bwd(f'{self.ep}transformer.h.{l}', e.fam, ctx.saves[l], dxf, dperm, B, T, 0, None, None)
            dx = self.materialize(dxf, dperm).view(B, T, d)
        else:
            ops.copy_rows(dcls, e.ncls * d, dx, T * d, B, e.ncls, d)
            masked = emb_mask is not None and e.L > 0
            self._blocks_bwd(self.ep, ctx.saves, dx.view(B * T, d), B, T, d, e.H, e.ff, e.causal, 0, None, lowest_dx_mask=emb_mask if masked else None)
        if ctx.emb_drop is not None and not masked:
            ops.dropout_apply(dx, B * T, d, ctx.emb_drop)
        ops.sum_over_batch(dx, T * d, a.G(f'{self.ep}cls_token'), B, e.ncls, d, accumulate=True)
        g = a.P(f'{self.ep}ln_input.weight')
        gg, gb = a.G(f'{self.ep}ln_input.weight'), a.G(f'{self.ep}ln_input.bias')
        dy1, dproj = self._empty(B, e.P2, d), self._empty(B, e.P2, d)
        ops.layernorm_nd_bwd(dx[:, e.ncls:], T * d, ctx.y1, a.P(f'{self.ep}transformer.wpe.weight'), g, ctx.st2, dy1, gg, gb,
                             a.G(f'{self.ep}transformer.wpe.weight'), B, e.P2, d)
        ops.layernorm_nd_bwd(dy1, e.P2 * d, ctx.proj, None, g, ctx.st1, dproj, gg, gb, None, B, e.P2, d)
        M0 = B * e.P2
        dpb = self._empty(M0, d, dtype=BF16)
        ops.cast_f32_bf16(dproj, dpb)
        dflat = self._empty(M0, self.input_d, dtype=BF16)
        self._linear_bwd(dpb, M0, d, self.input_d, ctx.flat, f'{self.ep}projector.weight',
                         f'{self.ep}projector.bias' if a.G(f'{self.ep}projector.bias') is not None else None, dx_out=dflat)
        dy = dflat.view(B, self.conv[-1][2], ctx.Hh, ctx.Ww)
        dy_layout = ops.LAYOUT_NCHW_BF16
        if self.conv_mfma and self.conv[-1][2] % 8 == 0:       # one coalesced transpose beats two 2-byte-scatter stagings
            dyt = self._empty(B, ctx.Hh, ctx.Ww, self.conv[-1][2], dtype=BF16)
            ops.nchw_to_nhwc(dy, dyt, B, self.conv[-1][2], ctx.Hh, ctx.Ww)
            dy, dy_layout = dyt, ops.LAYOUT_NHWC_BF16
        for i in reversed(range(len(self.conv))):
            nm, cin, cout = self.conv[i]
            xin = ctx.images if i == 0 else ctx.acts[i - 1]
            if self.conv_mfma:
                ops.conv6_bwd_weight(dy, dy_layout, xin, ops.LAYOUT_NCHW_F32 if i == 0 else ops.LAYOUT_NHWC_BF16, i > 0,
                                     a.G(f'{nm}.weight'), a.G(f'{nm}.bias'), self._conv_scratch, B, cin, cout, ctx.Hh, ctx.Ww)
                if i > 0:
                    dxi = self._empty(B, ctx.Hh, ctx.Ww, cin, dtype=BF16)
                    ops.conv6_bwd_data(dy, dy_layout, a.P(f'{nm}.weight'), ctx.acts[i - 1], dxi, self._conv_ws, B, cin, cout,
                                       ctx.Hh, ctx.Ww)
                    dy, dy_layout = dxi, ops.LAYOUT_NHWC_BF16
                continue
            ops.conv_bwd_weight(dy, xin, i > 0, a.G(f'{nm}.weight'), a.G(f'{nm}.bias'), B, cin, cout, ctx.Hh, ctx.Ww, e.k)
            if i > 0:
                dxi = self._empty(B, cin, ctx.Hh, ctx.Ww, dtype=BF16)
                ops.conv_bwd_data(dy, a.P(f'{nm}.weight'), ctx.acts[i - 1], True, dxi, self._conv_ws, B, cin, cout, ctx.Hh, ctx.Ww, e.k)
                dy = dxi

    # ------------------------------------------------------------------------------------------------ decoder
    def _mem_bf16(self, enc_out: torch.Tensor):
        mem = self._empty(enc_out.shape[0] * enc_out.shape[1], enc_out.shape[2], dtype=BF16)
        ops.cast_f32_bf16(enc_out.contiguous(), mem)
        return mem

    def decode_segment(self, B: int, T: int, mem_bf, S: int, save: bool, ids=None, embeds=None, pos_offset: int = 0, vl=None,
                       dropout_without_save: bool = False, drop_plan=None, split: int = 0):
        """One causal segment through the decoder blocks + ln_f.  Returns (hidden fp32 [M,d], hidden bf16, ctx), M = B*T, or
        M = vl.total for packed variable-length rows (vl = namespace(cu, pos, total): ids is then the packed 1-D id list)."""
        a, dc = self.arena, self.dec
        if T + pos_offset > dc.block:
            raise AssertionError(f'Cannot forward sequence of length {T + pos_offset}, block size is only {dc.block}')
        if dc.llama is not None:
            return self.llama_decode_fwd(B, T, save, ids, embeds, pos_offset, vl)
        if split and (save or vl is not None or dc.fam is None):
            raise I2TError('split visibility is a forward-only feature of the grouped attention kernels on dense rows')
        d, M = dc.d, (vl.total if vl is not None else B * T)
        x = self._empty(M, d)
        wpe = None if dc.advpos else a.P(f'{self.dp}transformer.wpe.weight')
        # (dropout_without_save: a forward that is never differentiated but runs in training mode -- the momentum twin)
        # (drop_plan: a second differentiated segment of the same step -- the prompt rows of the contrastive loss -- draws its own masks)
        plan = (drop_plan if drop_plan is not None else self.dec_drop) if (save or dropout_without_save) else None
        emb_drop = plan.get(0, 'emb') if plan is not None else None
        pos_ctx = None
        if ids is not None:
            ids = ids.to(device=a.device, dtype=torch.long).contiguous()
            if vl is not None:
                ops.embed_fwd(ids, a.P(self.n_wte), wpe, x, M, 1, d, pos_offset, dc.V, pos=vl.pos)
            else:
                ops.embed_fwd(ids, a.P(self.n_wte), wpe, x, B, T, d, pos_offset, dc.V)
            if dc.advpos:
                x, pos_ctx = self.posmlp_fwd(x, B, T, pos_offset, vl, save)
            if emb_drop is not None:
                ops.dropout_apply(x, M, d, emb_drop)
        else:
            if dc.advpos:
                x, pos_ctx = self.posmlp_fwd(embeds.to(device=a.device, dtype=F32).contiguous().view(M, d), B, T, pos_offset, vl, save)
            elif vl is not None:      # packed rows: each row's own position (the prefixed sequence of a Hugging Face decoder)
                torch.index_select(wpe, 0, vl.pos.long() + pos_offset, out=x)
                ops.add_(x, embeds.to(device=a.device, dtype=F32).contiguous())
            else:
                ops.bcast_rows(wpe[pos_offset:pos_offset + T], x, T * d, B, T, d)
                ops.add_(x, embeds.to(device=a.device, dtype=F32).contiguous())
            if emb_drop is not None:                          # decoder.py:233/238: the embedding dropout covers the prompt rows too
                ops.dropout_apply(x, M, d, emb_drop)
        saves, cur, perm = [], x, None
        for l in range(dc.L):
            m = mem_bf if (mem_bf is not None and (self.dec_cross[l] or not self.dcfg.skip_alternate_cross_attn)) else None
            if dc.fam is not None:
                cur, sv, perm = self.fam_layer_fwd(f'{self.dp}transformer.h.{l}', dc.fam, cur, perm, B, T, m, S, save, plan, l, vl,
                                                   self.sparse_subset('dec', l, B, T, pos_offset, vl), split)
            else:
                cur, sv = self.block_fwd(f'{self.dp}transformer.h.{l}', cur, B, T, d, dc.H, dc.ff, dc.causal, m, S, save, plan, l, vl)
            saves.append(sv)
        cur = self.materialize(cur, perm)
        hid, mf, rf = self._empty(M, d), self._empty(M), self._empty(M)
        ops.layernorm_fwd(cur, a.P(f'{self.dp}transformer.ln_f.weight'), a.P(f'{self.dp}transformer.ln_f.bias'), hid, mf, rf, M, d)
        hb = self._empty(M, d, dtype=BF16)
        ops.cast_f32_bf16(hid, hb)
        ctx = SimpleNamespace(ids=ids, saves=saves, xl=cur, mf=mf, rf=rf, hb=hb, B=B, T=T, S=S, pos_offset=pos_offset, vl=vl, M=M,
                              emb_drop=emb_drop, pos_ctx=pos_ctx) if save else None
        return hid, hb, ctx

    # ---- Hugging Face decoders with a soft prompt: ONE causal sequence [encoder outputs | text] (the reference passes no mask to
    # transformers, decoder.py:349-361: text rows see the prompt rows, prompt rows see their predecessors) instead of the two
    # independent segments the nanoGPT decoder's mask leaves (vision_encoder_decoder.py:84-113)
    def decode_prefixed(self, B: int, T: int, enc_out, mem_bf, save: bool, ids, dropout_without_save: bool = False, text_mask=None):
        """ids [B, T] after the n_p = min(n_cls, block) encoder outputs.  Returns (hidden fp32 [B, n_p + T, d], bf16 hidden of the
        text rows [B * T, d], ctx).  text_mask (bool [B, T], a prefix of every row True: the live text positions of a training step,
        wrapper._pack_rows): the sequences are packed -- n_p + len_b rows each -- and both outputs hold the packed rows only."""
        a, dc = self.arena, self.dec
        ncls = enc_out.shape[1]
        n_p = min(ncls, dc.block)
        T = min(T, dc.block - n_p)
        ids = ids[:, :T].to(device=a.device, dtype=torch.long).contiguous()
        emb = self._empty(B, n_p + T, dc.d)
        emb[:, :n_p].copy_(enc_out[:, :n_p])
        if T:
            tok = self._empty(B * T, dc.d)
            ops.embed_fwd(ids, a.P(self.n_wte), None, tok, B, T, dc.d, 0, dc.V)
            emb[:, n_p:].copy_(tok.view(B, T, dc.d))
        if text_mask is not None:
            full = torch.cat((torch.ones(B, n_p, dtype=torch.bool, device=a.device), text_mask[:, :T]), dim=1)
            lens = full.sum(dim=1)
            cu = torch.zeros(B + 1, dtype=torch.int32, device=a.device)
            cu[1:] = torch.cumsum(lens, 0)
            pos = torch.arange(n_p + T, dtype=torch.int32, device=a.device).expand(B, n_p + T)[full].contiguous()
            vl = SimpleNamespace(cu=cu, pos=pos, total=int(pos.numel()), mask=full)
            is_text = pos >= n_p
            hid, hb, ctx = self.decode_segment(B, n_p + T, mem_bf, ncls, save, embeds=emb[full], pos_offset=0, vl=vl,
                                               dropout_without_save=dropout_without_save)
            hb_text = hb[is_text]
            if ctx is not None:
                ctx.prefixed = SimpleNamespace(ids=ids[text_mask[:, :T]].contiguous(), n_p=n_p, T=T, hb_text=hb_text, is_text=is_text)
            return hid, hb_text, ctx
        hid, hb, ctx = self.decode_segment(B, n_p + T, mem_bf, ncls, save, embeds=emb.view(B * (n_p + T), dc.d), pos_offset=0,
                                           dropout_without_save=dropout_without_save)
        hb_text = hb.view(B, n_p + T, dc.d)[:, n_p:].contiguous().view(B * T, dc.d)
        if ctx is not None:
            ctx.prefixed = SimpleNamespace(ids=ids, n_p=n_p, T=T, hb_text=hb_text, is_text=None)
        return hid.view(B, n_p + T, dc.d), hb_text, ctx

    def decode_prefixed_backward(self, ctx, dlogits_bf, dhid, dmem):
        """dlogits_bf: bf16 [B * T, Vp] over the TEXT rows or None; dhid: fp32 [B, n_p + T, d] or None.  The prompt rows' input
        gradient is added to dmem (they ARE the encoder outputs), the text rows' is scattered into wte."""
        a, dc, px = self.arena, self.dec, ctx.prefixed
        B, n_p, T, d = ctx.B, px.n_p, px.T, dc.d
        wte, head = self.n_wte, self.n_head
        if px.is_text is not None:        # packed rows (training step): dlogits_bf covers the packed text rows, dhid is not offered
            assert dhid is None
            Mt = int(px.hb_text.shape[0])
            dh = torch.zeros(ctx.M, d, dtype=F32, device=a.device)
            if dlogits_bf is not None and Mt:
                if a.trainable(head):
                    ops.gemm(dlogits_bf, px.hb_text, a.G(head), dc.V, d, Mt, a_kmajor=True, b_kmajor=True, accumulate=True)
                dht = self._empty(Mt, d)
                ops.gemm(dlogits_bf, a.W(head), dht, Mt, d, dc.V, b_kmajor=True)
                dh[px.is_text] = dht
            dx = self.decode_backward(ctx, None, dh, dmem)
            dmem.view(B, -1, d)[:, :n_p] += dx[~px.is_text].view(B, n_p, d)
            if Mt and a.trainable(wte):
                ops.embed_bwd(px.ids, dx[px.is_text].contiguous(), a.G(wte), None, Mt, 1, d, 0, dc.V)
            return
        dh = torch.zeros(B, n_p + T, d, dtype=F32, device=a.device) if dhid is None else dhid.to(F32).reshape(B, n_p + T, d).clone()
        if dlogits_bf is not None and T:
            if a.trainable(head):
                ops.gemm(dlogits_bf, px.hb_text, a.G(head), dc.V, d, B * T, a_kmajor=True, b_kmajor=True, accumulate=True)   # lm_head (tied or not)
            dht = self._empty(B * T, d)
            ops.gemm(dlogits_bf, a.W(head), dht, B * T, d, dc.V, b_kmajor=True)
            dh[:, n_p:] += dht.view(B, T, d)
        dx = self.decode_backward(ctx, None, dh.view(B * (n_p + T), d), dmem).view(B, n_p + T, d)
        dmem.view(B, -1, d)[:, :n_p] += dx[:, :n_p]
        if T and a.trainable(wte):
            ops.embed_bwd(px.ids, dx[:, n_p:].contiguous().view(B * T, d), a.G(wte), None, B, T, d, 0, dc.V)

    def logits_f32(self, hb: torch.Tensor, M: int):
        out = self._empty(M, self.dec.V)
        ops.gemm(hb, self.arena.W(self.n_head), out, M, self.dec.V, self.dec.d)
        return out

    def logits_bf16(self, hb: torch.Tensor, M: int, capacity: Optional[int] = None):
        """bf16 logits [M, Vp] in a cached buffer (of `capacity` >= M rows) whose pad columns are zero and never written."""
        cap = capacity or M
        buf = self._logits_cache.get(cap)
        if buf is None:
            self._logits_cache.clear()
            buf = torch.zeros(cap, self.dec.Vp, dtype=BF16, device=self.arena.device)
            self._logits_cache[cap] = buf
        buf = buf[:M]
        ops.gemm(hb, self.arena.W(self.n_head), buf, M, self.dec.V, self.dec.d)
        return buf

    @staticmethod
    def _lockstep(*gens):
        """Run backward generators side by side.  Each yields (key, its sum(g^2) part) at a gradient-normaliser site (keys
        descend); generators stopped at the SAME key belong to one tensor in the reference (the prompt rows and the text rows of a
        decoder block) and are sent the sum of their parts, everything else gets its own part back.  Returns their return values."""
        state, results = {}, [None] * len(gens)
        for i, g in enumerate(gens):
            try:
                state[i] = next(g)
            except StopIteration as stop:
                results[i] = stop.value
        while state:
            top = max(k for k, _ in state.values())
            group = [i for i, (k, _) in state.items() if k == top]
            joint = state[group[0]][1]
            for i in group[1:]:
                joint = joint + state[i][1]                    # 1-float device tensors: no host sync
            for i in group:
                try:
                    state[i] = gens[i].send(joint)
                except StopIteration as stop:
                    results[i] = stop.value
                    del state[i]
        return results

    def decode_backward(self, ctx, dlogits_bf: Optional[torch.Tensor], dhid: Optional[torch.Tensor], dmem):
        """dlogits_bf: bf16 [M, Vp] (pads zero) or None; dhid: fp32 [M, d] or None; dmem: fp32 [B*S, d] accumulator.  Returns the
        gradient w.r.t. ``embeds`` for an embeds-driven segment (None when the segment was driven by token ids)."""
        return self._lockstep(self._decode_backward_steps(ctx, dlogits_bf, dhid, dmem, paired=False))[0]

    def decode_backward_pair(self, ctx_a, dlogits_a, dhid_a, ctx_b, dlogits_b, dhid_b, dmem):
        """Backward of two segments of ONE decoder sequence (text rows a, prompt rows b: reference vision_encoder_decoder.py:84-113)
        whose block outputs the reference normalises as one tensor (layers.py:606-607): the two run layer by layer in lock step and
        share every normaliser's sum.  Returns (None | d embeds_a, None | d embeds_b)."""
        return self._lockstep(self._decode_backward_steps(ctx_a, dlogits_a, dhid_a, dmem, paired=True),
                              self._decode_backward_steps(ctx_b, dlogits_b, dhid_b, dmem, paired=True))

    def _decode_backward_steps(self, ctx, dlogits_bf, dhid, dmem, paired: bool):
        a, dc = self.arena, self.dec
        B, T, d, M = ctx.B, ctx.T, dc.d, ctx.M
        wte, head = self.n_wte, self.n_head
        dh = torch.zeros(M, d, dtype=F32, device=a.device) if dlogits_bf is None else self._empty(M, d)
        if dlogits_bf is not None:
            if a.trainable(head):
                ops.gemm(dlogits_bf, ctx.hb, a.G(head), dc.V, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)   # lm_head (tied or not)
            ops.gemm(dlogits_bf, a.W(head), dh, M, d, dc.V, b_kmajor=True)
        if dhid is not None:
            ops.add_(dh, dhid.contiguous())
        if dc.llama is not None:
            dx = self.llama_decode_bwd(ctx, dh)
            if ctx.ids is not None:
                if not a.trainable(wte):
                    return None
                if ctx.vl is not None:
                    ops.embed_bwd(ctx.ids, dx, a.G(wte), None, M, 1, d, ctx.pos_offset, dc.V, pos=ctx.vl.pos)
                else:
                    ops.embed_bwd(ctx.ids, dx, a.G(wte), None, B, T, d, ctx.pos_offset, dc.V)
                return None
            return dx
        dx = self._empty(M, d)
        ops.layernorm_bwd(dh, ctx.xl, a.P(f'{self.dp}transformer.ln_f.weight'), ctx.mf, ctx.rf, dx,
                          a.G(f'{self.dp}transformer.ln_f.weight'), a.G(f'{self.dp}transformer.ln_f.bias'), M, d)
        masked = False            # (the embedding dropout's backward already applied by the lowest block)
        if dc.fam is not None:
            dperm = None
            for l in reversed(range(dc.L)):
                dx, dperm = yield from self.fam_layer_bwd_steps(l, f'{self.dp}transformer.h.{l}', dc.fam, ctx.saves[l], dx, dperm, B, T, ctx.S,
                                                                dmem, ctx.vl)
            dx = self.materialize(dx, dperm)
        elif not dc.grad_norm:
            self._blocks_bwd(self.dp, ctx.saves, dx, B, T, d, dc.H, dc.ff, dc.causal, ctx.S, dmem, ctx.vl, normalize=False)
        elif paired:        # the un-fused form of _blocks_bwd: the normaliser's sum comes from outside
            dxb = self._empty(M, d, dtype=BF16)
            for l in reversed(range(dc.L)):
                joint = yield l, ops.sumsq(dx, self._empty(1))
                ops.grad_normalize(dx, joint, dxb, bf16_drop=ctx.saves[l].dr['mlp'], presummed=True)
                self.block_bwd(f'{self.dp}transformer.h.{l}', ctx.saves[l], dx, dxb, B, T, d, dc.H, dc.ff, dc.causal, ctx.S, dmem,
                               emit_last_bf16=False, vl=ctx.vl)
        else:
            emb_mask = ctx.emb_drop if (ctx.emb_drop is not None and EMB_DROP_FUSED and int(ctx.emb_drop[0]) == 1 and M * d < 2 ** 32 and dc.L > 0) else None
            self._blocks_bwd(self.dp, ctx.saves, dx, B, T, d, dc.H, dc.ff, dc.causal, ctx.S, dmem, ctx.vl, lowest_dx_mask=emb_mask)
            masked = emb_mask is not None
        if ctx.emb_drop is not None and not masked:
            ops.dropout_apply(dx, M, d, ctx.emb_drop)
        if ctx.ids is not None:
            dwpe = None if dc.advpos else a.G(f'{self.dp}transformer.wpe.weight')
            if dc.advpos:
                dx = self.posmlp_bwd(ctx.pos_ctx, dx)
            if ctx.vl is not None:
                ops.embed_bwd(ctx.ids, dx, a.G(wte), dwpe, M, 1, d, ctx.pos_offset, dc.V, pos=ctx.vl.pos)
            else:
                ops.embed_bwd(ctx.ids, dx, a.G(wte), dwpe, B, T, d, ctx.pos_offset, dc.V)
            return None
        if dc.advpos:
            return self.posmlp_bwd(ctx.pos_ctx, dx)
        if ctx.vl is not None:      # packed rows: position of row m = vl.pos[m] (the same scatter kernel the id-driven path uses; no token part)
            ops.embed_bwd(None, dx, None, a.G(f'{self.dp}transformer.wpe.weight'), M, 1, d, ctx.pos_offset, dc.V, pos=ctx.vl.pos)
            return dx
        ops.sum_over_batch(dx, T * d, a.G(f'{self.dp}transformer.wpe.weight')[ctx.pos_offset:ctx.pos_offset + T], B, T, d, accumulate=True)
        return dx
