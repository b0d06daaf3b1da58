"""VisionEncoderDecoder: the plugin surface of the captioning hot path (reference models/vision_encoder_decoder.py).

Same constructor, attributes, ``forward`` / ``generate`` signatures, output record and state-dict keys as the
reference, so ``trainer.py`` / ``training/*`` / the notebook drive it unchanged.  Underneath, ``forward`` is one
autograd node around ``engine.HotPath`` (hand-written HIP forward + backward) and greedy ``generate`` is a static
KV-cache decode loop replayed from a hipGraph (``decoding.GreedyDecoder``).
"""
import weakref
from typing import Optional

import torch
import torch.nn as nn

from ..configs.models import VisionEncoderDecoderConfig
from ..engine import BF16, F32, HotPath
from ..lib import I2TError
from ..object_models import VisionEncoderDecoderModelOutput
from .decoder import Decoder
from .encoder import Encoder
from .utils import update_state_dict_from_partial_checkpoint


class _BridgedEncoder(nn.Sequential):
    """nn.Sequential(encoder, Linear(bias=False)) as the reference builds it when widths differ (:33-37) -- kept for the
    ``encoder.0.* / encoder.1.weight`` state-dict keys; calling it runs the HIP encoder + bridge."""

    def forward(self, images):
        owner = self._owner() if getattr(self, '_owner', None) else None
        if owner is None:
            raise I2TError('bridged encoder detached from its VisionEncoderDecoder')
        return owner.encode(images)


class _HotPathFunction(torch.autograd.Function):
    """(images | encoder_output, ids) -> (encoder_output, text logits, text hidden); parameters receive their
    gradients as a side effect of ``backward`` (views of the flat gradient arena become ``p.grad``)."""

    @staticmethod
    def forward(ctx, hook, model, images, ids, enc_in, save):
        eng: HotPath = model._engine
        eng.prepare(model.training and save)
        B, L = ids.shape
        cfg = model.config
        enc_ctx = None
        if enc_in is None:
            enc_out, enc_ctx = eng.encode(images, save)
        else:
            enc_out = enc_in.to(device=eng.arena.device, dtype=F32).contiguous()
        ncls = enc_out.shape[1]
        mem = eng._mem_bf16(enc_out) if eng.cross_inputs else None
        off = ncls if cfg.use_soft_prompting else 0
        T = min(L, eng.dec.block - off)                       # the reference crops inputs to block_size (:88)
        if eng.dec.prefixed:      # Hugging Face decoder + soft prompt: one causal sequence [encoder outputs | text] (engine.decode_prefixed)
            hid, hb, dctx = eng.decode_prefixed(B, T, enc_out, mem, save, ids)
            ctx.set_materialize_grads(False)
            ctx.model, ctx.enc_ctx, ctx.dctx, ctx.has_enc_in = model, enc_ctx, dctx, enc_in is not None
            ctx.pctx, ctx.shapes = None, (B, T, ncls, hid.shape[1] - T)
            return enc_out, eng.logits_f32(hb, B * T).view(B, T, -1), hid
        hid, hb, dctx = eng.decode_segment(B, T, mem, ncls, save, ids=ids[:, :T], pos_offset=off)
        logits = eng.logits_f32(hb, B * T).view(B, T, -1)
        hid = hid.view(B, T, -1)
        # hidden_state of the reference also carries the prompt rows (it is not sliced, :133).  Under a causal decoder they are an
        # independent causal segment (text never attends to them); differentiable like everything else: a loss on those rows
        # back-propagates through this segment, in lock step with the text rows (shared gradient normaliser, layers.py:606-607)
        pctx, n_p = None, 0
        if cfg.use_soft_prompting and eng.dec.causal:
            n_p = min(ncls, eng.dec.block)
            ph, _, pctx = eng.decode_segment(B, n_p, mem, ncls, save, embeds=enc_out[:, :n_p].reshape(B * n_p, -1), pos_offset=0,
                                             drop_plan=eng.dec_drop_prompt)
            hid = torch.cat((ph.view(B, n_p, -1), hid), dim=1)
        ctx.set_materialize_grads(False)                     # an output the loss does not use arrives as None, not as zeros
        ctx.model, ctx.enc_ctx, ctx.dctx, ctx.has_enc_in = model, enc_ctx, dctx, enc_in is not None
        ctx.pctx, ctx.shapes = pctx, (B, T, ncls, n_p)
        return enc_out, logits, hid

    @staticmethod
    def backward(ctx, d_enc, d_logits, d_hid):
        model = ctx.model
        eng: HotPath = model._engine
        a = eng.arena
        B, T, ncls, n_p = ctx.shapes
        eng.notify_grads_ready('begin')          # e.g. the DP exchange drains whatever is still in flight on the arena
        a.begin_backward()
        dmem = torch.zeros(B * ncls, eng.dec.d, dtype=F32, device=a.device)
        dl = None
        if d_logits is not None:
            dl = torch.zeros(B * T, eng.dec.Vp, dtype=BF16, device=a.device)
            dl[:, :eng.dec.V] = d_logits.reshape(B * T, -1)
        dh = dph = None
        if eng.dec.prefixed:
            eng.decode_prefixed_backward(ctx.dctx, dl, d_hid, dmem)
            d_hid = None
        elif d_hid is not None:
            d_hid = d_hid.to(F32)
            dh = d_hid[:, n_p:].reshape(B * T, -1).contiguous()
            dph = d_hid[:, :n_p].reshape(B * n_p, -1).contiguous() if ctx.pctx is not None else None
        if eng.dec.prefixed:
            pass
        elif dph is not None:                     # the prompt rows of hidden_state carry gradient: both segments, one normaliser
            _, dxp = eng.decode_backward_pair(ctx.dctx, dl, dh, ctx.pctx, None, dph, dmem)
            dmem.view(B, ncls, -1)[:, :n_p].add_(dxp.view(B, n_p, -1))
        else:
            eng.decode_backward(ctx.dctx, dl, dh, dmem)
        eng.notify_grads_ready('decoder')
        if d_enc is not None:
            dmem += d_enc.reshape(B * ncls, -1)
        d_enc_in = None
        if ctx.has_enc_in:
            d_enc_in = dmem.view(B, ncls, -1)
        else:
            eng.encode_backward(ctx.enc_ctx, dmem)
            eng.notify_grads_ready('encoder')
        a.attach_grads()
        ctx.enc_ctx = ctx.dctx = ctx.pctx = None
        return None, None, None, None, d_enc_in, None


class VisionEncoderDecoder(nn.Module):
    """Encoder Decoder model for conditional generation (reference vision_encoder_decoder.py:17-182)."""

    def __init__(self, config: VisionEncoderDecoderConfig, encoder: Optional[Encoder] = None,
                 decoder: Optional[Decoder] = None):
        super().__init__()
        self.config = config
        encoder = encoder if encoder is not None else Encoder.from_config(config.vision_encoder_config)
        self.space_for_prompt = encoder.num_outputs if config.use_soft_prompting else 0
        self.decoder = decoder if decoder is not None else Decoder.from_config(
            config=config.decoder_config, loose=config.loose_match_decoder_state_dict, space_for_prompt=self.space_for_prompt)
        decoder_n_embd = self.decoder.n_embd
        self.has_bridge = encoder.output_embed_dim != decoder_n_embd
        if self.has_bridge:
            self.encoder = _BridgedEncoder(encoder, nn.Linear(encoder.output_embed_dim, decoder_n_embd, bias=False))
        else:
            self.encoder = encoder
        object.__setattr__(self.encoder, '_owner', weakref.ref(self))
        object.__setattr__(encoder, '_owner', weakref.ref(self))
        object.__setattr__(self.decoder, '_owner', weakref.ref(self))
        self._processor = None
        self.use_cross_attn = config.use_cross_attn
        self.use_soft_prompting = config.use_soft_prompting
        if not (self.use_cross_attn or self.use_soft_prompting):
            raise ValueError('Misconfigured!!! Need to either use cross attn or soft prompting or both')
        object.__setattr__(self, '_engine', HotPath(self))
        object.__setattr__(self, '_hook', None)
        object.__setattr__(self, '_greedy', None)
        if config.chkpt_path is not None:
            update_state_dict_from_partial_checkpoint(self, config.chkpt_path, map_location=None)

    # -- reference attribute: HF logits processors (used by sampling modes and by BeamSearchTokenGenerator)
    @property
    def processor(self):
        if self._processor is None:
            from transformers import LogitsProcessorList, NoRepeatNGramLogitsProcessor
            self._processor = LogitsProcessorList([NoRepeatNGramLogitsProcessor(ngram_size=n) for n in self.config.no_repeat_n_grams])
        return self._processor

    def _grad_hook(self, device):
        if self._hook is None or self._hook.device != device:
            object.__setattr__(self, '_hook', torch.zeros(1, device=device, requires_grad=True))
        return self._hook

    def _check_mask(self, attn_msk, bs: int, L: int):
        """The reference expands the mask (einops ``repeat``, :61-72) and then loses it (bool masked_fill, :97-98):
        shapes are validated, values are inert (golden fixtures *_mask == nomask)."""
        if attn_msk is None:
            return
        if attn_msk.dim() == 2:
            ok = attn_msk.shape[1] == L if attn_msk.shape[0] == bs else tuple(attn_msk.shape) == (L, L)
        elif attn_msk.dim() == 3:
            ok = tuple(attn_msk.shape[1:]) == (L, L)
        else:
            ok = attn_msk.dim() == 4 and tuple(attn_msk.shape[2:]) == (L, L)
        if not ok:
            raise RuntimeError(f'attn_msk of shape {tuple(attn_msk.shape)} does not broadcast to (bs, h, {L}, {L})')

    def encode(self, images: torch.Tensor) -> torch.Tensor:
        """``self.encoder(images)`` of the reference: (B, n_cls, decoder_n_embd), no autograd graph."""
        with torch.no_grad():
            self._engine.prepare(False)
            out, _ = self._engine.encode(images, False)
        return out

    def forward(self, images: Optional[torch.FloatTensor], ids: torch.LongTensor,
                attn_msk: Optional[torch.BoolTensor] = None,
                encoder_output: Optional[torch.Tensor] = None) -> VisionEncoderDecoderModelOutput:
        dev = next(self.parameters()).device
        ids = ids.to(dev)
        self._check_mask(attn_msk, ids.shape[0], ids.shape[-1])
        save = torch.is_grad_enabled()
        enc_out, logits, hid = _HotPathFunction.apply(self._grad_hook(dev), self, images, ids, encoder_output, save)
        if self.use_soft_prompting and not self._engine.dec.causal:
            # a NON-causal decoder: the prompt rows see every column, text included (:93-95), so they need the whole sequence in one
            # pass; text rows still never see the prompt (split visibility of the grouped kernels).  Forward only: these rows carry
            # no gradient.
            with torch.no_grad():
                eng = self._engine
                B, ncls = enc_out.shape[0], enc_out.shape[1]
                n_p = min(ncls, eng.dec.block)
                mem = eng._mem_bf16(enc_out) if eng.cross_inputs else None
                Tt = hid.shape[1]
                emb = torch.cat((enc_out[:, :n_p], self.decoder.get_inputs_embeds(ids[:, :Tt])), dim=1)
                full, _, _ = eng.decode_segment(B, n_p + Tt, mem, ncls, False, embeds=emb.reshape(B * (n_p + Tt), -1), pos_offset=0, split=n_p)
                ph = full.view(B, n_p + Tt, -1)[:, :n_p]
            hid = torch.cat((ph.reshape(B, n_p, -1), hid), dim=1)
        return VisionEncoderDecoderModelOutput(encoder_output=enc_out, logits=logits, hidden_state=hid)

    @torch.no_grad()
    def generate(self, images, prompt_ids, max_new_tokens=128, temperature=1.0, top_k=None, nucleus_p=None) -> torch.LongTensor:
        """Autoregressive generation (reference :136-182) on the static KV cache under hipGraph replay, one replay per token.
        ``top_k=1`` without a nucleus is greedy decoding (argmax after the n-gram ban; the temperature cannot change an argmax);
        every other mode draws each token on the device from the reference's filtered distribution (temperature -> n-gram
        ban -> top-k -> softmax -> nucleus), see ``decoding.Sampling``.  The draws are reproducible under ``torch.manual_seed``."""
        blk_size = self.decoder.block_size - self.space_for_prompt
        assert max_new_tokens <= blk_size - prompt_ids.size(-1)
        dev = next(self.parameters()).device
        prompt_ids = prompt_ids.to(dev)
        from ..decoding import GreedyDecoder, Sampling, generate_by_recompute
        if not self._engine.dec.causal:      # bidirectional attention: every new token changes the state of the earlier ones
            return generate_by_recompute(self, images, prompt_ids, max_new_tokens,
                                         None if (top_k == 1 and nucleus_p is None) else Sampling(temperature, top_k, nucleus_p))
        if self._greedy is None:
            object.__setattr__(self, '_greedy', GreedyDecoder(self))
        if top_k == 1 and nucleus_p is None:
            return self._greedy.generate(images, prompt_ids, max_new_tokens)
        return self._greedy.generate(images, prompt_ids, max_new_tokens, sampling=Sampling(temperature, top_k, nucleus_p))


def _owner_of(module):
    """The VisionEncoderDecoder whose engine runs ``module``'s arithmetic.  A FREE-STANDING ``Encoder.from_config(...)`` /
    ``Decoder.from_config(...)`` module (the reference's factories return usable nn.Modules, encoder.py:34-45, decoder.py:39-42) gets
    a private holder on first use: the module itself plus the smallest counterpart that satisfies the engine (a one-layer decoder as
    wide as the encoder's output / a one-layer 64-wide encoder), kept alive on the module.  Only the module's own tower ever runs."""
    ref = getattr(module, '_owner', None)
    owner = ref() if ref is not None else None
    if owner is None:
        owner = _standalone_holder(module)
        object.__setattr__(module, '_standalone_holder', owner)         # strong reference: the weak _owner must not outlive its target
    if getattr(module, '_standalone_holder', None) is owner:             # the module may have been moved since: the stand-in tower follows
        dev = next(module.parameters()).device
        if any(p.device != dev for p in owner.parameters()):
            owner.to(dev)
    return owner


def _standalone_holder(module):
    from ..configs.models import (ImageInputSpec, MLPConfig, SelfAttentionConfig, SelfAttentionType, TransformerConfig,
                                  TransformerDecoderConfig, VisionEncoderDecoderConfig, VisionTransformerEncoderConfig)

    def tf(d, heads, causal, cross):
        return TransformerConfig(rotator_config=MLPConfig(ff_mult=1), is_causal=causal, is_cross_attn=cross,
                                 attn_config=SelfAttentionConfig(attn_dropout=0.0, bias=True, dropout=0.0, n_head=heads, n_embd=d,
                                                                 attn_type=SelfAttentionType.MULTI_HEAD))
    if isinstance(module, Encoder):
        d = module.output_embed_dim
        if d % 64:
            raise NotImplementedError('a free-standing encoder needs an output width that is a multiple of 64 (its stand-in decoder)')
        dcfg = TransformerDecoderConfig(transformer_config=tf(d, d // 64, True, False), n_layer=1, block_size=module.num_outputs + 8, vocab_size=8)
        cfg = VisionEncoderDecoderConfig(vision_encoder_config=module.config, decoder_config=dcfg, use_cross_attn=False, use_soft_prompting=True)
        dev = next(module.parameters()).device
        holder = VisionEncoderDecoder(cfg, encoder=module)
        holder.decoder.to(dev)
        return holder
    if isinstance(module, Decoder):
        ecfg = VisionTransformerEncoderConfig(transformer_config=tf(64, 1, False, False), input=ImageInputSpec(n_channels=3, width=8, height=8),
                                              n_layer=1, n_cls=1, num_patches=2, n_channels=8, feature_extractor_gate_sizes=None,
                                              feature_extractor_kernel_size=(4, 4))
        cross = bool(getattr(module, 'use_cross_attn', True)) and \
            (not hasattr(module.config, 'transformer_config') or module.config.transformer_config.is_cross_attn)
        cfg = VisionEncoderDecoderConfig(vision_encoder_config=ecfg, decoder_config=module.config, use_cross_attn=cross,
                                         use_soft_prompting=not cross)
        dev = next(module.parameters()).device
        holder = VisionEncoderDecoder(cfg, decoder=module)
        holder.encoder.to(dev)
        return holder
    raise NotImplementedError(f'{type(module).__name__} has no HIP path of its own')


def run_encoder_standalone(encoder, images):
    """``VisionTransformerEncoder.forward``: the un-bridged (B, n_cls, d_enc) output when a bridge exists is not
    separately exposed by the fused path, so standalone calls are supported for bridge-less models only."""
    owner = _owner_of(encoder)
    if owner.has_bridge:
        raise NotImplementedError('call model.encoder(images) (encoder + bridge); the un-bridged output is internal to the HIP path')
    return owner.encode(images)


def run_decoder_standalone(decoder, idx, inputs_embeds, cross_attn_embeds, attn_msk):
    """``TransformerDecoder.forward(idx | inputs_embeds, cross_attn_embeds, attn_msk) -> (logits, hidden)`` for one
    causal segment (``attn_msk`` must be None: arbitrary additive masks are not supported)."""
    owner = _owner_of(decoder)
    assert not (idx is None and inputs_embeds is None)
    assert idx is None or inputs_embeds is None
    if hasattr(decoder, 'hf_config'):        # Hugging Face decoders: always causal, cross inputs dropped without the layers (decoder.py:341-361)
        attn_msk = None
        cross_attn_embeds = cross_attn_embeds if decoder.use_cross_attn else None
    if attn_msk is not None:
        raise NotImplementedError('TransformerDecoder.forward with an explicit additive mask is not supported by the HIP kernels')
    eng: HotPath = owner._engine
    with torch.no_grad():
        eng.prepare(False)
        mem, S = None, 0
        if cross_attn_embeds is not None:
            S = cross_attn_embeds.shape[1]
            mem = eng._mem_bf16(cross_attn_embeds.to(eng.arena.device, F32))
        if idx is not None:
            B, T = idx.shape
            hid, hb, _ = eng.decode_segment(B, T, mem, S, False, ids=idx)
        else:
            B, T, _ = inputs_embeds.shape
            hid, hb, _ = eng.decode_segment(B, T, mem, S, False, embeds=inputs_embeds.reshape(B * T, -1))
        return eng.logits_f32(hb, B * T).view(B, T, -1), hid.view(B, T, -1)
