"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A plain-PyTorch fp32 restatement of the reference's captioning hot path, written as pure functions over a
state-dict (``{name: tensor}``) so that it cannot be mistaken for, or imported by, the product modules.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package.

Parity status: PINNED.  ``tests/test_oracle_golden.py`` checks every function here against fixtures produced by
running the reference itself in the build container (``tools/gen_goldens.py`` -> ``tests/golden/*.npz``): forward
outputs for four mask forms and three prompt/cross-attention modes, the train-step loss and the gradient of every
parameter, and greedy token ids -- all to <= 1e-5 (fp32) / token-exact.

Each function cites the reference lines it restates (paths relative to the reference repo).
"""
import math
from typing import Dict, Optional, Sequence

import torch
import torch.nn.functional as F

NEG_INF = float('-inf')
SD = Dict[str, torch.Tensor]


# --------------------------------------------------------------------------------------------------------------
# gradient normaliser (models/functions.py:4-27): identity forward, g / (||g||_2 + 1e-6) backward over the tensor
# --------------------------------------------------------------------------------------------------------------
class _UnitNormGrad(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g / (torch.linalg.vector_norm(g) + 1e-6)


def _get(sd: SD, name: str) -> Optional[torch.Tensor]:
    return sd.get(name)


def _drop(x, p, training):
    return F.dropout(x, p, training) if (training and p > 0) else x


def _planned(entry, x, per_row_third=None):
    """Dropout with the EXACT mask the HIP path uses for a site (``entry`` = (mode, key, thr, scale) from
    image2text_amd.engine.DropPlan.get, rebuilt on the host by image2text_amd.rng.keep_mask).  torch's own RNG stream
    cannot be matched, so parity under dropout is defined as: same masks -> same loss / gradients."""
    if entry is None:
        return x
    from image2text_amd import rng
    if len(entry) == 5:     # the HIP path computes (and indexes) this site on the first `live` rows of every sequence only: the
        live = entry[4]     # rows past them are dead in the reference too (nothing reads them), so they keep an all-ones mask
        out = x.clone()
        if x.dim() == 4:    # attention probabilities [B, H, Tq, Tk]: live query rows
            out[:, :, :live] = _planned(entry[:4], x[:, :, :live].contiguous())
        else:               # [B, T, C]
            out[:, :live] = _planned(entry[:4], x[:, :live].contiguous())
        return out
    mode, key, thr, scale = entry
    if mode == 2:       # per (row, third): x is (B, 1, T, 1) ones for third `per_row_third`
        B, _, T, _ = x.shape
        m = rng.keep_mask((key + per_row_third) & 0xFFFFFFFF, B * T, thr).view(B, 1, T, 1)
    else:
        m = rng.keep_mask(key, x.numel(), thr).view(x.shape)
    return x * m.to(x.dtype) * scale


# --------------------------------------------------------------------------------------------------------------
# blocks (models/layers.py)
# --------------------------------------------------------------------------------------------------------------
def layer_norm(x, w, b):
    """layers.py:349-370 -- LayerNorm / LayerNormND: normalise over the trailing dims covered by ``w``, eps 1e-5."""
    return F.layer_norm(x, tuple(w.shape), w, b, 1e-5)


def conv_stack(sd: SD, prefix: str, x):
    """layers.py:258-282 -- Conv2d('same') [-> GELU(tanh) -> Conv2d('same')]*.

    Sequential indices are 0,2,4,... (GELUs sit at the odd slots).  Even kernels pad asymmetrically:
    (k-1)//2 before and k//2 after, which is what torch's padding='same' does.
    """
    i = 0
    while f'{prefix}.model.{i}.weight' in sd:
        if i > 0:
            x = F.gelu(x, approximate='tanh')
        w = sd[f'{prefix}.model.{i}.weight']
        kh, kw = w.shape[-2:]
        x = F.pad(x, ((kw - 1) // 2, kw // 2, (kh - 1) // 2, kh // 2))
        x = F.conv2d(x, w, sd[f'{prefix}.model.{i}.bias'])
        i += 2
    return x


def softmax_attention(q, k, v, add_mask, dropout_p=0.0, training=False, planned=None):
    """F.scaled_dot_product_attention semantics as used at layers.py:465 and inside nn.MultiheadAttention:
    softmax(q k^T / sqrt(dh) + mask) v with the torch>=2.5 'safe softmax' rule: a fully masked row yields zeros."""
    s = (q @ k.transpose(-1, -2)) / math.sqrt(q.size(-1))
    if add_mask is not None:
        s = s + add_mask
    m = s.amax(dim=-1, keepdim=True)
    m = torch.where(torch.isinf(m), torch.zeros_like(m), m)
    p = torch.exp(s - m)
    z = p.sum(dim=-1, keepdim=True)
    p = p / torch.where(z == 0, torch.ones_like(z), z)
    p = _planned(planned, p) if planned is not None else _drop(p, dropout_p, training)
    return p @ v


def self_attention(sd: SD, p: str, x, n_head: int, add_mask, dropout=0.0, attn_dropout=0.0, training=False, plan=None, layer=0):
    """layers.py:433-470 MultiHeadAttention: fused c_attn, per-token dropout multipliers, SDPA, c_proj."""
    B, T, C = x.shape
    qkv = F.linear(x, sd[f'{p}.c_attn.weight'], _get(sd, f'{p}.c_attn.bias'))
    q, k, v = qkv.split(C, dim=2)
    ones = torch.ones((B, 1, T, 1), dtype=x.dtype)
    if plan is not None:      # HIP-path masks: thirds 0/1/2 of the fused c_attn output = q/k/v
        e = plan.get(layer, 'qkv')
        q_do, k_do, v_do = (_planned(e, ones, t) for t in range(3))
    else:
        k_do, q_do, v_do = (_drop(ones, attn_dropout, training) for _ in range(3))   # layers.py:454-457 order
    heads = lambda t: t.view(B, T, n_head, C // n_head).transpose(1, 2)
    y = softmax_attention(q_do * heads(q), k_do * heads(k), v_do * heads(v), add_mask, dropout, training,
                          planned=plan.get(layer, 'sdpa') if plan is not None else None)
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    out = F.linear(y, sd[f'{p}.c_proj.weight'], _get(sd, f'{p}.c_proj.bias'))
    return _planned(plan.get(layer, 'resid'), out) if plan is not None else _drop(out, dropout, training)


def cross_attention(sd: SD, p: str, x, mem, n_head: int, dropout=0.0, training=False, plan=None, layer=0):
    """layers.py:537-542,600-605 -- nn.MultiheadAttention(batch_first) with packed in_proj (3d,d), always biased:
    q from x, k/v from the encoder output, no mask, dropout on the attention weights, out_proj."""
    B, T, C = x.shape
    S = mem.size(1)
    w, b = sd[f'{p}.in_proj_weight'], sd[f'{p}.in_proj_bias']
    q = F.linear(x, w[:C], b[:C])
    k = F.linear(mem, w[C:2 * C], b[C:2 * C])
    v = F.linear(mem, w[2 * C:], b[2 * C:])
    hq = q.view(B, T, n_head, C // n_head).transpose(1, 2)
    hk = k.view(B, S, n_head, C // n_head).transpose(1, 2)
    hv = v.view(B, S, n_head, C // n_head).transpose(1, 2)
    y = softmax_attention(hq, hk, hv, None, dropout, training,
                          planned=plan.get(layer, 'xattn') if plan is not None else None).transpose(1, 2).contiguous().view(B, T, C)
    return F.linear(y, sd[f'{p}.out_proj.weight'], sd[f'{p}.out_proj.bias'])


def gelu_mlp(sd: SD, p: str, x, dropout=0.0, training=False, plan=None, layer=0):
    """layers.py:473-486 -- Linear d->4d, GELU(tanh), Linear 4d->d, dropout."""
    h = F.gelu(F.linear(x, sd[f'{p}.c_fc.weight'], _get(sd, f'{p}.c_fc.bias')), approximate='tanh')
    out = F.linear(h, sd[f'{p}.c_proj.weight'], _get(sd, f'{p}.c_proj.bias'))
    return _planned(plan.get(layer, 'mlp'), out) if plan is not None else _drop(out, dropout, training)


def multi_query_attention(sd: SD, p: str, x, n_head: int, add_mask, dropout=0.0, attn_dropout=0.0, training=False, plan=None, layer=0):
    """layers.py:391-430 MultiQueryAttention: q_proj for all heads, ONE key/value head from kv_proj shared by every query
    head, the same per-token dropout multipliers as the multi-head form, SDPA, out_proj."""
    B, T, C = x.shape
    hd = C // n_head
    q = F.linear(x, sd[f'{p}.q_proj.weight'], _get(sd, f'{p}.q_proj.bias'))
    k, v = F.linear(x, sd[f'{p}.kv_proj.weight'], _get(sd, f'{p}.kv_proj.bias')).split(hd, dim=-1)
    ones = torch.ones((B, 1, T, 1), dtype=x.dtype)
    if plan is not None:      # HIP-path masks: sections 0/1/2 = q/k/v token multipliers
        e = plan.get(layer, 'qkv')
        q_do, k_do, v_do = (_planned(e, ones, t) for t in range(3))
    else:
        k_do, q_do, v_do = (_drop(ones, attn_dropout, training) for _ in range(3))   # layers.py:414-416 order
    qh = q_do * q.view(B, T, n_head, hd).transpose(1, 2)
    kh = k_do * k.view(B, T, 1, hd).transpose(1, 2)
    vh = v_do * v.view(B, T, 1, hd).transpose(1, 2)
    y = softmax_attention(qh, kh, vh, add_mask, dropout, training, planned=plan.get(layer, 'sdpa') if plan is not None else None)
    y = y.transpose(1, 2).contiguous().view(B, T, C)
    out = F.linear(y, sd[f'{p}.out_proj.weight'], _get(sd, f'{p}.out_proj.bias'))
    return _planned(plan.get(layer, 'resid'), out) if plan is not None else _drop(out, dropout, training)


def gate_mlp(sd: SD, p: str, x):
    """layers.py:222-255 MLP without residual connector: Linear [GELU(tanh) Linear]* at Sequential slots 0, 2, 4, ..."""
    i = 0
    while f'{p}.model.{i}.weight' in sd:
        if i > 0:
            x = F.gelu(x, approximate='tanh')
        x = F.linear(x, sd[f'{p}.model.{i}.weight'], _get(sd, f'{p}.model.{i}.bias'))
        i += 2
    return x


def moe_linear(sd: SD, p: str, x, top_k: int, moe_io: Optional[dict] = None):
    """layers.py:301-346 MoELinear: gates = softmax(gate_mlp(x) / sqrt(in)); the top-k gate VALUES (not renormalised) weight the
    outputs of the chosen low-rank experts l2(gelu(l1(x))).

    moe_io (tests): {'forced': {site: LongTensor [N, top_k]}} replaces the top-k choice at a site by the given experts (their
    weights are still this function's gate values) -- the HIP path's bf16 gate can pick differently at a near-tie, and the
    comparison is then made on the same choice; {'record': {}} receives (gates [N, E], indices [N, top_k]) per site."""
    shape = x.shape
    in_f = shape[-1]
    xf = x.reshape(-1, in_f)
    gates = (gate_mlp(sd, f'{p}.expert_gates', xf) / math.sqrt(in_f)).softmax(dim=-1)
    w, idx = torch.topk(gates, top_k, dim=-1)
    if moe_io is not None:
        forced = moe_io.get('forced', {}).get(p)
        if forced is not None:
            idx = forced.to(torch.long)
            w = gates.gather(1, idx)
        if 'record' in moe_io:
            moe_io['record'][p] = (gates.detach().clone(), idx.clone())
    E = gates.size(-1)
    outs = torch.stack([F.linear(F.gelu(F.linear(xf, sd[f'{p}.experts.{e}.l1.weight'], sd[f'{p}.experts.{e}.l1.bias']), approximate='tanh'),
                                 sd[f'{p}.experts.{e}.l2.weight'], sd[f'{p}.experts.{e}.l2.bias']) for e in range(E)], dim=1)   # [N, E, out]
    sel = outs.gather(1, idx.unsqueeze(-1).expand(-1, -1, outs.size(-1)))
    y = (sel * w.unsqueeze(-1)).sum(dim=1)
    return y.view(*shape[:-1], -1)


def moe_mlp(sd: SD, p: str, x, top_k: int, dropout=0.0, training=False, plan=None, layer=0, moe_io=None):
    """layers.py:489-518 -- MoELinear d->ff d, GELU(tanh), MoELinear ff d->d, dropout."""
    h = F.gelu(moe_linear(sd, f'{p}.c_fc', x, top_k, moe_io), approximate='tanh')
    out = moe_linear(sd, f'{p}.c_proj', h, top_k, moe_io)
    return _planned(plan.get(layer, 'mlp'), out) if plan is not None else _drop(out, dropout, training)


def transformer_block(sd: SD, p: str, x, n_head: int, causal: bool, mem, add_mask, dropout=0.0, attn_dropout=0.0,
                      training=False, plan=None, layer=0, top_k=1, moe_io=None, pos_offset=0):
    """layers.py:565-614: pre-LN residual wiring attn -> (cross) -> mlp, then the gradient normaliser.  The variant is read off
    the state dict: ``attn.q_proj`` = multi-query, ``mlp.c_fc.experts`` = MoE rotator, ``input_mask_idx`` = sparse block (the
    block runs on the kept positions < T only; every other position takes x + null_connector(x); <= 1 kept position: the
    whole input takes the null path, :570-573).  pos_offset (text-segment form, see lm_step_text_segment): x holds positions
    pos_offset .. pos_offset + T - 1 of a longer sequence whose earlier rows it never attends to."""
    x_orig, idx, not_idx = x, None, None
    if f'{p}.input_mask_idx' in sd:
        T = pos_offset + x_orig.size(1)
        idx = sd[f'{p}.input_mask_idx'][sd[f'{p}.input_mask_idx'] < T]
        null = lambda t: F.linear(t, sd[f'{p}.null_connector.weight'], _get(sd, f'{p}.null_connector.bias'))
        if idx.numel() <= 1:
            return x_orig + null(x_orig)
        not_idx = sd[f'{p}.input_mask_not_idx'][sd[f'{p}.input_mask_not_idx'] < T]
        idx, not_idx = idx[idx >= pos_offset] - pos_offset, not_idx[not_idx >= pos_offset] - pos_offset
        if idx.numel() == 0:
            return x_orig + null(x_orig)
        x = x_orig[:, idx]
        add_mask = add_mask[..., idx, :][..., idx] if add_mask is not None else None
    if causal:
        L = x.size(-2)
        tri = torch.ones((L, L), dtype=torch.bool).tril()
        cm = torch.zeros((L, L), dtype=x.dtype).masked_fill(~tri, NEG_INF)[None, None]
        add_mask = cm if add_mask is None else add_mask + cm
    attn = multi_query_attention if f'{p}.attn.q_proj.weight' in sd else self_attention
    x = x + attn(sd, f'{p}.attn', layer_norm(x, sd[f'{p}.ln_1.weight'], _get(sd, f'{p}.ln_1.bias')),
                 n_head, add_mask, dropout, attn_dropout, training, plan, layer)
    if mem is not None:
        if f'{p}.cross_attn.in_proj_weight' not in sd:
            raise ValueError('Model not configured for cross attn inputs!!!')        # layers.py:598-599
        x = x + cross_attention(sd, f'{p}.cross_attn',
                                layer_norm(x, sd[f'{p}.ln_3.weight'], _get(sd, f'{p}.ln_3.bias')), mem, n_head,
                                dropout, training, plan, layer)
    h2 = layer_norm(x, sd[f'{p}.ln_2.weight'], _get(sd, f'{p}.ln_2.bias'))
    if f'{p}.mlp.c_fc.experts.0.l1.weight' in sd:
        x = x + moe_mlp(sd, f'{p}.mlp', h2, top_k, dropout, training, plan, layer, moe_io)
    else:
        x = x + gelu_mlp(sd, f'{p}.mlp', h2, dropout, training, plan, layer)
    x = _UnitNormGrad.apply(x)
    if idx is None:
        return x
    out = torch.zeros_like(x_orig)
    out[:, idx] = x
    out[:, not_idx] = x_orig[:, not_idx] + null(x_orig[:, not_idx])
    return out


# --------------------------------------------------------------------------------------------------------------
# encoder (models/encoder.py:130-195) and decoder (models/decoder.py:161-282)
# --------------------------------------------------------------------------------------------------------------
def _sub(sd: SD, prefix: str) -> SD:
    n = len(prefix)
    return {k[n:]: v for k, v in sd.items() if k.startswith(prefix)}


class _PrefixedDict(dict):
    """View of a dict whose keys are stored with a prefix (moe_io sites are keyed by full parameter paths)."""

    def __init__(self, base: dict, prefix: str):
        super().__init__()
        self.base, self.prefix = base, prefix

    def get(self, k, default=None):
        return self.base.get(self.prefix + k, default)

    def __setitem__(self, k, v):
        self.base[self.prefix + k] = v


def _moe_sub(moe_io, prefix: str):
    if moe_io is None:
        return None
    return {k: _PrefixedDict(v, prefix) for k, v in moe_io.items()}


def _top_k(tcfg) -> int:
    return getattr(tcfg.rotator_config, 'top_k', 1)


def vit_encoder(sd: SD, cfg, images, training=False, plan=None, moe_io=None):
    """encoder.py:163-178.  ``sd`` keys are relative to the VisionTransformerEncoder module.

    conv stack -> FLAT reshape to (n, P^2, C*ph*pw) (a chunking of the contiguous CHW buffer, not spatial patches,
    encoder.py:166) -> projector -> LayerNormND -> +wpe -> the SAME LayerNormND again -> prepend CLS -> blocks ->
    ln_f on the CLS rows only.
    """
    ac = cfg.transformer_config.attn_config
    P2 = cfg.num_patches ** 2
    x = conv_stack(sd, 'feature_extractor', images)
    n = x.size(0)
    x = x.reshape(n, P2, -1)
    x = F.linear(x, sd['projector.weight'], _get(sd, 'projector.bias'))
    x = layer_norm(x, sd['ln_input.weight'], _get(sd, 'ln_input.bias'))
    x = x + sd['transformer.wpe.weight'][:P2].unsqueeze(0)
    x = layer_norm(x, sd['ln_input.weight'], _get(sd, 'ln_input.bias'))
    x = torch.cat((sd['cls_token'].expand(n, -1, -1), x), dim=1)
    x = _planned(plan.get(0, 'emb'), x) if plan is not None else _drop(x, ac.dropout, training)
    for i in range(cfg.n_layer):
        x = transformer_block(sd, f'transformer.h.{i}', x, ac.n_head, cfg.transformer_config.is_causal, None, None,
                              ac.dropout, ac.attn_dropout, training, plan, i, _top_k(cfg.transformer_config), moe_io)
    return layer_norm(x[:, :cfg.n_cls].contiguous(), sd['transformer.ln_f.weight'], _get(sd, 'transformer.ln_f.bias'))


def encode(sd: SD, cfg, images, training=False, plan=None, moe_io=None):
    """vision_encoder_decoder.py:26-39,58-59: encoder, then the bias-free bridge Linear when the widths differ
    (state-dict keys then carry the nn.Sequential prefixes ``encoder.0.`` / ``encoder.1.``)."""
    if not hasattr(cfg.vision_encoder_config, 'transformer_config'):      # PretrainedViTConfig: oracle/vit.py (encoder.py:56-127)
        from . import vit
        if 'encoder.1.weight' in sd:
            return F.linear(vit.pretrained_vit(_sub(sd, 'encoder.0.'), cfg.vision_encoder_config, images), sd['encoder.1.weight'])
        return vit.pretrained_vit(_sub(sd, 'encoder.'), cfg.vision_encoder_config, images)
    if 'encoder.1.weight' in sd:
        y = vit_encoder(_sub(sd, 'encoder.0.'), cfg.vision_encoder_config, images, training, plan, _moe_sub(moe_io, 'encoder.0.'))
        return F.linear(y, sd['encoder.1.weight'])
    return vit_encoder(_sub(sd, 'encoder.'), cfg.vision_encoder_config, images, training, plan, _moe_sub(moe_io, 'encoder.'))


def gpt_decoder(sd: SD, cfg, idx=None, inputs_embeds=None, cross_attn_embeds=None, attn_msk=None, training=False, plan=None,
                pos_offset=0, moe_io=None):
    """decoder.py:214-256.  ``sd`` keys relative to TransformerDecoder.  Returns (logits, hidden)."""
    assert (idx is None) != (inputs_embeds is None)
    ac = cfg.transformer_config.attn_config
    if inputs_embeds is None:
        inputs_embeds = sd['transformer.wte.weight'][idx]
    t = inputs_embeds.size(1)
    assert t <= cfg.block_size, f'Cannot forward sequence of length {t}, block size is only {cfg.block_size}'
    if 'transformer.wpe.models.0.model.0.weight' in sd:
        # decoder.py:231-232 + layers.py:617-638 AdvancedPositionalBiasMLP: position p has its OWN MLP (Linear [GELU Linear]*, biased)
        # with an identity residual connector: x_p = MLP_p(e_p) + e_p
        x = torch.stack([gate_mlp(sd, f'transformer.wpe.models.{pos_offset + p}', inputs_embeds[..., p, :]) + inputs_embeds[..., p, :]
                         for p in range(t)], dim=-2)
    else:
        x = inputs_embeds + sd['transformer.wpe.weight'][pos_offset:pos_offset + t]
    x = _planned(plan.get(0, 'emb'), x) if plan is not None else _drop(x, ac.dropout, training)
    for depth in range(cfg.n_layer):
        mem = cross_attn_embeds if (depth % 2 == 0 or not cfg.skip_alternate_cross_attn) else None
        x = transformer_block(sd, f'transformer.h.{depth}', x, ac.n_head, cfg.transformer_config.is_causal, mem,
                              attn_msk, ac.dropout, ac.attn_dropout, training, plan, depth, _top_k(cfg.transformer_config), moe_io,
                              pos_offset)
    x = layer_norm(x, sd['transformer.ln_f.weight'], _get(sd, 'transformer.ln_f.bias'))
    return F.linear(x, sd['transformer.wte.weight']), x          # lm_head is tied to wte (decoder.py:189-204)


# --------------------------------------------------------------------------------------------------------------
# glue (models/vision_encoder_decoder.py:51-134)
# --------------------------------------------------------------------------------------------------------------
def expand_user_mask(attn_msk, bs: int):
    """vision_encoder_decoder.py:61-72: bool mask -> (bs|1, h|1, s, l).  A 2-D (bs, s) mask is broadcast along the
    KEY axis, i.e. it masks query rows."""
    if attn_msk is None:
        return None
    if attn_msk.dim() == 2:
        s = attn_msk.size(1)
        if attn_msk.size(0) == bs:
            return attn_msk[:, None, :, None].expand(bs, 1, s, s)
        return attn_msk[None, None].expand(bs, 1, *attn_msk.shape)
    if attn_msk.dim() == 3:
        if attn_msk.size(0) == bs:
            return attn_msk[:, None]
        return attn_msk[None].expand(bs, *attn_msk.shape)
    return attn_msk


def _mask_to_additive(allowed):
    """vision_encoder_decoder.py:97-98 / 118-119 as they actually execute: ``bool_mask.masked_fill(~bool_mask, -inf)``
    fills a BOOL tensor, so -inf is cast to True; after ``.float()`` every entry is 1.0 and the next line rewrites
    1 -> 0.  The additive image of the (user AND causal) mask is therefore ALL ZEROS: user masks are shape-checked
    but numerically inert (golden fixtures row_mask/sl_mask/bsl_mask == nomask bit for bit), and causality comes
    only from TransformerBlock (layers.py:581-595)."""
    filled = allowed.masked_fill(~allowed, NEG_INF).float()      # bool fill: all True -> all 1.0
    filled[filled == 1] = 0
    return filled


def forward(sd: SD, cfg, images, ids, attn_msk=None, encoder_output=None, training=False, moe_io=None):
    """VisionEncoderDecoder.forward -> (encoder_output, logits, hidden_state)."""
    dcfg = cfg.decoder_config
    dio = _moe_sub(moe_io, 'decoder.')
    if encoder_output is None:
        encoder_output = encode(sd, cfg, images, training, moe_io=moe_io)
    bs, ncls, _ = encoder_output.shape
    L = ids.size(-1)
    allowed = torch.ones((L, L), dtype=torch.bool).tril()[None, None]
    user = expand_user_mask(attn_msk, bs)
    if user is not None:
        allowed = torch.logical_and(user, allowed)
    dsd = _sub(sd, 'decoder.')
    if cfg.use_soft_prompting:
        emb = torch.cat((encoder_output, dsd['transformer.wte.weight'][ids]), dim=-2)[..., :dcfg.block_size, :]
        h, s = allowed.size(1), L
        add = torch.full((bs, h, ncls + s, ncls + s), NEG_INF)
        add[..., :ncls, :] = 0                                  # prompt rows see every column (:93-95)
        add[..., ncls:, ncls:] = _mask_to_additive(allowed)     # (:96-99)
        add = add[..., :dcfg.block_size, :dcfg.block_size]      # text rows never see prompt columns
        logits, hidden = gpt_decoder(dsd, dcfg, inputs_embeds=emb,
                                     cross_attn_embeds=encoder_output if cfg.use_cross_attn else None,
                                     attn_msk=add, training=training, moe_io=dio)
        return encoder_output, logits[..., ncls:, :], hidden
    add = _mask_to_additive(allowed)                            # (:117-119)
    logits, hidden = gpt_decoder(dsd, dcfg, idx=ids,
                                 cross_attn_embeds=encoder_output if cfg.use_cross_attn else None,
                                 attn_msk=add, training=training, moe_io=dio)
    return encoder_output, logits, hidden


def lm_step_text_segment(sd: SD, cfg, images, labels, tokenizer, plans=(None, None), ignore_index=-100, temperature=1.0, moe_io=None):
    """The factorisation the HIP path runs (engine.py): because text rows never see prompt columns and prompt-row
    logits are sliced off, the loss only needs the TEXT segment -- a plain causal pass over the ids with the position
    embedding offset by n_cls and cross-attention on the encoder output.  Without dropout this equals ``lm_step``
    (test_oracle_golden); with ``plans`` = (encoder DropPlan, decoder DropPlan) it applies the HIP path's exact masks."""
    ids, _ = shifted_inputs(labels, tokenizer.bos_token_id, tokenizer.eos_token_id, ignore_index)
    enc = encode(sd, cfg, images, True, plans[0], moe_io)
    off = enc.size(1) if cfg.use_soft_prompting else 0
    logits, _ = gpt_decoder(_sub(sd, 'decoder.'), cfg.decoder_config, idx=ids,
                            cross_attn_embeds=enc if cfg.use_cross_attn else None, attn_msk=None, training=True,
                            plan=plans[1], pos_offset=off, moe_io=_moe_sub(moe_io, 'decoder.'))
    w = loss_weights(labels, ignore_index)
    ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)) / temperature, labels.reshape(-1), ignore_index=ignore_index,
                         reduction='none')
    return (ce * w.reshape(-1)).sum()


# --------------------------------------------------------------------------------------------------------------
# training step (training/wrapper.py:80-96,120-151,153-214; default ``trainer: {}`` path = weighted CE only)
# --------------------------------------------------------------------------------------------------------------
def loss_weights(labels, ignore_index=-100, weight_fn='constant', eos_token_id=None, eos_token_weight=None):
    """wrapper.py:80-96: per-token weights, normalised per sequence (1e-3 in the denominator) and divided by B."""
    if weight_fn == 'constant':
        w = torch.ones_like(labels, dtype=torch.float)
    elif weight_fn == 'inverse_sqrt_position':
        w = (1.0 / torch.sqrt(torch.arange(1, labels.size(1) + 1, dtype=torch.float))).expand(labels.size(0), -1).clone()
    else:
        raise ValueError(f'unknown weight_fn: {weight_fn}')
    if eos_token_weight is not None:
        w[labels == eos_token_id] = eos_token_weight
    w[labels == ignore_index] = 0.0
    return (w / (1e-3 + w.sum(dim=-1, keepdim=True))) / w.size(0)


def shifted_inputs(labels, bos: int, eos: int, ignore_index=-100):
    """wrapper.py:154-159,185-196: ids = labels with ignore->EOS, BOS prepended, last dropped; same for the mask."""
    ids = torch.where(labels != ignore_index, labels, torch.full_like(labels, eos))
    msk = labels != ignore_index
    bs, sl = ids.shape
    ids = torch.cat((torch.full((bs, 1), bos, dtype=torch.long), ids), dim=1)[:, :sl]
    msk = torch.cat((torch.ones((bs, 1), dtype=torch.bool), msk), dim=1)[:, :sl]
    return ids, msk


def contrastive_loss(hidden_state, labels, wte, ignore_index=-100, temperature=1.0, weight_fn='constant', eos_token_id=None,
                     eos_token_weight=None):
    """wrapper.py:98-118: row i of hidden_state (flattened over batch and position; with a soft prompt the first n_cls positions
    are the PROMPT rows) must pick out the target embedding wte[label_i] among the embeddings of every labelled position of the
    batch; ignored positions are masked out as columns and carry zero weight as rows (their infinite loss is replaced by 0)."""
    labels = labels[..., :hidden_state.size(-2)]
    hidden_state = hidden_state[..., :labels.size(-1), :]
    w = loss_weights(labels, ignore_index, weight_fn, eos_token_id, eos_token_weight)
    keep = labels != ignore_index
    target = wte[torch.where(keep, labels, torch.zeros_like(labels))]
    pred = hidden_state.reshape(-1, hidden_state.size(-1)) @ target.reshape(-1, target.size(-1)).t()
    pred = torch.where(keep.reshape(1, -1), pred, torch.full_like(pred, NEG_INF))
    losses = F.cross_entropy(pred / temperature, torch.arange(pred.size(0)), reduction='none')
    losses = torch.where(losses.isinf(), torch.zeros_like(losses), losses)
    return (losses * w.reshape(-1)).sum()


def lm_step(sd: SD, cfg, images, labels, tokenizer, training: bool, ignore_index=-100, temperature=1.0,
            weight_fn='constant', eos_token_weight=None, moe_io=None, contrastive_temperature=None, return_parts=False):
    """ModelTrainerWrapper.train_step / val_step -> scalar loss: weighted cross-entropy, plus the contrastive term when
    contrastive_temperature is given (trainer option add_contrastive_loss, wrapper.py:206-209)."""
    ids, msk = shifted_inputs(labels, tokenizer.bos_token_id, tokenizer.eos_token_id, ignore_index)
    _, logits, hidden = forward(sd, cfg, images, ids, msk, training=training, moe_io=moe_io)
    lab = labels[..., :logits.size(-2)]
    logits = logits[..., :lab.size(-1), :]
    w = loss_weights(lab, ignore_index, weight_fn, tokenizer.eos_token_id, eos_token_weight)
    ce = F.cross_entropy(logits.reshape(-1, logits.size(-1)) / temperature, lab.reshape(-1),
                         ignore_index=ignore_index, reduction='none')
    loss = (ce * w.reshape(-1)).sum()
    if contrastive_temperature is None:
        return loss
    lc = contrastive_loss(hidden, labels, sd['decoder.transformer.wte.weight'], ignore_index, contrastive_temperature, weight_fn,
                          tokenizer.eos_token_id, eos_token_weight)
    return (loss + lc, loss, lc) if return_parts else loss + lc


# --------------------------------------------------------------------------------------------------------------
# greedy decode = generate(top_k=1, temperature=1) (vision_encoder_decoder.py:136-182) with the HF
# NoRepeatNGramLogitsProcessor (transformers 5.15.0 logits_process.py:1021-1070) restated
# --------------------------------------------------------------------------------------------------------------
def banned_next_tokens(row: Sequence[int], n: int):
    """Tokens t such that (last n-1 ids)+(t,) already occurs in ``row``; nothing is banned while len+1 < n."""
    cur = len(row)
    if cur + 1 < n:
        return []
    tail = tuple(row[cur + 1 - n:])
    return [row[i + n - 1] for i in range(cur - n + 1) if tuple(row[i:i + n - 1]) == tail]


def apply_ngram_ban(ids, logits, ngram_sizes):
    for b, row in enumerate(ids.tolist()):
        for n in ngram_sizes:
            banned = banned_next_tokens(row, n)
            if banned:
                logits[b, banned] = NEG_INF
    return logits


@torch.no_grad()
def generate_greedy(sd: SD, cfg, images, prompt_ids, max_new_tokens: int, return_margins=False):
    """Cache-free loop exactly as the reference runs it: full re-forward per token, ban, argmax, append."""
    blk = cfg.decoder_config.block_size - (cfg.vision_encoder_config.n_cls if cfg.use_soft_prompting else 0)
    assert max_new_tokens <= blk - prompt_ids.size(-1)
    enc, ids, margins = None, prompt_ids, []
    for _ in range(max_new_tokens):
        cond = ids if ids.size(-1) <= blk else ids[..., -blk:]
        enc, logits, _ = forward(sd, cfg, images, cond, None, encoder_output=enc)
        last = apply_ngram_ban(ids, logits[..., -1, :].clone(), cfg.no_repeat_n_grams)
        if return_margins:
            t2 = torch.topk(last, 2, dim=-1).values
            margins.append(t2[:, 0] - t2[:, 1])
        ids = torch.cat((ids, last.argmax(dim=-1, keepdim=True)), dim=-1)
    return (ids, torch.stack(margins, dim=1)) if return_margins else ids


# --------------------------------------------------------------------------------------------------------------
# sampling modes of generate() (vision_encoder_decoder.py:150-180): the distribution the next token is drawn from
# --------------------------------------------------------------------------------------------------------------
def sampling_distribution(last_logits, ids, ngram_sizes, temperature=1.0, top_k=None, nucleus_p=None):
    """``last_logits`` (B, V) of the final position, ``ids`` (B, t) everything decoded so far -> (B, V) probabilities in
    vocabulary order, zero outside the kept set.  Order of operations as the reference: / temperature (:152), n-gram ban
    (:153), top-k crop keeping ties at the k-th value (:155-157), softmax (:159), nucleus: sort descending, drop every
    entry whose cumulative mass exceeds max(nucleus_p, largest probability), renormalise (:160-171)."""
    logits = last_logits.clone().float() / temperature
    logits = apply_ngram_ban(ids, logits, ngram_sizes)
    if top_k is not None:
        kth = torch.topk(logits, min(top_k, logits.size(-1)), dim=-1).values[:, -1:]
        logits[logits < kth] = NEG_INF
    probs = logits.softmax(dim=-1)
    if nucleus_p is None:
        return probs
    sp, si = torch.sort(probs, descending=True, dim=-1)
    cum = torch.cumsum(sp, dim=-1)
    thr = torch.maximum(torch.full_like(sp[:, 0], nucleus_p), sp[:, 0]).unsqueeze(1)
    sp = sp.masked_fill(cum > thr, 0.0)
    sp = sp / sp.sum(dim=-1, keepdim=True)
    return torch.zeros_like(probs).scatter_(1, si, sp)


def inverse_cdf_token(dist, u):
    """The HIP sampler's draw rule (not the reference's Philox multinomial, which cannot be matched): walk the kept
    distribution in vocabulary order and take the first token whose cumulative mass reaches u * total (u in [0, 1))."""
    cdf = torch.cumsum(dist.double(), dim=-1)
    target = (u.double() * cdf[:, -1]).unsqueeze(1)
    tok = (cdf <= target).sum(dim=-1)
    kept = dist > 0
    last_kept = (kept * torch.arange(1, dist.size(1) + 1)).amax(dim=1) - 1
    return torch.minimum(tok, last_kept)


# --------------------------------------------------------------------------------------------------------------
# BeamSearchTokenGenerator (models/generation_utils.py:10-148)
# --------------------------------------------------------------------------------------------------------------
@torch.no_grad()
def beam_search(sd: SD, cfg, images, prompt_ids, beam_width=3, temperature=1.0, top_k=None, max_new_tokens=64, no_repeat_n_grams=(2, 3, 4),
                beam_expansion_factor=4, eos_token_id=None, consolidation_temperature=1.0, length_boost=1.0, draw=None):
    """-> (ids (B, W, L), cumulative log scores (B, W)).  ``draw(probs, n)`` stands in for torch.multinomial (the tests replay the
    draws the reference recorded).  Rows are kept beam-major (W, B) as the reference keeps them (:38-42), so that replayed draws
    line up call by call: per step one draw of E candidates per (beam, caption) row (:72) and one of W survivors per caption (:143).

    Step (:56-93): last-position logits -> n-gram ban -> top-k crop (strictly below the k-th value) -> candidates: temperature <= 0
    the E largest raw scores with log_softmax(scores), else E draws without replacement from softmax(scores / temperature);
    a beam whose LAST token is EOS re-emits EOS at log-score 0 for every candidate whose boosted log-score is negative, all
    other candidates get + log(length_boost).  Consolidation (:95-148): of the W x E candidates of a caption (beam-major) keep W --
    the top-W cumulative scores, sorted (consolidation temperature <= 0), or W draws from softmax(cumulative / temperature).
    Loop test (:46-47): stop at max_new_tokens + provided - 1 ... tokens or when EVERY beam contains EOS anywhere (prompt included)."""
    draw = draw or torch.multinomial
    W, E, lb = beam_width, beam_expansion_factor, math.log(length_boost)
    B = images.shape[0]
    enc = encode(sd, cfg, images)                                            # (B, n_cls, d)
    mem = enc.repeat(W, 1, 1)                                                # beam-major rows w * B + b
    provided = prompt_ids.size(-1) - 1
    beams = prompt_ids.unsqueeze(0).expand(W, -1, -1)                        # (W, B, L)
    cum = torch.zeros(W, B)
    while not (beams.size(-1) >= max_new_tokens + provided or bool(((beams == eos_token_id).sum(dim=-1) > 0).all())):
        flat = beams.reshape(W * B, -1)
        ended = (flat[:, -1:] == eos_token_id)
        _, logits, _ = forward(sd, cfg, None, flat, None, encoder_output=mem)
        scores = apply_ngram_ban(flat, logits[:, -1, :].clone(), no_repeat_n_grams)
        if top_k is not None:
            kth = torch.topk(scores, min(top_k, scores.size(-1)), dim=-1).values[:, -1:]
            scores[scores < kth] = NEG_INF
        if temperature <= 0:
            logp = scores.log_softmax(dim=-1)
            nxt = scores.topk(k=E, dim=-1).indices
        else:
            logp = (scores / temperature).log_softmax(dim=-1)
            nxt = draw(logp.exp(), E)
        lp = logp.gather(-1, nxt)
        stay = ended & (lp + lb < 0)
        nxt = torch.where(stay, torch.full_like(nxt, eos_token_id), nxt)
        lp = torch.where(stay, torch.zeros_like(lp), lp + lb)
        nxt, lp = nxt.view(W, B, E), lp.view(W, B, E)
        total = (cum.unsqueeze(2) + lp).permute(1, 0, 2).reshape(B, W * E)   # candidate w * E + e of caption b
        if consolidation_temperature <= 0:
            pick = total.topk(k=W, dim=-1).indices
        else:
            pick = draw((total / consolidation_temperature).softmax(dim=-1), W)
        bw, ce = pick // E, pick % E                                         # (B, W): surviving beam / its candidate
        ar = torch.arange(B).unsqueeze(1)
        beams = torch.cat((beams.permute(1, 0, 2)[ar, bw], nxt.permute(1, 0, 2)[ar, bw, ce].unsqueeze(-1)), dim=-1).permute(1, 0, 2)
        cum = (cum.t()[ar, bw] + lp.permute(1, 0, 2)[ar, bw, ce]).t()
    return beams.permute(1, 0, 2), cum.t()


# --------------------------------------------------------------------------------------------------------------
# SNRAdam (models/optimizer.py:56-113)
# --------------------------------------------------------------------------------------------------------------
def snradam_step(param, grad, state, lr, betas, weight_decay, eps):
    """One step on one tensor, in place; ``state`` = {} before the first step.  Decoupled decay first (:86-87); the
    second moment tracks (g - bias-corrected previous mean)^2 (:98-108); update m_hat / (sqrt(v_hat) + eps) (:110-111)."""
    b1, b2 = betas
    if weight_decay != 0:
        param.mul_(1 - lr * weight_decay)
    if not state:
        state.update(t=1, m=torch.zeros_like(param), v=torch.zeros_like(param))
    t, m, v = state['t'], state['m'], state['v']
    dev = grad - (m if t == 1 else m / (1 - b1 ** (t - 1)))
    m.mul_(b1).add_(grad, alpha=1 - b1)
    v.mul_(b2).add_(dev * dev, alpha=1 - b2)
    param.addcdiv_(m / (1 - b1 ** t), (v / (1 - b2 ** t)).sqrt() + eps, value=-lr)
    state['t'] = t + 1
    return param


# --------------------------------------------------------------------------------------------------------------
# trainer extras (training/wrapper.py:46-59,134-144,161-196): MLM corruption of the decoder inputs, momentum distillation
# --------------------------------------------------------------------------------------------------------------
def lm_inputs(labels, bos: int, eos: int, ignore_index=-100, mask_id=None, mask_fraction=0.0, random_fraction=0.0, u_mask=None,
              u_rand=None, r_ids=None):
    """Decoder inputs of a step (wrapper.py:154-196): labelled token -> itself, or with probability mask_fraction (draw u_mask) the
    MASK id -- or, for a share random_fraction of those (draw u_rand), the random id r_ids; ignored -> EOS; then BOS in front, last
    position dropped.  The draws are explicit arguments (torch's RNG stream is not part of the contract)."""
    ids = torch.where(labels != ignore_index, labels, torch.full_like(labels, eos))
    if mask_fraction > 0:
        repl = torch.where(u_rand <= random_fraction, r_ids, torch.full_like(labels, mask_id))
        ids = torch.where(u_mask <= mask_fraction, repl, ids)
        ids = torch.where(labels != ignore_index, ids, torch.full_like(labels, eos))
    bs, sl = ids.shape
    return torch.cat((torch.full((bs, 1), bos, dtype=torch.long), ids), dim=1)[:, :sl]


def lm_step_distill(sd: SD, sd_m: SD, cfg, images, labels, tokenizer, alpha: float, ignore_index=-100, temperature=1.0, ids=None):
    """train_step with momentum distillation (wrapper.py:134-144,197-214): the momentum model's logits (no gradient) give soft targets
    alpha * softmax(z_m / T) + (1 - alpha) * onehot(label) (all-zero onehot for ignored labels, whose weight is zero anyway)."""
    if ids is None:
        ids, _ = shifted_inputs(labels, tokenizer.bos_token_id, tokenizer.eos_token_id, ignore_index)
    _, logits, _ = forward(sd, cfg, images, ids, None, training=True)
    with torch.no_grad():
        _, logits_m, _ = forward(sd_m, cfg, images, ids, None, training=True)
    labels = labels[..., :logits.size(-2)]
    logits, logits_m = logits[..., :labels.size(-1), :], logits_m[..., :labels.size(-1), :]
    w = loss_weights(labels, ignore_index)
    V = logits.size(-1)
    onehot = F.one_hot(torch.where(labels == ignore_index, torch.full_like(labels, V), labels), V + 1)[..., :-1]
    target = alpha * F.softmax(logits_m / temperature, dim=-1) + (1 - alpha) * onehot
    return -((F.log_softmax(logits / temperature, dim=-1) * target).sum(dim=-1) * w).sum()


def ema(p_m, p, momentum: float):
    """wrapper.py:52-59"""
    return p_m * momentum + p * (1.0 - momentum)
