"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

Plain-PyTorch fp32 restatement of the reference's ``PretrainedViT`` encoder (reference models/encoder.py:56-127) as pure
functions over a state dict whose keys are relative to the PretrainedViT module.

Parity status
  * heads (per-slot MLP + normalisation, PEER lookup, LSH cosine embeddings): PINNED -- ``tests/test_vit_oracle.py`` checks them
    against fixtures made by running the reference's own head modules on recorded 768-wide features
    (``tools/gen_goldens_vit.py`` -> ``tests/golden/vit_heads.npz``; the torchvision backbone is absent from the image, a
    stand-in module returning the recorded features takes its place there).
  * backbone (torchvision ``vit_b_16``): torchvision is a dependency of the reference that is NOT in the image (unpinned in the
    reference's requirements.txt), so its published ``VisionTransformer.forward`` is restated from the architecture it
    implements and pinned to an INDEPENDENT implementation of the same architecture, transformers' ``ViTModel``
    (``tests/test_vit_oracle.py::test_backbone_matches_transformers_vit``) -- "parity unpinned" with respect to torchvision
    itself.  What is restated (torchvision/models/vision_transformer.py, v0.12+ key names): conv_proj (p x p, stride p) ->
    [class_token | patches] + pos_embedding -> num_layers x EncoderBlock (x + attn(ln_1 x); x + mlp(ln_2 x) with
    mlp = Linear, exact-erf GELU, Linear; LayerNorm eps 1e-6; nn.MultiheadAttention with packed in_proj) -> encoder.ln -> row 0;
    ``heads`` is replaced by Identity (reference encoder.py:61).
"""
import math
from typing import Dict

import torch
import torch.nn.functional as F

SD = Dict[str, torch.Tensor]
VIT_B16 = dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072)


def spec_of(sd: SD, pfx: str = 'model.'):
    """The architecture constants a torchvision-format state dict implies (heads are 64 wide in every torchvision ViT)."""
    w = sd[f'{pfx}conv_proj.weight']
    L = 0
    while f'{pfx}encoder.layers.encoder_layer_{L}.ln_1.weight' in sd:
        L += 1
    side = int(round(math.sqrt(sd[f'{pfx}encoder.pos_embedding'].shape[1] - 1)))
    return dict(image_size=side * w.shape[-1], patch_size=w.shape[-1], num_layers=L, num_heads=w.shape[0] // 64, hidden_dim=w.shape[0],
                mlp_dim=sd[f'{pfx}encoder.layers.encoder_layer_0.mlp.0.weight'].shape[0])


def vit_backbone(sd: SD, images, spec=None, pfx: str = 'model.'):
    """-> features (B, hidden_dim): the class-token row after the final LayerNorm."""
    spec = spec or spec_of(sd, pfx)
    p, d, H = spec['patch_size'], spec['hidden_dim'], spec['num_heads']
    x = F.conv2d(images, sd[f'{pfx}conv_proj.weight'], sd[f'{pfx}conv_proj.bias'], stride=p)           # (B, d, h/p, w/p)
    B = x.shape[0]
    x = x.reshape(B, d, -1).permute(0, 2, 1)                                                           # (B, P^2, d), row-major patches
    x = torch.cat((sd[f'{pfx}class_token'].expand(B, -1, -1), x), dim=1) + sd[f'{pfx}encoder.pos_embedding']
    T = x.shape[1]
    for i in range(spec['num_layers']):
        q = f'{pfx}encoder.layers.encoder_layer_{i}.'
        h = F.layer_norm(x, (d,), sd[q + 'ln_1.weight'], sd[q + 'ln_1.bias'], 1e-6)
        qkv = F.linear(h, sd[q + 'self_attention.in_proj_weight'], sd[q + 'self_attention.in_proj_bias'])
        qq, kk, vv = (t.view(B, T, H, d // H).transpose(1, 2) for t in qkv.split(d, dim=-1))
        att = torch.softmax(qq @ kk.transpose(-1, -2) / math.sqrt(d // H), dim=-1) @ vv
        att = att.transpose(1, 2).reshape(B, T, d)
        x = x + F.linear(att, sd[q + 'self_attention.out_proj.weight'], sd[q + 'self_attention.out_proj.bias'])
        h = F.layer_norm(x, (d,), sd[q + 'ln_2.weight'], sd[q + 'ln_2.bias'], 1e-6)
        h = F.gelu(F.linear(h, sd[q + 'mlp.0.weight'], sd[q + 'mlp.0.bias']))
        x = x + F.linear(h, sd[q + 'mlp.3.weight'], sd[q + 'mlp.3.bias'])
    x = F.layer_norm(x, (d,), sd[f'{pfx}encoder.ln.weight'], sd[f'{pfx}encoder.ln.bias'], 1e-6)
    return x[:, 0]


def head_slot_mlp(sd: SD, feat, n_cls: int, pfx: str = 'proj.'):
    """encoder.py:118-119 + layers.py:222-255,617-638: normalize(x) -> one private MLP per slot (Linear [GELU-tanh Linear]* plus the
    residual connector: a Linear when the widths differ, else the identity) -> normalize over the output width.  -> (B, n_cls, out)"""
    x = F.normalize(feat, p=2.0, dim=-1)
    outs = []
    for s in range(n_cls):
        h, i = x, 0
        while f'{pfx}models.{s}.model.{i}.weight' in sd:
            h = F.linear(h, sd[f'{pfx}models.{s}.model.{i}.weight'], sd[f'{pfx}models.{s}.model.{i}.bias'])
            if f'{pfx}models.{s}.model.{i + 2}.weight' in sd:
                h = F.gelu(h, approximate='tanh')
            i += 2
        rw = sd.get(f'{pfx}models.{s}.residual_connector.weight')
        r = x if rw is None else F.linear(x, rw, sd[f'{pfx}models.{s}.residual_connector.bias'])
        outs.append(h + r)
    return F.normalize(torch.stack(outs, dim=-2), p=2.0, dim=-1)


def peer_lookup(sd: SD, inp, topk: int, nhead: int, pfx: str = 'peer.', trace=None):
    """layers.py:37-109 (product-key expert retrieval).  inp (B, S, in) -> (B, S, out).  Note the reference's expert index:
    ``left_index * topk + right_index`` (:93-96, topk -- not the number of query units -- as the stride), kept as is.
    trace (dict): receives 'final_indices' and the top-k margins the choice hangs on."""
    B, S, din = inp.shape
    qd = sd[f'{pfx}query_left.linear.weight'].shape[1]
    x = F.linear(inp, sd[f'{pfx}query_linear.weight']).view(B, S, nhead, qd)
    inp_proj = F.linear(inp, sd[f'{pfx}key_linear.weight']).view(B, S, nhead, din)
    residual = F.linear(inp, sd[f'{pfx}residual.weight'])
    ls, rs = F.linear(x, sd[f'{pfx}query_left.linear.weight']), F.linear(x, sd[f'{pfx}query_right.linear.weight'])
    left, right = torch.topk(ls, topk, dim=-1), torch.topk(rs, topk, dim=-1)
    cross = (left.values.unsqueeze(-1) + right.values.unsqueeze(-2)).view(B, S, nhead, topk * topk)
    y = torch.topk(cross, topk, dim=-1)
    scores = F.softmax(y.values, dim=-1)
    li = left.indices.gather(-1, y.indices // topk)
    ri = right.indices.gather(-1, y.indices % topk)
    final = li * topk + ri
    if trace is not None:
        trace['final_indices'] = final
        def margin(v, k):
            t = torch.topk(v, k + 1, dim=-1).values
            return (t[..., k - 1] - t[..., k])
        trace['margin_left'], trace['margin_right'] = margin(ls, topk), margin(rs, topk)
        trace['margin_cross'] = margin(cross, topk)
    e_in = F.embedding(final, sd[f'{pfx}emb_in.weight'])
    e_out = F.embedding(final, sd[f'{pfx}emb_out.weight'])
    act = F.gelu(torch.einsum('bshkd,bshd->bshk', e_in, inp_proj), approximate='tanh')
    return torch.einsum('bshk,bshkd->bsd', scores * act, e_out) + residual


def head_peer(sd: SD, feat, topk: int, nhead: int, trace=None):
    """encoder.py:115-116: peer(einsum('bd,des->bse', x, peer_proj_wt)) on the raw backbone feature."""
    return peer_lookup(sd, torch.einsum('bd,des->bse', feat, sd['peer_proj_wt']), topk, nhead, 'peer.', trace)


def lsh_bucket_ids(sd: SD, feat, slot: int, k: int):
    """layers.py:138-143 of CosineVectorEmbedding k of slot ``slot``: bucket index of every projection, before the table offset.
    -> (ids (B, n_proj) int64, z (B, n_proj) the cosine projections the buckets come from)"""
    q = f'lsh_emb.{slot}.emb.{k}.'
    z = F.normalize(feat, p=2.0, dim=-1) @ sd[q + 'projection_mat']
    return torch.bucketize(z, sd[q + 'grid']), z


def head_lsh(sd: SD, feat, n_cls: int):
    """encoder.py:117-118 + layers.py:112-143,190-219 (learnable = False): per slot, the sum over bin resolutions of
    EmbeddingBag(mean) of the rows [bucket(cos projection j) + (num_bins + 1) j].  -> (B, n_cls, out)"""
    outs = []
    for s in range(n_cls):
        acc, k = None, 0
        while f'lsh_emb.{s}.emb.{k}.emb.weight' in sd:
            q = f'lsh_emb.{s}.emb.{k}.'
            ids, _ = lsh_bucket_ids(sd, feat, s, k)
            rows = ids + sd[q + 'pos_offset'].view(1, -1)
            e = F.embedding(rows, sd[q + 'emb.weight']).mean(dim=1)
            acc = e if acc is None else acc + e
            k += 1
        outs.append(acc)
    return torch.stack(outs, dim=1)


def pretrained_vit(sd: SD, cfg, images=None, spec=None, features=None, trace=None):
    """PretrainedViT.forward (encoder.py:109-119); ``features`` short-cuts the backbone (head-only checks).  The backbone output
    is detached unless ``refine`` (:110-114; LSH forces refine off, :73)."""
    use_peer = cfg.peer_config is not None
    use_lsh = (not use_peer) and cfg.lsh_config is not None
    refine = cfg.refine_base_model and not use_lsh
    if features is None:
        if refine:
            features = vit_backbone(sd, images, spec)
        else:
            with torch.no_grad():
                features = vit_backbone(sd, images, spec)
    if use_peer:
        return head_peer(sd, features, cfg.peer_config.topk, cfg.peer_config.nhead, trace)
    if use_lsh:
        if cfg.lsh_config.learnable:
            raise NotImplementedError('learnable LSH embeddings are not restated')
        return head_lsh(sd, features, cfg.n_cls)
    return head_slot_mlp(sd, features, cfg.n_cls)
