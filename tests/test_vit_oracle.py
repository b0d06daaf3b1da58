"""Pin oracle/vit.py (the PretrainedViT encoder, reference models/encoder.py:56-127): the three heads against fixtures made by the
reference's own head modules (tools/gen_goldens_vit.py), the torchvision backbone restatement against two independent
implementations of the same architecture (transformers' ViTModel; torch's own nn.MultiheadAttention / nn.LayerNorm / nn.GELU run
through this package's parameter tree).  CPU only."""
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from image2text_amd.configs.models import PretrainedViTConfig
from image2text_amd.synth import det_init_
from oracle import vit as ovit

os.environ.setdefault('I2T_VIT_B16_CHECKPOINT', 'random')
SMALL = dict(image_size=32, patch_size=16, num_layers=2, num_heads=2, hidden_dim=128, mlp_dim=256)
HEAD_CASES = {      # tools/gen_goldens_vit.py::CASES
    'mlp_res': dict(n_cls=4, n_embd_out_vit=96, gate_sizes=(64,), refine_base_model=True),
    'mlp_id': dict(n_cls=3, n_embd_out_vit=768, gate_sizes=(32,), refine_base_model=True),
    'mlp_deep': dict(n_cls=2, n_embd_out_vit=64, gate_sizes=(64, 32), refine_base_model=True),
    'peer': dict(n_cls=3, n_embd_out_vit=64, refine_base_model=True, peer_config=dict(num_units_sqrt=16, topk=4, nhead=2, query_dim=32)),
    'lsh': dict(n_cls=3, n_embd_out_vit=64, refine_base_model=True, lsh_config=dict(num_bins=(4, 8, 20), num_proj=32, learnable=False)),
}


def head_module(name, gold, spec=None):
    """This package's PretrainedViT with the fixture's head weights: det_init_ seed 7 (same names as the reference's module ->
    same draws), the PEER tables rescaled as the generator did, the LSH buffers from the fixture."""
    from image2text_amd.models.encoder import PretrainedViT

    class _Enc(PretrainedViT):
        backbone_spec = spec or dict(image_size=32, patch_size=16, num_layers=1, num_heads=12, hidden_dim=768, mlp_dim=64)
    enc = _Enc(PretrainedViTConfig.model_validate(HEAD_CASES[name]))
    det_init_(enc, seed=7)
    with torch.no_grad():
        if name == 'peer':
            enc.peer.emb_in.weight.mul_(4.0)
            enc.peer.emb_out.weight.mul_(10.0)
        for n, b in enc.named_buffers():
            if f'{name}.buffer.{n}' in gold:
                b.copy_(torch.from_numpy(gold[f'{name}.buffer.{n}']))
    return enc


def summarise(t):
    t = t.detach().reshape(-1)
    return np.concatenate(([float(t.norm())], t[:256].numpy()))


@pytest.mark.parametrize('name', list(HEAD_CASES))
def test_heads_match_the_reference(name):
    g = load_golden('vit_heads.npz')
    enc = head_module(name, g)
    cfg = enc.config
    sd = {k: v.detach().clone().requires_grad_(v.dtype.is_floating_point) for k, v in enc.state_dict().items()}
    feats = torch.from_numpy(g[f'{name}.features']).requires_grad_(True)
    y = ovit.pretrained_vit(sd, cfg, features=feats)
    assert y.shape == g[f'{name}.output'].shape
    assert float((y.detach() - torch.from_numpy(g[f'{name}.output'])).abs().max()) <= 2e-5 * max(1.0, float(np.abs(g[f'{name}.output']).max()))
    (y * torch.from_numpy(g[f'{name}.G'])).sum().backward()
    if f'{name}.grad_features' in g:
        ref = g[f'{name}.grad_features']
        assert float(np.abs(feats.grad.numpy() - ref).max()) <= 1e-4 * max(1e-3, float(np.abs(ref).max()))
    else:
        assert feats.grad is None or float(feats.grad.abs().max()) == 0.0          # LSH: bucketize passes no gradient
    checked = 0
    for key in g:
        if not key.startswith(f'{name}.grad.'):
            continue
        pname = key[len(f'{name}.grad.'):]
        got, ref = sd[pname].grad, g[key]
        got = summarise(got) if (ref.ndim == 1 and ref.shape[0] == 257 and sd[pname].numel() > 200_000) else got.numpy()
        assert float(np.abs(got - ref).max()) <= 1e-4 * max(1e-3, float(np.abs(ref).max())), pname
        checked += 1
    assert checked >= 4


def _tv_state(spec, seed=3):
    from image2text_amd.models.encoder import TorchvisionViT
    m = TorchvisionViT(spec)
    det_init_(m, seed=seed)
    return m


def test_backbone_matches_transformers_vit():
    """transformers.ViTModel = an independent implementation of the architecture torchvision's vit_b_16 implements (pre-LN blocks,
    exact GELU, class token, learned positions); with eps 1e-6 and the weights mapped name by name it must agree with the restatement."""
    from transformers import ViTConfig, ViTModel
    spec = SMALL
    m = _tv_state(spec)
    sd = {f'model.{k}': v.detach() for k, v in m.state_dict().items()}
    d = spec['hidden_dim']
    hf = ViTModel(ViTConfig(hidden_size=d, num_hidden_layers=spec['num_layers'], num_attention_heads=spec['num_heads'],
                            intermediate_size=spec['mlp_dim'], hidden_act='gelu', hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0,
                            layer_norm_eps=1e-6, image_size=spec['image_size'], patch_size=spec['patch_size'], num_channels=3, qkv_bias=True),
                  add_pooling_layer=False).eval()
    h = {}
    h['embeddings.cls_token'] = sd['model.class_token']
    h['embeddings.position_embeddings'] = sd['model.encoder.pos_embedding']
    h['embeddings.patch_embeddings.projection.weight'] = sd['model.conv_proj.weight']
    h['embeddings.patch_embeddings.projection.bias'] = sd['model.conv_proj.bias']
    for i in range(spec['num_layers']):          # (key names of transformers 5.x)
        q, t = f'model.encoder.layers.encoder_layer_{i}.', f'layers.{i}.'
        w, b = sd[q + 'self_attention.in_proj_weight'], sd[q + 'self_attention.in_proj_bias']
        for j, nm in enumerate(('q_proj', 'k_proj', 'v_proj')):
            h[f'{t}attention.{nm}.weight'], h[f'{t}attention.{nm}.bias'] = w[j * d:(j + 1) * d], b[j * d:(j + 1) * d]
        h[t + 'attention.o_proj.weight'], h[t + 'attention.o_proj.bias'] = sd[q + 'self_attention.out_proj.weight'], sd[q + 'self_attention.out_proj.bias']
        h[t + 'layernorm_before.weight'], h[t + 'layernorm_before.bias'] = sd[q + 'ln_1.weight'], sd[q + 'ln_1.bias']
        h[t + 'layernorm_after.weight'], h[t + 'layernorm_after.bias'] = sd[q + 'ln_2.weight'], sd[q + 'ln_2.bias']
        h[t + 'mlp.fc1.weight'], h[t + 'mlp.fc1.bias'] = sd[q + 'mlp.0.weight'], sd[q + 'mlp.0.bias']
        h[t + 'mlp.fc2.weight'], h[t + 'mlp.fc2.bias'] = sd[q + 'mlp.3.weight'], sd[q + 'mlp.3.bias']
    h['layernorm.weight'], h['layernorm.bias'] = sd['model.encoder.ln.weight'], sd['model.encoder.ln.bias']
    hf.load_state_dict(h, strict=True)
    images = torch.randn(3, 3, spec['image_size'], spec['image_size'], generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        ref = hf(pixel_values=images).last_hidden_state[:, 0]
        got = ovit.vit_backbone(sd, images)
    assert float((ref - got).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))


def test_backbone_matches_torch_modules_in_the_parameter_tree():
    """The parameter tree holds real nn.MultiheadAttention / nn.LayerNorm(eps 1e-6) / nn.GELU modules under torchvision's names;
    composing THEM the way torchvision's forward does is a third statement of the same function."""
    spec = SMALL
    m = _tv_state(spec, seed=4).eval()
    sd = {f'model.{k}': v.detach() for k, v in m.state_dict().items()}
    images = torch.randn(2, 3, spec['image_size'], spec['image_size'], generator=torch.Generator().manual_seed(6))
    with torch.no_grad():
        x = torch.nn.functional.conv2d(images, m.conv_proj.weight, m.conv_proj.bias, stride=spec['patch_size'])
        x = x.reshape(2, spec['hidden_dim'], -1).permute(0, 2, 1)
        x = torch.cat((m.class_token.expand(2, -1, -1), x), dim=1) + m.encoder.pos_embedding
        for blk in m.encoder.layers:
            h = blk.ln_1(x)
            h, _ = blk.self_attention(h, h, h, need_weights=False)
            x = x + h
            y = blk.ln_2(x)
            for layer in blk.mlp:
                y = layer(y)
            x = x + y
        ref = m.encoder.ln(x)[:, 0]
        got = ovit.vit_backbone(sd, images)
    assert float((ref - got).abs().max()) <= 2e-5 * max(1.0, float(ref.abs().max()))
    assert ovit.spec_of(sd) == spec


def test_checkpoint_rule(tmp_path, monkeypatch):
    """A torchvision-format file is loaded (its classification head dropped), a missing one is an error, 'random' keeps the init."""
    from image2text_amd.models import encoder as E

    class _Enc(E.PretrainedViT):
        backbone_spec = SMALL
    src = _tv_state(SMALL, seed=9)
    sd = dict(src.state_dict())
    sd['heads.head.weight'], sd['heads.head.bias'] = torch.zeros(10, 128), torch.zeros(10)
    path = tmp_path / 'vit.pth'
    torch.save(sd, path)
    cfg = PretrainedViTConfig(n_cls=2, n_embd_out_vit=64, gate_sizes=(32,))
    monkeypatch.setenv('I2T_VIT_B16_CHECKPOINT', str(path))
    enc = _Enc(cfg)
    assert torch.equal(enc.model.conv_proj.weight, src.conv_proj.weight)
    monkeypatch.setenv('I2T_VIT_B16_CHECKPOINT', str(tmp_path / 'absent.pth'))
    with pytest.raises(FileNotFoundError):
        _Enc(cfg)
