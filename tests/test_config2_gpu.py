"""BASELINE.json configs[2] as ONE model on the MI355X: the torchvision-shaped ViT-B/16 ``PretrainedViT`` (12 x 768, 12 heads, 224 x 224 /
16 -> 197 tokens; reference models/encoder.py:56-127) in front of ``GPT2HuggingfaceDecoder`` at gpt2-small's shape (12 x 768, 12 heads,
vocabulary 50257 + 2, cross-attention + soft prompt; reference models/decoder.py:285-382, training_configs/local/gpt2.yaml) -- frozen
and refined backbone, with and without the yaml's LoRA spec.  Random initialisation (no checkpoint can be fetched here); the checker is
the composition the reference runs: oracle/vit.py backbone + oracle head (pinned in tests/test_vit_oracle.py) into transformers' own
GPT2LMHeadModel on the CPU in fp32 (+ peft's LoRA layer restated around the targeted Conv1Ds), composed as
vision_encoder_decoder.py:74-134 composes them.  Logits, the trainer's loss, every trainable gradient, and the first greedy tokens."""
import os

import pytest
import torch
import torch.nn.functional as F

from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch
from test_host_cpu import _hf_decoder_config, _local_hf_gpt2, _lora_spec
from test_hf_decoder_gpu import _hf_twin, _lora_twin, _reference_forward
from test_model_gpu import grad_close
from test_vit_gpu import vit_model_config

pytestmark = pytest.mark.gpu
REPORT = {}
VIT_B16 = dict(image_size=224, patch_size=16, num_layers=12, num_heads=12, hidden_dim=768, mlp_dim=3072)
V = 50257


@pytest.fixture(scope='module', autouse=True)
def write_report():
    yield
    import json
    import test_model_gpu
    REPORT.update({k: v for k, v in test_model_gpu.REPORT.items() if k.startswith('grad.config2')})
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_config2.json', 'w') as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


def dev():
    return torch.device('cuda:0')


@pytest.mark.timeout(900)
@pytest.mark.parametrize('refine,lora', [(False, False), (True, False), (False, True), (True, True)],
                         ids=['frozen', 'refine', 'frozen+lora', 'refine+lora'])
def test_vit_b16_with_gpt2_small_decoder_as_one_model(tmp_path, monkeypatch, refine, lora):
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.models.decoder import Decoder
    from image2text_amd.models.encoder import PretrainedViT
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    from oracle import reference_model as orc
    tag = f'config2.{"refine" if refine else "frozen"}{".lora" if lora else ""}'
    monkeypatch.setenv('I2T_VIT_B16_CHECKPOINT', 'random')
    _local_hf_gpt2(tmp_path, monkeypatch, name='gpt2-small-rand', n_layer=12, n_head=12, n_embd=768, n_positions=1024, vocab_size=V,
                   bos_token_id=V - 1, eos_token_id=V - 1)
    spec = _lora_spec(r=16, lora_alpha=64) if lora else None                 # gpt2.yaml:57-63 (its input dropout off: masks are tested elsewhere)
    dcfg = _hf_decoder_config(name='gpt2-small-rand', vocab_size=V, extra_tokens=2, use_cross_attn=True, lora_spec=spec)
    vit_kw = dict(n_embd_out_vit=768, n_cls=16, gate_sizes=[1024], refine_base_model=refine)           # gpt2.yaml:18-22
    cfg = vit_model_config(vit_kw).model_copy(update=dict(decoder_config=dcfg, use_cross_attn=True, use_soft_prompting=True))
    Vp = V + 2
    tok = fake_tokenizer(Vp)
    old = PretrainedViT.backbone_spec
    PretrainedViT.backbone_spec = VIT_B16
    try:
        w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    finally:
        PretrainedViT.backbone_spec = old
    m = w.model
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    if lora:
        with torch.no_grad():
            g = torch.Generator().manual_seed(9)
            for n, p in m.decoder.lora_params.items():
                if n.endswith('_B'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.02)
        hf, wraps = _lora_twin(m)
    else:
        hf, wraps = _hf_twin(m, True), {}
    esd = {k: (v.detach().clone().requires_grad_(True) if v.dtype.is_floating_point else v.detach().clone())
           for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    w = w.to(dev()).train()
    eng = m._engine
    assert eng.vit and eng.enc.refine == refine and eng.dec.prefixed and eng.cross_inputs and eng.dec.L == 12 and eng.dec.d == 768
    B, T = 2, 12
    images, labels = synthetic_batch(B, 224, T, Vp, seed=17)
    ids = labels.clamp(min=0)
    # ---- forward
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    enc, ologits, ohid = _reference_forward(orc, esd, hf, cfg, images, ids, True, True)
    for name, got, ref, tol in (('logits', out.logits, ologits, 1.25e-2), ('hidden', out.hidden_state, ohid, 1.5e-2),
                                ('encoder_output', out.encoder_output, enc, 1.25e-2)):
        diff = got.detach().float().cpu() - ref.detach()
        err, scale = float(diff.abs().max()), max(1.0, float(ref.detach().abs().max()))
        REPORT[f'{tag}.{name}'] = {'max_abs_err': err, 'tol': tol * scale, 'rel_l2': float(diff.norm() / ref.detach().norm())}
        assert err <= tol * scale, (name, err, tol * scale)
    # ---- the trainer's loss and every trainable gradient
    for p in m.parameters():
        p.grad = None
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    sids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id, -100)
    _, sl, _ = _reference_forward(orc, esd, hf, cfg, images, sids, True, True)
    ce = F.cross_entropy(sl.reshape(-1, Vp), labels.reshape(-1), ignore_index=-100, reduction='none')
    oloss = (ce * orc.loss_weights(labels, -100).reshape(-1)).sum()
    oloss.backward()
    REPORT[f'{tag}.loss'] = {'got': float(loss.detach()), 'ref': float(oloss)}
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * float(oloss)
    ref_grads = {k: v.grad for k, v in esd.items() if torch.is_tensor(v) and v.dtype.is_floating_point and v.grad is not None}
    shell = Decoder.from_config(_hf_decoder_config(name='gpt2-small-rand', vocab_size=V, extra_tokens=2, use_cross_attn=True))
    hf_g = {'backbone.' + k.replace('.base.', '.'): (p.grad if p.grad is not None else torch.zeros_like(p))
            for k, p in hf.named_parameters() if not (k.endswith('.A') or k.endswith('.B'))}
    hf_g['backbone.lm_head.weight'] = hf_g['backbone.transformer.wte.weight']
    shell.load_state_dict(hf_g, strict=True)
    ref_grads.update({'decoder.' + k: v.detach() for k, v in shell.named_parameters()})
    for (l, site), wr in wraps.items():
        ref_grads[f'decoder.lora_params.h{l}_{site}_A'], ref_grads[f'decoder.lora_params.h{l}_{site}_B'] = wr.A.grad, wr.B.grad
    fails, checked, none = [], 0, 0
    for name, p in m.named_parameters():
        if name in frozen or (name.startswith(eng.ep + 'model.') and not refine):
            assert p.grad is None, f'{name} is frozen but received a gradient'
            none += 1
            continue
        checked += 1
        try:
            grad_close(f'{tag}.{name}', p.grad, ref_grads[name].numpy(), rel=0.12, cos=0.985)
        except (AssertionError, KeyError) as e:
            fails.append(f'{name}: {e}')
    REPORT[f'{tag}.gradients'] = {'checked': checked, 'frozen_without_gradient': none, 'failed': len(fails)}
    assert checked >= (20 if lora else 100) and not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    assert none >= (150 if not refine else 0) + (12 * 10 if lora else 0), none     # the backbone's 152 tensors; the adapted decoder's base weights
    # ---- the first greedy tokens: the KV-cache decoder against the reference composition's argmax wherever its margin is clear
    w.eval()
    with torch.no_grad():
        prompt = torch.full((B, 1), tok.bos_token_id, dtype=torch.long)
        gen = m.generate(images.to(dev()), prompt.to(dev()), max_new_tokens=4, temperature=1.0, top_k=1).cpu()
        cur, agree, total = prompt, 0, 0
        for t in range(4):
            _, lg, _ = _reference_forward(orc, esd, hf, cfg, images, cur, True, True)
            lg = orc.apply_ngram_ban(cur, lg[:, -1].detach(), cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same = (gen[:, :cur.shape[1]] == cur).all(dim=1)
            agree += int(((gen[:, cur.shape[1]] == lg.argmax(-1)) & clear & same).sum())
            total += int((clear & same).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True)), dim=1)
        REPORT[f'{tag}.greedy_vs_reference'] = {'agree': agree, 'of': total}
        assert agree == total, (agree, total)
