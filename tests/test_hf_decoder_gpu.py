"""GPT2HuggingfaceDecoder on the MI355X (reference models/decoder.py:285-382): the plugin loads a Hugging Face
GPT-2 checkpoint from a local directory and runs it on the HIP hot path.  The checker is what the reference itself calls for this
decoder -- transformers' GPT2LMHeadModel, evaluated on the CPU in fp32 -- composed with the CPU oracle's encoder exactly as
vision_encoder_decoder.py:84-134 composes them (soft prompt = encoder outputs concatenated in front of the token embeddings, NO mask
handed to transformers: one causal sequence; cross-attention on the encoder outputs when the decoder has the layers)."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch, tiny_config
from test_host_cpu import _hf_decoder_config, _local_hf_gpt2
from test_model_gpu import grad_close

pytestmark = pytest.mark.gpu
REPORT = {}


@pytest.fixture(scope='module', autouse=True)
def write_report():
    yield
    import json
    import test_model_gpu
    REPORT.update({k: v for k, v in test_model_gpu.REPORT.items() if k.startswith('grad.hf_')})     # grad_close records there
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_hf_decoder.json', 'w') as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


def dev():
    return torch.device('cuda:0')


def _model_config(cross: bool, soft: bool, **hf_kw):
    cfg = tiny_config(dec_d=128, dec_heads=2, dec_layers=2, block_size=64)
    return cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(use_cross_attn=cross, **hf_kw), use_cross_attn=cross or not soft,
                                      use_soft_prompting=soft))


def _hf_twin(model, cross: bool):
    """a transformers GPT-2 holding the plugin's weights (through the plugin's own Hugging Face-keyed state dict)"""
    from transformers import GPT2Config, GPT2LMHeadModel
    hc = model.decoder.hf_config
    hf = GPT2LMHeadModel(GPT2Config(n_layer=hc.n_layer, n_head=hc.n_head, n_embd=hc.n_embd, n_positions=hc.n_positions,
                                    vocab_size=model._engine.dec.V, add_cross_attention=cross, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0)).eval()
    hf.load_state_dict({k[len('backbone.'):]: v.detach().cpu().clone() for k, v in model.decoder.state_dict().items()}, strict=True)
    return hf


def _reference_forward(orc, osd, hf, cfg, images, ids, cross: bool, soft: bool):
    """vision_encoder_decoder.py:74-134 with a HuggingfaceDecoder: (encoder_output, text logits, hidden_state)"""
    enc = orc.encode(osd, cfg, images, training=False)
    mem = enc if (cfg.use_cross_attn and cross) else None
    if soft:
        ncls = enc.shape[1]
        emb = torch.cat((enc, hf.transformer.wte(ids)), dim=-2)[..., :hf.config.n_positions, :]
        out = hf(inputs_embeds=emb, encoder_hidden_states=mem, output_hidden_states=True)
        return enc, out.logits[..., ncls:, :], out.hidden_states[-1]
    out = hf(input_ids=ids, encoder_hidden_states=mem, output_hidden_states=True)
    return enc, out.logits, out.hidden_states[-1]


def _build(tmp_path, monkeypatch, cross, soft, **hf_kw):
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    _local_hf_gpt2(tmp_path, monkeypatch)
    cfg = _model_config(cross, soft, **hf_kw)
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)                          # deterministic, LayerNorm parameters off their 1 / 0 initialisation (encoder)
    m.decoder.load_state_dict(keep)               # ... the decoder keeps the checkpoint's weights
    return cfg, m


@pytest.mark.parametrize('cross,soft', [(True, True), (False, True), (True, False)])
def test_gpt2_hf_decoder_forward_and_gradients(tmp_path, monkeypatch, cross, soft):
    """forward(): logits, hidden_state (prompt rows included) and encoder_output; then a loss over logits AND hidden_state back-propagated
    -- every parameter's gradient against autograd through oracle encoder + transformers' GPT-2 (which has no gradient normaliser in
    its blocks: the hot path's decoder must not apply one here, while the encoder blocks keep theirs)."""
    from oracle import reference_model as orc
    tag = f'hf_gpt2.{"cross" if cross else "nocross"}.{"soft" if soft else "ids"}'
    cfg, m = _build(tmp_path, monkeypatch, cross, soft)
    hf = _hf_twin(m, cross)
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    m = m.to(dev()).train()                       # the checkpoint's dropout rates are 0: train mode only enables the backward
    assert m._engine.dec.prefixed == soft and not m._engine.dec.grad_norm
    images, labels = synthetic_batch(3, 32, 12, 384, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls if soft else 0
    g = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, 128, generator=g) * 0.05
    wl = torch.randn(3, 12, 384, generator=g) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    enc, ologits, ohid = _reference_forward(orc, esd, hf, cfg, images, ids, cross, soft)
    assert tuple(out.hidden_state.shape) == tuple(ohid.shape) == (3, n_p + 12, 128)
    for name, got, ref, tol in (('logits', out.logits, ologits, 1e-2), ('hidden', out.hidden_state, ohid, 1.5e-2),
                                ('encoder_output', out.encoder_output, enc, 1e-2)):
        err, scale = float((got.float().cpu() - ref.detach()).abs().max()), max(1.0, float(ref.detach().abs().max()))
        REPORT[f'{tag}.{name}'] = {'max_abs_err': err, 'tol': tol * scale}
        assert err <= tol * scale, (name, err, tol * scale)
    loss = (out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()
    loss.backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    # transformers' gradients in the plugin's internal (nn.Linear) layout: through a plugin-shaped container's state-dict hooks
    from image2text_amd.models.decoder import Decoder
    shell = Decoder.from_config(cfg.decoder_config)
    hf_g = {'backbone.' + k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in hf.named_parameters()}
    hf_g['backbone.lm_head.weight'] = hf_g['backbone.transformer.wte.weight']
    shell.load_state_dict(hf_g, strict=True)
    for k, v in shell.named_parameters():
        ref_grads['decoder.' + k] = v.detach()
    fails, checked = [], 0
    for name, p in m.named_parameters():
        if name not in ref_grads:
            continue
        checked += 1
        try:
            grad_close(f'{tag}.{name}', p.grad, ref_grads[name].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert checked == len(list(m.named_parameters())), (checked, len(list(m.named_parameters())))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])


@pytest.mark.parametrize('cross,soft', [(True, True), (False, True)])
def test_gpt2_hf_decoder_train_step_and_generate(tmp_path, monkeypatch, cross, soft):
    """ModelTrainerWrapper.train_step / val_step (weighted cross-entropy over the text rows) against the same composition, and
    generate(): the KV cache opened by the prompt rows' keys/values + hipGraph replay == greedy decoding by repeated full forwards"""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    tag = f'hf_gpt2.step.{"cross" if cross else "nocross"}'
    _local_hf_gpt2(tmp_path, monkeypatch)
    cfg = _model_config(cross, soft)
    tok = fake_tokenizer(384)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    keep = {k: v.detach().clone() for k, v in w.model.decoder.state_dict().items()}
    det_init_(w.model, seed=0)
    w.model.decoder.load_state_dict(keep)
    hf = _hf_twin(w.model, cross)
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in w.model.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    w = w.to(dev()).train()
    images, labels = synthetic_batch(4, 32, 14, 384, seed=5)
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    ids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id, -100)
    _, ologits, _ = _reference_forward(orc, esd, hf, cfg, images, ids, cross, soft)
    wts = orc.loss_weights(labels, -100)
    ce = F.cross_entropy(ologits.reshape(-1, ologits.size(-1)), labels.reshape(-1), ignore_index=-100, reduction='none')
    oloss = (ce * wts.reshape(-1)).sum()
    oloss.backward()
    REPORT[f'{tag}.loss'] = {'got': float(loss.detach()), 'ref': float(oloss)}
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * float(oloss)
    from image2text_amd.models.decoder import Decoder
    shell = Decoder.from_config(cfg.decoder_config)
    hf_g = {'backbone.' + k: (p.grad if p.grad is not None else torch.zeros_like(p)) for k, p in hf.named_parameters()}
    hf_g['backbone.lm_head.weight'] = hf_g['backbone.transformer.wte.weight']
    shell.load_state_dict(hf_g, strict=True)
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.' + k: v.detach() for k, v in shell.named_parameters()})
    fails = []
    for name, p in w.model.named_parameters():
        try:
            grad_close(f'{tag}.{name}', p.grad, ref_grads[name].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    w.eval()
    with torch.no_grad():
        vloss, _ = w.val_step(images.to(dev()), labels.to(dev()))
        assert abs(float(vloss) - float(oloss)) <= 1e-2 * float(oloss)
        # generation: KV cache (prompt rows first) vs the reference's loop of full forwards, on the device
        prompt = torch.full((4, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = w.model.generate(images.to(dev()), prompt, max_new_tokens=12, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(12):
            lg = w.model(images=images.to(dev()), ids=cur).logits[:, -1].float().cpu()
            lg = orc.apply_ngram_ban(cur.cpu(), lg, cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same_prefix = (gen[:, :cur.shape[1]].cpu() == cur.cpu()).all(dim=1)
            ok = gen[:, cur.shape[1]].cpu() == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True).to(dev())), dim=1)
        REPORT[f'{tag}.generate_vs_forward'] = {'agree': agree, 'of': total}
        assert total >= 12 and agree == total, (agree, total)
        # ... and the first generated token against transformers itself
        ids0 = prompt.cpu()
        _, l0, _ = _reference_forward(orc, esd, hf, cfg, images, ids0, cross, soft)
        l0 = l0[:, -1].detach()
        t2 = l0.topk(2, dim=-1).values
        sure = (t2[:, 0] - t2[:, 1]) > 3e-2 * l0.abs().max()
        assert bool((gen[:, 1].cpu()[sure] == l0.argmax(-1)[sure]).all())


def test_gpt2_hf_decoder_standalone_call(tmp_path, monkeypatch):
    """decoder(idx | inputs_embeds, cross_attn_embeds, attn_msk) as the reference's HuggingfaceDecoder.forward (decoder.py:332-361):
    the mask argument is accepted and ignored, cross inputs are used only when the decoder has the layers"""
    cfg, m = _build(tmp_path, monkeypatch, True, True)
    hf = _hf_twin(m, True)
    m = m.to(dev()).eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 380, (3, 40), generator=g)
    mem = torch.randn(3, 8, 128, generator=g) * 0.5
    emb = torch.randn(3, 20, 128, generator=g) * 0.1
    with torch.no_grad():
        for kind, kw, hkw in (('ids', dict(idx=ids.to(dev())), dict(input_ids=ids)),
                              ('embeds', dict(inputs_embeds=emb.to(dev())), dict(inputs_embeds=emb))):
            ref = hf(encoder_hidden_states=mem, output_hidden_states=True, **hkw)
            logits, hidden = m.decoder(cross_attn_embeds=mem.to(dev()), attn_msk=torch.ones(40, 40, dtype=torch.bool), **kw)
            err = float((logits.float().cpu() - ref.logits).abs().max())
            REPORT[f'hf_gpt2.standalone.{kind}'] = {'max_abs_err': err, 'ref_absmax': float(ref.logits.abs().max())}
            assert err <= 1e-2 * max(1.0, float(ref.logits.abs().max())), (kind, err)
            assert float((hidden.float().cpu() - ref.hidden_states[-1]).abs().max()) <= 1.5e-2 * max(1.0, float(ref.hidden_states[-1].abs().max()))


# ------------------------------------------------------------------------------------------------------------------------------
# Llama2HuggingfaceDecoder / Qwen2HuggingfaceDecoder (reference decoder.py:404-440): RMSNorm, rotary embedding, grouped K/V heads,
# SwiGLU (image2text_amd/engine_llama.py, csrc/llama.hip).  Checker: the checkpoint's own transformers module on the CPU in fp32.
# ------------------------------------------------------------------------------------------------------------------------------
def _llama_model(tmp_path, monkeypatch, kind):
    import copy
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from test_host_cpu import _local_hf_llama
    _, name, vocab = _local_hf_llama(tmp_path, monkeypatch, kind)
    extra = 4 if kind == 'llama' else 0
    cfg = tiny_config(dec_d=256, dec_heads=4, dec_layers=2, block_size=64)
    cfg = cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra),
                                     use_cross_attn=False, use_soft_prompting=True))
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    m.decoder.tie_weights()
    hf = copy.deepcopy(m.decoder.backbone).float().eval()
    return cfg, m, hf, vocab + extra


def _llama_reference(orc, esd, hf, cfg, images, ids):
    enc = orc.encode(esd, cfg, images, training=False)
    emb = torch.cat((enc, hf.model.embed_tokens(ids)), dim=-2)
    out = hf(inputs_embeds=emb, output_hidden_states=True)
    return enc, out.logits[..., enc.shape[1]:, :], out.hidden_states[-1]


@pytest.mark.parametrize('kind', ['llama', 'qwen'])
def test_llama_family_decoder_forward_gradients_generate(tmp_path, monkeypatch, kind):
    """forward() -- logits, hidden_state over [prompt rows | text rows], encoder_output -- a loss over both back-propagated (every
    parameter's gradient: fused q|k|v and gate|up projections, their biases, RMSNorm gains, tied or untied head, the encoder through
    the prompt rows), then the trainer's train_step loss, and generate() on the KV cache against greedy decoding by full forwards."""
    from oracle import reference_model as orc
    tag = f'hf_{kind}'
    cfg, m, hf, V = _llama_model(tmp_path, monkeypatch, kind)
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    m = m.to(dev()).train()
    eng = m._engine
    assert eng.dec.prefixed and eng.dec.llama is not None and not eng.cross_inputs
    images, labels = synthetic_batch(3, 32, 12, V, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls
    g = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, 256, generator=g) * 0.05
    wl = torch.randn(3, 12, V, generator=g) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    enc, ologits, ohid = _llama_reference(orc, esd, hf, cfg, images, ids)
    for name, got, ref, tol in (('logits', out.logits, ologits, 1e-2), ('hidden', out.hidden_state, ohid, 1.5e-2),
                                ('encoder_output', out.encoder_output, enc, 1e-2)):
        err, scale = float((got.float().cpu() - ref.detach()).abs().max()), max(1.0, float(ref.detach().abs().max()))
        REPORT[f'{tag}.{name}'] = {'max_abs_err': err, 'tol': tol * scale}
        assert err <= tol * scale, (name, err, tol * scale)
    ((out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()).backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.backbone.' + k: p.grad for k, p in hf.named_parameters()})
    fails, names = [], [n for n, _ in m.named_parameters()]
    assert set(names) == set(ref_grads), set(names) ^ set(ref_grads)
    for name, p in m.named_parameters():
        try:
            grad_close(f'{tag}.{name}', p.grad, ref_grads[name].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    # trainer step on the same model
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    tok = fake_tokenizer(V)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    w.model.load_state_dict({k: v.detach().cpu() for k, v in m.state_dict().items()})
    w.model.decoder.tie_weights()
    w = w.to(dev()).train()
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    assert w.pack_rows                                   # ragged captions: the step runs on packed rows (n_p + len_b per sequence)
    sids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id, -100)
    for t in list(esd.values()) + list(hf.parameters()):
        t.grad = None
    _, sl, _ = _llama_reference(orc, esd, hf, cfg, images, sids)
    ce = F.cross_entropy(sl.reshape(-1, V), labels.reshape(-1), ignore_index=-100, reduction='none')
    oloss_t = (ce * orc.loss_weights(labels, -100).reshape(-1)).sum()
    oloss_t.backward()
    oloss = float(oloss_t.detach())
    REPORT[f'{tag}.train_loss'] = {'got': float(loss.detach()), 'ref': oloss}
    assert abs(float(loss.detach()) - oloss) <= 1e-2 * oloss
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.backbone.' + k: p.grad for k, p in hf.named_parameters()})
    fails = []
    for name, p in w.model.named_parameters():
        try:
            grad_close(f'{tag}.step.{name}', p.grad, ref_grads[name].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} train-step gradients out of tolerance: ' + '; '.join(fails[:6])
    # generation
    w.eval()
    with torch.no_grad():
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = w.model.generate(images.to(dev()), prompt, max_new_tokens=12, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(12):
            lg = w.model(images=images.to(dev()), ids=cur).logits[:, -1].float().cpu()
            lg = orc.apply_ngram_ban(cur.cpu(), lg, cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same_prefix = (gen[:, :cur.shape[1]].cpu() == cur.cpu()).all(dim=1)
            ok = gen[:, cur.shape[1]].cpu() == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True).to(dev())), dim=1)
        REPORT[f'{tag}.generate_vs_forward'] = {'agree': agree, 'of': total}
        assert total >= 9 and agree == total, (agree, total)
        _, l0, _ = _llama_reference(orc, esd, hf, cfg, images, prompt.cpu())
        l0 = l0[:, -1].detach()
        t2 = l0.topk(2, dim=-1).values
        sure = (t2[:, 0] - t2[:, 1]) > 3e-2 * l0.abs().max()
        assert bool((gen[:, 1].cpu()[sure] == l0.argmax(-1)[sure]).all())


@pytest.mark.parametrize('kind', ['llama', 'qwen'])
def test_llama_family_long_sequence_forward(tmp_path, monkeypatch, kind):
    """8 prompt rows + 110 text rows (the checkpoints' 128 positions): several key tiles per head in the grouped attention kernels, rotary
    angles far from zero -- logits and hidden state of forward() against the transformers module"""
    from oracle import reference_model as orc
    cfg, m, hf, V = _llama_model(tmp_path, monkeypatch, kind)
    esd = {k: v.detach().clone() for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    m = m.to(dev()).eval()
    images, labels = synthetic_batch(2, 32, 110, V, seed=29, min_len=100)
    ids = labels.clamp(min=0)
    with torch.no_grad():
        out = m(images=images.to(dev()), ids=ids.to(dev()))
        _, ologits, ohid = _llama_reference(orc, esd, hf, cfg, images, ids)
    for name, got, ref, tol in (('logits', out.logits, ologits, 1e-2), ('hidden', out.hidden_state, ohid, 1.5e-2)):
        err, scale = float((got.float().cpu() - ref).abs().max()), max(1.0, float(ref.abs().max()))
        REPORT[f'hf_{kind}.long.{name}'] = {'max_abs_err': err, 'tol': tol * scale}
        assert err <= tol * scale, (name, err, tol * scale)
    assert (out.logits.argmax(-1).cpu() == ologits.argmax(-1)).float().mean() > 0.9


class _LoraLinear(torch.nn.Module):
    """peft's LoRA layer around an nn.Linear, restated (peft is not in this image): base(x) + lora_B(lora_A(dropout(x))) * alpha / r"""

    def __init__(self, base, A, B, scale):
        super().__init__()
        self.base, self.scale, self.mask = base, scale, None
        self.A, self.B = torch.nn.Parameter(A.clone()), torch.nn.Parameter(B.clone())

    def forward(self, x):
        xd = x if self.mask is None else x * self.mask.view(x.shape)
        return self.base(x) + (xd @ self.A.t() @ self.B.t()) * self.scale


@pytest.mark.parametrize('kind,p_lora', [('llama', 0.0), ('llama', 0.25), ('qwen', 0.0)])
def test_llama_family_decoder_lora(tmp_path, monkeypatch, kind, p_lora):
    """lora_spec on the Llama-2 / Qwen2 plugins with the targets of the reference's gpu/llama2-13b.yaml (q, k, v, o, up, down): the adapters
    of a fused projection ride block-diagonally in the K panel of its ONE GEMM, frozen base weights get no gradient (and no dW GEMM), the
    adapter input dropout uses the same mask forward and backward, generation runs on merged weights.  Checker: the checkpoint's own
    transformers module on the CPU in fp32 with peft's published LoRA layer restated around the targeted linears."""
    import copy
    from oracle import reference_model as orc
    from image2text_amd import rng
    from image2text_amd.configs.models import LoraSpec
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from test_host_cpu import _local_hf_llama
    tag = f'hf_{kind}_lora.p{p_lora}'
    _, name, vocab = _local_hf_llama(tmp_path, monkeypatch, kind)
    extra = 4 if kind == 'llama' else 0
    V = vocab + extra
    spec = LoraSpec(r=4, lora_alpha=16, lora_dropout=p_lora, target_modules=['q_proj', 'k_proj', 'v_proj', 'o_proj', 'up_proj', 'down_proj'],
                    force_enable_update_modules=['*.norm.*'] if kind == 'qwen' else None)
    cfg = tiny_config(dec_d=256, dec_heads=4, dec_layers=2, block_size=64)
    cfg = cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=extra, lora_spec=spec),
                                     use_cross_attn=False, use_soft_prompting=True))
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    m.decoder.tie_weights()
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for n, p in m.decoder.lora_params.items():
            if n.endswith('_B'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)          # lora_B starts at zero: give the adapters something to do
    dec = m.decoder
    hf = copy.deepcopy(dec.backbone).float().eval()
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    wraps = {}
    for l in range(2):
        for site, members in dec.lora.members.items():
            for mod_path in members:
                parent_name, leaf = mod_path.split('.')
                parent = getattr(hf.model.layers[l], parent_name)
                key = f'backbone.model.model.layers.{l}.{mod_path}.lora_'
                w = _LoraLinear(getattr(parent, leaf), sd[key + 'A.default.weight'], sd[key + 'B.default.weight'], dec.lora.scale)
                setattr(parent, leaf, w)
                wraps[(l, site, mod_path)] = w
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    assert 'decoder.backbone.model.layers.0.self_attn.q_proj.weight' in frozen and 'decoder.lora_params.h0_qkv_A' not in frozen
    assert ('decoder.backbone.model.norm.weight' in frozen) == (kind == 'llama')
    m = m.to(dev()).train()
    eng = m._engine
    images, labels = synthetic_batch(3, 32, 12, V, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls
    gg = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, 256, generator=gg) * 0.05
    wl = torch.randn(3, 12, V, generator=gg) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    if p_lora > 0:          # the masks this step drew (one per fused site), replicated on the host for the checker's adapters
        plan = eng.dec_drop
        for (l, site, _), w in wraps.items():
            _, key, thr, scale = plan.get(l, f'lora_{site}')
            rows, K = 3 * (n_p + 12), w.A.shape[1]
            w.mask = rng.keep_mask(key, rows * K, thr).view(rows, K).float() * scale
    enc, ologits, ohid = _llama_reference(orc, esd, hf, cfg, images, ids)
    # (max over 3 x 12 x V logits of bf16 rounding noise: the un-adapted tests sit at 0.85 of 1e-2 x max|logit|; the adapters add two
    # bf16 roundings (u and s B) per site, so the max-norm bound is 1.25e-2 here and the rel-L2 error is bounded beside it)
    for name_, got, ref, tol in (('logits', out.logits, ologits, 1.25e-2), ('hidden', out.hidden_state, ohid, 1.5e-2)):
        diff = got.float().cpu() - ref.detach()
        err, scale_ = float(diff.abs().max()), max(1.0, float(ref.detach().abs().max()))
        rel = float(diff.norm() / ref.detach().norm())
        REPORT[f'{tag}.{name_}'] = {'max_abs_err': err, 'tol': tol * scale_, 'rel_l2': rel}
        assert err <= tol * scale_ and rel <= 2e-2, (name_, err, tol * scale_, rel)
    ((out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()).backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.backbone.' + k.replace('.base.', '.'): p.grad for k, p in hf.named_parameters()
                      if not (k.endswith('.A') or k.endswith('.B'))})
    r = dec.lora.r
    for l in range(2):
        for site, members in dec.lora.members.items():
            ref_grads[f'decoder.lora_params.h{l}_{site}_A'] = torch.cat([wraps[(l, site, mp)].A.grad for mp in members], 0)
            for mp in members:
                ref_grads[f'decoder.lora_params.h{l}_{dec._LORA_TAGS[mp]}_B'] = wraps[(l, site, mp)].B.grad
    fails, checked = [], 0
    for name_, p in m.named_parameters():
        if name_ in frozen:
            assert p.grad is None, f'{name_} is frozen but received a gradient'
            continue
        checked += 1
        try:
            grad_close(f'{tag}.{name_}', p.grad, ref_grads[name_].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert checked >= 2 * 10 and not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    if p_lora > 0:
        return
    # generation on merged weights against greedy decoding by full (adapter-in-the-K-panel) forwards
    m.eval()
    tok = fake_tokenizer(V)
    with torch.no_grad():
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = m.generate(images.to(dev()), prompt, max_new_tokens=10, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(10):
            lg = m(images=images.to(dev()), ids=cur).logits[:, -1].float().cpu()
            lg = orc.apply_ngram_ban(cur.cpu(), lg, cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same_prefix = (gen[:, :cur.shape[1]].cpu() == cur.cpu()).all(dim=1)
            ok = gen[:, cur.shape[1]].cpu() == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True).to(dev())), dim=1)
        REPORT[f'{tag}.generate_vs_forward'] = {'agree': agree, 'of': total}
        assert total >= 8 and agree == total, (agree, total)
        _, l0, _ = _llama_reference(orc, esd, hf, cfg, images, prompt.cpu())
        l0 = l0[:, -1].detach()
        t2 = l0.topk(2, dim=-1).values
        sure = (t2[:, 0] - t2[:, 1]) > 3e-2 * l0.abs().max()
        assert bool((gen[:, 1].cpu()[sure] == l0.argmax(-1)[sure]).all())


# ------------------------------------------------------------------------------------------------------------------------------
# FalconHuggingfaceDecoder (reference decoder.py:122-123, 383-400; gpu/falcon-7b.yaml): parallel attention + MLP behind one LayerNorm,
# multi-query attention, rotary embedding, exact GELU.  Checker: transformers' FalconForCausalLM on the CPU in fp32 (+ the restated peft
# layer for LoRA).  'ids' is the mode the reference itself can run (decoder.py:352-361: the decoder sees token ids only); 'soft' is the
# mode its yaml asks for and its get_inputs_embeds cannot deliver (models/decoder.py FalconHuggingfaceDecoder docstring).
# ------------------------------------------------------------------------------------------------------------------------------
def test_falcon_decoder_id_driven_forward_matches_transformers(tmp_path, monkeypatch):
    """``Decoder.from_config(cfg)(idx=ids) -> (logits, hidden_states[-1])``: the call the reference's FalconHuggingfaceDecoder can make
    (decoder.py:332-361; inside a VisionEncoderDecoder the model needs cross-attention or a soft prompt, vision_encoder_decoder.py:38-39, and
    Falcon has neither there), against transformers' FalconForCausalLM(input_ids=ids)."""
    import copy
    from image2text_amd.models.decoder import Decoder
    from test_host_cpu import _local_hf_falcon
    _, name, vocab = _local_hf_falcon(tmp_path, monkeypatch)
    dec = Decoder.from_config(_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1))
    hf = copy.deepcopy(dec.backbone).float().eval()
    dec = dec.to(dev()).eval()
    g = torch.Generator().manual_seed(5)
    for B, T in ((3, 12), (2, 100)):                    # (100 rows: several key tiles per head in the grouped attention kernel)
        ids = torch.randint(0, vocab + 1, (B, T), generator=g)
        logits, hidden = dec(idx=ids.to(dev()))
        with torch.no_grad():
            out = hf(input_ids=ids, output_hidden_states=True)
        for name_, got, ref, tol in (('logits', logits, out.logits, 1.25e-2), ('hidden', hidden, out.hidden_states[-1], 1.5e-2)):
            diff = got.float().cpu() - ref
            err, scale_ = float(diff.abs().max()), max(1.0, float(ref.abs().max()))
            REPORT[f'hf_falcon.ids.T{T}.{name_}'] = {'max_abs_err': err, 'tol': tol * scale_, 'rel_l2': float(diff.norm() / ref.norm())}
            assert err <= tol * scale_, (name_, T, err, tol * scale_)


@pytest.mark.parametrize('mode', ['soft', 'soft_lora'])
def test_falcon_decoder_forward_gradients_generate(tmp_path, monkeypatch, mode):
    import copy
    from oracle import reference_model as orc
    from image2text_amd.configs.models import LoraSpec
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from test_host_cpu import _local_hf_falcon
    tag = f'hf_falcon.{mode}'
    _, name, vocab = _local_hf_falcon(tmp_path, monkeypatch)
    V, soft = vocab + 1, mode != 'ids'
    spec = LoraSpec(r=4, lora_alpha=16, lora_dropout=0.0, target_modules=['query_key_value', 'dense', 'dense_h_to_4h', 'dense_4h_to_h'],
                    force_enable_update_modules=['*.word_embeddings.*', '*.lm_head.*']) if mode == 'soft_lora' else None      # gpu/falcon-7b.yaml:55-60
    cfg = tiny_config(dec_d=256, dec_heads=4, dec_layers=2, block_size=64)
    cfg = cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=1, lora_spec=spec),
                                     use_cross_attn=False, use_soft_prompting=soft))
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    m.decoder.tie_weights()
    dec = m.decoder
    wraps = {}
    if spec is not None:
        with torch.no_grad():
            g = torch.Generator().manual_seed(9)
            for n, p in dec.lora_params.items():
                if n.endswith('_B'):
                    p.copy_(torch.randn(p.shape, generator=g) * 0.05)
    hf = copy.deepcopy(dec.backbone).float().eval()
    if spec is not None:
        sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
        for l in range(2):
            for site, members in dec.lora.members.items():
                for mod_path in members:
                    parent_name, leaf = mod_path.split('.')
                    parent = getattr(hf.transformer.h[l], parent_name)
                    key = f'backbone.model.transformer.h.{l}.{mod_path}.lora_'
                    w = _LoraLinear(getattr(parent, leaf), sd[key + 'A.default.weight'], sd[key + 'B.default.weight'], dec.lora.scale)
                    setattr(parent, leaf, w)
                    wraps[(l, site, mod_path)] = w
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    m = m.to(dev()).train()
    eng = m._engine
    assert eng.dec.llama.arch == 'falcon' and eng.dec.prefixed == soft and not eng.cross_inputs
    images, labels = synthetic_batch(3, 32, 12, V, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls if soft else 0

    def reference(ids_):
        enc = orc.encode(esd, cfg, images, training=False)
        if soft:
            out = hf(inputs_embeds=torch.cat((enc, hf.transformer.word_embeddings(ids_)), dim=-2), output_hidden_states=True)
        else:
            out = hf(input_ids=ids_, output_hidden_states=True)             # reference decoder.py:352-361
        return enc, out.logits[..., n_p:, :], out.hidden_states[-1]
    gg = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, 256, generator=gg) * 0.05
    wl = torch.randn(3, 12, V, generator=gg) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    enc, ologits, ohid = reference(ids)
    for name_, got, ref, tol in (('logits', out.logits, ologits, 1.25e-2), ('hidden', out.hidden_state, ohid, 1.5e-2),
                                 ('encoder_output', out.encoder_output, enc, 1e-2)):
        diff = got.float().cpu() - ref.detach()
        err, scale_ = float(diff.abs().max()), max(1.0, float(ref.detach().abs().max()))
        REPORT[f'{tag}.{name_}'] = {'max_abs_err': err, 'tol': tol * scale_, 'rel_l2': float(diff.norm() / ref.detach().norm())}
        assert err <= tol * scale_, (name_, err, tol * scale_)
    ((out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()).backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.backbone.' + k.replace('.base.', '.'): p.grad for k, p in hf.named_parameters()
                      if not (k.endswith('.A') or k.endswith('.B'))})
    for (l, site, mp), w in wraps.items():
        ref_grads[f'decoder.lora_params.h{l}_{site}_A'] = w.A.grad
        ref_grads[f'decoder.lora_params.h{l}_{dec._LORA_TAGS[mp]}_B'] = w.B.grad
    fails, checked = [], 0
    for name_, p in m.named_parameters():
        if name_ in frozen:
            assert p.grad is None, f'{name_} is frozen but received a gradient'
            continue
        if not soft and not name_.startswith('decoder.'):        # the decoder never saw the image: nothing flows back to the encoder
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name_
            continue
        checked += 1
        try:
            grad_close(f'{tag}.{name_}', p.grad, ref_grads[name_].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert checked >= 12 and not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    # generation on the KV cache (merged adapter weights under LoRA) against greedy decoding by full forwards
    m.eval()
    tok = fake_tokenizer(V)
    with torch.no_grad():
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = m.generate(images.to(dev()), prompt, max_new_tokens=10, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(10):
            lg = m(images=images.to(dev()), ids=cur).logits[:, -1].float().cpu()
            lg = orc.apply_ngram_ban(cur.cpu(), lg, cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same_prefix = (gen[:, :cur.shape[1]].cpu() == cur.cpu()).all(dim=1)
            ok = gen[:, cur.shape[1]].cpu() == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True).to(dev())), dim=1)
        REPORT[f'{tag}.generate_vs_forward'] = {'agree': agree, 'of': total}
        assert total >= 8 and agree == total, (agree, total)
        _, l0, _ = reference(prompt.cpu())
        l0 = l0[:, -1].detach()
        t2 = l0.topk(2, dim=-1).values
        sure = (t2[:, 0] - t2[:, 1]) > 3e-2 * l0.abs().max()
        assert bool((gen[:, 1].cpu()[sure] == l0.argmax(-1)[sure]).all())


# ------------------------------------------------------------------------------------------------------------------------------
# The same plugins at PRODUCTION WIDTH, one layer deep (the cases above are toy shapes): Llama-2-7B's block
# (4096 wide, 32 heads of 128, SwiGLU 11008, vocabulary 32000), DeepSeek-R1-Distill-Qwen-1.5B's (1536, 12 heads of 128 on 2 K/V heads,
# 8960, q / k / v biases, vocabulary 151936) and Falcon-7B's (4544 = 71 heads of 64 on ONE K/V head, 18176, vocabulary 65024: K is not a
# multiple of 128, rows are wider than the one-wave LayerNorm).  Checker: the checkpoint's own transformers module, CPU, fp32.
# ------------------------------------------------------------------------------------------------------------------------------
def _production_width_checkpoint(kind, tmp_path, monkeypatch):
    import transformers
    torch.manual_seed(3)
    if kind == 'llama2-7b':
        name, vocab = 'meta-llama/Llama-2-7b-1layer', 32000
        hf = transformers.LlamaForCausalLM(transformers.LlamaConfig(hidden_size=4096, intermediate_size=11008, num_hidden_layers=1, num_attention_heads=32,
                                                                     num_key_value_heads=32, vocab_size=vocab, max_position_embeddings=128, rms_norm_eps=1e-5))
    elif kind == 'qwen2-1.5b':
        name, vocab = 'Qwen2-1.5B-1layer', 151936
        hf = transformers.Qwen2ForCausalLM(transformers.Qwen2Config(hidden_size=1536, intermediate_size=8960, num_hidden_layers=1, num_attention_heads=12,
                                                                     num_key_value_heads=2, vocab_size=vocab, max_position_embeddings=128, rms_norm_eps=1e-6,
                                                                     tie_word_embeddings=False))
    else:
        name, vocab = 'tiiuae/falcon-7b-1layer', 65024
        hf = transformers.FalconForCausalLM(transformers.FalconConfig(hidden_size=4544, num_attention_heads=71, num_hidden_layers=1, vocab_size=vocab,
                                                                       multi_query=True, parallel_attn=True, new_decoder_architecture=False, bias=False,
                                                                       alibi=False, max_position_embeddings=128))
    with torch.no_grad():
        for n_, p_ in hf.named_parameters():
            if n_.endswith('.bias') or 'norm' in n_ or 'ln_f' in n_:
                p_.add_(0.05 * torch.randn_like(p_))
    hf.save_pretrained(str(tmp_path / name))
    monkeypatch.chdir(tmp_path)
    return name, vocab


@pytest.mark.timeout(900)
@pytest.mark.parametrize('kind', ['llama2-7b', 'qwen2-1.5b', 'falcon-7b'])
def test_hf_decoder_block_at_production_width(tmp_path, monkeypatch, kind):
    import copy
    from oracle import reference_model as orc
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    tag = f'hf_width.{kind}'
    name, V = _production_width_checkpoint(kind, tmp_path, monkeypatch)
    cfg = tiny_config(dec_d=256, dec_heads=4, dec_layers=1, block_size=64)
    cfg = cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(name=name, vocab_size=V, extra_tokens=0), use_cross_attn=False,
                                     use_soft_prompting=True))
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    m.decoder.tie_weights()
    hf = copy.deepcopy(m.decoder.backbone).float().eval()
    d = m.decoder.n_embd
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    m = m.to(dev()).train()
    assert m.has_bridge and m._engine.dec.llama.d == d
    images, labels = synthetic_batch(3, 32, 12, V, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls
    embed = hf.get_input_embeddings()
    g = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, d, generator=g) * 0.02
    wl = torch.randn(3, 12, V, generator=g) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    enc = orc.encode(esd, cfg, images, training=False)
    ref = hf(inputs_embeds=torch.cat((enc, embed(ids)), dim=-2), output_hidden_states=True)
    ologits, ohid = ref.logits[..., n_p:, :], ref.hidden_states[-1]
    # bf16 operands over K = 1536 ... 18176 reductions: the error of a logit is noise of rel-L2 ~ 1 % (measured 1.1 % on the Llama block),
    # and the max over 3 x 12 x V ~ 10^6 samples of it sits ~5 sigma out -- the bound that means something here is the rel-L2 one
    # (<= 2e-2); the max-norm bound is 2e-2 x max|value| (the toy shapes above hold 1.25e-2)
    for name_, got, want, tol in (('logits', out.logits, ologits, 2e-2), ('hidden', out.hidden_state, ohid, 2e-2)):
        diff = got.float().cpu() - want.detach()
        err, scale = float(diff.abs().max()), max(1.0, float(want.detach().abs().max()))
        REPORT[f'{tag}.{name_}'] = {'max_abs_err': err, 'tol': tol * scale, 'rel_l2': float(diff.norm() / want.detach().norm())}
        assert err <= tol * scale and float(diff.norm() / want.detach().norm()) <= 2e-2, (name_, err, tol * scale)
    ((out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()).backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.backbone.' + k: p.grad for k, p in hf.named_parameters()})
    fails, names = [], [n for n, _ in m.named_parameters()]
    assert set(names) == set(ref_grads), set(names) ^ set(ref_grads)
    for name_, p in m.named_parameters():
        try:
            grad_close(f'{tag}.{name_}', p.grad, ref_grads[name_].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    # KV-cache generation (hipGraph) at this width: the first token against the checker's own argmax where its margin is clear
    m.eval()
    tok = fake_tokenizer(V)
    with torch.no_grad():
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = m.generate(images.to(dev()), prompt, max_new_tokens=4, temperature=1.0, top_k=1)
        l0 = hf(inputs_embeds=torch.cat((enc, embed(prompt.cpu())), dim=-2)).logits[:, -1].detach()
        t2 = l0.topk(2, dim=-1).values
        sure = (t2[:, 0] - t2[:, 1]) > 3e-2 * l0.abs().max()
        REPORT[f'{tag}.generate_first_token'] = {'clear_margins': int(sure.sum()), 'of': 3}
        assert bool((gen[:, 1].cpu()[sure] == l0.argmax(-1)[sure]).all())


def test_llama_decoder_frozen_by_prepare_for_kbit_training(tmp_path, monkeypatch):
    """prepare_for_kbit_training: True without 4-bit loading (reference local/llama2-7b.yaml; peft freezes the base model): no decoder
    parameter receives a gradient, the weight-gradient GEMMs are skipped, and the encoder's gradients -- through the soft prompt rows of
    the frozen decoder -- still match autograd through oracle encoder + transformers"""
    import copy
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    from test_host_cpu import _local_hf_llama
    _, name, vocab = _local_hf_llama(tmp_path, monkeypatch, 'llama')
    cfg = tiny_config(dec_d=256, dec_heads=4, dec_layers=2, block_size=64)
    cfg = cfg.model_copy(update=dict(decoder_config=_hf_decoder_config(name=name, vocab_size=vocab, extra_tokens=0, prepare_for_kbit_training=True),
                                     use_cross_attn=False, use_soft_prompting=True))
    tok = fake_tokenizer(vocab)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    keep = {k: v.detach().clone() for k, v in w.model.decoder.state_dict().items()}
    det_init_(w.model, seed=0)
    w.model.decoder.load_state_dict(keep)
    hf = copy.deepcopy(w.model.decoder.backbone).float().eval()
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in w.model.state_dict().items() if not k.startswith('decoder.')}
    w = w.to(dev()).train()
    images, labels = synthetic_batch(3, 32, 12, vocab, seed=17)
    from image2text_amd import ops
    calls = []
    orig = ops.gemm
    monkeypatch.setattr(ops, 'gemm', lambda *a, **k: (calls.append(bool(k.get('a_kmajor'))), orig(*a, **k))[1])
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    monkeypatch.setattr(ops, 'gemm', orig)
    ids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id, -100)
    _, sl, _ = _llama_reference(orc, esd, hf, cfg, images, ids)
    ce = F.cross_entropy(sl.reshape(-1, vocab), labels.reshape(-1), ignore_index=-100, reduction='none')
    oloss = (ce * orc.loss_weights(labels, -100).reshape(-1)).sum()
    oloss.backward()
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * float(oloss)
    fails = []
    for n, p in w.model.named_parameters():
        if n.startswith('decoder.'):
            assert p.grad is None and not p.requires_grad, n
            continue
        try:
            grad_close(f'hf_llama_frozen.{n}', p.grad, esd[n].grad.numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, '; '.join(fails[:6])
    # weight-gradient GEMMs (k-major A operand) ran for the encoder only: none of the decoder's 2 x 4 + lm_head
    enc_dw = sum(calls)
    w.model.decoder.requires_grad_(True)
    calls.clear()
    monkeypatch.setattr(ops, 'gemm', lambda *a, **k: (calls.append(bool(k.get('a_kmajor'))), orig(*a, **k))[1])
    w.model.zero_grad(set_to_none=True)
    w.train_step(images.to(dev()), labels.to(dev()))[0].backward()
    monkeypatch.setattr(ops, 'gemm', orig)
    assert sum(calls) == enc_dw + 2 * 4 + 1, (sum(calls), enc_dw)


def test_llama_row_kernels():
    """i2t_rmsnorm_fwd / _bwd, i2t_rope (forward, inverse, position sources) and i2t_swiglu_fwd / _bwd against torch fp32"""
    from image2text_amd import ops
    g = torch.Generator().manual_seed(0)
    for M, d in ((70, 2304), (70, 1536), (33, 256)):        # two column panels (> 2048) and one; ragged row groups
        _rmsnorm_case(ops, g, M, d)
    _rope_swiglu_cases(ops, g)


def _rmsnorm_case(ops, g, M, d):
    x = torch.randn(M, d, generator=g).to(dev())
    w = (1 + 0.1 * torch.randn(d, generator=g)).to(dev())
    dy = torch.randn(M, d, generator=g).to(dev())
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    yr = wr * xr * torch.rsqrt(xr.pow(2).mean(-1, keepdim=True) + 1e-5)
    yr.backward(dy)
    y, y32, rs = torch.empty(M, d, dtype=torch.bfloat16, device=dev()), torch.empty(M, d, device=dev()), torch.empty(M, device=dev())
    ops.rmsnorm_fwd(x, w, y, rs, M, d, 1e-5, y_f32=y32)
    assert float((y32 - yr.detach()).abs().max()) < 1e-5 and float((y.float() - yr.detach()).abs().max()) < 3e-2
    for dy_in in (dy, dy.to(torch.bfloat16)):
        dx, dxb, dw = torch.full((M, d), 0.5, device=dev()), torch.empty(M, d, dtype=torch.bfloat16, device=dev()), torch.zeros(d, device=dev())
        ops.rmsnorm_bwd(dy_in, x, w, rs, dx, dw, M, d, dx_accumulate=True, dx_bf16=dxb)
        tol = 1e-4 if dy_in.dtype == torch.float32 else 3e-2
        assert float((dx - 0.5 - xr.grad).abs().max()) < tol * max(1.0, float(xr.grad.abs().max()))
        assert float((dw - wr.grad).abs().max()) < tol * float(wr.grad.abs().max()) + 1e-3
        assert float((dxb.float() - dx).abs().max()) < 2e-2 * float(dx.abs().max())


def _rope_swiglu_cases(ops, g):
    # rotary embedding
    B, T, H, hd = 2, 9, 3, 64
    rsz = H * hd + 32
    q = torch.randn(B * T, rsz, generator=g).to(torch.bfloat16).to(dev())
    inv = 1.0 / (10000 ** (torch.arange(0, hd, 2).float() / hd))
    ang = torch.outer(torch.arange(40).float(), inv)
    tab = torch.cat((ang.cos(), ang.sin()), -1).to(dev())

    def ref_rope(t, pos, sign=1.0):
        t = t.float()[:, :H * hd].view(-1, H, hd)
        c, s_ = tab[pos, :hd // 2][:, None], sign * tab[pos, hd // 2:][:, None]
        a, b = t[..., :hd // 2], t[..., hd // 2:]
        return torch.cat((a * c - b * s_, b * c + a * s_), -1).reshape(-1, H * hd)
    pos_dense = (3 + torch.arange(B * T, device=dev()) % T)
    for kw, pos in ((dict(pos_offset=3, T=T), pos_dense),
                    (dict(pos=torch.arange(B * T, dtype=torch.int32, device=dev()).flip(0).contiguous()), torch.arange(B * T, device=dev()).flip(0)),
                    (dict(pos_ptr=torch.tensor([5], dtype=torch.int32, device=dev()), pos_offset=2), torch.full((B * T,), 7, device=dev()))):
        for inverse in (False, True):
            z = q.clone()
            ops.rope(z, rsz, 0, H, hd, tab, B * T, inverse=inverse, **kw)
            want = ref_rope(q, pos, -1.0 if inverse else 1.0)
            assert float((z[:, :H * hd].float() - want).abs().max()) < 2e-2 * float(want.abs().max())
            assert torch.equal(z[:, H * hd:], q[:, H * hd:])                       # columns past the heads untouched
    # SwiGLU
    M, ff = 37, 96
    gu = torch.randn(M, 2 * ff, generator=g).to(torch.bfloat16).to(dev())
    dh = torch.randn(M, ff, generator=g).to(torch.bfloat16).to(dev())
    gr = gu.float().clone().requires_grad_(True)
    hr = F.silu(gr[:, :ff]) * gr[:, ff:]
    hr.backward(dh.float())
    h, dgu = torch.empty(M, ff, dtype=torch.bfloat16, device=dev()), torch.empty(M, 2 * ff, dtype=torch.bfloat16, device=dev())
    ops.swiglu_fwd(gu, h, M, ff)
    ops.swiglu_bwd(dh, gu, dgu, M, ff)
    assert float((h.float() - hr.detach()).abs().max()) < 2e-2 * float(hr.abs().max())
    assert float((dgu.float() - gr.grad).abs().max()) < 2e-2 * float(gr.grad.abs().max())


def test_gpt2_hf_decoder_dropout_sites_train_mode(tmp_path, monkeypatch):
    """a checkpoint with resid_pdrop = embd_pdrop = attn_pdrop = 0.1: the train step draws masks at transformers' sites (the DropPlan
    carries the cross-attention residual site, the per-token q/k/v multipliers of the nanoGPT block are off), is reproducible under
    torch.manual_seed, differs from the eval-mode loss, and back-propagates finite gradients; unequal rates are refused by name"""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.models.decoder import Decoder
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    _local_hf_gpt2(tmp_path, monkeypatch, resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.1)
    cfg = _model_config(True, True)
    w = ModelTrainerWrapper(cfg, fake_tokenizer(384), TrainerWrapperConfig(), ignore_index=-100).to(dev()).train()
    eng = w.model._engine
    assert eng.dec.dropout == 0.1 and eng.dec.attn_dropout == 0.0
    images, labels = synthetic_batch(4, 32, 14, 384, seed=5)
    images, labels = images.to(dev()), labels.to(dev())
    losses = []
    for seed in (3, 4, 3):        # a change of torch's seed restarts the step-seed sequence (engine.prepare)
        torch.manual_seed(seed)
        loss, _ = w.train_step(images, labels)
        losses.append(float(loss.detach()))
    plan = eng.dec_drop
    assert plan.get(0, 'xresid') is not None and plan.get(0, 'qkv') is None and plan.get(0, 'sdpa') is not None
    loss.backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in w.model.parameters())
    w.eval()
    with torch.no_grad():
        vloss, _ = w.val_step(images, labels)
    # (same masks -> same loss up to the order of the cross-entropy's atomic row sums)
    assert abs(losses[0] - losses[2]) < 1e-5 * losses[0] and abs(losses[0] - losses[1]) > 1e-4 and abs(losses[0] - float(vloss)) > 1e-4
    assert abs(losses[0] - float(vloss)) < 0.2 * float(vloss)
    _local_hf_gpt2(tmp_path, monkeypatch, name='gpt2-uneven', resid_pdrop=0.1, embd_pdrop=0.1, attn_pdrop=0.0)
    with pytest.raises(NotImplementedError, match='pdrop'):
        Decoder.from_config(_hf_decoder_config(name='gpt2-uneven'))


# ------------------------------------------------------------------------------------------------------------------------------
# LoRA on the GPT-2 plugin (reference decoder.py:133-134 -> models/utils.py:46-65 -> peft LoraModel; peft is not in this image, so
# the checker restates its published layer: y = base(x) + lora_B(lora_A(dropout(x))) * lora_alpha / r around transformers' Conv1D)
# ------------------------------------------------------------------------------------------------------------------------------
class _LoraConv1D(torch.nn.Module):
    def __init__(self, base, A, B, scale):
        super().__init__()
        self.base, self.scale, self.mask = base, scale, None
        self.A, self.B = torch.nn.Parameter(A.clone()), torch.nn.Parameter(B.clone())

    def forward(self, x):
        xd = x if self.mask is None else x * self.mask.view(x.shape)           # dropout on the adapter's input only
        return self.base(x) + (xd @ self.A.t() @ self.B.t()) * self.scale


_LORA_MODS = {'attn_c_attn': ('attn', 'c_attn'), 'xattn_c_attn': ('crossattention', 'c_attn'), 'mlp_c_fc': ('mlp', 'c_fc'), 'mlp_c_proj': ('mlp', 'c_proj')}


def _lora_twin(model):
    """transformers GPT-2 + the adapters, from the plugin's (LoraModel-keyed) state dict"""
    from transformers import GPT2Config, GPT2LMHeadModel
    dec = model.decoder
    hc = dec.hf_config
    hf = GPT2LMHeadModel(GPT2Config(n_layer=hc.n_layer, n_head=hc.n_head, n_embd=hc.n_embd, n_positions=hc.n_positions,
                                    vocab_size=model._engine.dec.V, add_cross_attention=True, resid_pdrop=0.0, embd_pdrop=0.0,
                                    attn_pdrop=0.0)).eval()
    sd = {k: v.detach().cpu().clone() for k, v in dec.state_dict().items()}
    hf.load_state_dict({k[len('backbone.model.'):].replace('.base_layer.', '.'): v for k, v in sd.items() if '.lora_' not in k}, strict=True)
    wraps = {}
    for l in range(hc.n_layer):
        for site in dec.lora.sites:
            parent, leaf = _LORA_MODS[site]
            mod = getattr(hf.transformer.h[l], parent)
            key = f'backbone.model.transformer.h.{l}.{parent}.{leaf}.lora_'
            w = _LoraConv1D(getattr(mod, leaf), sd[key + 'A.default.weight'], sd[key + 'B.default.weight'], dec.lora.scale)
            setattr(mod, leaf, w)
            wraps[(l, site)] = w
    return hf, wraps


@pytest.mark.parametrize('p_lora', [0.0, 0.25])
def test_gpt2_hf_decoder_lora(tmp_path, monkeypatch, p_lora):
    """lora_spec on the GPT-2 plugin: the adapters in the K panel of their layers' GEMMs (forward), the four thin adapter GEMMs and the
    skipped base-weight GEMMs (backward), frozen parameters without gradient, the adapter's input dropout with the SAME mask in forward
    and backward (p = 0.25: the oracle is handed the masks of the step's DropPlan), merged weights in KV-cache generation."""
    from oracle import reference_model as orc
    from image2text_amd import rng
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from test_host_cpu import _lora_spec
    tag = f'hf_gpt2_lora.p{p_lora}'
    _local_hf_gpt2(tmp_path, monkeypatch)
    cfg = _model_config(True, True, lora_spec=_lora_spec(lora_dropout=p_lora))
    m = VisionEncoderDecoder(cfg)
    keep = {k: v.detach().clone() for k, v in m.decoder.state_dict().items()}
    det_init_(m, seed=0)
    m.decoder.load_state_dict(keep)
    with torch.no_grad():
        g = torch.Generator().manual_seed(9)
        for n, p in m.decoder.lora_params.items():
            if n.endswith('_B'):
                p.copy_(torch.randn(p.shape, generator=g) * 0.05)          # lora_B starts at zero: give the adapters something to do
    hf, wraps = _lora_twin(m)
    esd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if not k.startswith('decoder.')}
    for p in hf.parameters():
        p.requires_grad_(True)
    frozen = {n for n, p in m.named_parameters() if not p.requires_grad}
    assert 'decoder.transformer.h.0.attn.c_attn.weight' in frozen and 'decoder.transformer.h.0.cross_attn.in_proj_weight' not in frozen
    m = m.to(dev()).train()
    eng = m._engine
    images, labels = synthetic_batch(3, 32, 12, 384, seed=17)
    ids = labels.clamp(min=0)
    n_p = cfg.vision_encoder_config.n_cls
    gg = torch.Generator().manual_seed(2)
    wh = torch.randn(3, n_p + 12, 128, generator=gg) * 0.05
    wl = torch.randn(3, 12, 384, generator=gg) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    if p_lora > 0:          # the masks this step drew, replicated on the host for the oracle's adapters
        plan = eng.dec_drop
        for (l, site), w in wraps.items():
            _, key, thr, scale = plan.get(l, f'lora_{site}')
            rows, K = (3 * n_p if site == 'xattn_c_attn' else 3 * (n_p + 12)), w.A.shape[1]
            w.mask = rng.keep_mask(key, rows * K, thr).view(rows, K).float() * scale
    enc, ologits, ohid = _reference_forward(orc, esd, hf, cfg, images, ids, True, True)
    for name, got, ref, tol in (('logits', out.logits, ologits, 1e-2), ('hidden', out.hidden_state, ohid, 1.5e-2)):
        err, scale_ = float((got.float().cpu() - ref.detach()).abs().max()), max(1.0, float(ref.detach().abs().max()))
        REPORT[f'{tag}.{name}'] = {'max_abs_err': err, 'tol': tol * scale_}
        assert err <= tol * scale_, (name, err, tol * scale_)
    # the oracle's own restatement of GPT-2 + peft's LoRA layer (oracle/hf_decoders.py) says the same as the wrapped transformers twin
    from oracle import hf_decoders as hfo
    with torch.no_grad():
        psd = {k.replace('.base.', '.'): v for k, v in hf.named_parameters() if not (k.endswith('.A') or k.endswith('.B'))}
        psd['lm_head.weight'] = psd['transformer.wte.weight']
        lmap = {f'transformer.h.{l}.{".".join(_LORA_MODS[site])}': (w.A, w.B, w.scale, w.mask) for (l, site), w in wraps.items()}
        rl, rh = hfo.soft_prompt_forward(lambda e_, m_: hfo.gpt2_decoder(psd, 2, 2, e_, m_, lora=lmap), psd['transformer.wte.weight'],
                                         enc.detach(), ids, 64, True)
        assert float((rl - ologits).abs().max()) < 1e-4 and float((rh - ohid).abs().max()) < 1e-4
    ((out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()).backward()
    ((ohid * wh).sum() + (ologits * wl).sum()).backward()
    from image2text_amd.models.decoder import Decoder
    shell = Decoder.from_config(_hf_decoder_config(use_cross_attn=True))              # un-adapted container: transformers keys -> internal layout
    hf_g = {'backbone.' + k.replace('.base.', '.'): (p.grad if p.grad is not None else torch.zeros_like(p))
            for k, p in hf.named_parameters() if not (k.endswith('.A') or k.endswith('.B'))}
    hf_g['backbone.lm_head.weight'] = hf_g['backbone.transformer.wte.weight']
    shell.load_state_dict(hf_g, strict=True)
    ref_grads = {k: v.grad for k, v in esd.items() if v.grad is not None}
    ref_grads.update({'decoder.' + k: v.detach() for k, v in shell.named_parameters()})
    for (l, site), w in wraps.items():
        ref_grads[f'decoder.lora_params.h{l}_{site}_A'], ref_grads[f'decoder.lora_params.h{l}_{site}_B'] = w.A.grad, w.B.grad
    fails, checked = [], 0
    for name, p in m.named_parameters():
        if name in frozen:
            assert p.grad is None, f'{name} is frozen but received a gradient'
            continue
        checked += 1
        try:
            grad_close(f'{tag}.{name}', p.grad, ref_grads[name].numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert checked > 20 and not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    if p_lora > 0:
        return
    # the fused optimizer leaves the frozen base weights alone; generation runs on merged weights
    from image2text_amd.training.optim import FusedAdamW
    before = {n: p.detach().clone() for n, p in m.named_parameters()}
    opt = FusedAdamW(m.parameters(), m, lr=1e-2, weight_decay=0.1)
    opt.step()
    moved = {n: not torch.equal(before[n], p.detach()) for n, p in m.named_parameters()}
    assert not any(moved[n] for n in frozen) and all(moved[n] for n in moved if n not in frozen)
    m.eval()
    tok = fake_tokenizer(384)
    with torch.no_grad():
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = m.generate(images.to(dev()), prompt, max_new_tokens=10, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(10):
            lg = m(images=images.to(dev()), ids=cur).logits[:, -1].float().cpu()
            lg = orc.apply_ngram_ban(cur.cpu(), lg, cfg.no_repeat_n_grams)
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg[torch.isfinite(lg)].abs().max().clamp(min=1.0)
            same_prefix = (gen[:, :cur.shape[1]].cpu() == cur.cpu()).all(dim=1)
            ok = gen[:, cur.shape[1]].cpu() == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True).to(dev())), dim=1)
        REPORT[f'{tag}.generate_vs_forward'] = {'agree': agree, 'of': total}
        assert total >= 8 and agree == total, (agree, total)


def test_hf_decoder_edges(tmp_path, monkeypatch):
    """captions longer than the position table (cropped like the reference crops its concatenated embeddings, v_e_d.py:88), the
    sampling modes of generate() on a prefixed KV cache (reproducible under torch.manual_seed), the MLM-corruption trainer option,
    and the refusals of this path (contrastive loss, momentum distillation)"""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    _local_hf_gpt2(tmp_path, monkeypatch)
    cfg = _model_config(True, True)
    tok = fake_tokenizer(384)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    hf = _hf_twin(w.model, True)
    esd = {k: v.detach().clone() for k, v in w.model.state_dict().items() if not k.startswith('decoder.')}
    w = w.to(dev()).eval()
    n_p = cfg.vision_encoder_config.n_cls                                   # 8 prompt rows + 56 text rows fill the 64 positions
    images, labels = synthetic_batch(3, 32, 70, 384, seed=23, min_len=60)
    with torch.no_grad():
        vloss, _ = w.val_step(images.to(dev()), labels.to(dev()))
        ids, _ = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id, -100)
        _, ologits, _ = _reference_forward(orc, esd, hf, cfg, images, ids, True, True)
        assert ologits.shape[1] == 64 - n_p
        lab = labels[:, :64 - n_p]
        ce = F.cross_entropy(ologits.reshape(-1, 384), lab.reshape(-1), ignore_index=-100, reduction='none')
        oloss = float((ce * orc.loss_weights(lab, -100).reshape(-1)).sum())
        assert abs(float(vloss) - oloss) <= 1e-2 * oloss, (float(vloss), oloss)
        prompt = torch.full((3, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        outs = []
        for seed in (5, 6, 5):
            torch.manual_seed(seed)
            outs.append(w.model.generate(images.to(dev()), prompt, max_new_tokens=20, temperature=0.9, top_k=8, nucleus_p=0.9).cpu())
        assert torch.equal(outs[0], outs[2]) and not torch.equal(outs[0], outs[1]) and int(outs[0].max()) < 384
        with pytest.raises(AssertionError):
            w.model.generate(images.to(dev()), prompt, max_new_tokens=64 - n_p)                # prompt + new tokens exceed the text window
    wm = ModelTrainerWrapper(cfg, tok.__class__(**{**tok.__dict__, 'mask_token_id': 380}), TrainerWrapperConfig(mask_fraction=0.2, random_mask_fraction=0.1),
                             ignore_index=-100).to(dev()).train()
    loss, _ = wm.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    assert torch.isfinite(loss) and all(p.grad is not None and torch.isfinite(p.grad).all() for p in wm.model.parameters())
    for kw in (dict(add_contrastive_loss=True), dict(moco_momentum=0.99, moco_alpha=0.4)):
        with pytest.raises(NotImplementedError, match='soft prompt'):
            ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(**kw), ignore_index=-100)
