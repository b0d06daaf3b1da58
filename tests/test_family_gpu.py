"""Model-level parity of the nano-mini family (multi-query attention, MoE rotators, sparse token subsets) on the MI355X against the
fixtures the reference produced (tests/golden/mini_*.npz, nano_mini_shapes.npz; tools/gen_goldens_mini.py) and the CPU oracle.

Expert routing is a discontinuous function of the gate logits: where two experts' gate values are closer than the bf16 noise of
the gate GEMM the HIP path may choose differently from the fp32 reference.  The tests therefore (a) require the device's choice to
equal the reference's wherever the reference's top-k margin exceeds MOE_MARGIN, and at >= 97 % of all (token, site) pairs, and
(b) compare values against the ORACLE run on the device's own choices (oracle moe_io 'forced'), which itself is pinned to the
reference (tests/test_family_oracle.py).  Tolerances otherwise as tests/test_model_gpu.py.
"""
import numpy as np
import pytest
import torch

from conftest import load_golden
from image2text_amd.synth import det_init_, fake_tokenizer, mini_config, nano_mini_config, sharpen_gates_, synthetic_batch
from test_family_oracle import VARIANTS, variant_config
from test_model_gpu import REPORT, dev, grad_close, hidden_tol, logits_tol, maxerr

pytestmark = pytest.mark.gpu
MOE_MARGIN = 4e-3


@pytest.fixture(scope='module', autouse=True)
def write_family_report():
    import json
    import os
    yield
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_family.json', 'w') as fh:
        json.dump({k: v for k, v in REPORT.items() if k.startswith(('mini', 'variant', 'nano_mini', 'unit_test', 'advpos'))}, fh, indent=1, sort_keys=True)


def build(cfg, train=False):
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    m = sharpen_gates_(det_init_(VisionEncoderDecoder(cfg), seed=0))
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev())
    return (m.train() if train else m.eval()), sd


def device_choices(trace, top_k_of, call=0):
    """engine.moe_trace -> {site: LongTensor [N, top_k] (expert indices, by descending weight)} of the call-th forward of each site"""
    out = {}
    for site, calls in trace.items():
        _gates, wsel = calls[min(call, len(calls) - 1)]
        k = top_k_of(site)
        out[site] = torch.topk(wsel.detach().float().cpu(), k, dim=-1).indices
    return out


def check_choices(tag, forced, record, rows=None):
    """device choice vs the oracle's own (fp32) choice: equal away from ties, and almost everywhere"""
    bad = total = 0
    for site, idx in forced.items():
        gates, own = record[site]
        k = idx.shape[1]
        srt = gates.sort(dim=1, descending=True).values
        margin = srt[:, k - 1] - srt[:, k] if k < gates.shape[1] else torch.ones(gates.shape[0])
        diff = (idx.sort(dim=1).values != own.sort(dim=1).values).any(dim=1)
        assert not bool((diff & (margin > MOE_MARGIN)).any()), f'{tag} {site}: expert choice differs at margin {float(margin[diff].max()):.4g}'
        bad += int(diff.sum())
        total += diff.numel()
    REPORT[f'{tag}.moe_choice_flips'] = {'flips': bad, 'of': total}
    assert bad <= 0.03 * max(total, 1), (tag, bad, total)


def top_k_fn(cfg):
    ek = getattr(cfg.vision_encoder_config.transformer_config.rotator_config, 'top_k', 1)
    dk = getattr(cfg.decoder_config.transformer_config.rotator_config, 'top_k', 1)
    return lambda site: ek if site.startswith('encoder.') else dk


def test_mini_forward_against_reference_and_oracle():
    from oracle import reference_model as orc
    f = load_golden('mini_forward.npz')
    cfg = mini_config()
    m, sd = build(cfg)
    images, ids = torch.from_numpy(f['images']), torch.from_numpy(f['ids'])
    m._engine.moe_trace = {}
    with torch.no_grad():
        out = m(images=images.to(dev()), ids=ids.to(dev()))
    trace = dict(m._engine.moe_trace)
    m._engine.moe_trace = None
    assert tuple(out.logits.shape) == f['logits'].shape and tuple(out.hidden_state.shape) == f['hidden_state'].shape
    # the reference's own numbers (choices may differ at near-ties -> looser, and a bound on how many)
    text_sites = {s: v for s, v in trace.items()}
    forced = device_choices(text_sites, top_k_fn(cfg))
    io = {'forced': {}, 'record': {}}
    # the device runs the TEXT rows of decoder blocks (the prompt rows go through a separate segment): force only the encoder
    # sites in the full-sequence oracle and compare the decoder through the text-segment oracle below
    io['forced'] = {s: i for s, i in forced.items() if s.startswith('encoder.')}
    with torch.no_grad():
        enc_o, logits_o, hidden_o = orc.forward(sd, cfg, images, ids, moe_io=io)
    check_choices('mini.encoder', io['forced'], {s: io['record'][s] for s in io['forced']})
    maxerr('mini.encoder_output', out.encoder_output, enc_o.numpy(), hidden_tol(enc_o.numpy()))
    flips = sum(int((forced[s].sort(1).values != torch.from_numpy(f[f'moe.{s}.idx']).sort(1).values).any(1).sum())
                for s in forced if s.startswith('encoder.'))
    REPORT['mini.encoder_flips_vs_reference'] = flips
    if flips == 0:
        maxerr('mini.encoder_output_vs_reference', out.encoder_output, f['encoder_output'], hidden_tol(f['encoder_output']))
    # logits / hidden: reference values where no routing decision differs anywhere, else the bulk statistic
    err = np.abs(out.logits.float().cpu().numpy() - f['logits'])
    REPORT['mini.logits_vs_reference'] = {'max': float(err.max()), 'median': float(np.median(err)), 'tol': logits_tol(f['logits'])}
    assert float(np.quantile(err, 0.9)) <= logits_tol(f['logits'])
    assert (out.logits.argmax(-1).cpu().numpy() == f['logits'].argmax(-1)).mean() > 0.9


def test_mini_train_step_loss_and_every_gradient_vs_oracle_on_device_routing():
    """train_step on the mini model: loss + the gradient of every parameter against the oracle evaluated with the device's own
    expert choices; the choices themselves against the oracle's fp32 gates; and the loss against the reference's."""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    f, tr = load_golden('mini_forward.npz'), load_golden('mini_train.npz')
    cfg = mini_config()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    sd = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    w = w.to(dev()).train()
    w.pack_rows = False                         # the oracle's text-segment form is dense (b, t); packing is checked separately below
    images, labels = torch.from_numpy(f['images']), torch.from_numpy(f['labels'])
    w.model._engine.moe_trace = {}
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    trace = dict(w.model._engine.moe_trace)
    w.model._engine.moe_trace = None
    loss.backward()
    ref = float(tr['loss'])
    REPORT['mini.train_loss'] = {'got': float(loss.detach()), 'ref': ref}
    assert abs(float(loss.detach()) - ref) <= 1e-2 * max(1.0, ref)
    forced = device_choices(trace, top_k_fn(cfg))
    # decoder sites on the device hold the rows of the sparse subset of the TEXT segment, exactly the oracle's text-segment rows
    io = {'forced': forced, 'record': {}}
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    oloss = orc.lm_step_text_segment(osd, cfg, images, labels, tok, moe_io=io)
    oloss.backward()
    check_choices('mini.train', forced, io['record'])
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * max(1.0, float(oloss))
    fails, n = [], 0
    for name, p in w.model.named_parameters():
        assert p.grad is not None, name
        g = osd[name].grad
        try:
            grad_close(f'mini.{name}', p.grad, (g if g is not None else torch.zeros_like(osd[name])).numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
        n += 1
    assert n >= 100
    assert not fails, f'{len(fails)} of {n} gradients out of tolerance: ' + '; '.join(fails[:8])
    # row packing (ragged captions) is result-preserving for the family too
    w2 = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    w2.model.load_state_dict(sd)
    w2 = w2.to(dev()).train()
    assert w2.pack_rows
    loss2, _ = w2.train_step(images.to(dev()), labels.to(dev()))
    loss2.backward()
    assert abs(float(loss2.detach()) - float(loss.detach())) <= 2e-3 * max(1.0, ref)
    for (name, p), (_, p2) in zip(w.model.named_parameters(), w2.model.named_parameters()):
        grad_close(f'mini.packed.{name}', p2.grad, p.grad.detach().float().cpu().numpy(), rel=5e-2, cos=0.995)


@pytest.mark.parametrize('name', list(VARIANTS))
def test_mini_variants_forward_and_loss(name):
    """one-feature variants (multi-query only, MoE only, sparse only, 128-wide multi-head, top-2 decoder, single-layer gate,
    16-wide heads, cross-attention in every layer): logits vs the oracle on the device's routing, loss vs the reference"""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    f = load_golden('mini_variants.npz')
    cfg = variant_config(name)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    sd = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    w = w.to(dev()).train()
    images, labels = torch.from_numpy(f['images']), torch.from_numpy(f['labels'])
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    ref = float(f[f'{name}.loss'])
    REPORT[f'variant.{name}.loss'] = {'got': float(loss.detach()), 'ref': ref}
    assert abs(float(loss.detach()) - ref) <= 1e-2 * max(1.0, ref)
    for n_, p in w.model.named_parameters():
        key = f'{name}.gradnorm.{n_}'
        assert p.grad is not None and torch.isfinite(p.grad).all(), n_
        if key in f and float(f[key]) > 1e-4 and '.expert' not in n_:         # (an expert's gradient depends on which tokens were routed to it)
            got = float(p.grad.norm())
            assert abs(got - float(f[key])) <= 0.15 * float(f[key]) + 1e-5, (n_, got, float(f[key]))
    w.eval()
    with torch.no_grad():
        out = w.model(images=images.to(dev()), ids=torch.from_numpy(f['ids']).to(dev()))
    err = np.abs(out.logits.float().cpu().numpy() - f[f'{name}.logits'])
    REPORT[f'variant.{name}.logits'] = {'max': float(err.max()), 'q90': float(np.quantile(err, 0.9))}
    assert float(np.quantile(err, 0.9)) <= logits_tol(f[f'{name}.logits'])
    if not any('experts' in k for k in sd):                                   # no routing: the reference's logits hold everywhere
        maxerr(f'variant.{name}.logits_all', out.logits, f[f'{name}.logits'], logits_tol(f[f'{name}.logits']))


def test_nano_mini_full_size_forward_and_loss():
    """gpu/nano-mini.yaml at full size (12 + 12 layers of 1024, 8 x 128 heads on one K/V head, 4 experts, half the positions
    attended), B = 2: the reference's recorded statistics"""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    g = load_golden('nano_mini_shapes.npz')
    cfg = nano_mini_config()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    w = w.to(dev()).eval()
    images, labels = synthetic_batch(2, 128, 48, cfg.decoder_config.vocab_size, seed=1)
    assert np.array_equal(labels.numpy(), g['labels'])
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    bos_ids = torch.cat((torch.full((2, 1), tok.bos_token_id, dtype=torch.long), ids), dim=1)[:, :48]
    eng = w.model._engine
    eng.moe_trace = {}
    with torch.no_grad():
        out = w.model(images=images.to(dev()), ids=bos_ids.to(dev()))
        vloss, _ = w.val_step(images.to(dev()), labels.to(dev()))
    trace = dict(eng.moe_trace)
    eng.moe_trace = None
    assert tuple(out.logits.shape) == (2, 48, cfg.decoder_config.vocab_size) and tuple(out.encoder_output.shape) == (2, 64, 1024)
    # routing vs the reference (encoder sites: same rows in the same order)
    flips = total = 0
    for site, calls in trace.items():
        if not site.startswith('encoder.'):
            continue
        _g, wsel = calls[0]
        want, margin = torch.from_numpy(g[f'moe.{site}.idx'].astype(np.int64)), g[f'moe.{site}.margin']
        got = torch.topk(wsel.float().cpu(), want.shape[1], dim=-1).indices
        diff = (got.sort(1).values != want.sort(1).values).any(1).numpy()
        flips += int(diff.sum())
        total += diff.size
    REPORT['nano_mini.encoder_moe_flips'] = {'flips': flips, 'of': total}
    assert flips <= 0.05 * total
    err_e = np.abs(out.encoder_output.float().cpu().numpy() - g['encoder_output'])
    REPORT['nano_mini.encoder_output_vs_reference'] = {'max': float(err_e.max()), 'q99': float(np.quantile(err_e, 0.99)),
                                                       'median': float(np.median(err_e))}
    assert float(np.median(err_e)) <= 2e-2
    # same routing -> same numbers: the oracle on the device's encoder choices
    from oracle import reference_model as orc
    sd = {k: v.detach().cpu().clone() for k, v in w.model.state_dict().items()}
    forced = {s: i for s, i in device_choices(trace, top_k_fn(cfg)).items() if s.startswith('encoder.')}
    with torch.no_grad():
        enc_o = orc.encode(sd, cfg, images, moe_io={'forced': forced})
    maxerr('nano_mini.encoder_output_vs_oracle_on_device_routing', out.encoder_output, enc_o.numpy(), hidden_tol(enc_o.numpy()))
    lse = torch.logsumexp(out.logits.float(), dim=-1).cpu().numpy()
    REPORT['nano_mini.val_loss'] = {'got': float(vloss), 'ref': float(g['val_loss'])}
    assert float(np.abs(lse - g['logits_lse']).max()) <= 5e-2
    assert abs(float(vloss) - float(g['val_loss'])) <= 1e-2 * float(g['val_loss'])
    w.train()
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    assert abs(float(loss.detach()) - float(g['train_loss'])) <= 1e-2 * float(g['train_loss'])
    worst = 0.0
    for n_, p in w.model.named_parameters():
        assert p.grad is not None and torch.isfinite(p.grad).all(), n_
        ref = float(g[f'gradnorm.{n_}'])
        if ref > 1e-3:
            worst = max(worst, abs(float(p.grad.norm()) - ref) / ref)
    REPORT['nano_mini.worst_gradnorm_rel_err'] = worst
    assert worst <= 0.35


def _decode_vs_forward(tag, m, images, gold, steps):
    """KV-cache decode step (sparse slot tables, grouped decode attention, MoE step) == the full forward, teacher-forced"""
    from image2text_amd.decoding import GreedyDecoder
    V = m.config.decoder_config.vocab_size
    with torch.no_grad():
        full = m(images=images, ids=torch.from_numpy(gold[:, :-1]).to(dev())).logits.float()
    dec = GreedyDecoder(m)
    worst = 0.0
    for t in steps:
        # (return_margins: the step in its logits form -- without it the greedy head leaves only the segment maxima of the row)
        dec.generate(images, torch.from_numpy(gold[:, :t + 1]).to(dev()), 1, use_graph=(t % 2 == 0), return_margins=True)
        lg = dec._state.logits[:, :V].float()
        err = (lg - full[:, t]).abs()
        tol = logits_tol(full[:, t].cpu().numpy())
        worst = max(worst, float(err.max()) / tol)
        assert float(err.quantile(0.99)) <= tol, (tag, t, float(err.quantile(0.99)), tol)
        assert float(err.max()) <= 4 * tol, (tag, t, float(err.max()), tol)          # (a routing near-tie may differ between the two paths)
    REPORT[f'{tag}.decode_vs_forward_worst_over_tol'] = worst


def test_mini_greedy_decode():
    from test_model_gpu import assert_greedy_matches
    f = load_golden('mini_decode.npz')
    cfg = mini_config()
    m, _ = build(cfg)
    images = torch.from_numpy(f['images']).to(dev())
    _decode_vs_forward('mini', m, images, f['ids'], (0, 1, 2, 3, 5, 8, 12, 16, 19))
    # the reference's greedy tokens wherever its top-1 margin is clear of the bf16 logit noise (untrained model: logits ~ +-1)
    r = assert_greedy_matches(m, images, f['ids'], f['margins'], 1, eps=0.08)
    REPORT['mini.greedy'] = {'low_margin_restarts': r, 'steps_with_margin_ge_eps': int((f['margins'] >= 0.08).sum()), 'steps': int(f['margins'].size)}
    prompt = torch.from_numpy(f['prompt']).to(dev())
    a = m.generate(images, prompt, max_new_tokens=20, temperature=1.0, top_k=1)
    b = m.generate(images, prompt, max_new_tokens=20, temperature=1.0, top_k=1)
    assert a.shape == (4, 21) and torch.equal(a, b)
    s = m.generate(images, prompt, max_new_tokens=8, temperature=0.7, nucleus_p=0.6)          # eval_model's call shape (trainer.py:41-56)
    assert s.shape == (4, 9) and int(s.min()) >= 0 and int(s.max()) < cfg.decoder_config.vocab_size


@pytest.mark.parametrize('name', ['mq_only', 'moe_only', 'sparse_only', 'mh128', 'heads16', 'all_cross', 'top2_dec', 'advpos'])
def test_variant_decode_matches_forward(name):
    f = load_golden('mini_decode.npz')
    m, _ = build(variant_config(name))
    _decode_vs_forward(f'variant.{name}', m, torch.from_numpy(f['images']).to(dev()), f['ids'], (0, 1, 4, 9, 15, 19))


def test_mini_train_step_with_dropout_matches_oracle_on_the_same_masks_and_routing():
    """dropout 0.1 / attn_dropout 0.1 in both towers of the mini model: the oracle is fed the step's exact masks (DropPlans rebuilt on
    the host; sparse blocks index their sites over the gathered rows) and the device's expert choices"""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    torch.manual_seed(20241004)
    f = load_golden('mini_forward.npz')
    cfg = mini_config(dropout=0.1)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    sd = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    w = w.to(dev()).train()
    w.pack_rows = False
    images, labels = torch.from_numpy(f['images']), torch.from_numpy(f['labels'])
    eng = w.model._engine
    eng.moe_trace = {}
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    trace = dict(eng.moe_trace)
    eng.moe_trace = None
    loss.backward()
    plans = (eng.enc_drop, eng.dec_drop)
    assert plans[0] is not None and plans[1] is not None
    forced = device_choices(trace, top_k_fn(cfg))
    io = {'forced': forced, 'record': {}}
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    oloss = orc.lm_step_text_segment(osd, cfg, images, labels, tok, plans, moe_io=io)
    oloss.backward()
    check_choices('mini_dropout', forced, io['record'])
    REPORT['mini_dropout.train_loss'] = {'got': float(loss.detach()), 'ref': float(oloss)}
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * float(oloss)
    fails = []
    for name, p in w.model.named_parameters():
        g = osd[name].grad
        try:
            grad_close(f'mini_dropout.{name}', p.grad, (g if g is not None else torch.zeros_like(osd[name])).numpy(), rel=0.1, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    with torch.no_grad():
        clean = orc.lm_step_text_segment(sd, cfg, images, labels, tok)
    assert abs(float(clean) - float(oloss)) > 1e-6


@pytest.mark.parametrize('packed', [False, True])
def test_advanced_positional_mlp_train_step_every_gradient(packed):
    """use_advanced_pos_emb (one MLP per position as the decoder's wpe, layers.py:617-638) on the mini family model: loss and the
    gradient of every parameter -- all per-position MLPs included -- against the oracle on the device's routing; with packed ragged
    caption rows the position groups are ragged too"""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    f = load_golden('mini_variants.npz')
    cfg = variant_config('advpos')
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    sd = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    w = w.to(dev()).train()
    w.pack_rows = packed
    images, labels = torch.from_numpy(f['images']), torch.from_numpy(f['labels'])
    eng = w.model._engine
    eng.moe_trace = {}
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    trace = dict(eng.moe_trace)
    eng.moe_trace = None
    loss.backward()
    ref = float(f['advpos.loss'])
    assert abs(float(loss.detach()) - ref) <= 1e-2 * max(1.0, ref)
    if packed:      # the oracle's dense rows differ from the packed ones in the decoder sites: compare with the dense device run instead
        w2 = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
        w2.model.load_state_dict(sd)
        w2 = w2.to(dev()).train()
        w2.pack_rows = False
        l2, _ = w2.train_step(images.to(dev()), labels.to(dev()))
        l2.backward()
        assert abs(float(l2.detach()) - float(loss.detach())) <= 2e-3 * max(1.0, ref)
        for (name, p), (_, p2) in zip(w.model.named_parameters(), w2.model.named_parameters()):
            grad_close(f'advpos.packed.{name}', p.grad, p2.grad.detach().float().cpu().numpy(), rel=5e-2, cos=0.995)
        return
    forced = device_choices(trace, top_k_fn(cfg))
    io = {'forced': forced, 'record': {}}
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    oloss = orc.lm_step_text_segment(osd, cfg, images, labels, tok, moe_io=io)
    oloss.backward()
    check_choices('advpos', forced, io['record'])
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * max(1.0, float(oloss))
    fails, n_pos = [], 0
    for name, p in w.model.named_parameters():
        g = osd[name].grad
        n_pos += '.wpe.models.' in name
        try:
            grad_close(f'advpos.{name}', p.grad, (g if g is not None else torch.zeros_like(osd[name])).numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert n_pos == 40 * 8
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:8])


def test_reference_unit_test_runs_and_matches_the_reference():
    """The reference's only unit test (models/vision_encoder_decoder_test.py) at its own sizes -- 96 images of 128 x 128, 192 ids, a
    random boolean mask, generate with a nucleus -- then the same model on the fixture the reference produced
    (tests/golden/unit_test_config.npz): logits, the full hidden_state (prompt rows of the non-causal decoder see the text rows),
    greedy tokens by re-evaluation."""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.synth import reference_unit_test_config
    cfg = reference_unit_test_config()
    torch.manual_seed(0)
    model = VisionEncoderDecoder(cfg).to(dev())
    inp = torch.randint(0, 256, (96, 3, 128, 128)).float()
    ids = torch.randint(0, 1024, (96, 192,))
    attn_mask = torch.randint(0, 2, (192, 192), dtype=torch.bool)
    outs = model(images=inp.to(dev()), ids=ids.to(dev()), attn_msk=attn_mask.to(dev()))
    assert (96, 24, 64) == tuple(outs.encoder_output.shape)
    assert (96, 192, 1024) == tuple(outs.logits.shape)
    assert (96, 24 + 192, 64) == tuple(outs.hidden_state.shape)
    assert torch.isfinite(outs.logits).all() and torch.isfinite(outs.hidden_state).all()
    generated_ids = model.generate(inp[:2].to(dev()), ids[:2].to(dev()), max_new_tokens=16, temperature=1.0, nucleus_p=0.5)
    assert (2, 192 + 16) == tuple(generated_ids.shape)
    assert torch.equal(generated_ids[:, :192].cpu(), ids[:2])
    # numbers: the reference's fixture on the same (det_init_ + sharpened gate) weights
    f = load_golden('unit_test_config.npz')
    m, sd = build(cfg)
    images, fids = torch.from_numpy(f['images']).to(dev()), torch.from_numpy(f['ids']).to(dev())
    eng = m._engine
    eng.moe_trace = {}
    with torch.no_grad():
        out = m(images=images, ids=fids, attn_msk=torch.from_numpy(f['attn_msk']).to(dev()))
    eng.moe_trace = None
    for key, got in (('encoder_output', out.encoder_output), ('logits', out.logits), ('hidden_state', out.hidden_state)):
        err = np.abs(got.float().cpu().numpy() - f[key])
        tol = 1.5e-2 * max(1.0, float(np.abs(f[key]).max()))
        REPORT[f'unit_test_config.{key}'] = {'max': float(err.max()), 'q99': float(np.quantile(err, 0.99)), 'tol': tol}
        assert float(np.quantile(err, 0.99)) <= tol, (key, float(np.quantile(err, 0.99)), tol)
        assert float(err.max()) <= 6 * tol, (key, float(err.max()))          # (a routing near-tie moves single tokens further)
    gen = m.generate(images, fids[:, :5], max_new_tokens=6, temperature=1.0, top_k=1).cpu().numpy()
    agree = disagree_clear = 0
    for b in range(gen.shape[0]):
        for t in range(6):
            if not np.array_equal(gen[b, :5 + t], f['greedy_ids'][b, :5 + t]):
                break                                   # a different earlier token: the conditioning differs from here on
            if gen[b, 5 + t] == f['greedy_ids'][b, 5 + t]:
                agree += 1
            elif f['greedy_margins'][b, t] >= 0.05:
                disagree_clear += 1
    REPORT['unit_test_config.greedy'] = {'agree': agree, 'clear_disagreements': disagree_clear}
    assert disagree_clear == 0 and agree >= 3


@pytest.mark.parametrize('extra', [{}, dict(advanced_pos_emb_gate_sizes=(32, 64, 32))])
def test_family_training_memorises_a_small_caption_set(extra):
    """End-to-end sanity of the family's train path over many steps (MoE routing that changes as the gates learn, the packed W2aug
    re-built from the updated parameters every step, sparse blocks with packed ragged rows, dropout, the fused AdamW over an arena
    with pad entries): 8 fixed (image, caption) pairs must be memorised, and the trained model must then write those captions."""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    torch.manual_seed(11)
    cfg = mini_config(dropout=0.05, **extra)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
    det_init_(w.model, seed=3, style='reference')
    w = w.to(dev()).train()
    images, labels = synthetic_batch(8, 32, 16, cfg.decoder_config.vocab_size, seed=21)
    images, labels = images.to(dev()), labels.to(dev())
    opt = FusedAdamW(w.model.parameters(), w.model, lr=2e-3, betas=(0.9, 0.95), weight_decay=0.0)
    losses = []
    for _ in range(300):
        loss, _ = w.train_step(images, labels)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    REPORT[f'mini.memorise.{"advpos" if extra else "plain"}'] = {'first': losses[0], 'last': losses[-1]}
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.2 * losses[0], (losses[0], losses[-1])
    a = w.model._engine.arena
    pads = [n for n in a.entries if n not in a.params]
    assert len(pads) == 4 and all(float(a.P(n).abs().max()) == 0.0 for n in pads), 'the pad entries behind bias-free gates must stay zero'
    w.eval()
    with torch.no_grad():
        vl, _ = w.val_step(images, labels)
        assert float(vl) < 0.25 * losses[0]
        # teacher-forced: the trained model predicts its captions
        ids = torch.cat((torch.full((8, 1), tok.bos_token_id, dtype=torch.long, device=dev()),
                         torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))), dim=1)[:, :16]
        logits = w.model(images=images, ids=ids).logits
        live = labels != -100
        acc = float(((logits.argmax(-1) == labels) & live).sum()) / float(live.sum())
        REPORT[f'mini.memorise.{"advpos" if extra else "plain"}.teacher_forced_accuracy'] = acc
        assert acc > 0.8, acc
        # the KV-cache generation of the TRAINED model == greedy decoding by repeated full forwards (the reference's loop)
        prompt = torch.full((8, 1), tok.bos_token_id, dtype=torch.long, device=dev())
        gen = w.model.generate(images, prompt, max_new_tokens=10, temperature=1.0, top_k=1)
        cur, agree, total = prompt, 0, 0
        for t in range(10):
            lg = w.model(images=images, ids=cur).logits[:, -1].float()
            top2 = lg.topk(2, dim=-1).values
            clear = (top2[:, 0] - top2[:, 1]) > 3e-2 * lg.abs().amax(dim=-1).clamp(min=1.0)      # 3 x the bf16 logit tolerance of the trained model
            same_prefix = (gen[:, :cur.shape[1]] == cur).all(dim=1)
            ok = gen[:, cur.shape[1]] == lg.argmax(-1)
            agree += int((ok & clear & same_prefix).sum())
            total += int((clear & same_prefix).sum())
            cur = torch.cat((cur, lg.argmax(-1, keepdim=True)), dim=1)
    REPORT[f'mini.memorise.{"advpos" if extra else "plain"}.generate_vs_forward'] = {'agree': agree, 'of': total}
    # (a trained MoE model: the cached single-row step and the full-sequence pass run GEMMs of different shapes, so a gate value near a
    # top-k tie can pick another expert in one of them -- a discontinuity no logit-margin filter covers; such a row leaves the
    # comparison at its first differing token.  Three of 67 steps did in one of ~10 full-suite runs of a non-deterministically trained
    # model; every other run agreed on all of them.)
    assert total >= 24 and agree >= 0.9 * total, (agree, total)


def test_family_contrastive_loss_lock_step_normaliser():
    """add_contrastive_loss on the mini family model: the prompt rows and the text rows run as two decoder segments whose sparse
    blocks gather DIFFERENT row subsets, yet every block's gradient normaliser must use the joint norm over both (one tensor in
    the reference).  Loss terms and every gradient against the oracle's full-sequence evaluation (same expert choices checked)."""
    from oracle import reference_model as orc
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    f = load_golden('mini_forward.npz')
    cfg = mini_config()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(add_contrastive_loss=True, training_contrastive_temperature=0.7), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0))
    sd = {k: v.detach().clone() for k, v in w.model.state_dict().items()}
    w = w.to(dev()).train()
    w.pack_rows = False                         # dense (b, t) rows, so that the device's routing can be laid out as the oracle's rows
    images, labels = torch.from_numpy(f['images']), torch.from_numpy(f['labels'])
    eng = w.model._engine
    eng.moe_trace = {}
    loss, metrics = w.train_step(images.to(dev()), labels.to(dev()))
    trace = dict(eng.moe_trace)
    eng.moe_trace = None
    loss.backward()
    osd = {k: (v.clone().requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    # the device's routing, laid out as the oracle's full-sequence rows: per sequence [kept prompt rows | kept text rows]
    B, n_p = images.shape[0], cfg.vision_encoder_config.n_cls
    text, prompt = device_choices(trace, top_k_fn(cfg), 0), device_choices(trace, top_k_fn(cfg), 1)
    forced = {s: c for s, c in text.items() if s.startswith('encoder.')}
    for s in (k for k in text if k.startswith('decoder.')):
        pr, tx = prompt[s].view(B, -1, prompt[s].shape[-1]), text[s].view(B, -1, text[s].shape[-1])
        assert pr.shape[1] == n_p                               # every prompt position is kept by every sparse decoder layer
        forced[s] = torch.cat((pr, tx), dim=1).reshape(-1, pr.shape[-1])
    io = {'forced': forced, 'record': {}}
    ol, olm, oc = orc.lm_step(osd, cfg, images, labels, tok, training=True, contrastive_temperature=0.7, return_parts=True, moe_io=io)
    ol.backward()
    check_choices('mini.contrastive', forced, io['record'])
    REPORT['mini.contrastive'] = {'lm': [float(metrics['train_loss_lm']), float(olm)], 'contrastive': [float(metrics['train_loss_contrastive']), float(oc)]}
    assert abs(float(metrics['train_loss_lm']) - float(olm)) <= 1e-2 * max(1.0, float(olm))
    assert abs(float(metrics['train_loss_contrastive']) - float(oc)) <= 1e-2 * max(1.0, float(oc))
    fails = []
    for name, p in w.model.named_parameters():
        g = osd[name].grad
        try:
            grad_close(f'mini.contrastive.{name}', p.grad, (g if g is not None else torch.zeros_like(osd[name])).numpy(), rel=0.1, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:8])


def test_sparse_position_sets_follow_the_state_dict():
    """The kept / skipped position sets are persistent buffers (layers.py:557-558): loading a checkpoint with other draws into a model
    that has already run must re-plan the sparse layers (row lists, decode slot tables) -- same results as a model that was built
    with those sets from the start"""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    f = load_golden('mini_decode.npz')
    cfg = mini_config()
    m1, sd = build(cfg)
    images = torch.from_numpy(f['images']).to(dev())
    ids = torch.from_numpy(f['ids'][:, :12]).to(dev())
    prompt = torch.from_numpy(f['prompt']).to(dev())
    with torch.no_grad():
        out1 = m1(images=images, ids=ids).logits.clone()
        gen1 = m1.generate(images, prompt, max_new_tokens=8, temperature=1.0, top_k=1)
    sd2 = {k: v.clone() for k, v in sd.items()}
    n_cls = cfg.vision_encoder_config.n_cls
    for tower, first in (('decoder', n_cls), ('encoder', 0)):
        for l in range(2):
            key = f'{tower}.transformer.h.{l}.input_mask_idx'
            size = sd[key].numel() + sd[key.replace('_idx', '_not_idx')].numel()
            gen = np.random.Generator(np.random.PCG64(seed=100 + l))
            full = np.concatenate((np.arange(first), gen.permutation(size - first) + first))
            k = sd[key].numel()
            sd2[key] = torch.from_numpy(np.sort(full[:k])).long()
            sd2[key.replace('_idx', '_not_idx')] = torch.from_numpy(np.sort(full[k:])).long()
    m1.load_state_dict(sd2)
    m2 = VisionEncoderDecoder(cfg)
    m2.load_state_dict(sd2)
    m2 = m2.to(dev()).eval()
    with torch.no_grad():
        a, b = m1(images=images, ids=ids).logits, m2(images=images, ids=ids).logits
        ga = m1.generate(images, prompt, max_new_tokens=8, temperature=1.0, top_k=1)
        gb = m2.generate(images, prompt, max_new_tokens=8, temperature=1.0, top_k=1)
    assert torch.equal(a, b) and torch.equal(ga, gb)
    assert float((a - out1).abs().max()) > 1e-3                   # other positions attended: other numbers


def _mixed_config(kind):
    """towers of different kinds in one model (each tower picks its own block path)"""
    from image2text_amd.synth import tiny_config
    fam, dense = mini_config(dropout=0.1), tiny_config(dropout=0.1, dec_d=256, dec_heads=4, enc_d=256, enc_heads=4, block_size=40)
    if kind == 'family_encoder_dense_decoder':
        return fam.model_copy(update=dict(decoder_config=dense.decoder_config))
    if kind == 'dense_encoder_family_decoder':
        return fam.model_copy(update=dict(vision_encoder_config=dense.vision_encoder_config))
    if kind == 'bridge':                                  # encoder width != decoder width: nn.Sequential(encoder, Linear)
        return mini_config(dropout=0.1).model_copy(update=dict(vision_encoder_config=mini_config(dropout=0.1, d=128, heads=1).vision_encoder_config))
    return fam


@pytest.mark.parametrize('kind', ['family', 'family_encoder_dense_decoder', 'dense_encoder_family_decoder', 'bridge'])
def test_every_trainer_option_on_every_tower_mix(kind):
    """Plumbing across the combinations nothing else exercises together: momentum distillation + MLM corruption + contrastive loss +
    dropout + packed ragged rows, on models whose towers mix the dense and the family block paths (and a bridged encoder): three
    optimizer steps with finite losses and gradients, an EMA'd twin, a validation step and a generation call."""
    from types import SimpleNamespace
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    torch.manual_seed(5)
    cfg = _mixed_config(kind)
    V = cfg.decoder_config.vocab_size
    tok = SimpleNamespace(eos_token_id=V - 1, bos_token_id=V - 1, mask_token_id=V - 2, vocab_size=V)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(moco_momentum=0.9, moco_alpha=0.4, mask_fraction=0.2, random_mask_fraction=0.3,
                                                           add_contrastive_loss=True, training_contrastive_temperature=0.7,
                                                           weight_fn='inverse_sqrt_position', eos_token_weight=2.0), ignore_index=-100)
    sharpen_gates_(det_init_(w.model, seed=0, style='reference'))
    w.copy_momentum_params()
    w = w.to(dev()).train()
    images, labels = synthetic_batch(6, 32, 24, V, seed=9)
    images, labels = images.to(dev()), labels.to(dev())
    opt = FusedAdamW(w.model.parameters(), w.model, lr=1e-3, betas=(0.9, 0.95), weight_decay=0.01)
    for step in range(3):
        loss, metrics = w.train_step(images, labels)
        loss.backward()
        assert torch.isfinite(loss) and torch.isfinite(metrics['train_loss_contrastive'])
        for n, p in w.model.named_parameters():
            assert p.grad is not None and torch.isfinite(p.grad).all(), (kind, step, n)
        opt.step()
        opt.zero_grad()
    for (n, p), (_, pm) in zip(w.model.named_parameters(), w.model_m.named_parameters()):
        assert torch.isfinite(pm).all() and (n.endswith('input_mask_idx') or pm.shape == p.shape)
    w.eval()
    with torch.no_grad():
        vloss, vm = w.val_step(images, labels)
        gen = w.model.generate(images, torch.full((6, 1), V - 1, dtype=torch.long, device=dev()), max_new_tokens=6, temperature=0.8, top_k=20)
    assert torch.isfinite(vloss) and 'val_loss_contrastive' in vm and gen.shape == (6, 7)


@pytest.mark.parametrize('B', [1, 3])
def test_family_edge_case_captions(B):
    """ragged extremes through the sparse / packed planning: a caption with no label at all, a one-token caption, a full-length
    one (and a batch of one): packed == dense, finite gradients, zero-weight rows contribute nothing"""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    cfg = mini_config()
    V = cfg.decoder_config.vocab_size
    tok = fake_tokenizer(V)
    images, labels = synthetic_batch(3, 32, 24, V, seed=13)
    labels[0, :] = -100                                  # nothing to predict
    labels[1, 0] = V - 1
    labels[1, 1:] = -100                                 # EOS only
    labels[2, :] = torch.randint(0, V - 1, (24,), generator=torch.Generator().manual_seed(1))       # no EOS, every position live
    if B == 1:
        images, labels = images[1:2], labels[1:2]
    res = []
    for packed in (True, False):
        w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100)
        sharpen_gates_(det_init_(w.model, seed=0))
        w = w.to(dev()).train()
        w.pack_rows = packed
        loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
        loss.backward()
        assert torch.isfinite(loss)
        res.append((float(loss.detach()), {n: p.grad.detach().float().cpu().numpy() for n, p in w.model.named_parameters()}))
    assert abs(res[0][0] - res[1][0]) <= 2e-3 * max(1.0, abs(res[1][0]))
    for n in res[0][1]:
        assert np.isfinite(res[0][1][n]).all(), n
        grad_close(f'edge.{B}.{n}', torch.from_numpy(res[0][1][n]), res[1][1][n], rel=6e-2, cos=0.99)
