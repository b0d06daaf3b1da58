"""Model-level parity on the MI355X: the HIP hot path behind VisionEncoderDecoder / ModelTrainerWrapper against the
golden fixtures the reference produced (tests/golden, see tools/gen_goldens.py) and against the CPU oracle.

Tolerances (bf16 operands, fp32 accumulation, fp32 residual stream):
  * logits: max |err| <= 1e-2 * max(1, max|logit|): 1e-2 absolute at the reference's init-scale logits (north star),
    the same relative precision for the briefly trained tiny model whose logits reach +-27;
  * loss: |err| <= 1e-2 * max(1, loss);  gradients (untrained, well-conditioned weights): relative L2 error <= 6e-2
    and cosine >= 0.995 per parameter; trained tiny model (loss 2e-3, gradient norms ~1e-5 where the normaliser's
    1e-6 epsilon and bf16 noise dominate): <= 0.15 / >= 0.985;
  * greedy token ids: exact wherever the oracle's top-1 margin exceeds MARGIN_EPS, and the run is re-synchronised on
    the golden prefix after a low-margin step so that every step is checked.
"""
import json
import os

import numpy as np
import pytest
import torch

from image2text_amd.lib import I2TError
from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch, tiny_config

pytestmark = pytest.mark.gpu
REPORT = {}


def dev():
    return torch.device('cuda:0')


def logits_tol(ref):
    return 1e-2 * max(1.0, float(np.abs(ref).max()))


def hidden_tol(ref):
    return 1.5e-2 * max(1.0, float(np.abs(ref).max()))


def maxerr(name, got, ref, tol):
    got = got.detach().float().cpu().numpy() if isinstance(got, torch.Tensor) else got
    err = float(np.abs(got - ref).max())
    REPORT[name] = {'max_abs_err': err, 'tol': tol, 'ref_absmax': float(np.abs(ref).max())}
    assert np.isfinite(got).all(), f'{name}: non-finite'
    assert err <= tol, f'{name}: max abs err {err:.4g} > {tol:.4g} (ref absmax {np.abs(ref).max():.4g})'


def grad_close(name, got, ref, rel=6e-2, cos=0.995):
    got = got.detach().float().cpu().numpy().ravel().astype(np.float64)
    ref = ref.ravel().astype(np.float64)
    nr = np.linalg.norm(ref)
    if nr < 1e-12:
        assert np.linalg.norm(got) < 1e-6, f'{name}: reference gradient is zero'
        return
    r = np.linalg.norm(got - ref) / nr
    c = float(got @ ref / (np.linalg.norm(got) * nr + 1e-30))
    REPORT[f'grad.{name}'] = {'rel_l2': float(r), 'cos': c}
    assert r <= rel and c >= cos, f'{name}: rel L2 err {r:.4g}, cosine {c:.6f}'


@pytest.fixture(scope='module')
def tiny_model(tiny_weights):
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    m = VisionEncoderDecoder(tiny_config())
    m.load_state_dict(tiny_weights)
    return m.to(dev()).eval()


@pytest.fixture(scope='module', autouse=True)
def write_report():
    yield
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report.json', 'w') as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


@pytest.mark.parametrize('tag', ['nomask', 'row_mask', 'sl_mask', 'bsl_mask'])
def test_tiny_forward(tiny_model, tiny_forward, tag):
    f = tiny_forward
    msk = None if tag == 'nomask' else torch.from_numpy(f[tag]).to(dev())
    with torch.no_grad():
        out = tiny_model(images=torch.from_numpy(f['images']).to(dev()), ids=torch.from_numpy(f['ids']).to(dev()), attn_msk=msk)
    assert tuple(out.logits.shape) == f[f'{tag}.logits'].shape and tuple(out.hidden_state.shape) == f[f'{tag}.hidden_state'].shape
    maxerr(f'tiny.{tag}.encoder_output', out.encoder_output, f[f'{tag}.encoder_output'], 2e-2)
    maxerr(f'tiny.{tag}.logits', out.logits, f[f'{tag}.logits'], logits_tol(f[f'{tag}.logits']))
    maxerr(f'tiny.{tag}.hidden_state', out.hidden_state, f[f'{tag}.hidden_state'], hidden_tol(f[f'{tag}.hidden_state']))
    assert (out.logits.argmax(-1).cpu().numpy() == f[f'{tag}.logits'].argmax(-1)).mean() > 0.97


@pytest.mark.parametrize('tag,kw', [('cross_only', dict(use_soft_prompting=False)), ('prompt_only', dict(use_cross_attn=False))])
def test_tiny_forward_modes(tiny_weights, tiny_forward, tag, kw):
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    m = VisionEncoderDecoder(tiny_config(**kw))
    m.load_state_dict(tiny_weights)
    m = m.to(dev()).eval()
    f = tiny_forward
    with torch.no_grad():
        out = m(images=torch.from_numpy(f['images']).to(dev()), ids=torch.from_numpy(f['ids']).to(dev()),
                attn_msk=torch.from_numpy(f['row_mask']).to(dev()))
    maxerr(f'tiny.{tag}.logits', out.logits, f[f'{tag}.logits'], logits_tol(f[f'{tag}.logits']))
    maxerr(f'tiny.{tag}.hidden_state', out.hidden_state, f[f'{tag}.hidden_state'], hidden_tol(f[f'{tag}.hidden_state']))


def test_bad_mask_shape_raises(tiny_model, tiny_forward):
    f = tiny_forward
    with pytest.raises(RuntimeError):
        tiny_model(images=torch.from_numpy(f['images']).to(dev()), ids=torch.from_numpy(f['ids']).to(dev()),
                   attn_msk=torch.ones(3, 7, dtype=torch.bool, device=dev()))


def _wrapper(cfg, weights=None):
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    w = ModelTrainerWrapper(cfg, fake_tokenizer(cfg.decoder_config.vocab_size), TrainerWrapperConfig(), ignore_index=-100)
    if weights is not None:
        w.model.load_state_dict(weights)
    return w.to(dev())


def test_tiny_train_step_loss_and_every_gradient(tiny_weights, tiny_forward, tiny_train):
    w = _wrapper(tiny_config(), tiny_weights).train()
    f = tiny_forward
    images, labels = torch.from_numpy(f['images']).to(dev()), torch.from_numpy(f['labels']).to(dev())
    loss, metrics = w.train_step(images, labels)
    loss.backward()
    ref = float(tiny_train['loss'])
    REPORT['tiny.train_loss'] = {'got': float(loss.detach()), 'ref': ref}
    assert abs(float(loss.detach()) - ref) <= 1e-2 * max(1.0, ref)
    assert 'train_loss_lm' in metrics
    check_all_grads('tiny', w.model, tiny_train, rel=0.15, cos=0.985)
    with torch.no_grad():
        vloss, _ = w.eval().val_step(images, labels)
    assert abs(float(vloss) - float(tiny_train['val_loss'])) <= 1e-2 * max(1.0, float(tiny_train['val_loss']))


def check_all_grads(tag, model, golden, rel, cos):
    fails, n = [], 0
    for name, p in model.named_parameters():
        key = f'grad.{name}' if f'grad.{name}' in golden else 'grad.decoder.lm_head.weight'
        assert p.grad is not None, name
        try:
            grad_close(f'{tag}.{name}', p.grad, golden[key], rel=rel, cos=cos)
        except AssertionError as e:
            fails.append(str(e))
        n += 1
    assert n >= 55
    assert not fails, f'{len(fails)} of {n} gradients out of tolerance: ' + '; '.join(fails[:6])


def test_tiny_untrained_train_step_every_gradient():
    """Well-conditioned gradient parity: det_init_ weights (regenerated here, bit-identical to the golden run)."""
    from conftest import load_golden
    g = load_golden('tiny_train_init.npz')
    w = _wrapper(tiny_config())
    det_init_(w.model, seed=0)
    w.train()
    loss, _ = w.train_step(torch.from_numpy(g['images']).to(dev()), torch.from_numpy(g['labels']).to(dev()))
    loss.backward()
    REPORT['tiny_init.train_loss'] = {'got': float(loss.detach()), 'ref': float(g['loss'])}
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-2 * float(g['loss'])
    check_all_grads('tiny_init', w.model, g, rel=6e-2, cos=0.995)


def test_gradient_accumulation_and_scaled_backward(tiny_forward):
    """Two backward calls without zero_grad accumulate; loss/2 backward halves (accelerate's grad-accum contract)."""
    w = _wrapper(tiny_config())
    det_init_(w.model, seed=0)
    w.train()
    f = tiny_forward
    images, labels = torch.from_numpy(f['images']).to(dev()), torch.from_numpy(f['labels']).to(dev())
    loss, _ = w.train_step(images, labels)
    loss.backward()
    g1 = {n: p.grad.clone() for n, p in w.model.named_parameters()}
    loss, _ = w.train_step(images, labels)
    (loss / 2).backward()
    for n, p in w.model.named_parameters():
        # the gradient normaliser at every block output makes everything below ln_f invariant to the loss scale:
        # only ln_f (and the tied head, which mixes both contributions and is skipped here) sees the factor 1/2
        if 'wte' in n:
            continue
        ref = g1[n] * (1.5 if n.startswith('decoder.transformer.ln_f') else 2.0)
        err = (p.grad - ref).norm() / (ref.norm() + 1e-12)
        assert float(err) < 3e-2, (n, float(err))


def test_fused_adamw_matches_torch_adamw(tiny_weights, tiny_forward):
    """Same gradients into FusedAdamW (one HIP launch over the arena) and torch.optim.AdamW on cloned parameters."""
    from image2text_amd.training.optim import FusedAdamW
    f = tiny_forward
    images, labels = torch.from_numpy(f['images']).to(dev()), torch.from_numpy(f['labels']).to(dev())
    wa = _wrapper(tiny_config(), tiny_weights).train()
    named = list(wa.model.named_parameters())
    groups = [{'params': [p for n, p in named if 'cross_attn' in n], 'lr': 3e-3, 'weight_decay': 0.0},
              {'params': [p for n, p in named if 'cross_attn' not in n], 'lr': 1e-3, 'weight_decay': 0.1}]
    oa = FusedAdamW(groups, wa.model, betas=(0.9, 0.95))
    clones = [p.detach().clone().requires_grad_(True) for _, p in named]
    by = {id(p): c for (_, p), c in zip(named, clones)}
    ob = torch.optim.AdamW([{'params': [by[id(p)] for p in g['params']], 'lr': g['lr'], 'weight_decay': g['weight_decay']}
                            for g in groups], betas=(0.9, 0.95))
    for _ in range(3):
        loss, _ = wa.train_step(images, labels)
        loss.backward()
        for (_, p), c in zip(named, clones):
            c.grad = p.grad.clone()
        oa.step()
        ob.step()
        oa.zero_grad()
    for (n, p), c in zip(named, clones):
        assert float((p - c).abs().max()) <= 1e-5 * max(1.0, float(c.abs().max())), n
    # the bf16 shadow the kernels read must follow the update
    eng = wa.model._engine
    assert float((eng.arena.pbf.float() - eng.arena.p32).abs().max()) <= float(eng.arena.p32.abs().max()) / 128


@pytest.mark.parametrize('large_tiles', [False, True])
def test_nano224_full_size_forward_and_loss(nano224_golden, large_tiles, monkeypatch):
    """large_tiles: route every eligible GEMM through the persistent 256 x 256 kernel (at batch 2 the default heuristic
    keeps the 128 x 128 one), so that the kernel the full-size bench runs on is held to the same golden."""
    if large_tiles:
        monkeypatch.setenv('I2T_G256_MIN_TILES', '1')
    g = nano224_golden
    cfg = nano224_config()
    w = _wrapper(cfg)
    det_init_(w.model, seed=0)
    w.eval()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    assert np.array_equal(labels.numpy(), g['labels'])
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    ids = torch.cat((torch.full((2, 1), tok.bos_token_id), ids), dim=1)[:, :64]
    with torch.no_grad():
        out = w.model(images=images.to(dev()), ids=ids.to(dev()))
        vloss, _ = w.val_step(images.to(dev()), labels.to(dev()))
    # 'stress' weights (every residual branch O(1), 12 layers of bf16 rounding): measured max 1.6-2e-2, rms 4e-3
    maxerr('nano224.encoder_output', out.encoder_output, g['encoder_output'], 3e-2)
    maxerr('nano224.logits_head', out.logits[:, :, :256], g['logits_head'], 2.5e-2)
    maxerr('nano224.logits_tail', out.logits[:, :, -64:], g['logits_tail'], 2.5e-2)
    rms = float((out.logits[:, :, :256].float().cpu() - torch.from_numpy(g['logits_head'])).pow(2).mean().sqrt())
    REPORT['nano224.logits_head_rms_err'] = {'got': rms}
    assert rms <= 5e-3
    maxerr('nano224.logits_lse', torch.logsumexp(out.logits.float(), -1), g['logits_lse'], 1e-2)
    maxerr('nano224.hidden_text', out.hidden_state[:, 64:], g['hidden_text'], hidden_tol(g['hidden_text']))
    REPORT['nano224.val_loss'] = {'got': float(vloss), 'ref': float(g['val_loss'])}
    assert abs(float(vloss) - float(g['val_loss'])) <= 1e-2 * float(g['val_loss'])
    # train step: loss + the stored gradients + every gradient norm
    w.train()
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    assert abs(float(loss) - float(g['train_loss'])) <= 1e-2 * float(g['train_loss'])
    named = dict(w.model.named_parameters())
    for k in g:
        if k.startswith('grad.'):
            grad_close(f'nano224.{k[5:]}', named[k[5:]].grad, g[k], rel=8e-2, cos=0.99)
    bad = []
    for n, p in named.items():
        ref = float(g[f'gradnorm.{n}'])
        got = float(p.grad.norm())
        REPORT[f'gradnorm.{n}'] = {'got': got, 'ref': ref}
        if abs(got - ref) > 6e-2 * ref + 1e-7:
            bad.append((n, got, ref))
    assert not bad, bad[:8]


def test_nano224_reference_init_logits_bf16_floor():
    """North-star criterion "logits within 1e-2 bf16" at the reference's own initial distributions (6.4 M logits, |logit| <= 2.5).

    What is asserted, strongest first:
      * REGRESSION PIN near the measured values: max |err| <= 1.9e-2 (measured 1.5-1.7e-2 across boxes), rms <= 4e-3
        (measured 3.4-3.6e-3), >= 98.5 % of the logits within 1e-2 absolute (measured 99.4 %);
      * the literal allclose form |err| <= 1e-2 + 1e-2 |ref| may fail for at most 0.2 % of the logits (measured 0.10 %):
        the max over millions of O(1) logits of a bf16-operand pipeline is NOT within 1e-2 -- see the attribution table in
        DESIGN.md section 2 (tools/diag_precision.py): the error is spread over conv stack (0.41 % rel-rms), encoder blocks,
        decoder blocks (0.39 %) and lm_head (0.24 %), adding in quadrature to 0.63 % of the logit rms = 1.7e-2 at the 5-sigma
        tail; no single stage can be fixed to reach 1e-2 (a hi+lo split of the lm_head operand moves the max by 3 %);
      * calibration: the REFERENCE ITSELF under bf16 autocast (how trainer.py runs precision 'bf16') deviates from its own fp32
        run by max 2.09e-2 / rms 4.1e-3 / 98.5 % within 1e-2 on the same inputs (stored in the fixture): this path must be at
        least that close.
    """
    from conftest import load_golden
    g = load_golden('nano224_refinit.npz')
    cfg = nano224_config()
    w = _wrapper(cfg)
    det_init_(w.model, seed=0, style='reference')
    w.eval()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    ids = torch.cat((torch.full((2, 1), tok.bos_token_id), ids), dim=1)[:, :64]
    with torch.no_grad():
        out = w.model(images=images.to(dev()), ids=ids.to(dev()))
        vloss, _ = w.val_step(images.to(dev()), labels.to(dev()))
    maxerr('nano224_refinit.encoder_output', out.encoder_output, g['encoder_output'], 2.2e-2)      # measured 1.7e-2 of absmax 2.6
    ref_max, ref_rms, ref_frac = (float(v) for v in g['reference_bf16_autocast_dev'])
    for key, sl in (('logits_head', slice(0, 256)), ('logits_tail', slice(-64, None))):
        got = out.logits[:, :, sl].float().cpu().numpy()
        err = np.abs(got - g[key])
        viol = float((err > 1e-2 + 1e-2 * np.abs(g[key])).mean())
        rms = float(np.sqrt((err ** 2).mean()))
        REPORT[f'nano224_refinit.{key}'] = {'max_abs_err': float(err.max()), 'rms_err': rms, 'frac_within_1e-2': float((err <= 1e-2).mean()),
                                             'frac_violating_allclose_1e-2_1e-2': viol, 'ref_absmax': float(np.abs(g[key]).max())}
        assert err.max() <= 1.9e-2, f'{key}: max abs err {err.max():.4g} (regression pin 1.9e-2)'
        assert rms <= 4e-3, f'{key}: rms {rms:.4g} (regression pin 4e-3)'
        assert (err <= 1e-2).mean() >= 0.985, f'{key}: only {(err <= 1e-2).mean():.4f} within 1e-2'
        assert viol <= 2e-3, f'{key}: {viol:.2e} of the logits violate allclose(atol=1e-2, rtol=1e-2)'
        assert err.max() <= ref_max and rms <= ref_rms and (err <= 1e-2).mean() >= ref_frac, 'worse than the reference under bf16 autocast'
    maxerr('nano224_refinit.logits_lse', torch.logsumexp(out.logits.float(), -1), g['logits_lse'], 1e-2)
    assert abs(float(vloss) - float(g['val_loss'])) <= 1e-3 * float(g['val_loss'])


def test_nano224_reference_init_logits_within_1e2_in_precise_mode(monkeypatch):
    """The literal north-star tolerance, met once: under I2T_PRECISE=1 (image2text_amd/ops.py: every GEMM operand produced from fp32
    data carries a second bf16 term, every GEMM runs hi.hi + lo.hi + hi.lo through the same MFMA kernels and its accumulate class;
    attention and the conv stack keep their bf16 operands) the nano-224 logits at the reference's initial distributions satisfy
    allclose(atol=1e-2, rtol=1e-2) against the fp32 reference fixture -- all of them.  So the 1.5-1.7e-2 of the shipped one-term
    pipeline (test above) is operand rounding and nothing else.  The mode is inference-only and never runs in bench.py."""
    from conftest import load_golden
    from image2text_amd import ops
    g = load_golden('nano224_refinit.npz')
    monkeypatch.setenv('I2T_PRECISE', '1')
    cfg = nano224_config()
    w = _wrapper(cfg)
    det_init_(w.model, seed=0, style='reference')
    w.eval()
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    ids = torch.cat((torch.full((2, 1), tok.bos_token_id), ids), dim=1)[:, :64]
    before = dict(ops.PRECISE_CALLS)
    with torch.no_grad():
        out = w.model(images=images.to(dev()), ids=ids.to(dev()))
    calls = {k: ops.PRECISE_CALLS[k] - before[k] for k in before}
    REPORT['nano224_refinit.precise.calls'] = calls
    L_e, L_d = cfg.vision_encoder_config.n_layer, cfg.decoder_config.n_layer
    assert calls['gemm'] >= 4 * L_e + 6 * L_d + 3 and calls['products'] >= 3 * calls['gemm'] - 2 * (L_e + 2 * L_d) - 1, calls
    maxerr('nano224_refinit.precise.encoder_output', out.encoder_output, g['encoder_output'], 1e-2)
    for key, sl in (('logits_head', slice(0, 256)), ('logits_tail', slice(-64, None))):
        got = out.logits[:, :, sl].float().cpu().numpy()
        err = np.abs(got - g[key])
        REPORT[f'nano224_refinit.precise.{key}'] = {'max_abs_err': float(err.max()), 'rms_err': float(np.sqrt((err ** 2).mean())),
                                                     'frac_violating_allclose_1e-2_1e-2': float((err > 1e-2 + 1e-2 * np.abs(g[key])).mean())}
        assert np.allclose(got, g[key], atol=1e-2, rtol=1e-2), f'{key}: max abs err {err.max():.4g}'
    with pytest.raises(I2TError, match='I2T_PRECISE'):          # inference only
        w.train()
        w.train_step(images.to(dev()), labels.to(dev()))


# ------------------------------------------------------------------------------------------------------ greedy decode
def assert_greedy_matches(model, images, gold_ids, gold_margins, P, eps, use_graph=True):
    """Token-exact wherever the oracle's top-1 margin >= eps; after a low-margin step the run is restarted from the
    golden prefix so that every later step is still checked.  Returns the number of low-margin restarts."""
    from image2text_amd.decoding import GreedyDecoder
    dec = GreedyDecoder(model)
    total = gold_ids.shape[1]
    start, restarts = P, 0
    while start < total:
        out = dec.generate(images, torch.from_numpy(gold_ids[:, :start]).to(dev()), total - start, use_graph=use_graph)
        out = out.cpu().numpy()
        assert np.array_equal(out[:, :start], gold_ids[:, :start])
        mism = out[:, start:] != gold_ids[:, start:]
        if not mism.any():
            break
        first = int(mism.any(0).argmax())
        col = start + first
        for b in np.nonzero(mism[:, first])[0]:
            assert gold_margins[b, col - P] < eps, (f'row {b} step {col - P}: token {out[b, col]} != {gold_ids[b, col]} although '
                                                    f'the oracle margin is {gold_margins[b, col - P]:.4f} >= {eps}')
        restarts += 1
        start = col + 1
    return restarts


def test_tiny_greedy_tokens(tiny_model, tiny_decode):
    d = tiny_decode
    images = torch.from_numpy(d['images']).to(dev())
    # trained tiny model: logits reach +-27, bf16 error up to ~0.2 there -> steps with margin < 0.5 may flip
    r = assert_greedy_matches(tiny_model, images, d['ids'], d['margins'], 1, eps=0.5)
    r3 = assert_greedy_matches(tiny_model, images, d['ids3'], d['margins3'], 3, eps=0.5)
    REPORT['tiny.greedy'] = {'low_margin_restarts': r, 'low_margin_restarts_prompt3': r3,
                             'steps_with_margin_ge_eps': int((d['margins'] >= 0.5).sum()), 'steps': int(d['margins'].size)}
    # public API + hipGraph replay == eager step sequence
    from image2text_amd.decoding import GreedyDecoder
    prompt = torch.from_numpy(d['prompt']).to(dev())
    a = tiny_model.generate(images, prompt, max_new_tokens=24, temperature=1.0, top_k=1)
    b = GreedyDecoder(tiny_model).generate(images, prompt, 24, use_graph=False)
    assert a.shape == (4, 25) and torch.equal(a, b)
    a2 = tiny_model.generate(images, prompt, max_new_tokens=24, temperature=1.0, top_k=1)     # replay of the cached graph
    assert torch.equal(a, a2)


def test_greedy_with_sliced_encoder_pass(tiny_model, tiny_decode, monkeypatch):
    """generate() runs the encoder in slices of ENC_CHUNK images (every image is independent there): same tokens and margins."""
    from image2text_amd import decoding
    images = torch.from_numpy(tiny_decode['images']).to(dev())
    prompt = torch.from_numpy(tiny_decode['prompt']).to(dev())
    assert images.shape[0] == 4
    whole, m0 = decoding.GreedyDecoder(tiny_model).generate(images, prompt, 24, return_margins=True, use_graph=False)
    monkeypatch.setattr(decoding, 'ENC_CHUNK', 3)                      # slices of 3 + 1 images
    sliced, m1 = decoding.GreedyDecoder(tiny_model).generate(images, prompt, 24, return_margins=True, use_graph=False)
    assert torch.equal(whole, sliced)
    assert torch.equal(m0, m1)


def test_concurrent_decoder_equals_sequential():
    """ConcurrentGreedyDecoder (one stream + one graph per lane, one shared HotPath) must emit exactly the tokens the
    sequential decoder emits for the same batches: the lanes share the weights (read-only) and nothing else -- in particular
    not the encoder's conv weight workspace, which each lane's conv launcher rewrites per layer (a cross-stream race when it
    was engine-wide).  nano-224 at a batch big enough that the lanes really overlap on the chip; repeated to catch a race."""
    from image2text_amd.decoding import ConcurrentGreedyDecoder, GreedyDecoder
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    cfg = nano224_config()
    V = cfg.decoder_config.vocab_size
    m = det_init_(VisionEncoderDecoder(cfg), seed=0).to(dev()).eval()
    lanes, B, N = 3, 96, 12
    imgs = [synthetic_batch(B, 224, 64, V, seed=40 + i)[0].to(dev()) for i in range(lanes)]
    prompts = [torch.full((B, 1), V - 1, dtype=torch.long, device=dev()) for _ in range(lanes)]
    seq = GreedyDecoder(m)
    want = [seq.generate(imgs[i], prompts[i], N).clone() for i in range(lanes)]
    assert not torch.equal(want[0], want[1])                   # different images -> different captions
    con = ConcurrentGreedyDecoder(m, lanes)
    for rep in range(4):
        got = con.generate(imgs, prompts, N)
        torch.cuda.synchronize()
        for i in range(lanes):
            assert torch.equal(got[i], want[i]), f'rep {rep} lane {i}: {(got[i] != want[i]).sum().item()} tokens differ'
    # the weights change under the decoders (an optimizer step on the torch side): the shadow refresh happens once, on the
    # parent stream, and every lane sees the new weights
    with torch.no_grad():
        m.decoder.transformer.wte.weight.mul_(1.5)
    want2 = [seq.generate(imgs[i], prompts[i], N).clone() for i in range(lanes)]
    with torch.no_grad():
        m.decoder.transformer.wte.weight.add_(0.0)                          # version bump only: forces a shadow re-cast
    got2 = con.generate(imgs, prompts, N)
    for i in range(lanes):
        assert torch.equal(got2[i], want2[i])


def test_sampling_modes_shapes(tiny_model, tiny_decode):
    """The reference unit test's contract (models/vision_encoder_decoder_test.py:88-92): shapes of sampled ids."""
    d = tiny_decode
    images = torch.from_numpy(d['images']).to(dev())[:2]
    ids = torch.from_numpy(d['ids3'][:2, :3]).to(dev())
    out = tiny_model.generate(images, ids, max_new_tokens=8, temperature=1.0, nucleus_p=0.5)
    assert tuple(out.shape) == (2, 11) and torch.equal(out[:, :3], ids)
    out = tiny_model.generate(images, ids, max_new_tokens=4, temperature=0.7, top_k=5)
    assert tuple(out.shape) == (2, 7)


def test_nano224_greedy_tokens(nano224_golden):
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    g = nano224_golden
    cfg = nano224_config()
    m = det_init_(VisionEncoderDecoder(cfg), seed=0).to(dev()).eval()
    images, _ = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    r = assert_greedy_matches(m, images.to(dev()), g['greedy_ids'], g['greedy_margins'], 1, eps=0.05)
    REPORT['nano224.greedy'] = {'low_margin_restarts': r, 'steps_with_margin_ge_eps': int((g['greedy_margins'] >= 0.05).sum()),
                                'steps': int(g['greedy_margins'].size)}
    # full-size property: 64 new tokens for 8 captions, n-gram constraint holds on the output, replay is deterministic
    images8, _ = synthetic_batch(8, 224, 64, cfg.decoder_config.vocab_size, seed=2)
    prompt = torch.full((8, 1), cfg.decoder_config.vocab_size - 1, dtype=torch.long, device=dev())
    out = m.generate(images8.to(dev()), prompt, max_new_tokens=64, top_k=1)
    out2 = m.generate(images8.to(dev()), prompt, max_new_tokens=64, top_k=1)
    assert tuple(out.shape) == (8, 65) and torch.equal(out, out2)
    for row in out.cpu().tolist():
        for n in (2, 3, 4, 5):
            grams = [tuple(row[i:i + n]) for i in range(len(row) - n + 1)]
            assert len(grams) == len(set(grams)), f'repeated {n}-gram in greedy output'


def test_train_step_with_dropout_matches_oracle_on_the_same_masks():
    """Dropout 0.1 / attn_dropout 0.1 in both towers: the oracle is fed the exact masks of the HIP step (rebuilt on the host
    from the step's DropPlans), so loss and every gradient must agree as in the dropout-free case; a second step must draw
    different masks; eval mode must be dropout-free."""
    from oracle import reference_model as orc
    torch.manual_seed(20240917)  # the step seed (hence every mask) derives from torch's seed: without this the masks -- and how
    #                              close to its tolerances the test lands -- changed from process to process (1 failure in ~10 runs)
    cfg = tiny_config(dropout=0.1)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    w = _wrapper(cfg)
    w.pack_rows = False          # the host-side mask replica indexes rows densely (b, t); packing only renumbers the draws
    det_init_(w.model, seed=0)
    sd = {k: v.detach().cpu().clone() for k, v in w.model.state_dict().items()}
    w.train()
    images, labels = synthetic_batch(4, 32, 16, cfg.decoder_config.vocab_size, seed=9)
    loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
    loss.backward()
    eng = w.model._engine
    plans = (eng.enc_drop, eng.dec_drop)
    assert plans[0] is not None and plans[1] is not None
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    oloss = orc.lm_step_text_segment(osd, cfg, images, labels, tok, plans)
    oloss.backward()
    REPORT['tiny_dropout.train_loss'] = {'got': float(loss.detach()), 'ref': float(oloss)}
    assert abs(float(loss.detach()) - float(oloss)) <= 1e-2 * float(oloss)
    fails = []
    for name, p in w.model.named_parameters():
        try:
            grad_close(f'tiny_dropout.{name}', p.grad, osd[name].grad.numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    # the dropout-free loss differs (masks really were applied) and a new step draws new masks
    with torch.no_grad():
        clean = orc.lm_step_text_segment(sd, cfg, images, labels, tok)
    assert abs(float(clean) - float(oloss)) > 1e-6          # not the same number (how far apart depends on the step seed: 3e-5 .. 1e-2 observed)
    w.zero_grad()
    loss2, _ = w.train_step(images.to(dev()), labels.to(dev()))
    assert abs(float(loss2.detach()) - float(loss.detach())) > 1e-4
    w.eval()
    with torch.no_grad():
        v1, _ = w.val_step(images.to(dev()), labels.to(dev()))
        v2, _ = w.val_step(images.to(dev()), labels.to(dev()))
    assert abs(float(v1) - float(v2)) < 1e-5 and abs(float(v1) - float(clean)) <= 1e-2 * float(clean)   # (atomic loss sum: last-bit jitter)


def test_training_memorises_a_small_caption_set():
    """End-to-end sanity of the whole train path on the GPU (forward, hand-written backward, dropout, gradient normaliser,
    fused AdamW, bf16 shadow refresh): 8 fixed (image, caption) pairs must be memorised -- the loss has to fall far below
    both its initial value and the unigram entropy, which only happens if the image-conditioned gradients are right."""
    from image2text_amd.training.optim import FusedAdamW
    torch.manual_seed(7)         # reproducible dropout masks (the step seeds derive from torch's seed)
    cfg = tiny_config(dropout=0.05)
    w = _wrapper(cfg).train()
    det_init_(w.model, seed=3, style='reference')
    images, labels = synthetic_batch(8, 32, 16, cfg.decoder_config.vocab_size, seed=21)
    images, labels = images.to(dev()), labels.to(dev())
    opt = FusedAdamW(w.model.parameters(), w.model, lr=3e-3, betas=(0.9, 0.95), weight_decay=0.0)
    losses = []
    for _ in range(120):
        loss, _ = w.train_step(images, labels)
        loss.backward()
        opt.step()
        opt.zero_grad()
        losses.append(float(loss.detach()))
    REPORT['memorise.loss_first_last'] = {'first': losses[0], 'last': losses[-1]}
    assert all(np.isfinite(losses))
    assert losses[-1] < 0.25 * losses[0], (losses[0], losses[-1])
    w.eval()
    with torch.no_grad():
        vl, _ = w.val_step(images, labels)
    assert float(vl) < 0.3 * losses[0]


def test_row_packing_is_result_preserving():
    """Skipping the caption rows past the last label (packed variable-length decoder pass) must not change the loss or any
    gradient: ragged captions, including one with no label at all and one that fills the window."""
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    images, labels = synthetic_batch(6, 32, 16, V, seed=21)
    labels[1, :] = -100                                   # nothing to predict
    labels[2, :] = torch.randint(0, V - 1, (16,))         # full window, no EOS
    labels[3, 5] = -100                                   # a hole inside a caption: rows before the LAST label stay live
    res = {}
    for pack in (True, False):
        w = _wrapper(cfg)
        det_init_(w.model, seed=0)
        w.pack_rows = pack
        w.train()
        loss, _ = w.train_step(images.to(dev()), labels.to(dev()))
        loss.backward()
        res[pack] = (float(loss.detach()), {n: p.grad.detach().float().cpu().numpy().copy() for n, p in w.model.named_parameters()})
    assert abs(res[True][0] - res[False][0]) <= 2e-4 * abs(res[False][0])
    for n, g in res[False][1].items():
        grad_close(f'pack.{n}', torch.from_numpy(res[True][1][n]), g, rel=3e-2, cos=0.999)
