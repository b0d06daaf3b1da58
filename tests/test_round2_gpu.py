"""Round-2 parity on the MI355X: trainer-side pieces around the hot path (fused SNRAdam, the train/val loops behind
trainer.py, truncated captions) and the benchmark-sized greedy decode, against the reference's fixtures (tests/golden,
tools/gen_goldens_r2.py) and the CPU oracle."""
import contextlib
import os

import numpy as np
import pytest
import torch

from conftest import load_golden
from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch, tiny_config
from test_model_gpu import grad_close

pytestmark = pytest.mark.gpu
REPORT = {}


@pytest.fixture(scope='module', autouse=True)
def write_report():
    yield
    import json
    os.makedirs('gpurun_out', exist_ok=True)
    with open('gpurun_out/parity_report_r2.json', 'w') as fh:
        json.dump(REPORT, fh, indent=1, sort_keys=True)


def dev():
    return torch.device('cuda:0')


def _wrapper(cfg, **trainer_kw):
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    return ModelTrainerWrapper(cfg, fake_tokenizer(cfg.decoder_config.vocab_size), TrainerWrapperConfig(**trainer_kw),
                               ignore_index=-100).to(dev())


class FakeAccelerator:
    """The members of accelerate.Accelerator the loops touch (single process, gradient accumulation over `accum` micro-batches)."""

    def __init__(self, accum=1):
        self.device, self.is_local_main_process, self.accum = dev(), True, accum
        self.sync_gradients, self._micro, self.saved = True, 0, []

    def autocast(self):
        return contextlib.nullcontext()

    @contextlib.contextmanager
    def accumulate(self, model):
        self._micro += 1
        self.sync_gradients = self._micro % self.accum == 0
        yield

    def backward(self, loss):
        (loss / self.accum).backward()

    def no_sync(self, model):
        return contextlib.nullcontext()

    def gather(self, x):
        return x

    def unwrap_model(self, m):
        return m

    def wait_for_everyone(self):
        pass

    def save(self, obj, f):
        torch.save(obj, f)
        self.saved.append(sorted(obj))

    def print(self, *a):
        pass


class AccumOptimizer:
    """accelerate's optimizer wrapper: step / zero_grad only act on sync micro-batches."""

    def __init__(self, opt, acc):
        self.opt, self.acc, self.steps = opt, acc, 0

    def step(self):
        if self.acc.sync_gradients:
            self.opt.step()
            self.steps += 1

    def zero_grad(self):
        if self.acc.sync_gradients:
            self.opt.zero_grad()


def test_snradam_fused_matches_reference_trajectory():
    """The fused arena SNRAdam against (a) the reference's recorded trajectory, replayed through three parameters of a model,
    and (b) the oracle on every parameter of the model."""
    from image2text_amd.models.optimizer import SNRAdam
    from oracle import reference_model as orc
    g = load_golden('snradam.npz')
    cfg = tiny_config()
    w = _wrapper(cfg)
    det_init_(w.model, seed=0)
    w.train()
    images, labels = synthetic_batch(4, 32, 16, cfg.decoder_config.vocab_size, seed=3)
    w.train_step(images.to(dev()), labels.to(dev()))[0].backward()            # builds the arena, attaches p.grad views
    named = dict(w.model.named_parameters())
    # three parameters whose sizes hold the fixture's tensors (35, 12, 24 elements) in their leading elements
    hosts = ['decoder.transformer.h.0.ln_1.weight', 'decoder.transformer.h.0.ln_1.bias', 'decoder.transformer.h.1.ln_2.weight']
    sizes = [g[f'init.{i}'].size for i in range(3)]
    groups = [{'params': [named[hosts[0]], named[hosts[1]]], 'lr': 3e-3, 'weight_decay': 0.1, 'betas': (0.9, 0.95)},
              {'params': [named[hosts[2]]], 'lr': 1e-3, 'weight_decay': 0.0, 'betas': (0.9, 0.95)}]
    rest = [p for n, p in named.items() if n not in hosts]
    groups.append({'params': rest, 'lr': 2e-3, 'weight_decay': 0.05, 'betas': (0.9, 0.95)})
    opt = SNRAdam(groups)
    with torch.no_grad():
        for i, h in enumerate(hosts):
            named[h].view(-1)[:sizes[i]].copy_(torch.from_numpy(g[f'init.{i}']).view(-1))
    ref = {n: p.detach().cpu().clone() for n, p in named.items()}
    states = {n: {} for n in named}
    hp = {n: dict(lr=2e-3, weight_decay=0.05) for n in named}
    hp[hosts[0]] = hp[hosts[1]] = dict(lr=3e-3, weight_decay=0.1)
    hp[hosts[2]] = dict(lr=1e-3, weight_decay=0.0)
    gen = torch.Generator().manual_seed(0)
    for step in range(6):
        grads = {n: 0.1 * torch.randn(p.shape, generator=gen) for n, p in named.items()}
        for i, h in enumerate(hosts):
            grads[h].view(-1)[:sizes[i]] = torch.from_numpy(g[f'grad.{step}.{i}']).view(-1)
        for n, p in named.items():
            p.grad.copy_(grads[n])
            orc.snradam_step(ref[n], grads[n], states[n], betas=(0.9, 0.95), eps=1e-8, **hp[n])
        opt.step()
        for i, h in enumerate(hosts):
            got = named[h].detach().cpu().view(-1)[:sizes[i]].numpy()
            assert np.abs(got - g[f'param.{step}.{i}'].ravel()).max() <= 2e-6, (step, h)
    for n, p in named.items():
        assert float((p.detach().cpu() - ref[n]).abs().max()) <= 2e-6 * max(1.0, float(ref[n].abs().max())), n
    eng = w.model._engine                                                     # the bf16 shadow follows the update
    assert float((eng.arena.pbf.float() - eng.arena.p32).abs().max()) <= float(eng.arena.p32.abs().max()) / 128


def test_truncated_captions_match_reference():
    """Captions longer than the text window: loss and every gradient against the reference fixture (the per-sequence weight
    normaliser must only cover the kept positions; weight_fn and the EOS weight are exercised too)."""
    g = load_golden('tiny_trunc.npz')
    cfg = tiny_config()
    w = _wrapper(cfg, weight_fn='inverse_sqrt_position', eos_token_weight=2.0)
    det_init_(w.model, seed=0)
    w.train()
    loss, _ = w.train_step(torch.from_numpy(g['images']).to(dev()), torch.from_numpy(g['labels']).to(dev()))
    loss.backward()
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-2 * float(g['loss'])
    bad = []
    for n, p in w.model.named_parameters():
        ref = g[f'grad.{n}'].ravel().astype(np.float64)
        got = p.grad.detach().float().cpu().numpy().ravel().astype(np.float64)
        rel = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        cos = got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)
        if rel > 8e-2 or cos < 0.995:          # (measured worst 6.4e-2: an 8-element conv bias gradient)
            bad.append((n, rel, cos))
    assert not bad, bad[:6]


def test_train_and_val_loops_with_accumulation(tmp_path):
    """training.utils.train_loop / val_loop as trainer.py drives them (reference training/utils.py:63-164), with a fake
    accelerator: 6 micro-batches at accumulation 2 = 3 optimizer steps; equal to the same 3 steps written out by hand;
    metrics reach the logging callback once per micro-batch with the right batch index; the partial checkpoint holds exactly
    the parameters the optimizer patterns select and loads back through update_state_dict_from_partial_checkpoint."""
    from image2text_amd.models.utils import PatternMatcher, update_state_dict_from_partial_checkpoint
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.utils import train_loop, val_loop
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    batches = [synthetic_batch(4, 32, 16, V, seed=50 + i) for i in range(6)]

    def build():
        w = _wrapper(cfg)
        det_init_(w.model, seed=0)
        return w, FusedAdamW(w.model.parameters(), lr=1e-3, betas=(0.9, 0.95), weight_decay=0.0)

    # Round 3: the comparison runs in deterministic mode (fixed-order reductions, ops.set_deterministic) and is EXACT.  In the default
    # mode Adam turns gradient elements that are zero up to atomics jitter into +-lr steps (0.14 % ... 1.5 % of wte's elements
    # differed from box to box), which this test used to absorb with allowances.
    from image2text_amd import ops
    ops.set_deterministic(True)
    try:
        # by hand
        w0, o0 = build()
        w0.train()
        for i in range(0, 6, 2):
            for im, lb in batches[i:i + 2]:
                (w0.train_step(im.to(dev()), lb.to(dev()))[0] / 2).backward()
            o0.step()
            o0.zero_grad()
        # through the loop
        w1, o1 = build()
        acc = FakeAccelerator(accum=2)
        opt = AccumOptimizer(o1, acc)
        seen = []
        ck = str(tmp_path / 'partial.pt')
        matchers = [PatternMatcher(['decoder*.transformer.h.*.cross_attn.*']), PatternMatcher(['encoder.1.*'])]
        stop = train_loop(w1, opt, iter(batches), epoch=0, num_steps=6, accelerator=acc, disable_flash=True,
                          logging_callback=lambda m, batch, epoch: seen.append((batch, m['train_loss_lm'])), chckpt_fname=ck,
                          matchers=matchers)
    finally:
        ops.set_deterministic(False)
    assert stop is False and opt.steps == 3
    assert [b for b, _ in seen] == list(range(6)) and all(np.isfinite(v) for _, v in seen)
    for (n, p0), (_, p1) in zip(w0.model.named_parameters(), w1.model.named_parameters()):
        assert torch.equal(p0, p1), (n, float((p0 - p1).abs().max()))
    saved = torch.load(ck, weights_only=True)
    want = sorted(n for n, _ in w1.model.named_parameters() if 'cross_attn' in n or n.startswith('encoder.1.'))
    assert sorted(saved) == want and len(want) >= 5
    w2, _ = build()
    update_state_dict_from_partial_checkpoint(w2.model, ck, map_location='cpu')
    for n in want:
        assert torch.equal(dict(w2.model.named_parameters())[n].detach().cpu(), saved[n].cpu())
    # an exhausted iterator stops the epoch
    assert train_loop(w1, opt, iter(batches[:2]), 1, 5, FakeAccelerator(), chckpt_fname=None) is True
    # validation: mean over steps of val_step
    vloss, vmetrics = val_loop(w1, iter(batches), 0, 3, FakeAccelerator())
    w1.eval()
    with torch.no_grad():
        ref = np.mean([float(w1.val_step(im.to(dev()), lb.to(dev()))[0]) for im, lb in batches[:3]])
    assert abs(vloss - ref) <= 1e-5 * ref and abs(vmetrics['val_loss_lm'] - ref) <= 1e-5 * ref


def test_nano224_greedy_64_tokens_8_captions():
    """The benchmark's decode workload at parity size: 8 captions x 64 new tokens at nano-224 against the reference's
    generate(top_k=1) (tests/golden/nano224_greedy64.npz).  Token-exact wherever the oracle's top-1 margin is >= 0.03
    = 2 x the measured max-abs logit error of the bf16 path (a flip needs both competitors to move by the full error);
    after a sub-margin flip the run is re-synchronised on the golden prefix so every later step is still checked."""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from test_model_gpu import assert_greedy_matches
    g = load_golden('nano224_greedy64.npz')
    cfg = nano224_config()
    m = det_init_(VisionEncoderDecoder(cfg), seed=0).to(dev()).eval()
    images, _ = synthetic_batch(8, 224, 64, cfg.decoder_config.vocab_size, seed=2)
    eps = 0.03
    r = assert_greedy_matches(m, images.to(dev()), g['ids'], g['margins'], 1, eps=eps)
    checked = int((g['margins'] >= eps).sum())
    REPORT['nano224.greedy64'] = {'low_margin_restarts': r, 'steps_with_margin_ge_eps': checked, 'steps': int(g['margins'].size), 'eps': eps}
    assert checked >= 0.5 * g['margins'].size
    # how many of ALL 512 decisions agree when the run is never re-synchronised by more than the flips themselves
    assert r <= int((g['margins'] < eps).sum())


# ------------------------------------------------------------------------------------------------ on-device sampling
SAMPLING_MODES = {'t07_k5': dict(temperature=0.7, top_k=5), 't10_p05': dict(temperature=1.0, nucleus_p=0.5),
                  't07_p06': dict(temperature=0.7, nucleus_p=0.6), 't13_k20_p09': dict(temperature=1.3, top_k=20, nucleus_p=0.9),
                  't10_plain': dict(temperature=1.0), 't20_p095': dict(temperature=2.0, nucleus_p=0.95)}


def _check_step_against_oracle(dec, model, ids_prefix, dist_dev, kw, tag):
    """The kept, renormalised distribution the device sampled from vs the oracle's on THE DEVICE'S OWN fp32 logits of that step
    (same logits -> same filter decisions; the oracle itself is pinned to the reference by tests/test_oracle_golden.py)."""
    from oracle import reference_model as orc
    cfg = model.config
    V = cfg.decoder_config.vocab_size
    logits = dec._state.logits[:, :V].float().cpu()
    want = orc.sampling_distribution(logits, ids_prefix.cpu(), cfg.no_repeat_n_grams, **kw).numpy()
    got = dist_dev.cpu().numpy()
    diff = (got > 0) != (want > 0)
    # an entry may sit within float noise of the nucleus cut (different summation order): at most one such entry per row
    assert diff.sum(axis=1).max() <= 1, (tag, diff.sum(axis=1))
    assert np.abs(got - want)[~diff].max() <= 1e-5 + 1e-3 * want.max(), (tag, float(np.abs(got - want)[~diff].max()))
    return want


@pytest.mark.parametrize('wtag', ['trained', 'init'])
def test_sampling_filters_match_reference_goldens(wtag, tiny_weights):
    """generate() sampling modes on the tiny model, teacher-forced on the reference's recorded ids (tests/golden/tiny_sampling.npz):
    per step (a) device filter == oracle filter on the device's logits (kept set exact, probabilities 1e-3), (b) the drawn token
    is the inverse-CDF token of (seed, step, row), (c) the device distribution is close to the REFERENCE's recorded one in total
    variation (bf16 logits: not bit-equal)."""
    from image2text_amd import rng
    from image2text_amd.decoding import GreedyDecoder, Sampling
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from oracle import reference_model as orc
    g = load_golden('tiny_sampling.npz')
    cfg = tiny_config()
    m = VisionEncoderDecoder(cfg)
    if wtag == 'trained':
        m.load_state_dict(tiny_weights)
    else:
        det_init_(m, seed=0)
    m = m.to(dev()).eval()
    images = torch.from_numpy(g['images']).to(dev())
    dec = GreedyDecoder(m)
    seed = 0x1234_5678_9ABC_DEF0
    tv_max = 0.0
    for tag, kw in SAMPLING_MODES.items():
        ids, dist = torch.from_numpy(g[f'{wtag}.{tag}.ids']), g[f'{wtag}.{tag}.dist']
        for s in range(0, dist.shape[1], 2):
            prefix = ids[:, :1 + s].contiguous()
            out, dd = dec.generate(images, prefix.to(dev()), 1, sampling=Sampling(seed=seed, **kw), return_dists=True, use_graph=(s % 4 == 0))
            want = _check_step_against_oracle(dec, m, prefix, dd[:, 0], kw, (wtag, tag, s))
            # (b) the draw: inverse CDF at u(seed, step = current length, row)
            u = torch.tensor([rng.sample_uniform(seed, 1 + s, b) for b in range(ids.shape[0])])
            tok = orc.inverse_cdf_token(dd[:, 0].cpu(), u)
            cdf = torch.cumsum(dd[:, 0].cpu().double(), -1)
            near = ((cdf - u.double().unsqueeze(1)).abs().min(dim=1).values < 1e-5)      # u within rounding of a CDF step
            assert bool(((out[:, -1].cpu() == tok) | near).all()), (wtag, tag, s, out[:, -1].tolist(), tok.tolist())
            assert bool((dd[:, 0].cpu().gather(1, out[:, -1:].cpu()) > 0).all()), 'token outside the kept set'
            # (c) vs the reference's own distribution
            tv_max = max(tv_max, float(0.5 * np.abs(dd[:, 0].cpu().numpy() - dist[:, s]).sum(axis=1).max()))
    REPORT[f'sampling.tiny.{wtag}.max_total_variation_vs_reference'] = tv_max
    assert tv_max <= (0.25 if wtag == "trained" else 0.1), tv_max      # bf16 logits: a token at the nucleus cut may change sides (measured 0.055); trained: |logit| <= 27


def test_sampling_nano224_full_vocabulary_and_statistics():
    """Full-size vocabulary (50257 = 100 register-resident values per lane): filters against the oracle on the device's logits
    for four modes; then the draw is checked statistically -- 4096 copies of one caption differ only in their row index, i.e.
    in their uniform, so their tokens are i.i.d. draws from one distribution."""
    from image2text_amd.decoding import GreedyDecoder, Sampling
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    cfg = nano224_config()
    V = cfg.decoder_config.vocab_size
    m = det_init_(VisionEncoderDecoder(cfg), seed=0).to(dev()).eval()
    with torch.no_grad():                                     # sharper logits: a handful of tokens carry most of the mass
        m.decoder.transformer.wte.weight.mul_(6.0)
    images, _ = synthetic_batch(8, 224, 64, V, seed=2)
    images = images.to(dev())
    dec = GreedyDecoder(m)
    prompt = torch.randint(0, V - 1, (8, 5), generator=torch.Generator().manual_seed(1))
    prompt[:, 3] = prompt[:, 1]                                # a repeated token: the n-gram ban has something to do
    for tag, kw in (('p06', dict(temperature=0.7, nucleus_p=0.6)), ('k50', dict(temperature=1.0, top_k=50)),
                    ('k20p09', dict(temperature=1.3, top_k=20, nucleus_p=0.9)), ('plain', dict(temperature=1.0))):
        out, dd = dec.generate(images, prompt.to(dev()), 1, sampling=Sampling(seed=7, **kw), return_dists=True, use_graph=False)
        want = _check_step_against_oracle(dec, m, prompt, dd[:, 0], kw, tag)
        REPORT[f'sampling.nano224.{tag}.kept'] = [int(x) for x in (want > 0).sum(axis=1)]
    # statistics: one image, one prompt, 4096 rows
    B = 4096
    img1 = images[:1].expand(B, -1, -1, -1).contiguous()
    p1 = prompt[:1].expand(B, -1).contiguous().to(dev())
    big = GreedyDecoder(m)
    out, dd = big.generate(img1, p1, 1, sampling=Sampling(temperature=1.0, top_k=8, seed=99), return_dists=True)
    p = dd[0, 0].cpu().double()
    assert float((dd[:, 0].cpu().double() - p).abs().max()) < 1e-6       # identical rows -> identical distributions
    freq = torch.bincount(out[:, -1].cpu(), minlength=V).double() / B
    kept = p > 0
    assert int(kept.sum()) == 8 and float(freq[~kept].sum()) == 0.0
    sigma = torch.sqrt(p * (1 - p) / B)
    assert bool(((freq - p).abs()[kept] <= 4.5 * sigma[kept] + 1e-9).all()), (freq[kept], p[kept])


def test_generate_sampling_api():
    """VisionEncoderDecoder.generate in the reference's sampling call shapes (unit test models/vision_encoder_decoder_test.py:88-92,
    eval_model trainer.py:41-56): shapes, prompt kept, reproducible under torch.manual_seed, different seeds differ, hipGraph
    replay == eager steps, and the no-repeat-n-gram constraint holds on sampled captions too."""
    from image2text_amd.decoding import GreedyDecoder, Sampling
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    m = det_init_(VisionEncoderDecoder(cfg), seed=0).to(dev()).eval()
    images, _ = synthetic_batch(4, 32, 16, V, seed=3)
    images = images.to(dev())
    prompt = torch.full((4, 1), V - 1, dtype=torch.long)
    torch.manual_seed(5)
    a = m.generate(images, prompt, max_new_tokens=30, temperature=0.7, nucleus_p=0.6)
    torch.manual_seed(5)
    b = m.generate(images, prompt, max_new_tokens=30, temperature=0.7, nucleus_p=0.6)
    c = m.generate(images, prompt, max_new_tokens=30, temperature=0.7, nucleus_p=0.6)
    assert tuple(a.shape) == (4, 31) and torch.equal(a[:, :1].cpu(), prompt) and torch.equal(a, b) and not torch.equal(a, c)
    for row in a.cpu().tolist():
        for n in cfg.no_repeat_n_grams:
            grams = [tuple(row[i:i + n]) for i in range(len(row) - n + 1)]
            assert len(grams) == len(set(grams)), f'repeated {n}-gram in a sampled caption'
    d = GreedyDecoder(m)
    s = Sampling(temperature=1.1, top_k=12, nucleus_p=0.8, seed=1234)
    assert torch.equal(d.generate(images, prompt.to(dev()), 20, sampling=s), d.generate(images, prompt.to(dev()), 20, sampling=s, use_graph=False))
    # top_k = 1 is greedy whatever the temperature
    assert torch.equal(m.generate(images, prompt, max_new_tokens=12, temperature=0.5, top_k=1),
                       m.generate(images, prompt, max_new_tokens=12, temperature=1.0, top_k=1))


# ------------------------------------------------------------------------------------------------ trainer extras (f4)
def test_momentum_distillation_matches_reference():
    """train_step with moco_momentum / moco_alpha (reference wrapper.py:30-33,46-59,134-144): loss and every gradient of the model
    against the reference fixture (twin at different weights, so the soft targets matter), the twin's parameters after the step's
    EMA launch, validation = plain CE, copy_momentum_params re-synchronises."""
    from types import SimpleNamespace
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    g = load_golden('tiny_moco.npz')
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    tok = SimpleNamespace(eos_token_id=V - 1, bos_token_id=V - 1, mask_token_id=V - 2, vocab_size=V)
    w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(moco_momentum=0.9, moco_alpha=0.4, training_temperature=1.3), ignore_index=-100)
    det_init_(w.model, seed=0)
    det_init_(w.model_m, seed=1)
    w = w.to(dev()).train()
    p_before = {n: p.detach().clone() for n, p in w.model.named_parameters()}
    pm_before = {n: p.detach().clone() for n, p in w.model_m.named_parameters()}
    images, labels = torch.from_numpy(g['images']).to(dev()), torch.from_numpy(g['labels']).to(dev())
    loss, _ = w.train_step(images, labels)
    loss.backward()
    assert abs(float(loss.detach()) - float(g['loss'])) <= 1e-2 * float(g['loss'])
    bad = []
    for n, p in w.model.named_parameters():
        ref = g[f'grad.{n}'].ravel().astype(np.float64)
        got = p.grad.detach().float().cpu().numpy().ravel().astype(np.float64)
        rel = np.linalg.norm(got - ref) / (np.linalg.norm(ref) + 1e-30)
        cos = got @ ref / (np.linalg.norm(got) * np.linalg.norm(ref) + 1e-30)
        if rel > 8e-2 or cos < 0.995:
            bad.append((n, rel, cos))
    assert not bad, bad[:6]
    for n, p in w.model_m.named_parameters():               # EMA: exact fp32 arithmetic on the arenas
        want = pm_before[n] * 0.9 + p_before[n] * (1.0 - 0.9)
        assert float((p.detach() - want).abs().max()) <= 1e-6 * max(1.0, float(want.abs().max())), n
        assert np.abs(p.detach().cpu().numpy() - g[f'ema.{n}']).max() <= 1e-6 * max(1.0, float(np.abs(g[f'ema.{n}']).max())), n
    em = w.model_m._engine                                    # the twin's bf16 shadow follows its fp32 arena
    assert float((em.arena.pbf.float() - em.arena.p32).abs().max()) <= float(em.arena.p32.abs().max()) / 128
    w.eval()
    with torch.no_grad():
        vloss, _ = w.val_step(images, labels)
    assert abs(float(vloss) - float(g['val_loss'])) <= 1e-2 * float(g['val_loss'])
    w.copy_momentum_params()
    for (n, p), (_, pm) in zip(w.model.named_parameters(), w.model_m.named_parameters()):
        assert torch.equal(p, pm), n
    # model_m never receives gradients and stays out of the optimizer (trainer.py:158-159 filters 'model_m.')
    assert all(p.grad is None for p in w.model_m.parameters())


def test_mlm_corruption_inputs():
    """The decoder-input kernel (BOS shift, ignored -> EOS, MLM corruption) against the oracle fed the kernel's own draws
    (rng.mlm_draws), the corruption statistics, reproducibility, and the clean path == the reference's shifted inputs."""
    from image2text_amd import ops, rng
    from oracle import reference_model as orc
    V, B, L = 50258, 64, 64
    _, labels = synthetic_batch(B, 32, L, V - 1, seed=8)
    lab = labels.to(dev())
    ids = torch.empty_like(lab)
    ops.lm_inputs(lab, ids, B, L, V - 2, V - 2, None, V, -100)
    assert torch.equal(ids.cpu(), orc.shifted_inputs(labels, V - 2, V - 2)[0])
    seed = 0xC0FFEE1234567
    ops.lm_inputs(lab, ids, B, L, V - 2, V - 2, V - 1, V, -100, mask_fraction=0.3, random_fraction=0.4, seed=seed)
    u_mask, u_rand, num = rng.mlm_draws(seed, B * L)
    r_ids = ((num * V) >> 32).view(B, L)
    want = orc.lm_inputs(labels, V - 2, V - 2, -100, mask_id=V - 1, mask_fraction=0.3, random_fraction=0.4, u_mask=u_mask.view(B, L),
                         u_rand=u_rand.view(B, L), r_ids=r_ids)
    assert torch.equal(ids.cpu(), want)
    live = (labels != -100)[:, :-1]
    changed = (ids.cpu()[:, 1:] != torch.where(labels != -100, labels, torch.full_like(labels, V - 2))[:, :-1]) & live
    n_live = int(live.sum())
    frac = float(changed.sum()) / n_live
    assert abs(frac - 0.3) < 4 * (0.3 * 0.7 / n_live) ** 0.5 + 1e-3, frac
    masked = (ids.cpu()[:, 1:] == V - 1) & live
    assert abs(float(masked.sum()) / max(1, int(changed.sum())) - 0.6) < 0.08
    ids2 = torch.empty_like(lab)
    ops.lm_inputs(lab, ids2, B, L, V - 2, V - 2, V - 1, V, -100, mask_fraction=0.3, random_fraction=0.4, seed=seed)
    assert torch.equal(ids, ids2)
    ops.lm_inputs(lab, ids2, B, L, V - 2, V - 2, V - 1, V, -100, mask_fraction=0.3, random_fraction=0.4, seed=seed + 1)
    assert not torch.equal(ids, ids2)


def test_mlm_train_step_runs_and_val_is_clean():
    """mask_fraction > 0: training steps see corrupted inputs (loss differs from the clean step and from step to step), validation
    steps never do (wrapper.py:161,183-185)."""
    from types import SimpleNamespace
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    tok = SimpleNamespace(eos_token_id=V - 1, bos_token_id=V - 1, mask_token_id=V - 2, vocab_size=V)
    images, labels = synthetic_batch(8, 32, 16, V - 2, seed=12)
    images, labels = images.to(dev()), labels.to(dev())
    losses = {}
    for tag, kw in (('clean', {}), ('mlm', dict(mask_fraction=0.5, random_mask_fraction=0.2))):
        w = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(**kw), ignore_index=-100)
        det_init_(w.model, seed=0)
        w = w.to(dev()).train()
        torch.manual_seed(3)
        l1, _ = w.train_step(images, labels)
        l1.backward()
        l2, _ = w.train_step(images, labels)
        w.eval()
        with torch.no_grad():
            v, _ = w.val_step(images, labels)
        losses[tag] = (float(l1.detach()), float(l2.detach()), float(v))
    assert abs(losses['clean'][0] - losses['clean'][1]) < 1e-4 * losses['clean'][0]
    assert abs(losses['mlm'][0] - losses['clean'][0]) > 1e-3 and abs(losses['mlm'][0] - losses['mlm'][1]) > 1e-4
    assert abs(losses['mlm'][2] - losses['clean'][2]) < 1e-4 * losses['clean'][2]


@pytest.mark.parametrize('tag,kw', [('prompt', {}), ('cross_only', dict(use_soft_prompting=False))])
@pytest.mark.parametrize('packed', [False, True])
def test_contrastive_loss_and_every_gradient(tag, kw, packed):
    """add_contrastive_loss (reference training/wrapper.py:98-118, temperature 0.7): both loss terms and the gradient of every parameter
    against the reference's fixture (tests/golden/tiny_contrastive.npz) -- with a soft prompt the term differentiates the prompt rows
    of hidden_state (their own decoder segment) -- dense and with packed caption rows; validation reports the term too."""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    g = load_golden('tiny_contrastive.npz')
    cfg = tiny_config(**kw)
    w = ModelTrainerWrapper(cfg, fake_tokenizer(cfg.decoder_config.vocab_size),
                            TrainerWrapperConfig(add_contrastive_loss=True, training_contrastive_temperature=0.7), ignore_index=-100)
    det_init_(w.model, seed=0)
    w = w.to(dev()).train()
    w.pack_rows = packed
    images, labels = torch.from_numpy(g['images']).to(dev()), torch.from_numpy(g['labels']).to(dev())
    loss, metrics = w.train_step(images, labels)
    loss.backward()
    for key, got in (('loss', loss), ('loss_lm', metrics['train_loss_lm']), ('loss_contrastive', metrics['train_loss_contrastive'])):
        ref = float(g[f'{tag}.{key}'])
        REPORT[f'contrastive.{tag}.{packed}.{key}'] = {'got': float(got.detach()), 'ref': ref}
        assert abs(float(got.detach()) - ref) <= 1e-2 * max(1.0, ref), (key, float(got.detach()), ref)
    fails = []
    for name, p in w.model.named_parameters():
        try:
            grad_close(f'contrastive.{tag}.{name}', p.grad, g[f'{tag}.grad.{name}'], rel=6e-2, cos=0.995)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    w.eval()
    with torch.no_grad():
        vloss, vm = w.val_step(images, labels)
    assert abs(float(vloss) - float(g[f'{tag}.val_loss'])) <= 1e-2 * float(g[f'{tag}.val_loss'])
    assert abs(float(vm['val_loss_contrastive']) - float(g[f'{tag}.val_loss_contrastive'])) <= 1e-2 * float(g[f'{tag}.val_loss_contrastive'])


def test_imported_gpt2_decoder_matches_hugging_face_forward(monkeypatch):
    """pretrained_model: gpt2 -> the nanoGPT decoder with imported weights (reference decoder.py:45-117).  An independent check of the
    whole decoder stack: a randomly initialised Hugging Face GPT-2 evaluated by transformers on the CPU in fp32 vs the same weights
    through the HIP path (standalone decoder call, no cross-attention input), logits and hidden states; and greedy generation by
    the KV-cache path vs transformers' generate."""
    from test_host_cpu import _gpt2_decoder_config, _tiny_hf_gpt2
    from image2text_amd.models.decoder import Decoder
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    hf = _tiny_hf_gpt2(monkeypatch)
    cfg = tiny_config(dec_d=128, dec_heads=2, dec_layers=2, block_size=64, use_soft_prompting=False)
    dec = Decoder.from_config(_gpt2_decoder_config(), loose=True)
    m = VisionEncoderDecoder(cfg, decoder=dec).to(dev()).eval()
    g = torch.Generator().manual_seed(1)
    ids = torch.randint(0, 383, (3, 40), generator=g)
    with torch.no_grad():
        ref = hf(input_ids=ids, output_hidden_states=True)
        logits, hidden = m.decoder(idx=ids.to(dev()))
    tol = 1e-2 * max(1.0, float(ref.logits.abs().max()))
    err = float((logits.float().cpu() - ref.logits).abs().max())
    REPORT['gpt2_import.logits'] = {'max_abs_err': err, 'tol': tol, 'ref_absmax': float(ref.logits.abs().max())}
    assert err <= tol, (err, tol)
    ref_h = hf.transformer(input_ids=ids).last_hidden_state
    assert float((hidden.float().cpu() - ref_h).abs().max()) <= 1.5e-2 * max(1.0, float(ref_h.abs().max()))
    assert (logits.argmax(-1).cpu() == ref.logits.argmax(-1)).float().mean() > 0.9


def test_hidden_state_prompt_rows_are_differentiable():
    """VisionEncoderDecoder.forward returns the reference's full hidden_state (prompt rows included, vision_encoder_decoder.py:133) and
    a custom loss on it back-propagates like the reference's autograd: prompt rows through their own decoder segment into the
    encoder, in lock step with the text rows (one gradient normaliser per block) -- every gradient against the oracle."""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from oracle import reference_model as orc
    cfg = tiny_config()
    m = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m = m.to(dev()).train()
    images, labels = synthetic_batch(3, 32, 12, cfg.decoder_config.vocab_size, seed=17)
    ids = labels.clamp(min=0)
    g = torch.Generator().manual_seed(2)
    n_p = cfg.vision_encoder_config.n_cls
    wh = torch.randn(3, n_p + 12, 128, generator=g) * 0.05           # a loss that touches prompt rows, text rows and logits
    wl = torch.randn(3, 12, cfg.decoder_config.vocab_size, generator=g) * 0.01
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    assert tuple(out.hidden_state.shape) == (3, n_p + 12, 128)
    loss = (out.hidden_state * wh.to(dev())).sum() + (out.logits * wl.to(dev())).sum()
    loss.backward()
    osd = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
    osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
    _, ologits, ohid = orc.forward(osd, cfg, images, ids, None, training=True)
    oloss = (ohid * wh).sum() + (ologits * wl).sum()
    oloss.backward()
    assert abs(float(loss.detach()) - float(oloss)) <= 0.1          # a random-signed sum over 21 k outputs, each within the bf16 tolerance
    fails = []
    for name, p in m.named_parameters():
        try:
            grad_close(f'hidden_prompt.{name}', p.grad, osd[name].grad.numpy(), rel=8e-2, cos=0.99)
        except AssertionError as e:
            fails.append(str(e))
    assert not fails, f'{len(fails)} gradients out of tolerance: ' + '; '.join(fails[:6])
    # a loss on the logits alone does not pay for the prompt segment's backward (its gradient arrives as None)
    m.zero_grad()
    out = m(images=images.to(dev()), ids=ids.to(dev()))
    (out.logits * wl.to(dev())).sum().backward()
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
