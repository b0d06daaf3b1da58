"""Pin the CPU oracle (oracle/reference_model.py) to the reference's own outputs (tests/golden, made by
tools/gen_goldens.py running the reference).  CPU only; fp32 tolerance 1e-5 abs on O(1) values, token-exact ids."""
import numpy as np
import pytest
import torch

from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch, tiny_config
from oracle import reference_model as orc

TOL = 1e-5


def close(a, b, tol=TOL):
    a = a.detach().numpy() if isinstance(a, torch.Tensor) else a
    err = np.abs(a - b).max()
    assert err <= tol, f'max abs err {err}'


@pytest.mark.parametrize('tag', ['nomask', 'row_mask', 'sl_mask', 'bsl_mask'])
def test_forward_masks(tiny_weights, tiny_forward, tag):
    cfg = tiny_config()
    f = tiny_forward
    m = None if tag == 'nomask' else torch.from_numpy(f[tag])
    with torch.no_grad():
        enc, logits, hidden = orc.forward(tiny_weights, cfg, torch.from_numpy(f['images']), torch.from_numpy(f['ids']), m)
    close(enc, f[f'{tag}.encoder_output'])
    close(logits, f[f'{tag}.logits'], 2e-5)
    close(hidden, f[f'{tag}.hidden_state'], 2e-5)


@pytest.mark.parametrize('tag,kw', [('cross_only', dict(use_soft_prompting=False)), ('prompt_only', dict(use_cross_attn=False))])
def test_forward_modes(tiny_weights, tiny_forward, tag, kw):
    cfg = tiny_config(**kw)
    f = tiny_forward
    with torch.no_grad():
        _, logits, hidden = orc.forward(tiny_weights, cfg, torch.from_numpy(f['images']), torch.from_numpy(f['ids']),
                                        torch.from_numpy(f['row_mask']))
    close(logits, f[f'{tag}.logits'], 2e-5)
    close(hidden, f[f'{tag}.hidden_state'], 2e-5)


def test_encoder_stages(tiny_weights, tiny_forward):
    cfg = tiny_config()
    f = tiny_forward
    sd = orc._sub(tiny_weights, 'encoder.0.')
    with torch.no_grad():
        conv = orc.conv_stack(sd, 'feature_extractor', torch.from_numpy(f['images']))
        enc = orc.vit_encoder(sd, cfg.vision_encoder_config, torch.from_numpy(f['images']))
    close(conv, f['inter.enc.conv'])
    close(enc, f['inter.enc.out'])


def test_train_step_loss_and_all_grads(tiny_weights, tiny_forward, tiny_train):
    cfg = tiny_config()
    f = tiny_forward
    sd = {k: v.clone().requires_grad_(True) for k, v in tiny_weights.items() if k != 'decoder.lm_head.weight'}
    sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    loss = orc.lm_step(sd, cfg, torch.from_numpy(f['images']), torch.from_numpy(f['labels']), tok, training=True)
    loss.backward()
    assert abs(loss.item() - float(tiny_train['loss'])) < 1e-6
    n = 0
    for k, v in sd.items():
        if k == 'decoder.lm_head.weight':
            continue
        key = f'grad.{k}' if f'grad.{k}' in tiny_train else None
        if key is None and k == 'decoder.transformer.wte.weight':
            key = 'grad.decoder.lm_head.weight'
        g = tiny_train[key]
        scale = max(np.abs(g).max(), 1e-6)
        assert np.abs(v.grad.numpy() - g).max() <= 1e-4 * scale + 1e-7, k
        n += 1
    assert n >= 40
    with torch.no_grad():
        vloss = orc.lm_step(sd, cfg, torch.from_numpy(f['images']), torch.from_numpy(f['labels']), tok, training=False)
    assert abs(vloss.item() - float(tiny_train['val_loss'])) < 1e-6


def test_greedy_tokens_exact(tiny_weights, tiny_decode):
    cfg = tiny_config()
    d = tiny_decode
    ids, margins = orc.generate_greedy(tiny_weights, cfg, torch.from_numpy(d['images']), torch.from_numpy(d['prompt']),
                                       d['ids'].shape[1] - 1, return_margins=True)
    assert np.array_equal(ids.numpy(), d['ids'])
    close(margins, d['margins'], 1e-4)
    ids3 = orc.generate_greedy(tiny_weights, cfg, torch.from_numpy(d['images']), torch.from_numpy(d['prompt3']), 12)
    assert np.array_equal(ids3.numpy(), d['ids3'])


def test_ngram_ban_known_answers():
    # hand-worked cases of the HF no-repeat-ngram rule (reference call sites vision_encoder_decoder.py:40-43,153)
    assert orc.banned_next_tokens([1, 2, 3, 1, 2], 3) == [3]
    assert orc.banned_next_tokens([5, 5, 5], 2) == [5, 5]
    assert orc.banned_next_tokens([7], 3) == []              # len+1 < n: nothing banned
    assert orc.banned_next_tokens([1, 2], 3) == []           # prefix (1,2) never seen followed by anything
    assert sorted(orc.banned_next_tokens([1, 2, 1, 3, 1], 2)) == [2, 3]


@pytest.mark.slow
def test_nano224_full_size(nano224_golden):
    """Full-size nano-224 with regenerated det_init_ weights: logits slices, LSE, argmax, loss, greedy prefix."""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    g = nano224_golden
    cfg = nano224_config()
    model = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd = dict(model.state_dict())
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    assert np.array_equal(labels.numpy(), g['labels'])
    ids, msk = orc.shifted_inputs(labels, tok.bos_token_id, tok.eos_token_id)
    with torch.no_grad():
        enc, logits, hidden = orc.forward(sd, cfg, images, ids, msk)
        vloss = orc.lm_step(sd, cfg, images, labels, tok, training=False)
    close(enc, g['encoder_output'], 5e-5)
    close(logits[:, :, :256], g['logits_head'], 5e-5)
    close(torch.logsumexp(logits, -1), g['logits_lse'], 5e-5)
    assert abs(vloss.item() - float(g['val_loss'])) < 1e-4
    gids = orc.generate_greedy(sd, cfg, images, torch.full((2, 1), tok.bos_token_id), 4)
    assert np.array_equal(gids.numpy(), g['greedy_ids'][:, :5])


def test_text_segment_factorisation_equals_full_sequence(tiny_weights, tiny_forward, tiny_train):
    """The text-only pass the HIP path runs is the same function as the reference's full prompt+text sequence."""
    cfg = tiny_config()
    f = tiny_forward
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    with torch.no_grad():
        a = orc.lm_step_text_segment(tiny_weights, cfg, torch.from_numpy(f['images']), torch.from_numpy(f['labels']), tok)
    assert abs(a.item() - float(tiny_train['loss'])) < 1e-6


# ---------------------------------------------------------------------------------------------------- round-2 fixtures
def test_snradam_trajectory():
    """oracle.snradam_step == the reference's SNRAdam (tools/gen_goldens_r2.py): 6 steps, 3 tensors, 2 param groups."""
    from conftest import load_golden
    g = load_golden('snradam.npz')
    hp = [dict(lr=3e-3, weight_decay=0.1), dict(lr=3e-3, weight_decay=0.1), dict(lr=1e-3, weight_decay=0.0)]
    params = [torch.from_numpy(g[f'init.{i}']).clone() for i in range(3)]
    states = [{} for _ in range(3)]
    for step in range(6):
        for i in range(3):
            orc.snradam_step(params[i], torch.from_numpy(g[f'grad.{step}.{i}']), states[i], betas=(0.9, 0.95), eps=1e-8, **hp[i])
            close(params[i], g[f'param.{step}.{i}'], 1e-6)


def test_truncated_captions_loss_and_grads():
    """Captions longer than the text window (block_size - n_cls): labels are cut BEFORE the loss weights are normalised."""
    from conftest import load_golden
    g = load_golden('tiny_trunc.npz')
    cfg = tiny_config()
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    model = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items() if k != 'decoder.lm_head.weight'}
    sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    loss = orc.lm_step(sd, cfg, torch.from_numpy(g['images']), torch.from_numpy(g['labels']), tok, training=True,
                       weight_fn='inverse_sqrt_position', eos_token_weight=2.0)
    loss.backward()
    assert abs(float(loss) - float(g['loss'])) <= 1e-5 * float(g['loss'])
    for k in g:
        if k.startswith('grad.') and k[5:] in sd:
            close(sd[k[5:]].grad, g[k], 2e-5 * max(1.0, float(np.abs(g[k]).max())))


@pytest.mark.parametrize('wtag', ['trained', 'init'])
def test_sampling_distributions(tiny_weights, wtag):
    """oracle.sampling_distribution == the distribution the reference's generate() hands to torch.multinomial, for every
    recorded mode and step (teacher-forced on the recorded ids)."""
    from conftest import load_golden
    g = load_golden('tiny_sampling.npz')
    cfg = tiny_config()
    if wtag == 'trained':
        sd = tiny_weights
    else:
        from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
        sd = {k: v.detach() for k, v in det_init_(VisionEncoderDecoder(cfg), seed=0).state_dict().items()}
    images = torch.from_numpy(g['images'])
    modes = {'t07_k5': dict(temperature=0.7, top_k=5), 't10_p05': dict(temperature=1.0, nucleus_p=0.5),
             't07_p06': dict(temperature=0.7, nucleus_p=0.6), 't13_k20_p09': dict(temperature=1.3, top_k=20, nucleus_p=0.9),
             't10_plain': dict(temperature=1.0), 't20_p095': dict(temperature=2.0, nucleus_p=0.95)}
    enc = None
    for tag, kw in modes.items():
        ids, dist = torch.from_numpy(g[f'{wtag}.{tag}.ids']), g[f'{wtag}.{tag}.dist']
        for s in range(0, dist.shape[1], 3):
            with torch.no_grad():
                enc, logits, _ = orc.forward(sd, cfg, images, ids[:, :1 + s], None, encoder_output=enc)
            got = orc.sampling_distribution(logits[:, -1, :], ids[:, :1 + s], cfg.no_repeat_n_grams, **kw).numpy()
            ref = dist[:, s]
            # kept sets: identical except where an entry sits within float noise of the nucleus cut
            diff = (got > 0) != (ref > 0)
            assert diff.sum() <= 1 and np.abs(got - ref)[~diff].max() <= 2e-5, (tag, s, int(diff.sum()))


def test_inverse_cdf_rule():
    d = torch.tensor([[0.0, 0.25, 0.0, 0.5, 0.25, 0.0]])
    for u, want in ((0.0, 1), (0.2499, 1), (0.25, 3), (0.7499, 3), (0.75, 4), (0.999999, 4)):
        assert int(orc.inverse_cdf_token(d, torch.tensor([u]))) == want, u


def test_nano224_greedy64_prefix():
    """The 8 x 64 greedy fixture (the benchmark's decode workload) against the oracle on a CPU-sized corner: captions 0-1,
    first 6 steps (the full run is the GPU parity test's job)."""
    from conftest import load_golden
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    g = load_golden('nano224_greedy64.npz')
    cfg = nano224_config()
    sd = {k: v.detach() for k, v in det_init_(VisionEncoderDecoder(cfg), seed=0).state_dict().items()}
    images, _ = synthetic_batch(8, 224, 64, cfg.decoder_config.vocab_size, seed=2)
    ids, margins = orc.generate_greedy(sd, cfg, images[:2], torch.from_numpy(g['ids'][:2, :1]), 6, return_margins=True)
    assert np.array_equal(ids.numpy(), g['ids'][:2, :7])
    assert np.abs(margins.numpy() - g['margins'][:2, :6]).max() <= 2e-5


def test_trainer_extras_moco_and_mlm():
    """Momentum distillation (loss, every gradient, the twin after the EMA update) and the MLM corruption of the decoder inputs
    against the reference (tests/golden/tiny_moco.npz, tiny_mlm.npz; tools/gen_goldens_r2.py trainer_extras)."""
    from conftest import load_golden
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    g = load_golden('tiny_moco.npz')
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    tok = fake_tokenizer(V)

    def state(seed):
        sd = {k: v.detach().clone() for k, v in det_init_(VisionEncoderDecoder(cfg), seed=seed).state_dict().items() if k != 'decoder.lm_head.weight'}
        sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
        return sd
    sd, sd_m = state(0), state(1)
    for v in sd.values():
        v.requires_grad_(True)
    loss = orc.lm_step_distill(sd, sd_m, cfg, torch.from_numpy(g['images']), torch.from_numpy(g['labels']), tok, alpha=0.4, temperature=1.3)
    loss.backward()
    assert abs(loss.item() - float(g['loss'])) <= 1e-5 * float(g['loss'])
    for k in g:
        if k.startswith('grad.') and k[5:] in sd:
            close(sd[k[5:]].grad, g[k], 2e-5 * max(1.0, float(np.abs(g[k]).max())))
        if k.startswith('ema.') and k[4:] in sd:
            close(orc.ema(sd_m[k[4:]], sd[k[4:]].detach(), 0.9), g[k], 1e-6)
    m = load_golden('tiny_mlm.npz')
    ids = orc.lm_inputs(torch.from_numpy(m['labels']), V - 1, V - 1, -100, mask_id=V - 2, mask_fraction=0.3, random_fraction=0.4,
                        u_mask=torch.from_numpy(m['u_mask']), u_rand=torch.from_numpy(m['u_rand']), r_ids=torch.from_numpy(m['r_ids']))
    assert np.array_equal(ids.numpy(), m['ids'])
    assert np.array_equal(orc.lm_inputs(torch.from_numpy(m['labels']), V - 1, V - 1).numpy(),
                          orc.shifted_inputs(torch.from_numpy(m['labels']), V - 1, V - 1)[0].numpy())


@pytest.mark.parametrize('tag,kw', [('prompt', {}), ('cross_only', dict(use_soft_prompting=False))])
def test_contrastive_loss_and_every_gradient(tag, kw):
    """add_contrastive_loss (wrapper.py:98-118,206-209): both loss terms and the gradient of every parameter vs the reference"""
    from conftest import load_golden
    g = load_golden('tiny_contrastive.npz')
    cfg = tiny_config(**kw)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    m = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in m.state_dict().items() if k != 'decoder.lm_head.weight'}
    sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
    images, labels = torch.from_numpy(g['images']), torch.from_numpy(g['labels'])
    loss, lm, lc = orc.lm_step(sd, cfg, images, labels, tok, training=True, contrastive_temperature=0.7, return_parts=True)
    loss.backward()
    assert abs(lm.item() - float(g[f'{tag}.loss_lm'])) <= 2e-5 and abs(lc.item() - float(g[f'{tag}.loss_contrastive'])) <= 2e-5
    assert abs(loss.item() - float(g[f'{tag}.loss'])) <= 3e-5
    for k, v in sd.items():
        if k == 'decoder.lm_head.weight':
            continue
        want = g[f'{tag}.grad.{k}']
        err = np.abs(v.grad.numpy() - want).max()
        assert err <= 1e-6 + 1e-3 * np.abs(want).max(), (k, err)
    with torch.no_grad():
        _, _, vc = orc.lm_step({k: v.detach() for k, v in sd.items()}, cfg, images, labels, tok, training=False, contrastive_temperature=0.7,
                               return_parts=True)
    assert abs(vc.item() - float(g[f'{tag}.val_loss_contrastive'])) <= 2e-5


BEAM_RUNS = {     # tools/gen_goldens_r3.py::gen_beam (the reference's BeamSearchTokenGenerator arguments of each recorded run)
    'det': dict(beam_width=3, temperature=0.0, top_k=None, max_new_tokens=12, no_repeat_n_grams=(2, 3, 4), beam_expansion_factor=4,
                eos_token_id='rare', consolidation_temperature=0.0, length_boost=1.0),
    'det_eos': dict(beam_width=3, temperature=0.0, top_k=5, max_new_tokens=14, no_repeat_n_grams=(2, 3), beam_expansion_factor=4,
                    eos_token_id='eos', consolidation_temperature=0.0, length_boost=1.5),
    'smp': dict(beam_width=3, temperature=2.5, top_k=None, max_new_tokens=10, no_repeat_n_grams=(2, 3, 4), beam_expansion_factor=4,
                eos_token_id='rare', consolidation_temperature=6.0, length_boost=1.0),
    'smp_eos': dict(beam_width=4, temperature=1.0, top_k=None, max_new_tokens=14, no_repeat_n_grams=(2, 3), beam_expansion_factor=3,
                    eos_token_id='eos', consolidation_temperature=1.0, length_boost=2.0),
}


def beam_replay(gold, tag):
    """-> draw(probs, n): hands back the reference's recorded torch.multinomial results call by call (rows in ITS order)."""
    state = {'i': 0}

    def draw(probs, n, *a, **k):
        r = torch.from_numpy(gold[f'{tag}.draw.{state["i"]}']).to(probs.device)
        state['i'] += 1
        assert r.shape == (probs.shape[0], n), (r.shape, probs.shape, n)
        assert bool((probs.gather(1, r) > 0).all()), 'a replayed draw has zero probability here'
        return r
    draw.state = state
    return draw


@pytest.mark.parametrize('tag', list(BEAM_RUNS))
def test_beam_search_replays_the_reference(tiny_weights, tag):
    from conftest import load_golden
    g = load_golden('tiny_beam.npz')
    kw = dict(BEAM_RUNS[tag])
    kw['eos_token_id'] = int(g[kw['eos_token_id']])
    draw = beam_replay(g, tag)
    ids, scores = orc.beam_search(tiny_weights, tiny_config(), torch.from_numpy(g['images']), torch.from_numpy(g['prompt']), draw=draw, **kw)
    assert draw.state['i'] == int(g[f'{tag}.n_draws'])
    assert np.array_equal(ids.numpy(), g[f'{tag}.ids'])
    close(scores, g[f'{tag}.scores'], 2e-4)
