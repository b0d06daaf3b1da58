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
