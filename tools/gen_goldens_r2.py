#!/usr/bin/env python3
"""Round-2 golden fixtures, produced by RUNNING THE REFERENCE (container-only tooling; same import recipe as
tools/gen_goldens.py, whose fixtures this script leaves untouched).

    python tools/gen_goldens_r2.py [snradam] [greedy64] [trunc] [sampling] [trainer_extras]      # default: all

  snradam.npz           reference models/optimizer.py::SNRAdam: 6 steps on 3 small tensors, two param groups with different
                        lr / weight decay -> parameters after every step
  nano224_greedy64.npz  nano-224 (det_init_ seed 0): generate(top_k=1) for 8 captions x 64 new tokens + the oracle's top-1
                        margin at every step (the benchmark's decode workload at parity size)
  tiny_trunc.npz        train_step with captions LONGER than the text window (block_size - n_cls): loss + every gradient
                        (the labels are truncated before the loss weights are normalised, wrapper.py:122-133)
  tiny_sampling.npz     generate() in its sampling modes on the trained tiny model, with torch.multinomial / torch.sort
                        wrapped by recorders: per step the filtered, renormalised distribution in vocabulary order that the
                        reference samples from (temperature / top-k / nucleus), teacher-forced on the recorded argmax
"""
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tools'))
sys.path.insert(0, REPO)
from gen_goldens import OUT, REF, greedy_with_margins, install_stubs, to_ref_config      # noqa: E402


def gen_snradam():
    from models.optimizer import SNRAdam as RefSNRAdam
    g = torch.Generator().manual_seed(5)
    shapes = [(7, 5), (12,), (3, 4, 2)]
    params = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    init = [p.detach().clone() for p in params]
    groups = [{'params': params[:2], 'lr': 3e-3, 'weight_decay': 0.1, 'betas': (0.9, 0.95)},
              {'params': params[2:], 'lr': 1e-3, 'weight_decay': 0.0, 'betas': (0.9, 0.95)}]
    opt = RefSNRAdam(groups)
    out = {f'init.{i}': p.numpy() for i, p in enumerate(init)}
    for step in range(6):
        for i, p in enumerate(params):
            # a consistent drift plus step-dependent noise: exercises both the mean and the variance track
            p.grad = 0.3 * torch.ones_like(p) + (0.5 + 0.2 * step) * torch.randn(p.shape, generator=g)
            out[f'grad.{step}.{i}'] = p.grad.numpy().copy()
        opt.step()
        for i, p in enumerate(params):
            out[f'param.{step}.{i}'] = p.detach().numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'snradam.npz'), **out)


def gen_greedy64():
    from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch
    from models.vision_encoder_decoder import VisionEncoderDecoder as RefVED
    cfg = nano224_config(dropout=0.0)
    V = cfg.decoder_config.vocab_size
    tok = fake_tokenizer(V)
    model = det_init_(RefVED(to_ref_config(cfg)), seed=0).eval()
    images, _ = synthetic_batch(8, 224, 64, V, seed=2)
    prompt = torch.full((8, 1), tok.bos_token_id, dtype=torch.long)
    t0 = time.time()
    with torch.no_grad():
        ref_ids = model.generate(images, prompt, max_new_tokens=64, temperature=1.0, top_k=1)
    print(f'nano224 generate(top_k=1) 8 x 64: {time.time() - t0:.1f}s')
    ids, margins = greedy_with_margins(model, images, prompt, 64)
    assert torch.equal(ref_ids, ids)
    print(f'margins: min {margins.min():.4f} median {np.median(margins):.4f}; below 0.05: {(margins < 0.05).sum()} of {margins.size}')
    np.savez_compressed(os.path.join(OUT, 'nano224_greedy64.npz'), ids=ids.numpy(), margins=margins)


def gen_trunc():
    from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch, tiny_config
    from configs.trainer import TrainerWrapperConfig as RefTrainerCfg
    from training.wrapper import ModelTrainerWrapper as RefWrapper
    cfg = tiny_config(dropout=0.0)                      # block 48, n_cls 8 -> text window 40
    V = cfg.decoder_config.vocab_size
    wrapper = RefWrapper(to_ref_config(cfg), fake_tokenizer(V), RefTrainerCfg(weight_fn='inverse_sqrt_position', eos_token_weight=2.0),
                         ignore_index=-100)
    det_init_(wrapper.model, seed=0)
    wrapper.train()
    images, labels = synthetic_batch(4, 32, 46, V, seed=13, min_len=36)      # lengths 36..45: captions run past position 40
    assert int((labels[:, 40:] != -100).sum()) > 0
    loss, _ = wrapper.train_step(images, labels)
    loss.backward()
    out = {'images': images.numpy(), 'labels': labels.numpy(), 'loss': np.float32(loss.item())}
    for n, p in wrapper.model.named_parameters():
        out[f'grad.{n}'] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'tiny_trunc.npz'), **out)
    print('trunc loss', loss.item())


def gen_sampling():
    from image2text_amd.synth import det_init_, fake_tokenizer, tiny_config
    from models.vision_encoder_decoder import VisionEncoderDecoder as RefVED
    cfg = tiny_config(dropout=0.0)
    V = cfg.decoder_config.vocab_size
    with np.load(os.path.join(OUT, 'tiny_decode.npz')) as z:
        images = torch.from_numpy(z['images'])
    prompt = torch.full((4, 1), fake_tokenizer(V).bos_token_id, dtype=torch.long)
    modes = {'t07_k5': dict(temperature=0.7, top_k=5), 't10_p05': dict(temperature=1.0, nucleus_p=0.5),
             't07_p06': dict(temperature=0.7, nucleus_p=0.6),             # trainer.py:50-54 eval_model's call
             't13_k20_p09': dict(temperature=1.3, top_k=20, nucleus_p=0.9), 't10_plain': dict(temperature=1.0),
             't20_p095': dict(temperature=2.0, nucleus_p=0.95)}
    out = {'images': images.numpy(), 'prompt': prompt.numpy()}
    real_multinomial, real_sort = torch.multinomial, torch.sort
    # two weight sets: the briefly trained tiny model (peaked distributions: nucleus keeps 1-2 tokens) and the untrained
    # det_init_ seed 0 model (flat distributions: nucleus keeps hundreds) -- the latter is regenerated in the test, not stored
    for wtag in ('trained', 'init'):
        model = RefVED(to_ref_config(cfg)).eval()
        if wtag == 'trained':
            with np.load(os.path.join(OUT, 'tiny_weights.npz')) as z:
                model.load_state_dict({k: torch.from_numpy(z[k]) for k in z.files})
        else:
            det_init_(model, seed=0)
        for tag, kw in modes.items():
            dists, last_sort = [], {}

            def rec_sort(x, *a, **k):
                r = real_sort(x, *a, **k)
                last_sort['idx'] = r.indices
                return r

            def rec_multinomial(p, num_samples=1, **k):
                if 'idx' in last_sort:                           # nucleus branch: p is in sorted order -> back to vocabulary order
                    full = torch.zeros_like(p).scatter_(1, last_sort.pop('idx'), p)
                else:
                    full = p.clone()
                dists.append((full / full.sum(-1, keepdim=True)).numpy().copy())
                # deterministic stand-in for the draw: the kept token at the MEDIAN of the kept CDF in vocabulary order (walks
                # through less likely tokens too, so the forced prefixes differ from greedy)
                cdf = torch.cumsum(full / full.sum(-1, keepdim=True), dim=-1)
                tok = (cdf < 0.5).sum(-1, keepdim=True).clamp(max=full.size(-1) - 1)
                return tok

            def gen_with(kw):
                # the reference indexes sorted order after a nucleus sort: translate the vocabulary-order token back
                def rec_multinomial_sorted(p, num_samples=1, **k):
                    idx = last_sort.get('idx')
                    tok = rec_multinomial(p, num_samples)
                    if idx is None:
                        return tok
                    return (idx == tok).int().argmax(dim=-1, keepdim=True)         # position of that token in sorted order
                torch.multinomial, torch.sort = rec_multinomial_sorted, rec_sort
                try:
                    with torch.no_grad():
                        return model.generate(images, prompt, max_new_tokens=10, **kw)
                finally:
                    torch.multinomial, torch.sort = real_multinomial, real_sort

            ids = gen_with(kw)
            out[f'{wtag}.{tag}.ids'] = ids.numpy()
            out[f'{wtag}.{tag}.dist'] = np.stack(dists, axis=1).astype(np.float32)          # (B, steps, V)
            kept = (out[f'{wtag}.{tag}.dist'] > 0).sum(-1)
            chosen = np.take_along_axis(out[f'{wtag}.{tag}.dist'], ids.numpy()[:, 1:, None], axis=-1)
            assert (chosen > 0).all(), 'forced token outside the kept set?'
            print(wtag, tag, 'kept tokens per step: min', kept.min(), 'max', kept.max())
    out['modes'] = np.array([f'{t}:{kw}' for t, kw in modes.items()])
    np.savez_compressed(os.path.join(OUT, 'tiny_sampling.npz'), **out)


def gen_trainer_extras():
    """tiny_moco.npz: train_step with momentum distillation (moco_momentum 0.9, moco_alpha 0.4, training_temperature 1.3), twin at
    DIFFERENT weights than the model (det_init_ seed 1 vs 0) so that the soft targets matter: loss, every gradient, and the twin's
    parameters after the step's EMA update.  tiny_mlm.npz: the decoder inputs the reference builds with mask_fraction 0.3 /
    random_mask_fraction 0.4 when torch.rand_like / randint_like are replaced by recorded draws (the draws are stored)."""
    from types import SimpleNamespace
    from image2text_amd.synth import det_init_, synthetic_batch, tiny_config
    from configs.trainer import TrainerWrapperConfig as RefTrainerCfg
    from training.wrapper import ModelTrainerWrapper as RefWrapper
    cfg = tiny_config(dropout=0.0)
    V = cfg.decoder_config.vocab_size
    tok = SimpleNamespace(eos_token_id=V - 1, bos_token_id=V - 1, mask_token_id=V - 2, vocab_size=V)
    images, labels = synthetic_batch(4, 32, 16, V, seed=31)
    # ---- momentum distillation
    w = RefWrapper(to_ref_config(cfg), tok, RefTrainerCfg(moco_momentum=0.9, moco_alpha=0.4, training_temperature=1.3), ignore_index=-100)
    det_init_(w.model, seed=0)
    det_init_(w.model_m, seed=1)
    w.train()
    pm_before = {n: p.detach().clone() for n, p in w.model_m.named_parameters()}
    loss, _ = w.train_step(images, labels)
    loss.backward()
    out = {'images': images.numpy(), 'labels': labels.numpy(), 'loss': np.float32(loss.item())}
    for n, p in w.model.named_parameters():
        out[f'grad.{n}'] = p.grad.numpy().copy()
    for n, p in w.model_m.named_parameters():
        out[f'ema.{n}'] = p.detach().numpy().copy()
        assert torch.allclose(p.detach(), 0.9 * pm_before[n] + 0.1 * dict(w.model.named_parameters())[n].detach(), atol=1e-7)
    w.eval()
    with torch.no_grad():
        vloss, _ = w.val_step(images, labels)               # validation: plain CE, no distillation
    out['val_loss'] = np.float32(vloss.item())
    np.savez_compressed(os.path.join(OUT, 'tiny_moco.npz'), **out)
    print('moco loss', loss.item(), 'val', vloss.item())
    # ---- MLM corruption: record the decoder inputs
    w2 = RefWrapper(to_ref_config(cfg), tok, RefTrainerCfg(mask_fraction=0.3, random_mask_fraction=0.4), ignore_index=-100)
    det_init_(w2.model, seed=0)
    w2.train()
    g = torch.Generator().manual_seed(77)
    u_rand = torch.rand(labels.shape, generator=g)          # first rand_like call: random-vs-mask choice (wrapper.py:165)
    u_mask = torch.rand(labels.shape, generator=g)          # second: which tokens are corrupted (wrapper.py:172)
    r_ids = torch.randint(0, V, labels.shape, generator=g)
    seen = {}
    real_rand_like, real_randint_like, real_forward = torch.rand_like, torch.randint_like, w2.forward
    draws = [u_rand, u_mask]
    torch.rand_like = lambda x, **k: draws.pop(0).to(k.get('dtype', torch.float))
    torch.randint_like = lambda x, low=0, high=None, **k: r_ids.clone()

    def spy(images, input_ids, attn_msk=None):
        seen['ids'] = input_ids.clone()
        return real_forward(images, input_ids, attn_msk)
    w2.forward = spy
    try:
        loss2, _ = w2.train_step(images, labels)
    finally:
        torch.rand_like, torch.randint_like = real_rand_like, real_randint_like
    np.savez_compressed(os.path.join(OUT, 'tiny_mlm.npz'), labels=labels.numpy(), u_rand=u_rand.numpy(), u_mask=u_mask.numpy(),
                        r_ids=r_ids.numpy(), ids=seen['ids'].numpy(), loss=np.float32(loss2.item()), images=images.numpy())
    print('mlm corrupted positions', int((seen['ids'][:, 1:] != torch.where(labels != -100, labels, torch.full_like(labels, V - 1))[:, :-1]).sum()))


def gen_contrastive():
    """tiny_contrastive.npz: train_step with add_contrastive_loss (training_contrastive_temperature 0.7) on the tiny model with soft
    prompt + cross-attention, and on its cross-attention-only variant (no prompt rows in hidden_state): the two loss terms and the
    gradient of every parameter (the contrastive term differentiates the PROMPT rows of hidden_state and the target embeddings)."""
    from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch, tiny_config
    from configs.trainer import TrainerWrapperConfig as RefTrainerCfg
    from training.wrapper import ModelTrainerWrapper as RefWrapper
    out = {}
    for tag, kw in (('prompt', {}), ('cross_only', dict(use_soft_prompting=False))):
        cfg = tiny_config(dropout=0.0, **kw)
        V = cfg.decoder_config.vocab_size
        tok = fake_tokenizer(V)
        images, labels = synthetic_batch(4, 32, 16, V, seed=5)
        w = RefWrapper(to_ref_config(cfg), tok, RefTrainerCfg(add_contrastive_loss=True, training_contrastive_temperature=0.7), ignore_index=-100)
        det_init_(w.model, seed=0)
        w.train()
        loss, metrics = w.train_step(images, labels)
        loss.backward()
        out.update({'images': images.numpy(), 'labels': labels.numpy(), f'{tag}.loss': np.float32(loss.item()),
                    f'{tag}.loss_lm': np.float32(metrics['train_loss_lm'].item()),
                    f'{tag}.loss_contrastive': np.float32(metrics['train_loss_contrastive'].item())})
        for n, p in w.model.named_parameters():
            out[f'{tag}.grad.{n}'] = p.grad.numpy().copy()
        w.eval()
        with torch.no_grad():
            vloss, vm = w.val_step(images, labels)
        out[f'{tag}.val_loss'], out[f'{tag}.val_loss_contrastive'] = np.float32(vloss.item()), np.float32(vm['val_loss_contrastive'].item())
        print(tag, 'loss', loss.item(), 'lm', metrics['train_loss_lm'].item(), 'contrastive', metrics['train_loss_contrastive'].item())
    np.savez_compressed(os.path.join(OUT, 'tiny_contrastive.npz'), **out)


def main():
    install_stubs()
    sys.path.insert(0, REF)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    want = sys.argv[1:] or ['snradam', 'greedy64', 'trunc', 'sampling', 'trainer_extras', 'contrastive']
    for name in want:
        t0 = time.time()
        globals()[f'gen_{name}']()
        print(f'{name}: done in {time.time() - t0:.1f}s')


if __name__ == '__main__':
    main()
