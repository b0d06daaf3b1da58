#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE (container-only tooling).

Runs only where /root/reference exists (this container).  It imports the reference's own Python modules from
there -- nothing of the reference is copied into the repo -- after registering name-only stand-ins for three
third-party packages that are absent from the image and never executed on the from-scratch path (peft, torchvision,
smart_open).  What gets committed is data only: inputs, weights and the reference's outputs.

    python tools/gen_goldens.py            # writes tests/golden/*.npz

Fixtures
  tiny_weights.npz   state_dict of the reference tiny model after a short CPU training run on a synthetic
                     class->caption task (gives the greedy decode real top-1 margins)
  tiny_forward.npz   images/ids/masks -> encoder_output, logits, hidden_state (+ per-stage intermediates via hooks)
  tiny_train.npz     labels -> ModelTrainerWrapper.train_step loss and the gradient of every parameter
  tiny_decode.npz    generate(top_k=1) ids (n-grams 2,3,4,5) + the oracle's top-1 margin at every step
  nano224.npz        full-size nano-224 (det_init_ seed 0 weights are regenerated, not stored): logits slice,
                     row log-sum-exp / argmax, loss, 12 greedy steps + margins for B=2
"""
import importlib.machinery
import os
import sys
import time
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = '/root/reference'
OUT = os.path.join(REPO, 'tests', 'golden')


def install_stubs():
    # transformers must resolve its own optional imports before a torchvision stand-in appears
    import transformers  # noqa: F401
    from transformers import (AutoModelForCausalLM, BitsAndBytesConfig, GPT2LMHeadModel,  # noqa: F401
                              LlamaForCausalLM, LogitsProcessorList, NoRepeatNGramLogitsProcessor,
                              PreTrainedModel, PreTrainedTokenizer, Qwen2ForCausalLM)

    def mod(name):
        m = types.ModuleType(name)
        m.__spec__ = importlib.machinery.ModuleSpec(name, None)
        sys.modules[name] = m
        return m

    peft = mod('peft')
    for n in ('TaskType', 'LoraModel', 'LoraConfig', 'prepare_model_for_kbit_training'):
        setattr(peft, n, type(n, (), {'FEATURE_EXTRACTION': 0, 'CAUSAL_LM': 1}))
    tuners = mod('peft.tuners')
    tuners.LoraModel = peft.LoraModel
    peft.tuners = tuners
    tv = mod('torchvision')
    tvm = mod('torchvision.models')
    tvm.vit_b_16 = None
    tvm.ViT_B_16_Weights = None
    tv.models = tvm
    so = mod('smart_open')
    so.open = open


def to_ref_config(cfg):
    from configs.models import VisionEncoderDecoderConfig as RefCfg
    return RefCfg.parse_obj(cfg.model_dump(mode='json'))


def sd_numpy(module):
    return {k: v.detach().cpu().numpy().copy() for k, v in module.state_dict().items()}


def class_task(n_classes, img, cap_len, vocab, seed=7):
    """Synthetic captioning task: a fixed template image and a fixed caption per class."""
    g = torch.Generator().manual_seed(seed)
    templates = torch.randn(n_classes, 3, img, img, generator=g)
    eos = vocab - 1
    caps = torch.full((n_classes, cap_len), -100, dtype=torch.long)
    for c in range(n_classes):
        n = int(torch.randint(cap_len - 5, cap_len - 1, (1,), generator=g))
        caps[c, :n] = torch.randperm(eos, generator=g)[:n]      # distinct tokens: no n-gram ban inside a caption
        caps[c, n] = eos
    return templates, caps


def greedy_with_margins(model, images, prompt, steps):
    """Reference generate(top_k=1) semantics re-run step by step to also record the top-1 margin after the ban."""
    enc = None
    ids = prompt
    margins = []
    with torch.no_grad():
        for _ in range(steps):
            out = model(images=images, ids=ids, encoder_output=enc)
            enc = out.encoder_output
            logits = model.processor(ids, out.logits[..., -1, :].clone())
            top2 = torch.topk(logits, 2, dim=-1).values
            margins.append((top2[:, 0] - top2[:, 1]).numpy().copy())
            ids = torch.cat((ids, logits.argmax(dim=-1, keepdim=True)), dim=-1)
    return ids, np.stack(margins, axis=1)


def main():
    sys.path.insert(0, REPO)
    from image2text_amd.synth import det_init_, fake_tokenizer, nano224_config, synthetic_batch, tiny_config
    install_stubs()
    sys.path.insert(0, REF)
    from configs.trainer import TrainerWrapperConfig as RefTrainerCfg
    from models.vision_encoder_decoder import VisionEncoderDecoder as RefVED
    from training.wrapper import ModelTrainerWrapper as RefWrapper

    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)

    # ------------------------------------------------------------------ tiny: train briefly for decode margins
    cfg = tiny_config(dropout=0.0)
    rcfg = to_ref_config(cfg)
    vocab = cfg.decoder_config.vocab_size
    T = 16
    tok = fake_tokenizer(vocab)
    wrapper = RefWrapper(rcfg, tok, RefTrainerCfg(), ignore_index=-100)
    det_init_(wrapper.model, seed=0)
    templates, caps = class_task(6, 32, T, vocab)
    # well-conditioned gradient fixture: untrained det_init_ weights (regenerated in the test, not stored)
    g0 = torch.Generator().manual_seed(11)
    cls0 = torch.tensor([0, 3, 5, 1])
    images0 = templates[cls0] + 0.1 * torch.randn(4, 3, 32, 32, generator=g0)
    wrapper.train()
    loss0, _ = wrapper.train_step(images0, caps[cls0])
    loss0.backward()
    init = {'loss': np.float32(loss0.item()), 'images': images0.numpy(), 'labels': caps[cls0].numpy()}
    for n, p in wrapper.model.named_parameters():
        init[f'grad.{n}'] = p.grad.numpy().copy()
    np.savez_compressed(os.path.join(OUT, 'tiny_train_init.npz'), **init)
    wrapper.zero_grad()
    opt = torch.optim.AdamW(wrapper.parameters(), lr=2e-3, betas=(0.9, 0.95))
    g = torch.Generator().manual_seed(3)
    wrapper.train()
    t0 = time.time()
    for step in range(500):
        cls = torch.randint(0, 6, (16,), generator=g)
        imgs = templates[cls] + 0.1 * torch.randn(16, 3, 32, 32, generator=g)
        loss, _ = wrapper.train_step(imgs, caps[cls])
        loss.backward()
        opt.step()
        opt.zero_grad()
        if step % 100 == 0 or step == 499:
            print(f'tiny train step {step} loss {loss.item():.4f} ({time.time() - t0:.1f}s)')
    wrapper.eval()
    model = wrapper.model
    np.savez_compressed(os.path.join(OUT, 'tiny_weights.npz'), **sd_numpy(model))

    # ------------------------------------------------------------------ tiny forward (+ intermediates)
    g = torch.Generator().manual_seed(11)
    cls = torch.tensor([0, 3, 5, 1])
    images = templates[cls] + 0.1 * torch.randn(4, 3, 32, 32, generator=g)
    labels = caps[cls].clone()
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    row_mask = torch.cat((torch.ones(4, 1, dtype=torch.bool), (labels != -100)[:, :-1]), dim=1)
    sl_mask = torch.randint(0, 2, (T, T), generator=g).bool()
    sl_mask[5, :] = False                                   # a fully masked query row (torch>=2.5 SDPA -> zeros)
    bsl_mask = torch.randint(0, 2, (4, T, T), generator=g).bool()
    fwd = {'images': images.numpy(), 'ids': ids.numpy(), 'labels': labels.numpy(), 'row_mask': row_mask.numpy(),
           'sl_mask': sl_mask.numpy(), 'bsl_mask': bsl_mask.numpy()}
    inter = {}
    hooks = []

    def grab(name):
        def fn(_m, _i, o):
            inter[name] = (o[0] if isinstance(o, tuple) else o).detach().numpy().copy()
        return fn

    enc0 = model.encoder[0]
    hooks.append(enc0.feature_extractor.register_forward_hook(grab('enc.conv')))
    hooks.append(enc0.projector.register_forward_hook(grab('enc.projector')))
    for i, blk in enumerate(enc0.transformer.h):
        hooks.append(blk.register_forward_hook(grab(f'enc.h{i}')))
    hooks.append(enc0.register_forward_hook(grab('enc.out')))
    for i, blk in enumerate(model.decoder.transformer.h):
        hooks.append(blk.register_forward_hook(grab(f'dec.h{i}')))
        hooks.append(blk.attn.register_forward_hook(grab(f'dec.h{i}.attn')))
        if blk.is_cross_attn:
            hooks.append(blk.cross_attn.register_forward_hook(grab(f'dec.h{i}.cross')))
        hooks.append(blk.mlp.register_forward_hook(grab(f'dec.h{i}.mlp')))
    with torch.no_grad():
        out = model(images=images, ids=ids, attn_msk=None)
    for h in hooks:
        h.remove()
    fwd.update({f'inter.{k}': v for k, v in inter.items()})
    for tag, m in (('nomask', None), ('row_mask', row_mask), ('sl_mask', sl_mask), ('bsl_mask', bsl_mask)):
        with torch.no_grad():
            out = model(images=images, ids=ids, attn_msk=m)
        fwd[f'{tag}.encoder_output'] = out.encoder_output.numpy()
        fwd[f'{tag}.logits'] = out.logits.numpy()
        fwd[f'{tag}.hidden_state'] = out.hidden_state.numpy()
    # cross-attention only / soft prompt only variants of the plugin surface (same weights)
    for tag, kw in (('cross_only', dict(use_soft_prompting=False)), ('prompt_only', dict(use_cross_attn=False))):
        c2 = tiny_config(dropout=0.0, **kw)
        m2 = RefVED(to_ref_config(c2)).eval()
        m2.load_state_dict(model.state_dict())
        with torch.no_grad():
            out = m2(images=images, ids=ids, attn_msk=row_mask)
        fwd[f'{tag}.logits'] = out.logits.numpy()
        fwd[f'{tag}.hidden_state'] = out.hidden_state.numpy()
    np.savez_compressed(os.path.join(OUT, 'tiny_forward.npz'), **fwd)

    # ------------------------------------------------------------------ tiny train_step: loss + every grad
    wrapper.train()                                          # dropout 0 => deterministic
    wrapper.zero_grad()
    loss, metrics = wrapper.train_step(images, labels)
    loss.backward()
    tr = {'loss': np.float32(loss.item()), 'train_loss_lm': np.float32(metrics['train_loss_lm'].item())}
    for n, p in model.named_parameters():
        tr[f'grad.{n}'] = p.grad.numpy().copy()
    wrapper.eval()
    with torch.no_grad():
        vloss, _ = wrapper.val_step(images, labels)
    tr['val_loss'] = np.float32(vloss.item())
    np.savez_compressed(os.path.join(OUT, 'tiny_train.npz'), **tr)
    wrapper.zero_grad()

    # ------------------------------------------------------------------ tiny greedy decode
    prompt = torch.full((4, 1), tok.bos_token_id, dtype=torch.long)
    steps = 24
    with torch.no_grad():
        ref_ids = model.generate(images, prompt, max_new_tokens=steps, temperature=1.0, top_k=1)
    my_ids, margins = greedy_with_margins(model, images, prompt, steps)
    assert torch.equal(ref_ids, my_ids), 'generate(top_k=1) is not argmax-after-ban?'
    prompt3 = ids[:, :3].contiguous()                        # a longer prompt: exercises the n-gram ban on prompt ids
    with torch.no_grad():
        ref_ids3 = model.generate(images, prompt3, max_new_tokens=12, temperature=1.0, top_k=1)
    _, margins3 = greedy_with_margins(model, images, prompt3, 12)
    print('tiny greedy:', ref_ids[0].tolist(), 'caption:', caps[0].tolist())
    print('tiny margins min/median:', margins.min(), np.median(margins))
    np.savez_compressed(os.path.join(OUT, 'tiny_decode.npz'), images=images.numpy(), prompt=prompt.numpy(),
                        ids=ref_ids.numpy(), margins=margins, prompt3=prompt3.numpy(), ids3=ref_ids3.numpy(),
                        margins3=margins3)

    # ------------------------------------------------------------------ nano-224 full size (weights regenerated)
    cfg = nano224_config(dropout=0.0)
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    wrapper = RefWrapper(to_ref_config(cfg), tok, RefTrainerCfg(), ignore_index=-100)
    det_init_(wrapper.model, seed=0)
    wrapper.eval()
    model = wrapper.model
    images, labels = synthetic_batch(2, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    ids = torch.where(labels != -100, labels, torch.full_like(labels, tok.eos_token_id))
    row_mask = torch.cat((torch.ones(2, 1, dtype=torch.bool), (labels != -100)[:, :-1]), dim=1)
    bos_ids = torch.cat((torch.full((2, 1), tok.bos_token_id, dtype=torch.long), ids), dim=1)[:, :64]
    t0 = time.time()
    with torch.no_grad():
        out = model(images=images, ids=bos_ids, attn_msk=row_mask)
        vloss, _ = wrapper.val_step(images, labels)
    print(f'nano224 forward {time.time() - t0:.1f}s  val_loss {vloss.item():.5f}')
    logits = out.logits
    nano = {
        'labels': labels.numpy(),
        'encoder_output': out.encoder_output.numpy(),
        'logits_head': logits[:, :, :256].numpy().copy(),
        'logits_tail': logits[:, :, -64:].numpy().copy(),
        'logits_lse': torch.logsumexp(logits, dim=-1).numpy(),
        'logits_argmax': logits.argmax(dim=-1).numpy(),
        'logits_absmax': np.float32(logits.abs().max().item()),
        'hidden_text': out.hidden_state[:, 64:, :].numpy().copy(),
        'val_loss': np.float32(vloss.item()),
    }
    prompt = torch.full((2, 1), tok.bos_token_id, dtype=torch.long)
    t0 = time.time()
    gids, gm = greedy_with_margins(model, images, prompt, 12)
    print(f'nano224 greedy 12 steps {time.time() - t0:.1f}s  margins min {gm.min():.4f} med {np.median(gm):.4f}')
    nano['greedy_ids'] = gids.numpy()
    nano['greedy_margins'] = gm
    # train-step loss + a few gradients (full grads would be 650 MB)
    wrapper.train()
    loss, _ = wrapper.train_step(images, labels)
    loss.backward()
    nano['train_loss'] = np.float32(loss.item())
    named = dict(model.named_parameters())
    for n in ('decoder.transformer.h.11.mlp.c_proj.bias', 'decoder.transformer.h.0.ln_3.weight',
              'decoder.transformer.h.0.cross_attn.in_proj_bias', 'decoder.transformer.ln_f.weight',
              'encoder.0.transformer.ln_f.weight', 'encoder.0.feature_extractor.model.0.bias',
              'encoder.0.feature_extractor.model.4.weight', 'encoder.0.cls_token',
              'decoder.transformer.h.6.attn.c_attn.bias', 'encoder.0.transformer.h.5.ln_2.weight'):
        nano[f'grad.{n}'] = named[n].grad.numpy().copy()
    for n, p in named.items():
        nano[f'gradnorm.{n}'] = np.float32(p.grad.norm().item())
    np.savez_compressed(os.path.join(OUT, 'nano224.npz'), **nano)

    # ------------------------------------------------------------------ nano-224 with the reference's own init
    # distributions (det_init_ style='reference'): the logit scale the north-star tolerance (1e-2) is stated for
    wrapper = RefWrapper(to_ref_config(cfg), tok, RefTrainerCfg(), ignore_index=-100)
    det_init_(wrapper.model, seed=0, style='reference')
    wrapper.eval()
    with torch.no_grad():
        out = wrapper.model(images=images, ids=bos_ids, attn_msk=row_mask)
        vloss, _ = wrapper.val_step(images, labels)
    logits = out.logits
    # how far does the REFERENCE ITSELF move when run the way trainer.py runs it with precision 'bf16'
    # (accelerate autocast; here torch.autocast on the CPU)?  Recorded to calibrate the bf16 tolerance.
    with torch.no_grad(), torch.autocast('cpu', dtype=torch.bfloat16):
        out_bf = wrapper.model(images=images, ids=bos_ids, attn_msk=row_mask)
    dev_bf = (out_bf.logits.float() - logits).abs()
    print(f'reference under bf16 autocast vs its fp32 run: logits max abs dev {dev_bf.max().item():.5f} '
          f'rms {dev_bf.pow(2).mean().sqrt().item():.5f} frac<=1e-2 {(dev_bf <= 1e-2).float().mean().item():.4f}')
    autocast_stats = np.array([dev_bf.max().item(), dev_bf.pow(2).mean().sqrt().item(), (dev_bf <= 1e-2).float().mean().item()],
                              dtype=np.float32)
    gids, gm = greedy_with_margins(wrapper.model, images, prompt, 8)
    print(f'nano224 refinit val_loss {vloss.item():.5f} logits absmax {logits.abs().max().item():.3f} '
          f'margins min {gm.min():.5f} med {np.median(gm):.5f}')
    np.savez_compressed(os.path.join(OUT, 'nano224_refinit.npz'),
                        encoder_output=out.encoder_output.numpy(), logits_head=logits[:, :, :256].numpy().copy(),
                        logits_tail=logits[:, :, -64:].numpy().copy(), logits_lse=torch.logsumexp(logits, dim=-1).numpy(),
                        logits_absmax=np.float32(logits.abs().max().item()), hidden_text=out.hidden_state[:, 64:, :].numpy().copy(),
                        val_loss=np.float32(vloss.item()), greedy_ids=gids.numpy(), greedy_margins=gm,
                        reference_bf16_autocast_dev=autocast_stats)
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
