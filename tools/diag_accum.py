#!/usr/bin/env python3
"""Diagnostic: how far two backward passes of the SAME data differ, and where the difference enters (tiny model).

    python tools/diag_accum.py [plain | gpt2_lora | poison]

Finding (round 2): passes usually agree to 1e-7 (fp32 atomics order in the gradient normaliser's sums and the dW slices), but now and
then one differs by 3e-4 .. 1e-3 in the gradients at the bottom of a tower (convolutions, embeddings).  The per-entry analysis shows the
perturbation entering where the normalised gradient is re-quantised to bf16 and growing by sqrt(eps x 2^-8) at every further bf16 stage
(1e-7 -> 1e-5 -> 2e-4 -> 1e-3); poisoning the allocator's free blocks with NaN / 1e4 changes nothing, i.e. no kernel reads memory it
did not write.  bf16 intermediates make the backward pass chaotic at the 1e-3 level; nothing to fix, but bit-level or 1e-5-level
reproducibility checks between passes must not be asserted."""
import os, sys, tempfile
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    from image2text_amd import ops
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch, tiny_config
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    dev = torch.device('cuda:0')
    which = sys.argv[1] if len(sys.argv) > 1 else 'plain'
    if which == 'gpt2_lora':
        from transformers import GPT2Config, GPT2LMHeadModel
        from image2text_amd.configs.models import HuggingfaceDecoderConfig, LoraSpec
        os.chdir(tempfile.mkdtemp(prefix='i2t_dp_'))
        torch.manual_seed(0)
        GPT2LMHeadModel(GPT2Config(n_layer=2, n_head=4, n_embd=256, n_positions=128, vocab_size=1000, resid_pdrop=0.0, embd_pdrop=0.0,
                                   attn_pdrop=0.0)).save_pretrained('gpt2-dp')
        lora = LoraSpec(r=8, lora_alpha=16, lora_dropout=0.0, target_modules=['c_attn', 'mlp.c_fc', 'mlp.c_proj'],
                        force_enable_update_modules=['*.wte.*', '*.crossattention.*'])
        dcfg = HuggingfaceDecoderConfig(vocab_size=1000, use_cross_attn=True, model_str='gpt2-dp', extra_tokens=0, load_in_4bit=False,
                                        prepare_for_kbit_training=False, lora_spec=lora)
        cfg = tiny_config(dec_d=256, dec_heads=4).model_copy(update=dict(decoder_config=dcfg, use_cross_attn=True, use_soft_prompting=True))
        V = 1000
    else:
        cfg = tiny_config()
        V = cfg.decoder_config.vocab_size
    w = ModelTrainerWrapper(cfg, fake_tokenizer(V), TrainerWrapperConfig(), ignore_index=-100)
    if which != 'gpt2_lora':
        det_init_(w.model, seed=3)
    w = w.to(dev).train()
    images, labels = synthetic_batch(16, 32, 24, V, seed=5)
    images, labels = images.to(dev), labels.to(dev)
    eng = w.model._engine

    def accum(reserve_second=False, whole=False):
        for p in w.model.parameters():
            p.grad = None
        if whole:
            w.train_step(images, labels)[0].backward()
        else:
            w.train_step(images[:8], labels[:8])[0].backward()
            if reserve_second:
                ops.gemm_reserve_cus(16)
            w.train_step(images[8:], labels[8:])[0].backward()
            ops.gemm_reserve_cus(0)
        torch.cuda.synchronize()
        return eng.arena.g32.clone()

    def scales(x, y):
        for name, (o, n, _) in eng.arena.entries.items():
            u, v = x[o:o + n].double(), y[o:o + n].double()
            if float(v.abs().max()) == 0:
                continue
            ratio = float((u * v).sum() / (v * v).sum())
            resid = float((u - ratio * v).norm() / v.norm())
            if abs(ratio - 1) > 1e-5 or resid > 1e-5:
                print(f'   {name:58s} scale {ratio - 1:+.3e}  residual after scaling {resid:.2e}', flush=True)

    def cmp(tag, x, y):
        worst = sorted(((float((x[o:o + n] - y[o:o + n]).abs().max()), name) for name, (o, n, _) in eng.arena.entries.items()), reverse=True)[:3]
        print(tag, 'rel', float((x - y).abs().max()) / float(y.abs().max()), worst, flush=True)

    a0 = accum()
    a1 = accum()
    a2 = accum(reserve_second=True)
    a3 = accum()
    cmp('plain vs plain      ', a1, a0)
    more = [accum() for _ in range(6)]
    for i, m_ in enumerate([a1, a2, a3] + more):
        if float((m_ - a0).abs().max()) / float(a0.abs().max()) > 1e-5:
            print('pass', i + 1, 'differs from pass 0:', flush=True)
            scales(m_, a0)
            break
    cmp('reserved vs plain   ', a2, a0)
    cmp('plain again vs plain', a3, a0)
    b0, b1 = accum(whole=True), accum(whole=True)
    cmp('whole vs whole      ', b1, b0)
    cmp('first accum vs whole', a0, b0)
    cmp('later accum vs whole', a1, b0)
    c0 = accum()
    cmp('accum after whole   ', c0, b0)


if __name__ == '__main__' and not (len(sys.argv) > 1 and sys.argv[1] == 'poison'):
    main()


def poison():
    """run one whole-batch pass on NaN-poisoned allocator blocks and report which gradients pick up NaNs"""
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import det_init_, fake_tokenizer, synthetic_batch, tiny_config
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    dev = torch.device('cuda:0')
    cfg = tiny_config()
    V = cfg.decoder_config.vocab_size
    w = ModelTrainerWrapper(cfg, fake_tokenizer(V), TrainerWrapperConfig(), ignore_index=-100)
    det_init_(w.model, seed=3)
    w = w.to(dev).train()
    images, labels = synthetic_batch(16, 32, 24, V, seed=5)
    images, labels = images.to(dev), labels.to(dev)
    eng = w.model._engine
    w.train_step(images, labels)[0].backward()          # builds the arena and the persistent buffers
    for p in w.model.parameters():
        p.grad = None
    torch.cuda.synchronize()
    for val in (float('nan'), 1e4):
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 26,), val, device=dev) for _ in range(4)]          # 1 GiB of poison, then back to the allocator
        del junk
        loss = w.train_step(images, labels)[0]
        loss.backward()
        torch.cuda.synchronize()
        g = eng.arena.g32
        bad = [name for name, (o, n, _) in eng.arena.entries.items() if not torch.isfinite(g[o:o + n]).all()]
        print('poison', val, 'loss', float(loss), 'non-finite grads in', len(bad), 'entries:', bad[:12], flush=True)
        ref = g.clone()
        for p in w.model.parameters():
            p.grad = None
        if val != val:
            continue
        torch.cuda.empty_cache()
        junk = [torch.zeros((1 << 26,), device=dev) for _ in range(4)]
        del junk
        w.train_step(images, labels)[0].backward()
        torch.cuda.synchronize()
        worst = sorted(((float((g[o:o + n] - ref[o:o + n]).abs().max()), name) for name, (o, n, _) in eng.arena.entries.items()), reverse=True)[:6]
        print('1e4-poisoned vs zero-poisoned:', worst, flush=True)
        for name, (o, n, _) in eng.arena.entries.items():
            x, y = g[o:o + n].double(), ref[o:o + n].double()
            if float(y.abs().max()) == 0:
                continue
            ratio = float((x * y).sum() / (y * y).sum())
            resid = float((x - ratio * y).norm() / y.norm())
            if abs(ratio - 1) > 1e-5 or resid > 1e-5:
                print(f'   {name:60s} scale {ratio - 1:+.3e}  residual after scaling {resid:.2e}', flush=True)
        for p in w.model.parameters():
            p.grad = None


if __name__ == '__main__' and len(sys.argv) > 1 and sys.argv[1] == 'poison':
    poison()
