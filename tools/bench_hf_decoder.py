#!/usr/bin/env python3
"""Train-step and greedy-decode throughput of a Hugging Face decoder behind the reference's plugin surface on one MI355X
(BASELINE.json configs[2]'s decoder side; reference training_configs/local/gpt2.yaml, local/qwen-1.5b-deepseek-distill.yaml without
the pieces this image cannot load: torchvision's pretrained ViT-B/16 -> the nano-224 from-scratch ViT; peft LoRA -> every parameter
trains).

    python tools/bench_hf_decoder.py [--size gpt2|gpt2-medium|qwen2-1.5b|llama2-7b|llama2-tiny|falcon-7b] [--lora] [--batch 1024] [--steps 6] [--warmup 2]
                                     [--decode-batch 1024] [--cpu]

There is no network: the 'checkpoint' is a randomly initialised model of the named shape written to a scratch directory and loaded
back through AutoModelForCausalLM.from_pretrained, exactly the path a real checkpoint takes (GPT-2: cross-attention layers added by
transformers, embeddings resized by extra_tokens = 2; soft prompt = the 64 encoder outputs in front of the text: one causal sequence
of 64 + 64 positions).  qwen2-1.5b is the shape of deepseek-ai/DeepSeek-R1-Distill-Qwen-1.5B (28 x 1536, 12 query heads of 128 on 2
K/V heads, SwiGLU 8960, vocabulary 151 936, untied head: 1.78 B parameters); llama2-7b is Llama-2-7B's shape (32 x 4096, 32 heads of
128, SwiGLU 11008: 6.74 B parameters -- fp32 master weights, gradients, bf16 shadow and Adam state are 121 GB of the GPU's 288).  Prints one JSON line; --cpu adds the reference
composition (oracle encoder + the transformers module, fp32, host cores) on a small batch.  Synthetic data.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

SIZES = {'gpt2': dict(n_layer=12, n_head=12, n_embd=768), 'gpt2-medium': dict(n_layer=24, n_head=16, n_embd=1024)}
LLAMA_SIZES = {      # name -> (checkpoint directory name satisfying decoder.py:120-127, config class, config, vocabulary)
    'qwen2-1.5b': ('Qwen2-1.5B-random', 'Qwen2Config', dict(hidden_size=1536, intermediate_size=8960, num_hidden_layers=28,
                                                             num_attention_heads=12, num_key_value_heads=2, max_position_embeddings=4096,
                                                             rms_norm_eps=1e-6, tie_word_embeddings=False), 151936),
    'llama2-7b': ('meta-llama/Llama-2-7b-random', 'LlamaConfig', dict(hidden_size=4096, intermediate_size=11008, num_hidden_layers=32,
                                                                      num_attention_heads=32, num_key_value_heads=32,
                                                                      max_position_embeddings=4096, rms_norm_eps=1e-5), 32000),
    'falcon-7b': ('tiiuae/falcon-7b-random', 'FalconConfig', dict(hidden_size=4544, num_hidden_layers=32, num_attention_heads=71, multi_query=True,
                                                                   parallel_attn=True, new_decoder_architecture=False, bias=False, alibi=False,
                                                                   max_position_embeddings=2048), 65024),
    'llama2-tiny': ('meta-llama/Llama-2-tiny-random', 'LlamaConfig', dict(hidden_size=1024, intermediate_size=2816, num_hidden_layers=8,
                                                                          num_attention_heads=8, num_key_value_heads=8,
                                                                          max_position_embeddings=4096, rms_norm_eps=1e-5), 32000),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--size', default='gpt2', choices=sorted(SIZES) + sorted(LLAMA_SIZES))
    ap.add_argument('--batch', type=int, default=1024)
    ap.add_argument('--steps', type=int, default=6)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--caption-len', type=int, default=64)
    ap.add_argument('--decode-batch', type=int, default=1024)
    ap.add_argument('--new-tokens', type=int, default=64)
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--cpu', action='store_true')
    ap.add_argument('--freeze-decoder', action='store_true', help='prepare_for_kbit_training: True without 4-bit loading (reference '
                    'local/llama2-7b.yaml): the decoder is frozen, only the encoder trains through the soft prompt')
    ap.add_argument('--vit', default='none', choices=['none', 'frozen', 'refine'], help="encoder = the reference's PretrainedViT (torchvision ViT-B/16 "
                    "shape, randomly initialised: the SWAG checkpoint is not in the image) with the slot-MLP head of local/gpt2.yaml (n_cls 16, "
                    "gate_sizes [1024], n_embd_out_vit 768): BASELINE.json configs[2].  frozen = refine_base_model: False")
    ap.add_argument('--fp8', action='store_true', help='I2T_FP8=1: e4m3 operands for the GEMMs of FROZEN decoder weights (use with --freeze-decoder or --lora; '
                    'BASELINE.json configs[4])')
    ap.add_argument('--gemm-breakdown', action='store_true', help='per-shape table of the bf16 GEMM launches of one step on stderr')
    ap.add_argument('--lora', action='store_true', help="GPT-2 sizes: the lora_spec of the reference's gpu/gpt2-xl.yaml (r 16, alpha 64, "
                    "dropout 0.1, c_attn / mlp.c_fc / mlp.c_proj, wpe / wte / crossattention / ln_cross_attn left trainable)")
    args = ap.parse_args()
    if args.fp8:
        os.environ['I2T_FP8'] = '1'
    from transformers import GPT2Config, GPT2LMHeadModel
    from image2text_amd import ops
    from image2text_amd.configs.models import HuggingfaceDecoderConfig
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import fake_tokenizer, nano224_config, synthetic_batch
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    dev = torch.device('cuda:0')
    scratch = tempfile.mkdtemp(prefix='i2t_gpt2_')
    os.chdir(scratch)
    torch.manual_seed(0)
    base = nano224_config(dropout=0.1)
    if args.vit != 'none':
        from image2text_amd.configs.models import PretrainedViTConfig
        os.environ.setdefault('I2T_VIT_B16_CHECKPOINT', 'random')
        base = base.model_copy(update=dict(vision_encoder_config=PretrainedViTConfig(
            n_cls=16, n_embd_out_vit=768, gate_sizes=(1024,), refine_base_model=args.vit == 'refine')))
    llama = args.size in LLAMA_SIZES
    if llama:
        import transformers
        name, cfg_cls, kw, vocab = LLAMA_SIZES[args.size]
        hf_cfg = getattr(transformers, cfg_cls)(vocab_size=vocab, **kw)
        big = kw['hidden_size'] * kw['num_hidden_layers'] >= 100000            # multi-billion-parameter shapes: bf16 on disk
        t_init = time.perf_counter()
        transformers.AutoModelForCausalLM.from_config(hf_cfg, dtype=torch.bfloat16 if big else torch.float32).save_pretrained(name)
        print(f'[bench_hf_decoder] random checkpoint written in {time.perf_counter() - t_init:.0f} s', file=sys.stderr, flush=True)
        lora = None
        if args.lora:            # the lora_spec of the reference's gpu/falcon-7b.yaml:55-60 / gpu/llama2-13b.yaml:35-39
            from image2text_amd.configs.models import LoraSpec
            if args.size.startswith('falcon'):
                lora = LoraSpec(r=16, lora_alpha=64, lora_dropout=0.1, target_modules=['query_key_value', 'dense', 'dense_h_to_4h', 'dense_4h_to_h'],
                                force_enable_update_modules=['*.word_embeddings.*', '*.lm_head.*'])
            else:
                lora = LoraSpec(r=16, lora_alpha=64, lora_dropout=0.1, target_modules=['q_proj', 'k_proj', 'v_proj', 'o_proj', 'up_proj', 'down_proj'])
        dcfg = HuggingfaceDecoderConfig(vocab_size=vocab, use_cross_attn=False, model_str=name, extra_tokens=0, load_in_4bit=False,
                                        prepare_for_kbit_training=args.freeze_decoder, lora_spec=lora)
        cfg = base.model_copy(update=dict(decoder_config=dcfg, use_cross_attn=False, use_soft_prompting=True))
        V, eos = vocab, vocab - 1
    else:
        name = args.size + '-random'                      # the reference dispatches on model_str.startswith('gpt2') (decoder.py:120)
        GPT2LMHeadModel(GPT2Config(vocab_size=50257, n_positions=1024, **SIZES[args.size])).save_pretrained(name)
        lora = None
        if args.lora:
            from image2text_amd.configs.models import LoraSpec
            lora = LoraSpec(r=16, lora_alpha=64, lora_dropout=0.1, target_modules=['c_attn', 'mlp.c_fc', 'mlp.c_proj'],
                            force_enable_update_modules=['*.wpe.*', '*.wte.*', '*.crossattention.*', '*.ln_cross_attn.*'])
        dcfg = HuggingfaceDecoderConfig(vocab_size=50257, use_cross_attn=True, model_str=name, extra_tokens=2, load_in_4bit=False,
                                        prepare_for_kbit_training=args.freeze_decoder, lora_spec=lora)
        cfg = base.model_copy(update=dict(decoder_config=dcfg, use_cross_attn=True, use_soft_prompting=True))
        V, eos = 50259, 50256
    tok = fake_tokenizer(V, eos=eos)
    t_init = time.perf_counter()
    wrapper = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100).to(dev).train()
    print(f'[bench_hf_decoder] model loaded and moved in {time.perf_counter() - t_init:.0f} s', file=sys.stderr, flush=True)
    eng = wrapper.model._engine
    n_params = sum(p.numel() for p in wrapper.model.parameters())
    n_train = sum(p.numel() for p in wrapper.model.parameters() if p.requires_grad)
    opt = FusedAdamW(wrapper.model.parameters(), wrapper.model, lr=6e-4, betas=(0.9, 0.95), weight_decay=0.0)
    images, labels = synthetic_batch(args.batch, 224, args.caption_len, V, seed=1, eos=eos)
    images, labels = images.to(dev), labels.to(dev)

    def step():
        loss, _ = wrapper.train_step(images, labels)
        loss.backward()
        opt.step()
        opt.zero_grad()
        return loss

    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / args.steps
    loss = loss.detach().clone()             # (the loss tensor's graph node holds the step's saved activations: let them go)
    kind = type(wrapper.model.decoder).__name__
    enc_name = 'nano-224 ViT (6x512, 224x224x3, 64 CLS)' if args.vit == 'none' else \
        f'PretrainedViT (ViT-B/16 shape 12x768, 224x224x3, random init, backbone {"trains" if args.vit == "refine" else "frozen"}, 16 slot MLPs 768-1024-768)'
    n_prompt = eng.enc.ncls
    out = {'workload': f'{enc_name} + {kind}({args.size}, randomly initialised checkpoint, '
                       + ('' if llama else 'cross-attention, dropout 0.1, ') +
                       f'soft prompt of {n_prompt} + {args.caption_len} text positions)' + (', LoRA r 16 (the lora_spec of the reference yaml of this decoder)' if args.lora else (', decoder frozen (prepare_for_kbit_training)' if args.freeze_decoder else ', every parameter trains')),
           'params_M': round(n_params / 1e6, 1), 'trainable_params_M': round(n_train / 1e6, 1), 'batch': args.batch, 'train_images_per_sec': round(args.batch / dt, 1),
           'ms_per_step': round(dt * 1e3, 2), 'final_loss': round(float(loss.detach()), 4),
           'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), 'dtype': 'bf16', 'data': 'synthetic',
           'decoder': {'prefixed': eng.dec.prefixed, 'grad_norm': eng.dec.grad_norm, 'layers': eng.dec.L, 'd': eng.dec.d, 'vocab': eng.dec.V}}
    from bench import GemmTimer
    f8rec, f8orig = [], ops.gemm_fp8

    def f8timed(a8, sa, b8, sb, o, M, N, K, **kw):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(torch.cuda.current_stream())
        r = f8orig(a8, sa, b8, sb, o, M, N, K, **kw)
        e1.record(torch.cuda.current_stream())
        f8rec.append((e0, e1, 2.0 * M * N * K))
        return r
    ops.gemm_fp8 = f8timed
    with GemmTimer(ops) as gt:
        step()
        gs = gt.summary()
        if args.gemm_breakdown:
            for line in gt.breakdown()[:24]:
                print('[bench_hf_decoder] ' + line, file=sys.stderr, flush=True)
    ops.gemm_fp8 = f8orig
    if f8rec:
        torch.cuda.synchronize()
        ms8 = sum(e0.elapsed_time(e1) for e0, e1, _ in f8rec)
        fl8 = sum(f for _, _, f in f8rec)
        out['fp8_gemm_family'] = {'launches': len(f8rec), 'ms_per_step': round(ms8, 2), 'tflops': round(fl8 / ms8 / 1e9, 1),
                                  'frac_of_5000_tflops': round(fl8 / ms8 / 1e9 / 5000.0, 3)}
    if args.fp8:
        out['fp8'] = 'e4m3 operands (block-scaled MFMA, per-row scales) on every frozen decoder GEMM, forward and dx'
    out['gemm_family'] = {'launches': gs['launches'], 'ms_per_step': round(gs['total_ms'], 2), 'tflops': round(gs['tflops'], 1),
                          'frac_of_2500_tflops': round(gs['tflops'] / 2500.0, 3)}
    if not args.no_decode:
        wrapper.eval()
        Bd = args.decode_batch
        dimgs = synthetic_batch(Bd, 224, args.caption_len, V, seed=7, eos=eos)[0].to(dev)
        prompt = torch.full((Bd, 1), tok.bos_token_id, dtype=torch.long, device=dev)
        with torch.no_grad():
            wrapper.model.generate(dimgs, prompt, max_new_tokens=args.new_tokens, temperature=1.0, top_k=1)      # capture
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ids = wrapper.model.generate(dimgs, prompt, max_new_tokens=args.new_tokens, temperature=1.0, top_k=1)
            torch.cuda.synchronize()
            dd = time.perf_counter() - t0
        assert ids.shape == (Bd, 1 + args.new_tokens)
        out['greedy_captions_per_sec'] = round(Bd / dd, 1)
        out['greedy'] = {'captions': Bd, 'new_tokens': args.new_tokens, 'seconds': round(dd, 3),
                         'includes': 'encoder forward + prompt-row prefill of the KV cache + hipGraph replay per token'}
    if args.cpu:
        import torch.nn.functional as F
        from oracle import reference_model as orc
        b = 4
        wrapper.train()
        sd = {k: v.detach().float().cpu().clone() for k, v in wrapper.model.state_dict().items()}
        if llama:
            hf = transformers.AutoModelForCausalLM.from_config(hf_cfg)
        else:
            hf = GPT2LMHeadModel(GPT2Config(vocab_size=V, n_positions=1024, add_cross_attention=True, **SIZES[args.size]))
        hf.load_state_dict({k[len('decoder.backbone.'):]: v for k, v in sd.items() if k.startswith('decoder.backbone.')}, strict=True)
        hf.train()
        esd = {k: v.requires_grad_(True) for k, v in sd.items() if not k.startswith('decoder.')}
        ci, cl = synthetic_batch(b, 224, args.caption_len, V, seed=1, eos=eos)
        ids, _ = orc.shifted_inputs(cl, tok.bos_token_id, tok.eos_token_id, -100)

        def cpu_step():
            enc = orc.encode(esd, cfg, ci, training=True)
            emb = torch.cat((enc, hf.get_input_embeddings()(ids)), dim=-2)
            logits = (hf(inputs_embeds=emb) if llama else hf(inputs_embeds=emb, encoder_hidden_states=enc)).logits[:, enc.shape[1]:]
            ce = F.cross_entropy(logits.reshape(-1, V), cl.reshape(-1), ignore_index=-100, reduction='none')
            (ce * orc.loss_weights(cl, -100).reshape(-1)).sum().backward()
        cpu_step()
        t0 = time.perf_counter()
        cpu_step()
        dc = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': round(b / dc, 3), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': f'1 forward+backward of {b} images, oracle encoder + the transformers module (fp32), no optimizer step'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
