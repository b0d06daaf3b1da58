#!/usr/bin/env python3
"""Train-step and greedy-decode throughput of the shipped nano-mini configuration (reference training_configs/gpu/nano-mini.yaml:
12x1024 ViT over 128x128 images + 12x1024 decoder; multi-query attention, MoE rotators, sparse token subsets) on one MI355X.

    python tools/bench_nano_mini.py [--batch 512] [--steps 8] [--warmup 3] [--decode-batch 1024] [--new-tokens 64] [--cpu]

Prints one JSON line: images/s of the full train step (forward + backward + fused AdamW, dropout 0.1 as in the yaml), captions/s of
generate(top_k=1) with the KV cache, the per-kernel-family time split of one step (HIP events around every C-ABI call family), and
with --cpu the oracle's train step on the host cores as the baseline (small batch).  Synthetic data, random-init weights.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--batch', type=int, default=512)
    ap.add_argument('--steps', type=int, default=8)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--caption-len', type=int, default=64)
    ap.add_argument('--decode-batch', type=int, default=1024)
    ap.add_argument('--new-tokens', type=int, default=64)
    ap.add_argument('--cpu', action='store_true')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--gemm-breakdown', action='store_true', help='per-shape GEMM table of one step on stderr')
    ap.add_argument('--split', action='store_true', help='time every ops.* call family of one step with HIP events')
    args = ap.parse_args()
    from image2text_amd import ops
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import fake_tokenizer, nano_mini_config, synthetic_batch
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.wrapper import ModelTrainerWrapper
    dev = torch.device('cuda:0')
    cfg = nano_mini_config(dropout=0.1)
    V = cfg.decoder_config.vocab_size
    tok = fake_tokenizer(V)
    torch.manual_seed(0)
    wrapper = ModelTrainerWrapper(cfg, tok, TrainerWrapperConfig(), ignore_index=-100).to(dev).train()
    n_params = sum(p.numel() for p in wrapper.model.parameters())
    opt = FusedAdamW(wrapper.model.parameters(), wrapper.model, lr=6e-4, betas=(0.9, 0.95), weight_decay=0.0)
    images, labels = synthetic_batch(args.batch, 128, args.caption_len, V, seed=1)
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
    out = {'workload': 'nano-mini (gpu/nano-mini.yaml): 12x1024 ViT, 128x128x3 images, 12x1024 decoder, 8x128 heads on 1 K/V head, '
                       '4 experts rank 16, half the positions per layer', 'params_M': round(n_params / 1e6, 1),
           'batch': args.batch, 'caption_len': args.caption_len, 'train_images_per_sec': round(args.batch / dt, 1),
           'ms_per_step': round(dt * 1e3, 2), 'final_loss': round(float(loss.detach()), 4),
           'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2 ** 30, 1), 'dtype': 'bf16', 'data': 'synthetic'}
    if args.split:
        fams = {}
        names = [n for n in dir(ops) if callable(getattr(ops, n)) and not n.startswith('_') and n not in ('Graph', 'Optional')]
        orig = {n: getattr(ops, n) for n in names}
        recs = []

        def wrap(n):
            def f(*a, **k):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                r = orig[n](*a, **k)
                e1.record()
                recs.append((n, e0, e1))
                return r
            return f
        for n in names:
            if n in ('moe_gate_bwd_blocks', 'gemm_reserved_cus', 'gemm_reserve_cus'):
                continue
            setattr(ops, n, wrap(n))
        t0 = time.perf_counter()
        step()
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        for n in names:
            setattr(ops, n, orig[n])
        for n, e0, e1 in recs:
            c = fams.setdefault(n, [0, 0.0])
            c[0] += 1
            c[1] += e0.elapsed_time(e1)
        tot = sum(v[1] for v in fams.values())
        out['step_split_ms'] = {n: {'launches': v[0], 'ms': round(v[1], 2)} for n, v in sorted(fams.items(), key=lambda kv: -kv[1][1])[:14]}
        out['step_split_total_ms'] = round(tot, 1)
        out['step_split_wall_ms'] = round(wall * 1e3, 1)
    # GEMM family of one step (HIP events around every i2t_gemm_bf16 call): FLOP rate against the bf16 MFMA peak and algorithmic
    # bytes against the HBM peak -- most of this model's GEMMs have N or K around 100 (experts, K/V head) and sit on the HBM roof
    from bench import GemmTimer
    with GemmTimer(ops) as gt:
        step()
        gs = gt.summary()
        if args.gemm_breakdown:
            for line in gt.breakdown()[:45]:
                print(line, file=sys.stderr)
    tbs = gs['bytes_per_launch'] * gs['launches'] / (gs['total_ms'] * 1e-3) / 1e12
    out['gemm_family'] = {'launches': gs['launches'], 'ms_per_step': round(gs['total_ms'], 2), 'tflops': round(gs['tflops'], 1),
                          'frac_of_2500_tflops': round(gs['tflops'] / 2500.0, 3), 'algorithmic_tb_per_s': round(tbs, 2),
                          'frac_of_8_tb_per_s': round(tbs / 8.0, 3)}
    if args.no_decode:
        print(json.dumps(out))
        return
    # greedy decode
    wrapper.eval()
    Bd = args.decode_batch
    dimgs = synthetic_batch(Bd, 128, args.caption_len, V, seed=7)[0].to(dev)
    prompt = torch.full((Bd, 1), tok.bos_token_id, dtype=torch.long, device=dev)
    with torch.no_grad():
        wrapper.model.generate(dimgs, prompt, max_new_tokens=args.new_tokens, temperature=1.0, top_k=1)      # capture
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ids = wrapper.model.generate(dimgs, prompt, max_new_tokens=args.new_tokens, temperature=1.0, top_k=1)
        torch.cuda.synchronize()
        dd = time.perf_counter() - t0
    out['greedy_captions_per_sec'] = round(Bd / dd, 1)
    out['greedy'] = {'captions': Bd, 'new_tokens': args.new_tokens, 'seconds': round(dd, 3), 'includes': 'encoder forward + KV-cache decode (hipGraph replay)'}
    assert ids.shape == (Bd, 1 + args.new_tokens)
    if args.cpu:
        from oracle import reference_model as orc
        b = 4
        sd = {k: (v.detach().float() if v.is_floating_point() else v.detach()).cpu().clone() for k, v in wrapper.model.state_dict().items()}
        ci, cl = synthetic_batch(b, 128, args.caption_len, V, seed=1)
        osd = {k: (v.requires_grad_(True) if v.is_floating_point() else v) for k, v in sd.items() if k != 'decoder.lm_head.weight'}
        osd['decoder.lm_head.weight'] = osd['decoder.transformer.wte.weight']
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            l = orc.lm_step(osd, cfg, ci, cl, tok, training=True)
            l.backward()
            ts.append(time.perf_counter() - t0)
        out['cpu_baseline'] = {'value': round(b / min(ts[1:]), 2), 'unit': 'images/s', 'cores': torch.get_num_threads(), 'kind': 'port',
                               'sample': f'oracle fp32 forward + backward of nano-mini, batch {b}, best of 2 after 1 warm-up (no optimizer step)'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
