#!/usr/bin/env python3
"""Headline benchmark: nano-224 captioning train step (images/s) + greedy decode (captions/s) on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = one full training step of the hot path on a resident synthetic batch: encoder + decoder forward, fused
weighted cross-entropy, hand-written backward, (N > 1: RCCL all-reduce of the flat gradient arena), fused AdamW.
Prints ONE JSON line (rank 0).  ``value`` is whole-job train images/s; the same line carries greedy captions/s, the
roofline of the dominant kernel (the bf16 MFMA GEMM, timed live with HIP events) and the CPU baseline (the fp32
oracle port on the host cores, rank 0 at N = 1 only, bounded sample).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# Algorithmic FLOPs of the path ("required-output" variant: the 64 dead prompt rows are not counted):
# forward 32.6 GFLOP / image, train step 3x.
FWD_GFLOP_PER_IMAGE = 32.6
TRAIN_GFLOP_PER_IMAGE = 3 * FWD_GFLOP_PER_IMAGE
MFMA_BF16_PEAK_TFLOPS = 2500.0          # MI355X_MICROARCH.md: ~2.5 PF dense bf16
HBM_PEAK_GBPS = 8000.0                  # MI355X_MICROARCH.md: HBM3E ~8 TB/s (spec)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=3072, help='images per GPU per step')
    ap.add_argument('--dropout', type=float, default=0.1, help='dropout = attn_dropout of both towers (nano.yaml: 0.1)')
    ap.add_argument('--decode-batch', type=int, default=4096, help='captions per GPU per greedy run')
    ap.add_argument('--decode-reps', type=int, default=3)
    ap.add_argument('--decode-streams', type=int, default=3, help='independent caption batches decoded concurrently')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-decode', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--gemm-breakdown', action='store_true', help='per-shape GEMM table on stderr (2 timed steps)')
    return ap.parse_args()


class GemmTimer:
    """Wraps ops.gemm with HIP events on the launch stream: per-launch duration and algorithmic FLOPs (2 M N K)."""

    def __init__(self, ops):
        self.ops, self.orig, self.records = ops, ops.gemm, []

    def __enter__(self):
        def timed(a, b, out, M, N, K, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = self.orig(a, b, out, M, N, K, **kw)
            e1.record(torch.cuda.current_stream())
            c_bytes = 4 if out.dtype == torch.float32 else 2
            # algorithmic bytes: both operands once, C once, plus every epilogue operand the call names
            extra = (4 if kw.get('residual') is not None else 0) + (4 if kw.get('accumulate') else 0) + \
                    (2 if kw.get('aux_in') is not None else 0) + (2 if kw.get('aux_out') is not None else 0)
            kind = ('A^T' if kw.get('a_kmajor') else 'A') + ('.B' if kw.get('b_kmajor') else '.B^T') + \
                   (' f32' if c_bytes == 4 else ' bf16') + ('+res' if kw.get('residual') is not None else '') + \
                   ('+acc' if kw.get('accumulate') else '') + {0: '', 1: '+gelu', 2: '+dgelu', 3: '+gelu_erf', 4: '+dgelu_erf', 5: '+gelu+dgelu_out', 6: '+mul_aux'}[int(kw.get('act', 0))] + \
                   ('+drop' if kw.get('drop') is not None else '')
            self.records.append((e0, e1, 2.0 * M * N * K, 2.0 * (M * K + N * K) + (c_bytes + extra) * M * N, (M, N, K, kind)))
            return r
        def timed_top2(a, b, top2, M, N, K):          # the greedy step's lm_head (segment maxima instead of the logits: 16 B per 64 columns)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = self.orig_top2(a, b, top2, M, N, K)
            e1.record(torch.cuda.current_stream())
            self.records.append((e0, e1, 2.0 * M * N * K, 2.0 * (M * K + N * K) + 16.0 * M * ((N + 63) // 64), (M, N, K, 'A.B^T top2')))
            return r
        self.ops.gemm = timed
        self.orig_top2 = self.ops.gemm_top2
        self.ops.gemm_top2 = timed_top2
        return self

    def __exit__(self, *exc):
        self.ops.gemm = self.orig
        self.ops.gemm_top2 = self.orig_top2

    def breakdown(self):
        """Per (shape, layout, epilogue) table: launches, avg us, TFLOP/s, algorithmic TB/s."""
        torch.cuda.synchronize()
        g = {}
        for e0, e1, fl, by, key in self.records:
            t = g.setdefault(key, [0, 0.0, 0.0, 0.0])
            t[0] += 1; t[1] += e0.elapsed_time(e1); t[2] += fl; t[3] += by
        lines = []
        for key, (n, ms, fl, by) in sorted(g.items(), key=lambda kv: -kv[1][1]):
            lines.append(f'{ms:8.3f} ms {n:4d}x avg {1e3 * ms / n:8.1f} us {fl / ms / 1e9:7.1f} TF {by / ms / 1e9:6.2f} TB/s  M={key[0]} N={key[1]} K={key[2]} {key[3]}')
        return lines

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        by = sum(r[3] for r in self.records)
        return dict(launches=len(self.records), total_ms=ms, avg_us=1e3 * ms / max(1, len(self.records)),
                    tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0, gflop_per_launch=fl / max(1, len(self.records)) / 1e9,
                    bytes_per_launch=by / max(1, len(self.records)))


class XattnTimer:
    """Wraps ops.xattn_kv_fused (the north star's fused cross-attention forward: K/V projection + attention in one launch) with HIP
    events on the launch stream.  Algorithmic FLOPs per call: 2 (B S) (2 d) d for the projection + 4 rows S d for QK^T and PV."""

    def __init__(self, ops):
        self.ops, self.orig, self.records = ops, ops.xattn_kv_fused, []

    def __enter__(self):
        def timed(mem, w_kv, bias_kv, q, kv, o, lse, B, S, H, Tq, **kw):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = self.orig(mem, w_kv, bias_kv, q, kv, o, lse, B, S, H, Tq, **kw)
            e1.record(torch.cuda.current_stream())
            d = 64 * H
            rows = int(kw.get('total_q') or 0) or B * Tq
            self.records.append((e0, e1, 2.0 * B * S * 2 * d * d + 4.0 * rows * S * d, (B, S, H, rows)))
            return r
        self.ops.xattn_kv_fused = timed
        return self

    def __exit__(self, *exc):
        self.ops.xattn_kv_fused = self.orig

    def summary(self):
        if not self.records:
            return None
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        fl = sum(r[2] for r in self.records)
        B, S, H, rows = self.records[0][3]
        return dict(launches=len(self.records), total_ms=ms, avg_us=1e3 * ms / len(self.records), tflops=fl / (ms * 1e-3) / 1e12,
                    gflop_per_launch=fl / len(self.records) / 1e9, shape=dict(images=B, memory_rows_per_image=S, heads=H, query_rows=rows))


def log(msg):
    """Progress on stderr (rank 0): the JSON line on stdout stays the only stdout output."""
    if int(os.environ.get('RANK', '0')) == 0:
        print(f'[bench {time.strftime("%H:%M:%S")}] {msg}', file=sys.stderr, flush=True)


def cpu_baseline(batch=8, steps=6, dropout=0.1):
    """The fp32 oracle port (oracle/reference_model.py) timed on the host cores: train step + torch AdamW."""
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.synth import fake_tokenizer, nano224_config, synthetic_batch
    from oracle import reference_model as orc
    try:       # the GPU box grants ~16 CPUs per GPU (cgroup quota) while the affinity mask shows every core: more threads than that
        torch.set_num_threads(min(16, len(os.sched_getaffinity(0))))      # only spin against the quota
    except Exception:
        pass
    cfg = nano224_config(dropout=dropout)
    torch.manual_seed(0)
    model = VisionEncoderDecoder(cfg)                       # parameter container only: supplies reference-style init
    sd = {k: v.detach().clone().requires_grad_(True) for k, v in model.state_dict().items() if k != 'decoder.lm_head.weight'}
    sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
    params = [v for k, v in sd.items() if k != 'decoder.lm_head.weight']
    opt = torch.optim.AdamW(params, lr=6e-4, betas=(0.9, 0.95))
    tok = fake_tokenizer(cfg.decoder_config.vocab_size)
    images, labels = synthetic_batch(batch, 224, 64, cfg.decoder_config.vocab_size, seed=1)
    times = []
    for i in range(steps + 1):
        log(f'cpu_baseline: oracle train step {i}/{steps}')
        t0 = time.perf_counter()
        loss = orc.lm_step(sd, cfg, images, labels, tok, training=True)
        loss.backward()
        opt.step()
        opt.zero_grad()
        times.append(time.perf_counter() - t0)
    dt = sum(times[1:]) / steps
    log('cpu_baseline: oracle greedy decode (4 captions x 64 new tokens)')
    t0 = time.perf_counter()
    with torch.no_grad():
        orc.generate_greedy({k: v.detach() for k, v in sd.items()}, cfg, images[:4], torch.full((4, 1), tok.bos_token_id), 64)
    dec = time.perf_counter() - t0
    return dict(value=batch / dt, unit='images/s', cores=torch.get_num_threads(), kind='port',
                sample=f'oracle fp32 train step (fwd+bwd+AdamW), nano-224, dropout {dropout}, batch {batch}, 1 warm-up + {steps} timed steps',
                greedy_captions_per_sec=4 / dec,
                greedy_sample='4 captions x 64 new tokens (the benchmark\'s decode workload), cache-free loop as the reference runs it '
                              '(full re-forward per token), 1 timed run')


class DecodeAttnTimer:
    """HIP events around every i2t_decode_attention launch of ONE eager (graph-free) greedy run on the launch stream, with the
    algorithmic bytes of each launch: the new query and output rows plus every cached K and V row it reads (the self-attention
    cache grows by one key per step; the cross-attention K/V of the 64 memory tokens are read in full every step)."""

    def __init__(self, ops, n_layers, n_cross):
        self.ops, self.orig, self.records = ops, ops.decode_attention, []
        self.self_calls, self.n_layers = 0, n_layers

    def __enter__(self):
        def timed(q, q_rs, kc, vc, cache_bs, cache_rs, o, o_rs, pos, n_keys_fixed, B, H, append_dm=0, **kw):
            if pos is not None:                   # self-attention: keys 0 .. step (the step's own key was just appended)
                keys = self.self_calls // self.n_layers + 1
                self.self_calls += 1
            else:
                keys = n_keys_fixed
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            r = self.orig(q, q_rs, kc, vc, cache_bs, cache_rs, o, o_rs, pos, n_keys_fixed, B, H, append_dm, **kw)
            e1.record(torch.cuda.current_stream())
            self.records.append((e0, e1, B * H * 64 * 2 * (2 * keys + 2)))
            return r
        self.ops.decode_attention = timed
        return self

    def __exit__(self, *exc):
        self.ops.decode_attention = self.orig

    def summary(self):
        torch.cuda.synchronize()
        ms = sum(r[0].elapsed_time(r[1]) for r in self.records)
        by = sum(r[2] for r in self.records)
        n = max(1, len(self.records))
        return dict(launches=len(self.records), total_ms=ms, avg_us=1e3 * ms / n, gbps=by / (ms * 1e-3) / 1e9 if ms > 0 else 0.0,
                    bytes_per_launch=by / n)


def xattn_width_leg(ops, dev, images=2048, d=1280, dropout=0.1, reps=20):
    """The north star's kernel on the decoder of the reference's own GPU config (training_configs/gpu/nano.yaml:70-89: n_embd 1280,
    20 heads of 64, cross-attention over the encoder's 64 output tokens): ONE cross-attention layer's fused K/V projection +
    attention launch (i2t_xattn_kv_fused) over `images` images with packed 8..63-row caption queries, timed with HIP events on the
    launch stream -- with the step's probability dropout and without.  Returns {variant: dict(avg_us, tflops, gflop_per_launch)}."""
    from image2text_amd import rng
    BF16 = torch.bfloat16
    H, S = d // 64, 64
    g = torch.Generator().manual_seed(7)
    lens = torch.randint(8, 64, (images,), generator=g)
    cu = torch.zeros(images + 1, dtype=torch.int32)
    cu[1:] = torch.cumsum(lens, 0)
    M = int(cu[-1])
    cu = cu.to(dev)
    mem = torch.randn(images * S, d, device=dev).to(BF16)
    w_kv = (torch.randn(2 * d, d, device=dev) * 0.03).to(BF16)
    b_kv = torch.randn(2 * d, device=dev) * 0.1
    q = torch.randn(M, d, device=dev).to(BF16)
    kv = torch.empty(images, S, 2 * d, dtype=BF16, device=dev)
    o = torch.empty(M, d, dtype=BF16, device=dev)
    lse = torch.empty(H * M, device=dev)
    flops = 2.0 * images * S * 2 * d * d + 4.0 * M * S * d
    out = {}
    for name, p in (('dropout', dropout), ('no_dropout', 0.0)):
        thr = rng.threshold(p) if p > 0 else 0
        drop = (1, rng.site_key(1234, 99), thr, rng.scale(thr)) if p > 0 else None
        run = lambda: ops.xattn_kv_fused(mem, w_kv, b_kv, q, kv, o, lse, images, S, H, 64, drop=drop, cu_q=cu, total_q=M)
        for _ in range(5):
            run()
        evs = []
        for _ in range(reps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(torch.cuda.current_stream())
            run()
            e1.record(torch.cuda.current_stream())
            evs.append((e0, e1))
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in evs) / reps
        out[name] = dict(avg_us=1e3 * ms, tflops=flops / (ms * 1e-3) / 1e12, gflop_per_launch=flops / 1e9, launches=reps,
                         shape=dict(images=images, memory_rows_per_image=S, heads=H, width=d, query_rows=M))
    return out


def main():
    args = parse()
    # stdout carries exactly one JSON line: libraries that write to fd 1 (RCCL prints a version banner when the first
    # communicator is created) are sent to stderr, the JSON goes to a private duplicate of the original stdout
    json_out = os.fdopen(os.dup(1), 'w')
    sys.stdout.flush()
    os.dup2(2, 1)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1 or os.environ.get('I2T_FORCE_DP') == '1':
        from image2text_amd.training.dp import configure_rccl_env
        configure_rccl_env()          # NCCL_MAX_NCHANNELS must be in the environment BEFORE the communicator is created
    # Control plane of the N > 1 run (rendezvous, barriers, the timing reductions): a gloo group on the HOST.  The gradient exchange runs
    # on the package's own RCCL communicator (csrc/comm.cpp, created by DataParallelGrads; its unique id travels through this group's
    # store): each process then holds exactly ONE RCCL communicator, the one NCCL_MAX_NCHANNELS is meant for.  I2T_BENCH_PG=nccl: the
    # old arrangement (torch's RCCL group beside it), for A/B runs.
    pg_backend = os.environ.get('I2T_BENCH_PG', 'gloo')
    if world > 1:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if pg_backend == 'nccl':
            dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        else:
            dist.init_process_group('gloo')
    dev = torch.device('cuda', local)
    torch.cuda.set_device(dev)

    from image2text_amd import ops
    from image2text_amd.configs.trainer import TrainerWrapperConfig
    from image2text_amd.synth import fake_tokenizer, nano224_config, synthetic_batch
    from image2text_amd.training.dp import DataParallelGrads
    from image2text_amd.training.optim import FusedAdamW
    from image2text_amd.training.wrapper import ModelTrainerWrapper

    cfg = nano224_config(dropout=args.dropout)
    V = cfg.decoder_config.vocab_size
    torch.manual_seed(0)
    wrapper = ModelTrainerWrapper(cfg, fake_tokenizer(V), TrainerWrapperConfig(), ignore_index=-100).to(dev).train()
    opt = FusedAdamW(wrapper.model.parameters(), wrapper.model, lr=6e-4, betas=(0.9, 0.95), weight_decay=0.0)
    # I2T_FORCE_DP=1: run the gradient exchange even at world size 1 (a 1-rank RCCL group) -- rehearsal of the N > 1 code path on one GPU
    force_dp = world == 1 and os.environ.get('I2T_FORCE_DP') == '1' and 'RANK' in os.environ
    if force_dp:
        import torch.distributed as dist
        if pg_backend == 'nccl':
            dist.init_process_group('nccl', device_id=dev)
        else:
            dist.init_process_group('gloo')
    dp = DataParallelGrads(wrapper.model) if (world > 1 or force_dp) else None
    images, labels = synthetic_batch(args.batch, 224, 64, V, seed=1 + rank)      # each rank draws its own shard
    images, labels = images.to(dev), labels.to(dev)

    def step():
        loss, _ = wrapper.train_step(images, labels)
        loss.backward()
        if dp is not None:
            dp.all_reduce_mean()
        opt.step()
        opt.zero_grad()
        return loss

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if dp is not None:
        with dp.no_sync():                                           # priming backward: builds the arena, exchanges nothing
            wrapper.train_step(images[:1], labels[:1])[0].backward()
        opt.zero_grad()
        dp.broadcast_parameters()
    # ---- the north star's kernel on the reference's GPU-config decoder width, stand-alone (before the training phase: the same launch
    # measured after 20 steps at 190 GB resident ran ~10 % slower -- the chip's clock state, not the kernel)
    xattn_wide = None
    if not args.no_kernel_timing and rank == 0:
        log('xattn: one cross-attention layer of the gpu/nano.yaml decoder (d = 1280, 2048 images)')
        xattn_wide = xattn_width_leg(ops, dev, dropout=args.dropout)
        torch.cuda.empty_cache()
    fence()
    log(f'train: batch {args.batch}/gpu, {args.warmup} warm-up + {args.steps} timed steps')
    for _ in range(args.warmup):
        loss = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    elapsed = time.perf_counter() - t0
    final_loss = float(loss.detach())
    log(f'train: {1e3 * elapsed / args.steps:.2f} ms/step')
    red_dev = dev if pg_backend == 'nccl' else 'cpu'           # the timing reductions travel on the control plane
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
    img_s = world * args.batch * args.steps / elapsed

    # ---- the same step with the batch handed over as HOST buffers (pinned): PCIe-inclusive rate, reported beside `value`
    pcie_img_s = None
    if not args.no_kernel_timing:
        h_images, h_labels = images.cpu().pin_memory(), labels.cpu().pin_memory()
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            images.copy_(h_images, non_blocking=True)
            labels.copy_(h_labels, non_blocking=True)
            step()
        fence()
        pcie_img_s = world * args.batch * 3 / (time.perf_counter() - t1)
        del h_images, h_labels

    # ---- per-kernel timing of the dominant kernel (bf16 MFMA GEMM) with HIP events, 2 extra steps
    gemm = xattn = None
    if not args.no_kernel_timing:
        with GemmTimer(ops) as gt, XattnTimer(ops) as xt:
            for _ in range(2):
                step()
        gemm = gt.summary()
        xattn = xt.summary()
        if args.gemm_breakdown and rank == 0:
            for ln in gt.breakdown():
                log(ln)
    fence()

    # ---- greedy decode: B captions x 64 new tokens per run (encoder + KV-cache decode under hipGraph replay)
    cap_s = dec_attn = dec_gemm = None
    if not args.no_decode:
        wrapper.eval()
        opt.zero_grad()
        torch.cuda.empty_cache()              # hand the training step's cached blocks back before the decode leg allocates its own
        from image2text_amd.decoding import ConcurrentGreedyDecoder
        Bd, S = args.decode_batch, args.decode_streams
        cdec = dimgs = prompts = None
        while True:       # the decode leg is rank-local: a rank that runs out of memory retries with half the captions per batch
            try:
                dimgs = [synthetic_batch(Bd, 224, 64, V, seed=100 + rank * S + i)[0].to(dev) for i in range(S)]
                prompts = [torch.full((Bd, 1), V - 1, dtype=torch.long, device=dev) for _ in range(S)]
                cdec = ConcurrentGreedyDecoder(wrapper.model, S)
                log(f'decode: {S} concurrent batches x {Bd} captions, warm-up + graph capture')
                cdec.generate(dimgs, prompts, 64)                              # warm-up + graph capture
                torch.cuda.synchronize()
                break
            except torch.OutOfMemoryError:
                cdec = dimgs = prompts = None
                torch.cuda.empty_cache()
                if Bd <= 256:
                    raise
                Bd //= 2
                log(f'decode: out of memory, retrying with {Bd} captions per batch')
        fence()
        log('decode: timed runs')
        t0 = time.perf_counter()
        for _ in range(args.decode_reps):
            outs = cdec.generate(dimgs, prompts, 64)
        fence()
        assert all(tuple(o.shape) == (Bd, 65) for o in outs)
        dt = time.perf_counter() - t0
        n_caps = float(Bd * S * args.decode_reps)
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=red_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t)
            c = torch.tensor([n_caps], dtype=torch.float64, device=red_dev)
            dist.all_reduce(c, op=dist.ReduceOp.SUM)
            n_caps = float(c)
        cap_s = n_caps / dt
        # ---- dominant decode kernel (decode_attention: K/V cache streaming, HBM-bound): one eager greedy run of lane 0
        if not args.no_kernel_timing:
            eng = wrapper.model._engine
            # (the same eager run also times every GEMM of the decode leg: by kernel time they are the larger family -- r02 rocprof:
            # 52 % of the leg against 29 % for decode_attention -- and MFMA-bound, so both are reported)
            with DecodeAttnTimer(ops, eng.dec.L, sum(eng.dec_cross)) as dt_, GemmTimer(ops) as dg_:
                cdec.lanes[0][0].generate(dimgs[0], prompts[0], 64, use_graph=False)
            dec_attn = dt_.summary()
            dec_gemm = dg_.summary()
        wrapper.train()

    if rank == 0:
        out = {
            'metric': 'train images/sec + greedy captions/sec, nano config',
            'value': round(img_s, 2), 'unit': 'images/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(1e3 * elapsed / args.steps, 3), 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'bf16', 'data': 'synthetic',
            'config': {'workload': 'nano-224 (6x512 ViT encoder + 12x768 nanoGPT decoder, 224x224x3 images, 64-token captions)',
                       'global_batch': world * args.batch, 'per_gpu_batch': args.batch, 'caption_len': 64,
                       'parallelism': f'dp{world}', 'dropout': args.dropout, 'optimizer': 'AdamW lr 6e-4 betas (0.9,0.95)',
                       'weights': 'random init (reference distributions)'},
            'greedy_captions_per_sec': None if cap_s is None else round(cap_s, 2),
            'greedy_config': {'captions_per_batch': (Bd if not args.no_decode else args.decode_batch), 'concurrent_batches_per_gpu': args.decode_streams,
                              'new_tokens': 64, 'ngrams': [2, 3, 4, 5], 'includes': 'encoder forward + KV-cache decode (hipGraph replay)'},
            'final_loss': round(final_loss, 4),
            'host_input_images_per_sec': None if pcie_img_s is None else round(pcie_img_s, 1),     # H2D copy of the batch inside the step
            # nominal = SURVEY 8(d) required-output count (98 GFLOP/image); the step executes less: padded caption rows are skipped
            'step_nominal_tflops': round(img_s * TRAIN_GFLOP_PER_IMAGE / 1e3, 1),
            'peak_mem_gb': round(torch.cuda.max_memory_allocated() / 2**30, 1),
        }
        traffic = None
        try:      # HBM bytes per GEMM launch from the committed PMC passes (tools/pmc_traffic.py), same batch only
            with open(os.path.join(ROOT, 'profiles', f'pmc_traffic_b{args.batch}.json')) as fh:
                traffic = json.load(fh)['gemm_all']['hbm_bytes_per_launch']
        except Exception:
            pass
        if gemm is not None:
            out['roofline'] = {'bound': 'mfma', 'kernel': 'gemm256_kernel + gemm_bf16_kernel (every i2t_gemm_bf16 launch of the step)', 'achieved': round(gemm['tflops'], 1),
                               'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s', 'frac': round(gemm['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4),
                               'traffic': traffic, 'traffic_unit': 'bytes/launch (FETCH_SIZE x2 + WRITE_SIZE, rocprofv3 --pmc)',
                               'algorithmic_bytes_per_launch': round(gemm['bytes_per_launch']), 'launches_per_step': gemm['launches'] // 2,
                               'avg_launch_us': round(gemm['avg_us'], 2), 'gflop_per_launch': round(gemm['gflop_per_launch'], 3),
                               'gemm_ms_per_step': round(gemm['total_ms'] / 2, 3)}
        if xattn is not None:      # the north star's kernel (BASELINE.json: >= 40 % bf16 MFMA utilisation on the fused cross-attention kernel)
            out['xattn_roofline'] = {'bound': 'mfma', 'kernel': 'gemm256_kernel<..., 8> = i2t_xattn_kv_fused (K/V projection + attention in one launch; '
                                                                'every launch of the timed steps)',
                                     'achieved': round(xattn['tflops'], 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                     'frac': round(xattn['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                                     'launches_per_step': xattn['launches'] // 2, 'avg_launch_us': round(xattn['avg_us'], 2),
                                     'gflop_per_launch': round(xattn['gflop_per_launch'], 3), 'shape': xattn['shape'],
                                     'note': 'decoder width 768 (K loop of 12 tiles), timed inside the training step; the same kernel on the '
                                             'decoder of the reference\'s gpu/nano.yaml: xattn_roofline_d1280'}
        if xattn_wide is not None:      # the same kernel, one layer of the reference's GPU-config decoder (training_configs/gpu/nano.yaml:70-89)
            xw, xn = xattn_wide['dropout'], xattn_wide['no_dropout']
            out['xattn_roofline_d1280'] = {'bound': 'mfma', 'kernel': 'gemm256_kernel<..., 8> = i2t_xattn_kv_fused, decoder width 1280 (20 heads of 64), '
                                                                      'stand-alone launches (timed before the training phase) with the step\'s probability dropout',
                                           'achieved': round(xw['tflops'], 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                           'frac': round(xw['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None, 'launches': xw['launches'],
                                           'avg_launch_us': round(xw['avg_us'], 2), 'gflop_per_launch': round(xw['gflop_per_launch'], 3),
                                           'shape': xw['shape'], 'without_dropout': {'achieved': round(xn['tflops'], 1),
                                                                                     'frac': round(xn['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4),
                                                                                     'avg_launch_us': round(xn['avg_us'], 2)}}
        if dec_attn is not None:
            out['decode_roofline'] = {'bound': 'hbm', 'kernel': 'decode_attention_kernel (every self- and cross-attention launch of one 64-token greedy run, eager)',
                                      'achieved': round(dec_attn['gbps'], 1), 'peak': HBM_PEAK_GBPS, 'unit': 'GB/s',
                                      'frac': round(dec_attn['gbps'] / HBM_PEAK_GBPS, 4), 'traffic': None,
                                      'algorithmic_bytes_per_launch': round(dec_attn['bytes_per_launch']),
                                      'launches': dec_attn['launches'], 'avg_launch_us': round(dec_attn['avg_us'], 2),
                                      'decode_attention_ms_per_run': round(dec_attn['total_ms'], 2), 'captions': Bd, 'new_tokens': 64}
        if dec_gemm is not None and dec_gemm['launches']:
            out['decode_gemm_roofline'] = {'bound': 'mfma', 'kernel': 'gemm256_kernel / gemm_bf16_kernel (every i2t_gemm_bf16 / i2t_gemm_bf16_top2 launch of the same eager run: '
                                           'encoder forward of the captions + 64 decode steps at M = captions rows)',
                                           'achieved': round(dec_gemm['tflops'], 1), 'peak': MFMA_BF16_PEAK_TFLOPS, 'unit': 'TFLOP/s',
                                           'frac': round(dec_gemm['tflops'] / MFMA_BF16_PEAK_TFLOPS, 4), 'traffic': None,
                                           'launches': dec_gemm['launches'], 'avg_launch_us': round(dec_gemm['avg_us'], 2),
                                           'gemm_ms_per_run': round(dec_gemm['total_ms'], 2),
                                           'share_of_timed_decode_kernels': round(dec_gemm['total_ms'] / (dec_gemm['total_ms'] + dec_attn['total_ms']), 3)
                                           if dec_attn is not None else None}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(dropout=args.dropout)
        print(json.dumps(out), file=json_out, flush=True)
    if dp is not None:
        dp.close()                      # destroys the package's RCCL communicator (collective teardown) before the control plane goes
    if world > 1 or force_dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
