"""Data-parallel exchange on CPU: world_size 2 over gloo.  The oracle for DP is the MEAN over ranks of
the per-shard gradients, each shard's gradient taken from the CPU oracle on that shard (per-replica gradient
normaliser, loss divided by the local batch)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from image2text_amd.synth import det_init_, fake_tokenizer, tiny_config


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _shard_grads(cfg, sd0, images, labels):
    from oracle import reference_model as orc
    sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items() if k != 'decoder.lm_head.weight'}
    sd['decoder.lm_head.weight'] = sd['decoder.transformer.wte.weight']
    loss = orc.lm_step(sd, cfg, images, labels, fake_tokenizer(cfg.decoder_config.vocab_size), training=True)
    loss.backward()
    return {k: v.grad for k, v in sd.items() if k != 'decoder.lm_head.weight'}


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from image2text_amd.engine import ParamArena
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.training.dp import DataParallelGrads
    from conftest import load_golden
    cfg = tiny_config()
    model = VisionEncoderDecoder(cfg)
    det_init_(model, seed=rank)                      # ranks start from DIFFERENT weights: broadcast must fix that
    arena = ParamArena(model, torch.device('cpu'))

    class Holder:                                    # the exchange only needs `.arena`
        pass
    h = Holder()
    h.arena = arena
    dp = DataParallelGrads(h, overlap=False)
    dp.broadcast_parameters(0)
    g = load_golden('tiny_train_init.npz')
    images, labels = torch.from_numpy(g['images']), torch.from_numpy(g['labels'])
    shard = slice(2 * rank, 2 * rank + 2)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    grads = _shard_grads(cfg, sd0, images[shard], labels[shard])
    for name, gr in grads.items():
        arena.G(name).copy_(gr)
    # emulate the engine's early decoder-slice reduction, then finish
    dp._on_grads_ready('decoder')
    dp.all_reduce_mean()
    torch.save({'p': arena.p32.clone(), 'g': arena.g32.clone(), 'entries': arena.entries}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_mean_of_per_shard_gradients(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'r{i}.pt', weights_only=False) for i in range(world)]
    assert torch.equal(r[0]['p'], r[1]['p']), 'parameters must be identical after broadcast'
    assert torch.equal(r[0]['g'], r[1]['g']), 'every rank must hold the same reduced gradients'
    # single-process oracle of the same thing
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from conftest import load_golden
    cfg = tiny_config()
    model = det_init_(VisionEncoderDecoder(cfg), seed=0)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    g = load_golden('tiny_train_init.npz')
    images, labels = torch.from_numpy(g['images']), torch.from_numpy(g['labels'])
    per = [_shard_grads(cfg, sd0, images[2 * i:2 * i + 2], labels[2 * i:2 * i + 2]) for i in range(world)]
    for name, (off, n, shape) in r[0]['entries'].items():
        want = (per[0][name] + per[1][name]) / 2
        got = r[0]['g'][off:off + n].view(shape)
        assert torch.allclose(got, want, rtol=1e-3, atol=1e-6), name
    # and it is NOT the gradient of the concatenated batch (per-replica normaliser): guards against "one big batch"
    full = _shard_grads(cfg, sd0, images, labels)
    name = 'decoder.transformer.h.0.mlp.c_fc.weight'
    off, n, shape = r[0]['entries'][name]
    assert not torch.allclose(r[0]['g'][off:off + n].view(shape), full[name], rtol=1e-3, atol=1e-7)


def _overlap_worker(rank, world, port, out_dir):
    """The engine-hooked exchange (overlap=True) through the event sequence a real training loop produces: a priming backward
    that nobody finishes (bench.py builds the arena that way), broadcast, then an accumulation window of two micro-batches --
    the first under no_sync(), the second starting the decoder slice's all-reduce from the 'decoder' hook."""
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from image2text_amd.engine import ParamArena
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.training.dp import DataParallelGrads
    from conftest import load_golden
    cfg = tiny_config()
    model = VisionEncoderDecoder(cfg)
    det_init_(model, seed=rank)
    arena = ParamArena(model, torch.device('cpu'))
    eng = SimpleNamespace(arena=arena, grad_ready_hooks=[])
    holder = SimpleNamespace(_engine=eng)
    dp = DataParallelGrads(holder, overlap=True)
    assert len(eng.grad_ready_hooks) == 1
    notify = lambda which: [h(which) for h in eng.grad_ready_hooks]

    def backward(grads):                             # what _LMLossFunction.backward does to the arena, with oracle gradients:
        notify('begin')                              # the decoder's gradients are final at 'decoder', the encoder's only afterwards
        for name, gr in grads.items():
            if name.startswith('decoder.'):
                arena.G(name).add_(gr)
        notify('decoder')
        for work, _, _ in dp._pending:               # a collective may finish at any time -- here: at once, so an exchange that
            work.wait()                              # (wrongly) covered the encoder's slice this early would miss its gradients
        for name, gr in grads.items():
            if not name.startswith('decoder.'):
                arena.G(name).add_(gr)
        notify('encoder')

    g = load_golden('tiny_train_init.npz')
    images, labels = torch.from_numpy(g['images']), torch.from_numpy(g['labels'])
    # priming backward with garbage gradients, never followed by all_reduce_mean
    backward({name: torch.full(shape, float(rank + 1)) for name, (_, _, shape) in arena.entries.items()})
    dp.broadcast_parameters(0)                       # drains the exchange the priming backward left in flight ...
    assert dp._reduced_upto is None and not dp._pending
    arena.g32.zero_()                                # ... before the gradients are cleared (ParamArena.begin_backward does that after the
    #                                                  'begin' hook's drain; zeroing first would race the collective still writing)
    sd0 = {k: v.detach().clone() for k, v in model.state_dict().items()}
    # accumulation window: this rank's micro-batches are samples [2r] and [2r+1]
    micro = [_shard_grads(cfg, sd0, images[2 * rank + i:2 * rank + i + 1], labels[2 * rank + i:2 * rank + i + 1]) for i in range(2)]
    with dp.no_sync():
        backward(micro[0])
    assert not dp._pending, 'no_sync must not start an exchange'
    backward(micro[1])
    assert dp._reduced_upto is not None, "the 'decoder' hook must have started the decoder slice"
    dp.all_reduce_mean()
    first = arena.g32.clone()
    # a second window right after (state fully reset): same inputs -> same result
    arena.g32.zero_()
    with dp.no_sync():
        backward(micro[0])
    backward(micro[1])
    dp.all_reduce_mean()
    assert torch.equal(first, arena.g32)
    # un-annotated accumulation (no no_sync): wasteful, still the same mean (all-reduce is linear)
    arena.g32.zero_()
    backward(micro[0])
    backward(micro[1])
    dp.all_reduce_mean()
    assert torch.allclose(first, arena.g32, rtol=1e-4, atol=1e-6 * float(first.abs().max()))      # other summation order
    torch.save({'p': arena.p32.clone(), 'g': first, 'entries': arena.entries,
                'local': {k: micro[0][k] + micro[1][k] for k in micro[0]}}, os.path.join(out_dir, f'r{rank}.pt'))
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_overlap_priming_and_accumulation(tmp_path):
    world = 2
    mp.spawn(_overlap_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'r{i}.pt', weights_only=False) for i in range(world)]
    assert torch.equal(r[0]['p'], r[1]['p'])
    assert torch.equal(r[0]['g'], r[1]['g']), 'every rank must hold the same reduced gradients (decoder AND encoder slices)'
    for name, (off, n, shape) in r[0]['entries'].items():
        want = (r[0]['local'][name] + r[1]['local'][name]) / 2      # mean over ranks of each rank's accumulated window
        got = r[0]['g'][off:off + n].view(shape)
        assert torch.allclose(got, want, rtol=1e-5, atol=1e-8), name


def _span_worker(rank, world, port, out_dir):
    """A model with FROZEN stretches in its arena (a frozen backbone + frozen decoder base weights, as under LoRA /
    prepare_for_kbit_training) on the bf16 wire: only the coalesced spans of trainable entries travel."""
    from types import SimpleNamespace
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), I2T_DP_WIRE='bf16')
    dist.init_process_group('gloo', rank=rank, world_size=world)
    torch.set_num_threads(2)
    from image2text_amd.engine import ParamArena
    from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
    from image2text_amd.training import dp as dpm
    dpm.GAP_FLOATS = 64                              # (the tiny model's entries are far smaller than the production 1 MB gap rule)
    cfg = tiny_config()
    model = VisionEncoderDecoder(cfg)
    det_init_(model, seed=0)
    frozen = [n for n, _ in model.named_parameters() if '.mlp.' in n or n.startswith('encoder.') and '.attn.' in n]
    for n, p in model.named_parameters():
        if n in frozen:
            p.requires_grad_(False)
    arena = ParamArena(model, torch.device('cpu'))
    eng = SimpleNamespace(arena=arena, grad_ready_hooks=[])
    dp = dpm.DataParallelGrads(SimpleNamespace(_engine=eng), overlap=True)
    g = torch.Generator().manual_seed(100 + rank)
    local = torch.randn(arena.total, generator=g)
    sentinel = 7.0 + rank
    for name, (off, n, _) in arena.entries.items():
        if name in arena.params and not arena.trainable(name):
            local[off:off + n] = sentinel            # what a frozen segment's (never written) gradient slot holds: must not travel
    arena.g32.copy_(local)
    lo, hi = dp._split(arena)
    spans = dp._spans(arena, 0, arena.total)
    assert len(spans) > 2, spans                     # the frozen stretches really split the arena
    for h in eng.grad_ready_hooks:
        h('decoder')
    assert dp._pending and dp._reduced_upto == (lo, hi)
    dp.all_reduce_mean()
    torch.save({'g': arena.g32.clone(), 'local': local, 'entries': arena.entries, 'spans': spans,
                'frozen': [n for n in arena.entries if n in arena.params and not arena.trainable(n)]}, os.path.join(out_dir, f'r{rank}.pt'))
    dp.close()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_world2_trainable_spans_on_the_bf16_wire(tmp_path):
    world = 2
    mp.spawn(_span_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f'r{i}.pt', weights_only=False) for i in range(world)]
    assert r[0]['spans'] == r[1]['spans'] and r[0]['frozen']
    covered = torch.zeros(r[0]['g'].numel(), dtype=torch.bool)
    for a, b in r[0]['spans']:
        covered[a:b] = True
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    want = bf(bf(r[0]['local']) + bf(r[1]['local'])) / 2            # rounded onto the wire, summed in bf16, mean in fp32 on arrival
    for i in range(world):
        assert torch.equal(r[i]['g'][covered], want[covered]), 'trainable spans: the bf16-wire mean, identical on every rank'
        assert torch.equal(r[i]['g'][~covered], r[i]['local'][~covered]), 'everything outside the spans stays local'
    true_mean = (r[0]['local'] + r[1]['local']) / 2
    err = (r[0]['g'][covered] - true_mean[covered]).abs().max() / true_mean[covered].abs().max()
    assert float(err) < 2 ** -6
    for name in r[0]['frozen']:                       # every frozen entry larger than the gap rule stayed off the wire
        off, n, _ = r[0]['entries'][name]
        if n > 64 + 16:
            assert not bool(covered[off:off + n].all()), name
