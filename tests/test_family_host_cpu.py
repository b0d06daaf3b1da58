"""Host-side planning of the nano-mini family (no GPU): the per-layer row lists of sparse blocks, the position-major order of the
per-position MLP, the contrastive loss's row maps and the MoE-aware arena order -- each against a brute-force restatement."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from image2text_amd.engine import _arena_order
from image2text_amd.engine_family import FamilyBlocks, _expand_rows, family_spec
from image2text_amd.models.vision_encoder_decoder import VisionEncoderDecoder
from image2text_amd.synth import mini_config, nano_mini_config


class _Host(FamilyBlocks):
    """FamilyBlocks with just enough state for the planning methods (they only touch the device to upload index lists)."""

    def __init__(self, idx_sets, block=64):
        self.arena = SimpleNamespace(device=torch.device('cpu'))
        self._sparse_idx = {'dec': idx_sets, 'enc': None}
        self._sub_cache = {}
        self.dec = SimpleNamespace(block=block)


def _sets(max_block, n_cls, frac, seed):
    gen = np.random.Generator(np.random.PCG64(seed=seed))
    full = np.concatenate((np.arange(n_cls), gen.permutation(max_block - n_cls) + n_cls))
    k = int(frac * max_block)
    return np.sort(full[:k]).astype(np.int64), np.sort(full[k:]).astype(np.int64)


@pytest.mark.parametrize('off,T,lens', [(8, 24, None), (8, 24, [24, 3, 0, 17, 1]), (0, 16, [16, 5, 9]), (8, 1, [1, 1]), (4, 40, [40, 0, 0, 12])])
def test_sparse_subset_row_lists(off, T, lens):
    idx, nidx = _sets(64, off, 0.5, 3)
    h = _Host([(idx, nidx)])
    B = 4 if lens is None else len(lens)
    vl = None if lens is None else SimpleNamespace(lens_host=lens)
    sub = h.sparse_subset('dec', 0, B, T, off, vl)
    kept_full = idx[idx < off + T]
    keep = set((kept_full[kept_full >= off] - off).tolist())
    if kept_full.size <= 1 or not keep:
        assert sub.all_null
        return
    rows_in, rows_out, cu = [], [], [0]
    base = 0
    for b in range(B):
        n = T if lens is None else lens[b]
        for t in range(n):
            (rows_in if t in keep else rows_out).append(base + t)
        cu.append(len(rows_in))
        base += T if lens is None else n
    assert sub.rows_in.tolist() == rows_in and sub.rows_out.tolist() == rows_out
    assert sub.n_in == len(rows_in) and sub.n_out == len(rows_out)
    if lens is None:
        assert sub.vl_in is None and sub.T_in == len(keep)
    else:
        assert sub.vl_in.cu.tolist() == cu and sub.vl_in.total == len(rows_in)
        assert sub.T_in == max(cu[i + 1] - cu[i] for i in range(B))


def test_sparse_subset_rejects_sequences_past_max_block_size():
    idx, nidx = _sets(32, 4, 0.5, 1)
    with pytest.raises(AssertionError):
        _Host([(idx, nidx)]).sparse_subset('dec', 0, 2, 40, 4, None)


@pytest.mark.parametrize('T,lens', [(6, None), (9, [9, 2, 0, 5, 9, 1])])
def test_position_major_plan(T, lens):
    h = _Host(None)
    B = 3 if lens is None else len(lens)
    plan = h._pos_plan(B, T, None if lens is None else SimpleNamespace(lens_host=lens))
    L = [T] * B if lens is None else lens
    cu = np.concatenate(([0], np.cumsum(L)))
    seg, rows = plan.seg.tolist(), plan.rows.tolist()
    assert len(seg) == T + 1 and seg[-1] == sum(L) == plan.M
    for t in range(T):
        got = sorted(rows[seg[t]:seg[t + 1]])
        assert got == sorted(int(cu[b]) + t for b in range(B) if L[b] > t), t
    assert sorted(rows) == list(range(sum(L)))                    # a permutation of the packed rows
    assert plan.max_rows == max(1, max(sum(1 for b in range(B) if L[b] > t) for t in range(T)))


@pytest.mark.parametrize('n_p,L,T,lens', [(8, 16, 16, None), (8, 16, 16, [16, 3, 0, 9]), (0, 12, 12, [12, 5]), (8, 6, 6, None), (3, 10, 10, [10, 10])])
def test_contrastive_row_maps(n_p, L, T, lens):
    from image2text_amd.training.wrapper import _contrastive_rows
    B = 3 if lens is None else len(lens)
    Lc = min(L, n_p + T)
    vl = None if lens is None else SimpleNamespace(lens_host=lens)
    src, dst, psrc, pdst = (x.tolist() for x in _contrastive_rows(B, T, n_p, Lc, vl, torch.device('cpu')))
    want_s, want_d, base = [], [], 0
    for b in range(B):
        n = T if lens is None else lens[b]
        for t in range(n):
            if n_p + t < Lc:
                want_s.append(base + t)
                want_d.append(b * Lc + n_p + t)
        base += T if lens is None else n
    assert src == want_s and dst == want_d
    assert psrc == [b * n_p + c for b in range(B) for c in range(min(n_p, Lc))]
    assert pdst == [b * Lc + c for b in range(B) for c in range(min(n_p, Lc))]
    assert len(set(dst) | set(pdst)) == len(dst) + len(pdst)      # no destination written twice


def test_expand_rows():
    out = _expand_rows(np.array([100, 200, 300]), np.array([1, 4, 6, 9]), np.array([2, 0, 3]))
    assert out.tolist() == [101, 104, 301, 304, 306]
    assert _expand_rows(np.array([5]), np.array([1]), np.array([0])).tolist() == []


def test_arena_order_groups_each_moe_linear():
    m = VisionEncoderDecoder(mini_config())
    order = _arena_order(list(m.named_parameters()))
    names = [n for n, _, _, _ in order]
    assert len(names) == len(set(names))
    real = [n for n, p, _, _ in order if p is not None]
    assert sorted(real) == sorted(n for n, _ in m.named_parameters())         # every parameter exactly once
    pads = [(n, numel) for n, p, numel, _ in order if p is None]
    assert len(pads) == 4 and all(n.startswith('encoder.') and n.endswith('expert_gates.model.0.bias') and numel == 32 for n, numel in pads)
    p = 'decoder.transformer.h.1.mlp.c_proj'
    i = names.index(f'{p}.experts.0.l1.weight')
    assert names[i:i + 5] == [f'{p}.experts.{e}.l1.weight' for e in range(4)] + [f'{p}.expert_gates.model.0.weight']
    assert names[i + 5:i + 10] == [f'{p}.experts.{e}.l1.bias' for e in range(4)] + [f'{p}.expert_gates.model.0.bias']
    assert names[i + 10:i + 18] == [f'{p}.experts.{e}.l2.weight' for e in range(4)] + [f'{p}.experts.{e}.l2.bias' for e in range(4)]
    # towers stay contiguous (the data-parallel exchange puts the decoder's range on the wire while the encoder still runs backward)
    is_dec = [n.startswith('decoder.') for n in names]
    assert sum(1 for a, b in zip(is_dec, is_dec[1:]) if a != b) == 1


def test_family_spec_and_refusals():
    cfg = nano_mini_config()
    e = family_spec(cfg.vision_encoder_config.transformer_config, 12)
    d = family_spec(cfg.decoder_config.transformer_config, 12)
    assert (e.hd, e.mqa, e.sparse, e.moe.E, e.moe.P, e.moe.G, e.moe.top_k, e.moe.Kp) == (128, True, True, 4, 16, 32, 2, 128)
    assert d.moe.top_k == 1 and d.causal and not e.causal
    from image2text_amd.synth import nano224_config
    assert family_spec(nano224_config().decoder_config.transformer_config, 12) is None      # the dense path keeps the benchmark model
    bad = mini_config(gate_sizes=(32, 16))
    with pytest.raises(NotImplementedError):
        family_spec(bad.decoder_config.transformer_config, 2)
    with pytest.raises(NotImplementedError):
        family_spec(mini_config(proj=12).decoder_config.transformer_config, 2)


def test_lockstep_driver_sums_parts_of_generators_at_the_same_key():
    """HotPath._lockstep: backward generators that stop at the same normaliser key share the sum of their parts, a generator that
    is alone at a key gets its own part back, keys are served in descending order, return values are collected"""
    from image2text_amd.engine import HotPath
    log = []

    def seg(name, keys, part):
        total = 0.0
        for k in keys:
            joint = yield k, torch.tensor([part])
            log.append((name, k, float(joint)))
            total += float(joint)
        return name, total

    out = HotPath._lockstep(seg('text', [3, 2, 1, 0], 1.0), seg('prompt', [3, 1, 0], 10.0))
    assert out == [('text', 11.0 + 1.0 + 11.0 + 11.0), ('prompt', 33.0)]
    assert [(n, k) for n, k, _ in log if k == 2] == [('text', 2)] and dict(((n, k), j) for n, k, j in log)[('text', 2)] == 1.0
    assert [k for n, k, _ in log if n == 'text'] == [3, 2, 1, 0]
    # a generator that never yields (the fused dense path) just returns
    def plain():
        return 'done'
        yield
    assert HotPath._lockstep(plain()) == ['done']


@pytest.mark.parametrize('cfg_fn', [mini_config, lambda: __import__('image2text_amd.synth', fromlist=['tiny_config']).tiny_config()])
def test_dp_decoder_range_of_the_arena(cfg_fn):
    """DataParallelGrads._split: the decoder's parameters (laid out FIRST by named_parameters) form one contiguous arena range that
    ends where the encoder's begin -- also with the MoE regrouping and its pad entries"""
    from image2text_amd.engine import ParamArena
    from image2text_amd.training.dp import DataParallelGrads
    m = VisionEncoderDecoder(cfg_fn())
    arena = ParamArena(m, torch.device('cpu'))
    lo, hi = DataParallelGrads._split(None, arena)
    dec = [(o, n) for name, (o, n, _) in arena.entries.items() if name.startswith('decoder.')]
    enc = [(o, n) for name, (o, n, _) in arena.entries.items() if not name.startswith('decoder.')]
    assert lo == 0 and 0 < hi < arena.total
    assert all(lo <= o and o + n <= hi for o, n in dec) and all(o >= hi for o, n in enc)
