"""The nano-mini block family on the HIP kernels: multi-query attention, MoE rotators and sparse token subsets
(reference models/layers.py:285-346, 391-430, 489-518, 545-614; shipped as training_configs/gpu/nano-mini.yaml).

``FamilyBlocks`` is mixed into ``engine.HotPath``; a tower whose ``TransformerConfig.is_family`` is true runs its blocks through
``fam_layer_fwd`` / ``fam_layer_bwd`` instead of the dense ``block_fwd`` / ``block_bwd`` (which stay reserved for the tuned
multi-head / 64-wide / dense-MLP case).

How each reference feature maps to the hardware
  * MultiQueryAttention: q_proj and kv_proj are two GEMMs; the attention kernels (csrc/attention_g.hip) take H query heads on one
    shared K/V head and sum dK / dV over the heads in registers.  nn.MultiheadAttention with 128-wide heads uses the same kernels
    with H K/V heads.
  * MoELinear: every expert is a rank-P pair (l1: in -> P, l2: P -> out), so ALL experts of a layer are two GEMMs:
    U = x [l1_0; ..; l1_{E-1}; gate layer 0]^T (N = E P + G, one weight view thanks to the arena order) and y = A W2aug^T with
    A = [w_e gelu(U_e) | w] (K = E P + E, padded to 64); the routing kernel between them (csrc/family.hip) evaluates the gate's
    second layer, the softmax and the top-k in fp32 and zeroes the unchosen experts' panels, which reproduces the reference's
    gather / scatter over the chosen experts exactly.  Biases of l2 ride in the K panel.
  * sparse blocks: the kept positions of a layer are a fixed index set, so a sparse block is a dense block on gathered rows
    (row gather -> block -> row scatter) and the skipped rows take one GEMM (x + null_connector(x)).  With packed variable-length
    caption rows the per-layer row lists are built on the host from the caption lengths (a few kB per step).
"""
from types import SimpleNamespace
from typing import Optional

import numpy as np
import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def family_spec(tcfg, n_layer: int, force: bool = False):
    """Static description of one tower's blocks (None for the dense multi-head path).  force: a block the dense path could run is
    still described here (a non-causal decoder: its [prompt | text] forward needs the split visibility of the grouped kernels)."""
    if not (tcfg.is_family or force):
        return None
    ac = tcfg.attn_config
    rc = tcfg.rotator_config
    moe = None
    if hasattr(rc, 'num_experts'):
        gs = tuple(rc.gate_sizes or ())
        if len(gs) > 1:
            raise NotImplementedError('MoE gates deeper than one hidden layer are outside the HIP hot path (the routing kernel '
                                      'evaluates Linear or Linear-GELU-Linear gates)')
        if rc.proj_features % 8 or (gs and gs[0] % 8):
            raise NotImplementedError('MoE proj_features / gate width must be multiples of 8 (contiguous bf16 GEMM panels)')
        moe = SimpleNamespace(E=rc.num_experts, P=rc.proj_features, G=gs[0] if gs else 0, top_k=rc.top_k,
                              Kp=_round_up(rc.num_experts * (rc.proj_features + 1), 64))
    return SimpleNamespace(d=ac.n_embd, H=ac.n_head, hd=ac.n_embd // ac.n_head, mqa=ac.attn_type.value == 'multi_query', moe=moe,
                           sparse=bool(tcfg.is_sparse_attn), max_block=tcfg.max_block_size, causal=tcfg.is_causal)


def _expand_rows(base, positions, counts):
    """rows = concat_b (base[b] + positions[:counts[b]])  (all numpy, vectorised)"""
    total = int(counts.sum())
    if total == 0:
        return np.zeros(0, dtype=np.int32)
    seq = np.repeat(np.arange(counts.size), counts)
    start = np.cumsum(counts) - counts
    within = np.arange(total) - np.repeat(start, counts)
    return (base[seq] + positions[within]).astype(np.int32)


class FamilyBlocks:
    """Mixin of engine.HotPath (uses self.arena, self._empty, self._linear_bwd)."""

    # ------------------------------------------------------------------------------------------------ MoE parameters
    def _moe_views(self, pfx: str, sp):
        """Contiguous operand views of one MoELinear (arena order: engine._arena_order)."""
        a, m = self.arena, sp
        key = (pfx, a.p32.data_ptr())
        mv = self._moe_cache.get(key)
        if mv is not None:
            return mv
        E, P, G = m.E, m.P, m.G
        NG = G if G else E
        o_w, n_w, shp = a.entries[f'{pfx}.experts.0.l1.weight']
        in_f = shp[1]
        N1 = E * P + NG
        o_b = a.entries[f'{pfx}.experts.0.l1.bias'][0]
        o_w2, _, shp2 = a.entries[f'{pfx}.experts.0.l2.weight']
        out_f = shp2[0]
        o_b2 = a.entries[f'{pfx}.experts.0.l2.bias'][0]
        # the layout this relies on, checked once
        assert a.entries[f'{pfx}.expert_gates.model.0.weight'][0] == o_w + E * P * in_f
        assert a.entries[f'{pfx}.expert_gates.model.0.bias'][0] == o_b + E * P
        assert a.entries[f'{pfx}.experts.{E - 1}.l2.weight'][0] == o_w2 + (E - 1) * out_f * P
        assert a.entries[f'{pfx}.experts.{E - 1}.l2.bias'][0] == o_b2 + (E - 1) * out_f
        gate_bias = f'{pfx}.expert_gates.model.0.bias' in a.params
        mv = SimpleNamespace(
            in_f=in_f, out_f=out_f, N1=N1, NG=NG, gate_bias=gate_bias,
            W1=a.pbf[o_w:o_w + N1 * in_f].view(N1, in_f), gW1=a.g32[o_w:o_w + N1 * in_f].view(N1, in_f),
            b1=a.p32[o_b:o_b + N1], gb1=a.g32[o_b:o_b + N1],
            l2w=a.pbf[o_w2:o_w2 + E * out_f * P].view(E, out_f, P), gl2w=a.g32[o_w2:o_w2 + E * out_f * P].view(E, out_f, P),
            l2b=a.p32[o_b2:o_b2 + E * out_f].view(E, out_f), gl2b=a.g32[o_b2:o_b2 + E * out_f].view(E, out_f),
            wg2=a.P(f'{pfx}.expert_gates.model.2.weight') if G else None, bg2=a.P(f'{pfx}.expert_gates.model.2.bias') if G else None,
            gwg2=a.G(f'{pfx}.expert_gates.model.2.weight') if G else None,
            gbg2=(a.G(f'{pfx}.expert_gates.model.2.bias') if f'{pfx}.expert_gates.model.2.bias' in a.params else None) if G else None,
            W2aug=torch.zeros(out_f, m.Kp, dtype=BF16, device=a.device))
        if G and f'{pfx}.expert_gates.model.2.bias' not in a.params:
            mv.bg2 = None
        self._moe_cache[key] = mv
        return mv

    def moe_fwd(self, pfx: str, sp, x_bf, M: int, act: int, residual, drop, save: bool, out=None):
        """y = MoELinear(x) [+ GELU when act == 1] [+ residual, dropout]: returns (y, saved).  y is bf16 with act, else fp32."""
        mv = self._moe_views(pfx, sp)
        E, P, G = sp.E, sp.P, sp.G
        U = self._empty(M, mv.N1)
        ops.gemm(x_bf, mv.W1, U, M, mv.N1, mv.in_f, bias=mv.b1)
        A = self._empty(M, sp.Kp, dtype=BF16)
        gates, wsel = self._empty(M, E), self._empty(M, E)
        ops.moe_gate_fwd(U, mv.wg2, mv.bg2, A, gates, wsel, M, E, P, G, sp.top_k, mv.in_f ** -0.5)
        ops.moe_pack_w2(mv.l2w, mv.l2b, mv.W2aug, mv.out_f, E, P)            # (parameters may have changed since the last step)
        y = out if out is not None else self._empty(M, mv.out_f, dtype=BF16 if act else F32)
        pre = self._empty(M, mv.out_f, dtype=BF16) if (act and save) else None
        ops.gemm(A, mv.W2aug, y, M, mv.out_f, sp.Kp, act=act, aux_out=pre, residual=residual, drop=drop)
        if self.moe_trace is not None:                                       # tests: the routing decisions of every site, in call order
            self.moe_trace.setdefault(pfx, []).append((gates, wsel))
        return y, (SimpleNamespace(x=x_bf, U=U, A=A, gates=gates, wsel=wsel, pre=pre) if save else None)

    def moe_bwd(self, pfx: str, sp, sv, dy_bf, M: int, dx_out, **dx_kw):
        """dy_bf bf16 [M, out]: gradient w.r.t. the MoELinear output (before its residual).  Accumulates every parameter gradient
        of the layer; fills dx_out (through the GEMM epilogue options dx_kw, e.g. the GELU derivative of the producer)."""
        mv = self._moe_views(pfx, sp)
        E, P, G, Kp = sp.E, sp.P, sp.G, sp.Kp
        dA = self._empty(M, Kp, dtype=BF16)
        ops.gemm(dy_bf, mv.W2aug, dA, M, Kp, mv.out_f, b_kmajor=True)
        dW2 = torch.zeros(mv.out_f, Kp, dtype=F32, device=self.arena.device)      # accumulate form: the split-K path (K = M rows)
        ops.gemm(dy_bf, sv.A, dW2, mv.out_f, Kp, M, a_kmajor=True, b_kmajor=True, accumulate=True)
        ops.moe_unpack_dw2(dW2, mv.gl2w, mv.gl2b, mv.out_f, E, P)
        N1p = _round_up(mv.N1, 8)
        D1 = self._empty(M, N1p, dtype=BF16)
        part = self._empty(ops.moe_gate_bwd_blocks(M), E * G + E) if G else None
        ops.moe_gate_bwd(dA, sv.U, sv.gates, sv.wsel, mv.wg2, D1, mv.gwg2, mv.gbg2, part, M, E, P, G, sp.top_k, mv.in_f ** -0.5)
        ops.gemm(D1, sv.x, mv.gW1, mv.N1, mv.in_f, M, a_kmajor=True, b_kmajor=True, accumulate=True, lda=N1p)
        if mv.gate_bias:
            ops.colsum(D1, mv.gb1, M, mv.N1, accumulate=True)
        else:                                                                # the pad entry behind a bias-free gate stays zero
            ops.colsum(D1, mv.gb1[:E * P], M, E * P, accumulate=True)
        if dx_out is not None:
            ops.gemm(D1, mv.W1, dx_out, M, mv.in_f, mv.N1, b_kmajor=True, lda=N1p, **dx_kw)
        return dx_out

    # ------------------------------------------------------------------------------------------------ per-position MLP (wpe)
    def _pos_views(self):
        """AdvancedPositionalBiasMLP parameters (layers.py:617-638) as (base view of position 0, constant stride between positions)
        per layer: consecutive positions' MLPs are laid out identically, one after the other, in the arena."""
        a = self.arena
        key = ('pos', a.p32.data_ptr())
        pv = self._moe_cache.get(key)
        if pv is not None:
            return pv
        pfx = f'{self.dp}transformer.wpe.models'
        layers, i = [], 0
        while f'{pfx}.0.model.{i}.weight' in a.entries:
            ow, _, shp = a.entries[f'{pfx}.0.model.{i}.weight']
            ob = a.entries[f'{pfx}.0.model.{i}.bias'][0]
            layers.append(SimpleNamespace(N=shp[0], K=shp[1], W=a.pbf[ow:], gW=a.g32[ow:], b=a.p32[ob:], gb=a.g32[ob:]))
            i += 2
        n_pos = self.dec.block
        stride = a.entries[f'{pfx}.1.model.0.weight'][0] - a.entries[f'{pfx}.0.model.0.weight'][0] if n_pos > 1 else 0
        for t in (1, n_pos - 1):                              # the layout this relies on, checked once
            for j, L in enumerate(layers):
                if n_pos > 1:
                    assert a.entries[f'{pfx}.{t}.model.{2 * j}.weight'][0] - a.entries[f'{pfx}.0.model.{2 * j}.weight'][0] == t * stride
                    assert a.entries[f'{pfx}.{t}.model.{2 * j}.bias'][0] - a.entries[f'{pfx}.0.model.{2 * j}.bias'][0] == t * stride
        pv = SimpleNamespace(layers=layers, stride=stride)
        self._moe_cache[key] = pv
        return pv

    def _pos_plan(self, B: int, T: int, vl):
        """Position-major row order of a batch: rows of position 0 first, then position 1, ...  -> (row index list, segment table)."""
        key = ('posplan', B, T) if vl is None else None
        if key is not None and key in self._sub_cache:
            return self._sub_cache[key]
        dev = self.arena.device
        if vl is None:
            rows = (np.arange(T, dtype=np.int64)[:, None] + np.arange(B, dtype=np.int64)[None] * T).ravel()
            seg = np.arange(T + 1, dtype=np.int64) * B
            max_rows = B
        else:
            lens = np.asarray(vl.lens_host, dtype=np.int64)
            cu = np.zeros(B + 1, dtype=np.int64)
            cu[1:] = np.cumsum(lens)
            order = np.argsort(-lens, kind='stable')                      # sequences by decreasing length: position t holds the first n_t
            n_t = (lens[None, :] > np.arange(T)[:, None]).sum(axis=1)       # sequences that reach position t
            seg = np.zeros(T + 1, dtype=np.int64)
            seg[1:] = np.cumsum(n_t)
            rows = _expand_rows(np.arange(T, dtype=np.int64), cu[order], n_t).astype(np.int64) if int(n_t.sum()) else np.zeros(0, dtype=np.int64)
            max_rows = int(n_t.max()) if T else 0
        plan = SimpleNamespace(rows=torch.from_numpy(rows.astype(np.int32)).to(dev), seg=torch.from_numpy(seg.astype(np.int32)).to(dev),
                               max_rows=max(max_rows, 1), G=T, M=int(rows.size))
        if key is not None:
            self._sub_cache[key] = plan
        return plan

    def posmlp_fwd(self, x_emb, B: int, T: int, pos_offset: int, vl, save: bool):
        """x[row] = MLP_p(e[row]) + e[row] with p = pos_offset + position of the row (decoder.py:231-232).  x_emb fp32 [M, d]."""
        pv = self._pos_views()
        plan = self._pos_plan(B, T, vl)
        M, d = plan.M, self.dec.d
        e32, ebf = self._empty(M, d), self._empty(M, d, dtype=BF16)
        ops.gather_rows(x_emb, plan.rows, M, d, out_f32=e32, out_bf16=ebf)
        h, hs, pres = ebf, [ebf], []
        kw = dict(seg=plan.seg, n_groups=plan.G, max_rows=plan.max_rows, group0=pos_offset, b_group_stride=pv.stride, bias_group_stride=pv.stride)
        for i, L in enumerate(pv.layers):
            last = i == len(pv.layers) - 1
            out = self._empty(M, L.N, dtype=F32 if last else BF16)
            pre = self._empty(M, L.N, dtype=BF16) if (save and not last) else None
            ops.grouped_gemm(0, h, L.W[:L.N * L.K].view(L.N, L.K), out, L.N, L.K, bias=L.b, act=0 if last else 1, aux_out=pre,
                             residual=e32 if last else None, **kw)
            if not last:
                hs.append(out)
                pres.append(pre)
            h = out
        x = self._empty(M, d)
        ops.scatter_rows(h, plan.rows, x, M, d)
        return x, (SimpleNamespace(plan=plan, hs=hs, pres=pres, pos_offset=pos_offset) if save else None)

    def posmlp_bwd(self, sv, dx):
        """dx fp32 [M, d] (packed order): accumulates the gradients of every position's MLP, returns d/d(embedding) fp32 [M, d]."""
        pv, plan = self._pos_views(), sv.plan
        M, d = plan.M, self.dec.d
        dy32, g = self._empty(M, d), self._empty(M, d, dtype=BF16)
        ops.gather_rows(dx, plan.rows, M, d, out_f32=dy32, out_bf16=g)
        kw = dict(seg=plan.seg, n_groups=plan.G, max_rows=plan.max_rows, group0=sv.pos_offset)
        de = None
        for i in reversed(range(len(pv.layers))):
            L = pv.layers[i]
            ops.grouped_gemm(2, g, sv.hs[i], L.gW[:L.N * L.K].view(L.N, L.K), L.N, L.K, c_group_stride=pv.stride, accumulate=True, **kw)
            ops.grouped_colsum(g, plan.seg, plan.G, L.gb, pv.stride, sv.pos_offset, L.N)
            W = L.W[:L.N * L.K].view(L.N, L.K)
            if i > 0:
                nxt = self._empty(M, L.K, dtype=BF16)
                ops.grouped_gemm(1, g, W, nxt, L.N, L.K, b_group_stride=pv.stride, act=2, aux_in=sv.pres[i - 1], **kw)
                g = nxt
            else:
                de = self._empty(M, d)
                ops.grouped_gemm(1, g, W, de, L.N, L.K, b_group_stride=pv.stride, residual=dy32, **kw)
        out = self._empty(M, d)
        ops.scatter_rows(de, plan.rows, out, M, d)
        return out

    # ------------------------------------------------------------------------------------------------ one block
    def fam_block_fwd(self, pfx: str, sp, x, B, T, mem_bf, S, save: bool, plan, layer: int, vl, split: int = 0, out=None):
        """out: fp32 [M, d] buffer the block output is written to (a sparse layer hands in its slice of the layer output)."""
        a = self.arena
        d, H, hd = sp.d, sp.H, sp.hd
        M = vl.total if vl is not None else B * T
        cu = vl.cu if vl is not None else None
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        dr = {k: (plan.get(layer, k) if plan is not None else None) for k in ('qkv', 'sdpa', 'resid', 'xattn', 'mlp')}
        sv = SimpleNamespace(x=x, cross=False, dr=dr)
        ln1, m1, r1 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x, a.P(f'{pfx}.ln_1.weight'), a.P(f'{pfx}.ln_1.bias'), ln1, m1, r1, M, d)
        if sp.mqa:
            q, kv = self._empty(M, d, dtype=BF16), self._empty(M, 2 * hd, dtype=BF16)
            ops.gemm(ln1, a.W(f'{pfx}.attn.q_proj.weight'), q, M, d, d, bias=a.P(f'{pfx}.attn.q_proj.bias'))
            ops.gemm(ln1, a.W(f'{pfx}.attn.kv_proj.weight'), kv, M, 2 * hd, d, bias=a.P(f'{pfx}.attn.kv_proj.bias'))
            ops.row_sections_dropout(q, M, d, d, dr['qkv'], 0)                 # per-token multipliers: q, then k | v
            ops.row_sections_dropout(kv, M, 2 * hd, hd, dr['qkv'], 1)
            kv3 = v3(kv, 2 * hd)
            qq, kk, vv, Hkv, out_name = v3(q, d), kv3[..., :hd], kv3[..., hd:], 1, 'attn.out_proj'
            sv.q, sv.kv = q, kv
        else:
            qkv = self._empty(M, 3 * d, dtype=BF16)
            ops.gemm(ln1, a.W(f'{pfx}.attn.c_attn.weight'), qkv, M, 3 * d, d, bias=a.P(f'{pfx}.attn.c_attn.bias'), drop=dr['qkv'])
            q3 = v3(qkv, 3 * d)
            qq, kk, vv, Hkv, out_name = q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], H, 'attn.c_proj'
            sv.qkv = qkv
        ao, lse = self._empty(M, d, dtype=BF16), self._empty(H * M)
        ops.gq_attention_fwd(qq, kk, vv, v3(ao, d), lse, B, H, Hkv, hd, T, T, sp.causal, drop=dr['sdpa'], cu_q=cu, cu_k=cu, total_q=M,
                             split=split)
        x1 = self._empty(M, d)
        ops.gemm(ao, a.W(f'{pfx}.{out_name}.weight'), x1, M, d, d, bias=a.P(f'{pfx}.{out_name}.bias'), residual=x, drop=dr['resid'])
        sv.ln1, sv.m1, sv.r1, sv.ao, sv.lse, sv.x1, sv.out_name, sv.Hkv = ln1, m1, r1, ao, lse, x1, out_name, Hkv
        x2 = x1
        if mem_bf is not None:
            if f'{pfx}.cross_attn.in_proj_weight' not in a.entries:
                raise ValueError('Model not configured for cross attn inputs!!!')         # reference layers.py:598-599
            win, bin_ = a.W(f'{pfx}.cross_attn.in_proj_weight'), a.P(f'{pfx}.cross_attn.in_proj_bias')
            ln3, m3, r3 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
            ops.layernorm_fwd(x1, a.P(f'{pfx}.ln_3.weight'), a.P(f'{pfx}.ln_3.bias'), ln3, m3, r3, M, d)
            qc = self._empty(M, d, dtype=BF16)
            ops.gemm(ln3, win[:d], qc, M, d, d, bias=bin_[:d])
            kvc = self._empty(B, S, 2 * d, dtype=BF16)
            ops.gemm(mem_bf, win[d:], kvc.view(B * S, 2 * d), B * S, 2 * d, d, bias=bin_[d:])
            co, lse_c = self._empty(M, d, dtype=BF16), self._empty(H * M)
            ops.gq_attention_fwd(v3(qc, d), kvc[..., :d], kvc[..., d:], v3(co, d), lse_c, B, H, H, hd, T, S, False, drop=dr['xattn'],
                                 cu_q=cu, total_q=M)
            x2 = self._empty(M, d)
            ops.gemm(co, a.W(f'{pfx}.cross_attn.out_proj.weight'), x2, M, d, d, bias=a.P(f'{pfx}.cross_attn.out_proj.bias'), residual=x1)
            sv.cross, sv.ln3, sv.m3, sv.r3, sv.qc, sv.kvc, sv.co, sv.lse_c, sv.mem = True, ln3, m3, r3, qc, kvc, co, lse_c, mem_bf
        ln2, m2, r2 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x2, a.P(f'{pfx}.ln_2.weight'), a.P(f'{pfx}.ln_2.bias'), ln2, m2, r2, M, d)
        if sp.moe is not None:
            h, sv.fc = self.moe_fwd(f'{pfx}.mlp.c_fc', sp.moe, ln2, M, 1, None, None, save)
            x3, sv.pj = self.moe_fwd(f'{pfx}.mlp.c_proj', sp.moe, h, M, 0, x2, dr['mlp'], save, out=out)
        else:
            ff = a.entries[f'{pfx}.mlp.c_fc.weight'][2][0]
            h = self._empty(M, ff, dtype=BF16)
            pre = self._empty(M, ff, dtype=BF16) if save else None
            ops.gemm(ln2, a.W(f'{pfx}.mlp.c_fc.weight'), h, M, ff, d, bias=a.P(f'{pfx}.mlp.c_fc.bias'), act=1, aux_out=pre)
            x3 = out if out is not None else self._empty(M, d)
            ops.gemm(h, a.W(f'{pfx}.mlp.c_proj.weight'), x3, M, d, ff, bias=a.P(f'{pfx}.mlp.c_proj.bias'), residual=x2, drop=dr['mlp'])
            sv.h, sv.pre, sv.ff = h, pre, ff
        sv.x2, sv.ln2, sv.m2, sv.r2 = x2, ln2, m2, r2
        return x3, (sv if save else None)

    def fam_block_bwd(self, pfx: str, sp, sv, dx, B, T, S, dmem, vl, ws=None):
        """dx fp32 [M, d]: gradient w.r.t. the block output on entry (un-normalised), w.r.t. the block input on return.
        ws: 1 float holding sum(g^2) of the WHOLE tensor the reference normalises here (None: dx is that tensor)."""
        a = self.arena
        d, H, hd = sp.d, sp.H, sp.hd
        M = vl.total if vl is not None else B * T
        cu = vl.cu if vl is not None else None
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        dr = sv.dr
        bias = lambda n: n if a.G(n) is not None and n in a.params else None
        # normalize_gradients at the block output (layers.py:606-607); the bf16 copy feeds the MLP branch -> carries its dropout mask
        dxb = self._empty(M, d, dtype=BF16)
        ops.grad_normalize(dx, self._empty(1) if ws is None else ws, dxb, bf16_drop=dr['mlp'], presummed=ws is not None)
        dln = self._empty(M, d, dtype=BF16)
        if sp.moe is not None:
            ff = sv.fc.pre.shape[1]
            dpre = self._empty(M, ff, dtype=BF16)
            self.moe_bwd(f'{pfx}.mlp.c_proj', sp.moe, sv.pj, dxb, M, dpre, act=2, aux_in=sv.fc.pre)
            self.moe_bwd(f'{pfx}.mlp.c_fc', sp.moe, sv.fc, dpre, M, dln)
        else:
            dpre = self._empty(M, sv.ff, dtype=BF16)
            self._linear_bwd(dxb, M, d, sv.ff, sv.h, f'{pfx}.mlp.c_proj.weight', bias(f'{pfx}.mlp.c_proj.bias'), dx_out=dpre, act=2,
                             aux_in=sv.pre)
            self._linear_bwd(dpre, M, sv.ff, d, sv.ln2, f'{pfx}.mlp.c_fc.weight', bias(f'{pfx}.mlp.c_fc.bias'), dx_out=dln)
        ops.layernorm_bwd(dln, sv.x2, a.P(f'{pfx}.ln_2.weight'), sv.m2, sv.r2, dx, a.G(f'{pfx}.ln_2.weight'), a.G(f'{pfx}.ln_2.bias'),
                          M, d, dx_accumulate=True, dx_bf16=dxb, bf16_drop=None if sv.cross else dr['resid'])
        ws = self._empty(H * M)
        if sv.cross:
            win = a.W(f'{pfx}.cross_attn.in_proj_weight')
            gin, gbin = a.G(f'{pfx}.cross_attn.in_proj_weight'), a.G(f'{pfx}.cross_attn.in_proj_bias')
            dco = self._empty(M, d, dtype=BF16)
            self._linear_bwd(dxb, M, d, d, sv.co, f'{pfx}.cross_attn.out_proj.weight', f'{pfx}.cross_attn.out_proj.bias', dx_out=dco)
            dq, dkv = self._empty(M, d, dtype=BF16), self._empty(B, S, 2 * d, dtype=BF16)
            ops.gq_attention_bwd(v3(sv.qc, d), sv.kvc[..., :d], sv.kvc[..., d:], v3(sv.co, d), v3(dco, d), sv.lse_c, ws, v3(dq, d),
                                 dkv[..., :d], dkv[..., d:], B, H, H, hd, T, S, False, drop=dr['xattn'], cu_q=cu, total_q=M)
            dkvf = dkv.view(B * S, 2 * d)
            ops.colsum(dq, gbin[:d], M, d, accumulate=True)
            ops.gemm(dq, sv.ln3, gin[:d], d, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            ops.gemm(dq, win[:d], dln, M, d, d, b_kmajor=True)
            ops.colsum(dkvf, gbin[d:], B * S, 2 * d, accumulate=True)
            ops.gemm(dkvf, sv.mem, gin[d:], 2 * d, d, B * S, a_kmajor=True, b_kmajor=True, accumulate=True)
            ops.gemm(dkvf, win[d:], dmem, B * S, d, 2 * d, b_kmajor=True, accumulate=True)
            ops.layernorm_bwd(dln, sv.x1, a.P(f'{pfx}.ln_3.weight'), sv.m3, sv.r3, dx, a.G(f'{pfx}.ln_3.weight'),
                              a.G(f'{pfx}.ln_3.bias'), M, d, dx_accumulate=True, dx_bf16=dxb, bf16_drop=dr['resid'])
        dao = self._empty(M, d, dtype=BF16)
        self._linear_bwd(dxb, M, d, d, sv.ao, f'{pfx}.{sv.out_name}.weight', bias(f'{pfx}.{sv.out_name}.bias'), dx_out=dao)
        if sp.mqa:
            dq, dkv = self._empty(M, d, dtype=BF16), self._empty(M, 2 * hd, dtype=BF16)
            kv3, g3 = v3(sv.kv, 2 * hd), v3(dkv, 2 * hd)
            ops.gq_attention_bwd(v3(sv.q, d), kv3[..., :hd], kv3[..., hd:], v3(sv.ao, d), v3(dao, d), sv.lse, ws, v3(dq, d), g3[..., :hd],
                                 g3[..., hd:], B, H, 1, hd, T, T, sp.causal, drop=dr['sdpa'], cu_q=cu, cu_k=cu, total_q=M,
                                 out_drop=dr['qkv'])
            dl32 = self._empty(M, d)                                         # two projections feed ln_1: summed in fp32
            self._linear_bwd(dq, M, d, d, sv.ln1, f'{pfx}.attn.q_proj.weight', bias(f'{pfx}.attn.q_proj.bias'), dx_out=dl32)
            self._linear_bwd(dkv, M, 2 * hd, d, sv.ln1, f'{pfx}.attn.kv_proj.weight', bias(f'{pfx}.attn.kv_proj.bias'), dx_out=dl32,
                             accumulate=True)
            dln1 = dl32
        else:
            dqkv = self._empty(M, 3 * d, dtype=BF16)
            q3, g3 = v3(sv.qkv, 3 * d), v3(dqkv, 3 * d)
            ops.gq_attention_bwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], v3(sv.ao, d), v3(dao, d), sv.lse, ws, g3[..., :d],
                                 g3[..., d:2 * d], g3[..., 2 * d:], B, H, H, hd, T, T, sp.causal, drop=dr['sdpa'], cu_q=cu, cu_k=cu,
                                 total_q=M, out_drop=dr['qkv'])
            self._linear_bwd(dqkv, M, 3 * d, d, sv.ln1, f'{pfx}.attn.c_attn.weight', bias(f'{pfx}.attn.c_attn.bias'), dx_out=dln)
            dln1 = dln
        ops.layernorm_bwd(dln1, sv.x, a.P(f'{pfx}.ln_1.weight'), sv.m1, sv.r1, dx, a.G(f'{pfx}.ln_1.weight'), a.G(f'{pfx}.ln_1.bias'),
                          M, d, dx_accumulate=True)

    # ------------------------------------------------------------------------------------------------ sparse layers
    def sparse_subset(self, tower: str, layer: int, B: int, T: int, off: int, vl):
        """Row lists of one sparse layer for a batch of B sequences whose rows hold positions off .. off + T - 1 (vl: packed rows
        with per-sequence lengths).  None for a dense layer."""
        idx_all, not_all = self._sparse_idx[tower][layer] if self._sparse_idx[tower] else (None, None)
        if idx_all is None:
            return None
        key = (tower, layer, B, T, off) if vl is None else None
        if key is not None and key in self._sub_cache:
            return self._sub_cache[key]
        dev = self.arena.device
        t_full = off + T
        if t_full > idx_all.size + not_all.size:
            raise AssertionError(f'sequence of {t_full} positions exceeds max_block_size {idx_all.size + not_all.size} of the sparse blocks')
        kept = idx_all[idx_all < t_full]
        sub = SimpleNamespace(all_null=False)
        idx_t = kept[kept >= off] - off
        if kept.size <= 1 or idx_t.size == 0:                                # layers.py:572-573: the whole input takes the null path
            sub.all_null = True
        else:
            not_t = not_all[(not_all < t_full) & (not_all >= off)] - off
            sub.idx_t = idx_t
            if vl is None:
                base = (np.arange(B, dtype=np.int64) * T)[:, None]
                rows_in, rows_out = (base + idx_t[None]).ravel(), (base + not_t[None]).ravel()
                sub.T_in, sub.vl_in = int(idx_t.size), None
            else:
                lens, cu = np.asarray(vl.lens_host, dtype=np.int64), np.zeros(B + 1, dtype=np.int64)
                cu[1:] = np.cumsum(lens)
                cnt_in, cnt_out = np.searchsorted(idx_t, lens), np.searchsorted(not_t, lens)
                rows_in, rows_out = _expand_rows(cu[:-1], idx_t, cnt_in), _expand_rows(cu[:-1], not_t, cnt_out)
                cu_in = np.zeros(B + 1, dtype=np.int32)
                cu_in[1:] = np.cumsum(cnt_in)
                sub.T_in = int(cnt_in.max()) if B else 0
                sub.vl_in = SimpleNamespace(cu=torch.from_numpy(cu_in).to(dev), total=int(rows_in.size), pos=None)
            sub.n_in, sub.n_out = int(rows_in.size), int(rows_out.size)
            sub.rows_in = torch.from_numpy(rows_in.astype(np.int32)).to(dev)
            sub.rows_out = torch.from_numpy(rows_out.astype(np.int32)).to(dev)
            # the layer's output lives as [block rows | null rows] ("pieces"); perm[r] = where row r of the sequence order sits in it
            perm = np.empty(sub.n_in + sub.n_out, dtype=np.int32)
            perm[rows_in] = np.arange(sub.n_in, dtype=np.int32)
            perm[rows_out] = sub.n_in + np.arange(sub.n_out, dtype=np.int32)
            sub.perm = torch.from_numpy(perm).to(dev)
        if key is not None:
            self._sub_cache[key] = sub
        return sub

    def _null_fwd(self, pfx: str, x, n: int, d: int, rows, out=None):
        """x + null_connector(x) on the rows `rows` of x (all rows when None) -> (fp32 [n, d] (= out when given), bf16 copy of the inputs)"""
        a = self.arena
        xb = self._empty(n, d, dtype=BF16)
        if rows is None:
            xn = x
            ops.cast_f32_bf16(x, xb)
        else:
            xn = self._empty(n, d)
            ops.gather_rows(x, rows, n, d, out_f32=xn, out_bf16=xb)
        yn = out if out is not None else self._empty(n, d)
        ops.gemm(xb, a.W(f'{pfx}.null_connector.weight'), yn, n, d, d, bias=a.P(f'{pfx}.null_connector.bias'), residual=xn)
        return yn, xb

    def _null_bwd(self, pfx: str, xb, dy, n: int, d: int, rows, out=None):
        """gradient of x + null_connector(x): parameter gradients accumulated, returns d/dx fp32 [n, d] (= out when given)"""
        a = self.arena
        dyb = self._empty(n, d, dtype=BF16)
        if rows is None:
            dyn = dy
            ops.cast_f32_bf16(dy, dyb)
        else:
            dyn = self._empty(n, d)
            ops.gather_rows(dy, rows, n, d, out_f32=dyn, out_bf16=dyb)
        dxn = out if out is not None else self._empty(n, d)
        nb = f'{pfx}.null_connector.bias'
        self._linear_bwd(dyb, n, d, d, xb, f'{pfx}.null_connector.weight', nb if nb in a.params else None, dx_out=dxn, residual=dyn)
        return dxn

    @staticmethod
    def _compose(perm, rows):
        """rows of the sequence order -> rows of a tensor stored in `perm` order (None = sequence order)"""
        return rows if perm is None else perm[rows.long()]

    def materialize(self, x, perm):
        """A residual stream stored as a sparse layer's pieces -> sequence order (one gather pass)."""
        if perm is None:
            return x
        out = self._empty(*x.shape)
        ops.gather_rows(x, perm, x.shape[0], x.shape[1], out_f32=out)
        return out

    def fam_layer_fwd(self, pfx: str, sp, x, xperm, B, T, mem_bf, S, save: bool, plan, layer: int, vl, sub, split: int = 0):
        """One layer on the residual stream (x, xperm): x fp32 [M, d] holds row r of the sequence order at x[xperm[r]] (xperm None:
        in place).  A sparse layer never scatters: its block rows and its null-connector rows are written next to each other
        ("pieces") and the NEXT consumer gathers through the composed index, so the stream costs one gather per path and layer
        instead of a gather and a scatter.  Returns (out, saved, out_perm).
        split > 0 (forward only): the rows are [prompt | text] of a non-causal decoder, text rows must not see prompt keys."""
        if sub is None:
            x = self.materialize(x, xperm)
            y, bsv = self.fam_block_fwd(pfx, sp, x, B, T, mem_bf, S, save, plan, layer, vl, split)
            return y, bsv, None
        if split and not sub.all_null:
            split = int((sub.idx_t < split).sum())          # the prompt positions kept by this layer come first in the subset
        d, M = sp.d, x.shape[0]
        if sub.all_null or sub.n_in == 0:
            y, xb = self._null_fwd(pfx, x, M, d, xperm)
            return y, (SimpleNamespace(only_null=True, xb=xb) if save else None), None
        pieces = self._empty(M, d)
        xs = self._empty(sub.n_in, d)
        ops.gather_rows(x, self._compose(xperm, sub.rows_in), sub.n_in, d, out_f32=xs)
        _, bsv = self.fam_block_fwd(pfx, sp, xs, B, sub.T_in, mem_bf, S, save, plan, layer, sub.vl_in, split, out=pieces[:sub.n_in])
        xnb = None
        if sub.n_out:
            _, xnb = self._null_fwd(pfx, x, sub.n_out, d, self._compose(xperm, sub.rows_out), out=pieces[sub.n_in:])
        return pieces, (SimpleNamespace(only_null=False, block=bsv, xnb=xnb, sub=sub) if save else None), sub.perm

    def fam_layer_bwd_steps(self, key, pfx: str, sp, sv, dx, dperm, B, T, S, dmem, vl):
        """Generator form of the layer backward on the gradient stream (dx, dperm) (same storage convention as the forward stream):
        yields (key, sum(g^2) of this segment's part of the block-output gradient) right before the gradient normaliser and is sent
        back the sum to normalise with (engine.HotPath._lockstep); returns (gradient w.r.t. the layer input, its perm)."""
        if not hasattr(sv, 'only_null'):
            dx = self.materialize(dx, dperm)
            joint = yield key, ops.sumsq(dx, self._empty(1))
            self.fam_block_bwd(pfx, sp, sv, dx, B, T, S, dmem, vl, ws=joint)
            return dx, None
        d, M = sp.d, dx.shape[0]
        if sv.only_null:
            return self._null_bwd(pfx, sv.xb, dx, M, d, dperm), None
        sub = sv.sub
        dpieces = self._empty(M, d)
        dys = dpieces[:sub.n_in]
        ops.gather_rows(dx, self._compose(dperm, sub.rows_in), sub.n_in, d, out_f32=dys)
        joint = yield key, ops.sumsq(dys, self._empty(1))
        self.fam_block_bwd(pfx, sp, sv.block, dys, B, sub.T_in, S, dmem, sub.vl_in, ws=joint)
        if sub.n_out:
            self._null_bwd(pfx, sv.xnb, dx, sub.n_out, d, self._compose(dperm, sub.rows_out), out=dpieces[sub.n_in:])
        return dpieces, sub.perm

    def fam_layer_bwd(self, pfx: str, sp, sv, dx, dperm, B, T, S, dmem, vl):
        return self._lockstep(self.fam_layer_bwd_steps(0, pfx, sp, sv, dx, dperm, B, T, S, dmem, vl))[0]
