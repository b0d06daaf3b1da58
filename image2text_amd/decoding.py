"""Greedy decoding with a static KV cache under hipGraph replay.

Replaces the reference's cache-free sampling loop (models/vision_encoder_decoder.py:136-182 with ``top_k=1``,
``temperature=1``): there every new token re-runs the whole decoder over all t tokens and synchronises with the host
for the n-gram ban (``.tolist()``).  Here
  * the encoder runs once, the cross-attention K/V of every cross layer are projected once per image;
  * one decode step = one token per caption through single-row kernels: LayerNorm -> QKV GEMM -> append K/V to the
    cache -> attention of the new query against keys 0..pos -> projections / MLP with fused residuals -> ln_f ->
    tied lm_head (fp32 logits) -> on-device no-repeat-n-gram ban + argmax that appends the token to the id buffer;
  * every position-dependent kernel reads ``pos`` / ``len`` from device memory, so the step is captured ONCE into a
    hipGraph and replayed per token: no host sync, no per-step launch overhead, token ids never leave the GPU until
    the end.
Because text rows never attend to the soft-prompt columns (see engine.py) the cache holds text positions only; the
prompt shifts the position embedding by n_cls.

The sampling modes of ``generate`` (temperature / top-k / nucleus, reference :152-180; ``eval_model``'s call shape
trainer.py:41-56) run on the same cached step: only its last kernel differs -- ``i2t_sample_token`` (csrc/sample.hip)
instead of the ban + argmax -- so a sampled token also costs one hipGraph replay and no host sync.  The draw is a
counter-based function of (seed, step, row): ``Sampling.seed`` comes from torch's CPU generator once per call, so
``torch.manual_seed`` reproduces a run.
"""
import os
from types import SimpleNamespace
from typing import NamedTuple, Optional

import torch

from . import ops
from .engine import BF16, F32, HotPath


ENC_CHUNK = int(os.environ.get('I2T_DECODE_ENC_CHUNK', '4096'))      # images per encoder pass inside generate()
TOP2_HEAD = os.environ.get('I2T_DECODE_TOP2', '1') not in ('', '0')    # greedy steps: lm_head + argmax through segment maxima (ops.gemm_top2)


class Sampling(NamedTuple):
    """How the next token is chosen when it is not the argmax (arguments of the reference's generate, :136-137)."""
    temperature: float = 1.0
    top_k: Optional[int] = None
    nucleus_p: Optional[float] = None
    seed: Optional[int] = None          # 64-bit; None: drawn from torch's CPU generator at the start of the call

    def key(self):
        return (float(self.temperature), int(self.top_k or 0), None if self.nucleus_p is None else float(self.nucleus_p))


class GreedyDecoder:
    """KV-cache decoder; greedy by default, ``generate(..., sampling=Sampling(...))`` draws tokens on the device instead."""

    def __init__(self, model):
        self.model = model
        self.eng: HotPath = model._engine
        self._state = None

    # ------------------------------------------------------------------------------------------------ buffers
    def _build(self, B: int, ids_ld: int):
        eng, a = self.eng, self.eng.arena
        dc = eng.dec
        dev = a.device
        cfg = self.model.config
        ncls = eng.enc.ncls
        off = ncls if cfg.use_soft_prompting else 0
        tmax = dc.block - off
        d, ff = dc.d, dc.ff
        if dc.llama is not None:        # i2t_gq_decode_attention walks at most 1024 cached keys per caption (prompt rows included)
            tmax = min(tmax, 1024 - min(ncls, dc.block))
        # Hugging Face decoder + soft prompt (engine.decode_prefixed): the encoder outputs are the first cache positions of every caption
        prefix = min(ncls, dc.block) if dc.prefixed else 0
        st = SimpleNamespace(B=B, ids_ld=ids_ld, off=off, tmax=tmax, arena=a, sparse_epoch=eng.sparse_epoch, prefix=prefix,
                             clen=tmax + prefix)
        e = lambda *s, dtype=BF16: torch.zeros(*s, dtype=dtype, device=dev)
        st.ids = torch.zeros(B, ids_ld, dtype=torch.long, device=dev)
        st.counters = torch.zeros(2, dtype=torch.int32, device=dev)        # [pos, len]
        st.counters_init = torch.tensor([prefix, 1], dtype=torch.int32, device=dev)
        st.x = e(B, d, dtype=F32)
        st.ln = e(B, d)
        st.qkv = e(B, 3 * d)
        st.ao = e(B, d)
        st.q = e(B, d)
        st.h = e(B, ff)
        st.hid = e(B, d)
        st.logits = e(B, dc.Vp, dtype=F32)                      # rows padded to 8 columns: 16-byte aligned rows for the GEMM epilogue
        st.margin = e(B, dtype=F32)
        # fp32 planes of the deterministic split-K GEMM form (ops.gemm(workspace=...)): the step's GEMMs with few output tiles and a
        # long K (attention / MLP output projections at anything but the largest caption batches) spread over K slices
        st.ws = torch.empty(16 << 20, dtype=F32, device=dev)
        if dc.llama is not None:
            ls = dc.llama
            st.qkv, st.ao, st.gu = e(B, (ls.H + 2 * ls.Hkv) * ls.hd), e(B, ls.H * ls.hd), e(B, 2 * ff)
            st.kc = [e(B, st.clen, ls.Hkv * ls.hd) for _ in range(dc.L)]
            st.vc = [e(B, st.clen, ls.Hkv * ls.hd) for _ in range(dc.L)]
        elif dc.fam is None:
            st.kc = [e(B, st.clen, d) for _ in range(dc.L)]
            st.vc = [e(B, st.clen, d) for _ in range(dc.L)]
        S = ncls
        st.cross_kv = {l: (e(B, S, 2 * d), S) for l in self._cross_layers()}
        if dc.fam is not None:
            self._build_family(st, e)
        if dc.advpos:
            st.pos_h = [e(B, g) for g in (self.eng.dcfg.advanced_pos_emb_gate_sizes or ())]
        st.ngrams = torch.tensor(list(cfg.no_repeat_n_grams), dtype=torch.int32, device=dev)
        st.graphs = {}                      # None -> prefill step, 'greedy' / Sampling.key() -> full step
        st.seed = torch.zeros(2, dtype=torch.int32, device=dev)
        st.dist = None                      # optional [B, V] f32: the distribution of the last sampled step (tests)
        return st

    def _build_family(self, st, e):
        """Buffers of the nano-mini family's decode step (engine_family.py): a K/V cache of Hkv heads per layer that, in a sparse
        layer, holds the kept positions only (slot = number of kept positions before the token), the per-layer slot / membership
        tables, and the MoE routing workspaces."""
        import numpy as np
        eng, dc = self.eng, self.eng.dec
        sp, dev = dc.fam, st.arena.device
        B, d, hd, tmax, off = st.B, dc.d, sp.hd, st.tmax, st.off
        st.Hkv = 1 if sp.mqa else dc.H
        st.slots = tmax
        st.sparse = sp.sparse
        if sp.sparse:
            rank, member = np.zeros((dc.L, tmax), dtype=np.int32), np.zeros((dc.L, tmax), dtype=np.int32)
            for l, (idx, _not) in enumerate(eng._sparse_idx['dec']):
                if int((idx < off + 1).sum()) < 2:
                    # layers.py:572-573 sends the WHOLE sequence through the null connector while <= 1 position is kept; a token's
                    # state would then depend on the current length and could not be cached
                    raise NotImplementedError('KV-cache generation with sparse decoder blocks needs >= 2 kept positions before the '
                                              'first text token (a soft prompt of >= 2 encoder outputs)')
                if off + tmax > idx.size + _not.size:
                    raise AssertionError('block_size + prompt exceeds max_block_size of the sparse decoder blocks')
                text = np.zeros(off + tmax, dtype=np.int32)
                text[idx[idx < off + tmax]] = 1
                member[l] = text[off:]
                rank[l] = np.cumsum(member[l]) - member[l]
            st.slots = max(int(member.sum(axis=1).max()), 1)
            st.rank, st.member = torch.from_numpy(rank).to(dev), torch.from_numpy(member).to(dev)
            st.lpos = torch.zeros(dc.L, dtype=torch.int32, device=dev)
            st.lmem = torch.zeros(dc.L, dtype=torch.int32, device=dev)
            st.xb, st.xn, st.xnb = e(B, d, dtype=F32), e(B, d, dtype=F32), e(B, d)
        w = st.Hkv * hd
        st.kc = [e(B, st.slots, w) for _ in range(dc.L)]
        st.vc = [e(B, st.slots, w) for _ in range(dc.L)]
        st.kvn = e(B, 2 * w)
        if sp.moe is not None:
            m = sp.moe
            st.U = e(B, m.E * m.P + (m.G if m.G else m.E), dtype=F32)
            st.A = e(B, m.Kp)
            st.gates, st.wsel = e(B, m.E, dtype=F32), e(B, m.E, dtype=F32)

    def _moe_step(self, st, pfx: str, x_bf, y, act: int, residual):
        eng, m = self.eng, self.eng.dec.fam.moe
        mv = eng._moe_views(pfx, m)
        ops.gemm(x_bf, mv.W1, st.U, st.B, mv.N1, mv.in_f, bias=mv.b1)
        ops.moe_gate_fwd(st.U, mv.wg2, mv.bg2, st.A, st.gates, st.wsel, st.B, m.E, m.P, m.G, m.top_k, mv.in_f ** -0.5)
        ops.gemm(st.A, mv.W2aug, y, st.B, mv.out_f, m.Kp, act=act, residual=residual)

    def _layers_family(self, st):
        """The decoder blocks of one decode step for the nano-mini family.  A sparse layer computes both of its paths for the new
        token -- the block (whose K/V land in the layer's next free cache slot: a skipped token's entry is overwritten by the next
        kept one before anything reads it) and x + null_connector(x) -- and keeps the one its position table selects; which
        one is a device-side flag, so the captured graph is the same for every position."""
        eng, a, dc = self.eng, self.eng.arena, self.eng.dec
        sp = dc.fam
        B, d, ff, H, hd, Hkv = st.B, dc.d, dc.ff, dc.H, sp.hd, st.Hkv
        w = Hkv * hd
        dp = eng.dp
        if st.sparse:
            ops.sparse_step_setup(st.counters[0:1], st.rank, st.member, st.lpos, st.lmem, dc.L, st.tmax)
        for l in range(dc.L):
            p = f'{dp}transformer.h.{l}'
            xo = st.xb if st.sparse else st.x
            pos_ptr = st.lpos[l:l + 1] if st.sparse else st.counters[0:1]
            ops.layernorm_fwd(st.x, a.P(f'{p}.ln_1.weight'), a.P(f'{p}.ln_1.bias'), st.ln, None, None, B, d)
            if sp.mqa:
                ops.gemm(st.ln, a.W(f'{p}.attn.q_proj.weight'), st.q, B, d, d, bias=a.P(f'{p}.attn.q_proj.bias'), workspace=st.ws)
                ops.gemm(st.ln, a.W(f'{p}.attn.kv_proj.weight'), st.kvn, B, 2 * hd, d, bias=a.P(f'{p}.attn.kv_proj.bias'), workspace=st.ws)
                qv, kn, vn, out_name = st.q, st.kvn[:, :hd], st.kvn[:, hd:], 'attn.out_proj'
            else:
                ops.gemm(st.ln, a.W(f'{p}.attn.c_attn.weight'), st.qkv, B, 3 * d, d, bias=a.P(f'{p}.attn.c_attn.bias'), workspace=st.ws)
                qv, kn, vn, out_name = st.qkv[:, :d], st.qkv[:, d:2 * d], st.qkv[:, 2 * d:], 'attn.c_proj'
            ops.gq_decode_attention(qv, kn, vn, st.kc[l], st.vc[l], st.slots * w, w, st.ao, pos_ptr, 0, st.slots, B, H, Hkv, hd)
            ops.gemm(st.ao, a.W(f'{p}.{out_name}.weight'), xo, B, d, d, bias=a.P(f'{p}.{out_name}.bias'), residual=st.x, workspace=st.ws)
            if l in st.cross_kv:
                kv, S = st.cross_kv[l]
                win, bin_ = a.W(f'{p}.cross_attn.in_proj_weight'), a.P(f'{p}.cross_attn.in_proj_bias')
                ops.layernorm_fwd(xo, a.P(f'{p}.ln_3.weight'), a.P(f'{p}.ln_3.bias'), st.ln, None, None, B, d)
                ops.gemm(st.ln, win[:d], st.q, B, d, d, bias=bin_[:d], workspace=st.ws)
                ops.gq_decode_attention(st.q, None, None, kv, kv.view(-1)[d:], S * 2 * d, 2 * d, st.ao, None, S, S, B, H, H, hd)
                ops.gemm(st.ao, a.W(f'{p}.cross_attn.out_proj.weight'), xo, B, d, d, bias=a.P(f'{p}.cross_attn.out_proj.bias'), residual=xo, workspace=st.ws)
            ops.layernorm_fwd(xo, a.P(f'{p}.ln_2.weight'), a.P(f'{p}.ln_2.bias'), st.ln, None, None, B, d)
            if sp.moe is not None:
                self._moe_step(st, f'{p}.mlp.c_fc', st.ln, st.h, 1, None)
                self._moe_step(st, f'{p}.mlp.c_proj', st.h, xo, 0, xo)
            else:
                ops.gemm(st.ln, a.W(f'{p}.mlp.c_fc.weight'), st.h, B, ff, d, bias=a.P(f'{p}.mlp.c_fc.bias'), act=1, workspace=st.ws)
                ops.gemm(st.h, a.W(f'{p}.mlp.c_proj.weight'), xo, B, d, ff, bias=a.P(f'{p}.mlp.c_proj.bias'), residual=xo, workspace=st.ws)
            if st.sparse:
                ops.cast_f32_bf16(st.x, st.xnb)
                ops.gemm(st.xnb, a.W(f'{p}.null_connector.weight'), st.xn, B, d, d, bias=a.P(f'{p}.null_connector.bias'), residual=st.x, workspace=st.ws)
                ops.select_rows(st.lmem[l:l + 1], st.xb, st.xn, st.x, B * d)

    def _cross_layers(self):
        cfg = self.model.config
        return [l for l in range(self.eng.dec.L)
                if self.eng.cross_inputs and (self.eng.dec_cross[l] or not self.eng.dcfg.skip_alternate_cross_attn)]

    # ------------------------------------------------------------------------------------------------ one token
    def _step(self, st, with_head: bool, sampling: Optional[Sampling] = None, top2: bool = False):
        """Consume the token at ids[:, pos]; when with_head also choose ids[:, len] (argmax after the n-gram ban, or a draw from
        the filtered distribution when ``sampling`` is given); then advance pos and len."""
        eng, a, dc = self.eng, self.eng.arena, self.eng.dec
        B, d, ff, H = st.B, dc.d, dc.ff, dc.H
        pos_ptr, len_ptr = st.counters[0:1], st.counters[1:2]
        dp = eng.dp
        ops.embed_step(st.ids, st.ids_ld, len_ptr, a.P(eng.n_wte),
                       None if (dc.advpos or dc.llama is not None) else a.P(f'{dp}transformer.wpe.weight'), st.x, B, d, st.off, dc.V)
        if dc.advpos:           # x = MLP_p(e) + e with p = off + pos read on the device: the same captured launches serve every position
            pv = eng._pos_views()
            ops.cast_f32_bf16(st.x, st.ln)
            h = st.ln
            for i, L in enumerate(pv.layers):
                last = i == len(pv.layers) - 1
                out = st.x if last else st.pos_h[i]
                ops.grouped_gemm(0, h, L.W[:L.N * L.K].view(L.N, L.K), out, L.N, L.K, b_group_stride=pv.stride, bias=L.b,
                                 bias_group_stride=pv.stride, act=0 if last else 1, residual=st.x if last else None, n_groups=1, max_rows=B,
                                 group_ptr=pos_ptr, group0=st.off)
                h = out
        if dc.llama is not None:
            self._layers_llama(st)
        elif dc.fam is not None:
            self._layers_family(st)
        else:
            self._layers_dense(st)
        if with_head:
            if dc.llama is not None and dc.llama.arch == 'falcon':
                wn = dp + dc.llama.norm_f
                ops.layernorm_fwd(st.x, a.P(wn + '.weight'), a.P(wn + '.bias'), st.hid, None, None, B, d, eps=dc.llama.eps)
            elif dc.llama is not None:
                ops.rmsnorm_fwd(st.x, a.P(dp + dc.llama.norm_f + '.weight'), st.hid, None, B, d, dc.llama.eps)
            else:
                ops.layernorm_fwd(st.x, a.P(f'{dp}transformer.ln_f.weight'), a.P(f'{dp}transformer.ln_f.bias'), st.hid, None, None, B, d)
            if top2 and sampling is None:
                # greedy, nobody asked for margins: the lm_head leaves the two largest logits of every 64-column segment of a row
                # instead of the row (51 MB instead of 823 MB written and read back at 4096 captions), the ban + argmax merges them
                ops.gemm_top2(st.hid, a.W(eng.n_head), st.top2, B, dc.V, d)
                ops.top2_ngram_argmax(st.top2, st.hid, a.W(eng.n_head), st.ids, st.ids_ld, len_ptr, st.ngrams, st.ngrams.numel(), B, dc.V, d)
                ops.advance(st.counters, 1)
                return
            ops.gemm(st.hid, a.W(eng.n_head), st.logits, B, dc.V, d, workspace=st.ws)
            if sampling is None:
                ops.ngram_ban_argmax(st.logits, dc.Vp, st.ids, st.ids_ld, len_ptr, st.ngrams, st.ngrams.numel(), B, dc.V, st.margin)
            else:
                ops.sample_token(st.logits, dc.Vp, st.ids, st.ids_ld, len_ptr, st.ngrams, st.ngrams.numel(), B, dc.V,
                                 sampling.temperature, sampling.top_k, sampling.nucleus_p, st.seed, dist_out=st.dist)
        ops.advance(st.counters, 1)                            # pos and len together

    def _w(self, l: int, site: str, name: str, rows=None):
        """bf16 weight of a decoder linear for the decode step: the arena's shadow, or -- under a LoRA adapter -- the merged
        W + s B A in its persistent buffer (engine_lora.py; generation has no dropout, so merging is exact)"""
        eng = self.eng
        if eng._lora_site(l, site) is not None:
            return eng.lora_merged(l, site, name, rows)
        w = eng.arena.W(name)
        return w if rows is None else w[rows]

    def _layers_dense(self, st):
        """The decoder blocks of one decode step for the dense multi-head model (64-wide heads, GELU-MLP)."""
        eng, a, dc = self.eng, self.eng.arena, self.eng.dec
        B, d, ff, H = st.B, dc.d, dc.ff, dc.H
        pos_ptr = st.counters[0:1]
        dp = eng.dp
        for l in range(dc.L):
            p = f'{dp}transformer.h.{l}'
            ops.layernorm_fwd(st.x, a.P(f'{p}.ln_1.weight'), a.P(f'{p}.ln_1.bias'), st.ln, None, None, B, d)
            ops.gemm(st.ln, self._w(l, 'attn_c_attn', f'{p}.attn.c_attn.weight'), st.qkv, B, 3 * d, d, bias=a.P(f'{p}.attn.c_attn.bias'),
                     workspace=st.ws)
            ops.decode_attention(st.qkv, 3 * d, st.kc[l], st.vc[l], st.clen * d, 64, st.ao, d, pos_ptr, 0, B, H, append_dm=d,
                                 cache_hs=st.clen * 64)           # head-major self-attention cache [B][H][prefix + tmax][64]
            ops.gemm(st.ao, a.W(f'{p}.attn.c_proj.weight'), st.x, B, d, d, bias=a.P(f'{p}.attn.c_proj.bias'), residual=st.x, workspace=st.ws)
            if l in st.cross_kv:
                kv, S = st.cross_kv[l]
                win, bin_ = a.W(f'{p}.cross_attn.in_proj_weight'), a.P(f'{p}.cross_attn.in_proj_bias')
                ops.layernorm_fwd(st.x, a.P(f'{p}.ln_3.weight'), a.P(f'{p}.ln_3.bias'), st.ln, None, None, B, d)
                ops.gemm(st.ln, win[:d], st.q, B, d, d, bias=bin_[:d], workspace=st.ws)
                ops.decode_attention(st.q, d, kv, kv.view(-1)[d:], S * 2 * d, 2 * d, st.ao, d, None, S, B, H)
                ops.gemm(st.ao, a.W(f'{p}.cross_attn.out_proj.weight'), st.x, B, d, d,
                         bias=a.P(f'{p}.cross_attn.out_proj.bias'), residual=st.x, workspace=st.ws)
            ops.layernorm_fwd(st.x, a.P(f'{p}.ln_2.weight'), a.P(f'{p}.ln_2.bias'), st.ln, None, None, B, d)
            ops.gemm(st.ln, self._w(l, 'mlp_c_fc', f'{p}.mlp.c_fc.weight'), st.h, B, ff, d, bias=a.P(f'{p}.mlp.c_fc.bias'), act=1, workspace=st.ws)
            ops.gemm(st.h, self._w(l, 'mlp_c_proj', f'{p}.mlp.c_proj.weight'), st.x, B, d, ff, bias=a.P(f'{p}.mlp.c_proj.bias'), residual=st.x,
                     workspace=st.ws)

    def _layers_llama(self, st):
        """The Llama-2 / Qwen2 blocks of one decode step (engine_llama.py): the rotary angle is looked up at the position counter on
        the device (absolute position = cache slot: prompt rows first), so one captured graph serves every step."""
        eng, dc, ls = self.eng, self.eng.dec, self.eng.dec.llama
        B, d, ff, H, G, hd = st.B, dc.d, dc.ff, ls.H, ls.Hkv, ls.hd
        pos_ptr = st.counters[0:1]
        cs = eng.rope_table()
        for l in range(dc.L):
            v = eng._llama_views(l)
            if getattr(dc, 'lora', None) is not None:          # merged W + s B A per adapted projection (engine_lora.lora_merged)
                nm = v.names
                Wqkv, Wo, Wgu, Wdn = (eng.lora_merged(l, site, names) if eng._llama_lora(l, site) is not None else W for site, names, W in
                                      (('qkv', nm.qkv, v.Wqkv), ('o', nm.o, v.Wo), ('gu', nm.gu, v.Wgu), ('dn', nm.dn, v.Wdn)))
                v = SimpleNamespace(**{**vars(v), 'Wqkv': Wqkv, 'Wo': Wo, 'Wgu': Wgu, 'Wdn': Wdn})
            if ls.arch == 'falcon':        # n = LN(x);  x += dense(attn(rope(qkv(n)))) + W2 gelu(W1 n)   (engine_llama.falcon_block_fwd)
                ops.layernorm_fwd(st.x, v.n1, v.b1, st.ln, None, None, B, d, eps=ls.eps)
                ops.gemm(st.ln, v.Wqkv, st.qkv, B, v.nq, d, workspace=st.ws)
                ops.rope(st.qkv, v.nq, 0, H + G, hd, cs, B, pos_ptr=pos_ptr)
                ops.gq_decode_attention(st.qkv[:, :H * hd], st.qkv[:, H * hd:(H + G) * hd], st.qkv[:, (H + G) * hd:], st.kc[l], st.vc[l],
                                        st.clen * G * hd, G * hd, st.ao, pos_ptr, 0, st.clen, B, H, G, hd)
                ops.gemm(st.ao, v.Wo, st.x, B, d, H * hd, residual=st.x, workspace=st.ws)
                ops.gemm(st.ln, v.Wgu, st.h, B, ff, d, act=ops.ACT_GELU_ERF)
                ops.gemm(st.h, v.Wdn, st.x, B, d, ff, residual=st.x, workspace=st.ws)
                continue
            ops.rmsnorm_fwd(st.x, v.n1, st.ln, None, B, d, ls.eps)
            ops.gemm(st.ln, v.Wqkv, st.qkv, B, v.nq, d, bias=v.bqkv, workspace=st.ws)
            ops.rope(st.qkv, v.nq, 0, H + G, hd, cs, B, pos_ptr=pos_ptr)
            ops.gq_decode_attention(st.qkv[:, :H * hd], st.qkv[:, H * hd:(H + G) * hd], st.qkv[:, (H + G) * hd:], st.kc[l], st.vc[l],
                                    st.clen * G * hd, G * hd, st.ao, pos_ptr, 0, st.clen, B, H, G, hd)
            ops.gemm(st.ao, v.Wo, st.x, B, d, H * hd, residual=st.x, workspace=st.ws)
            ops.rmsnorm_fwd(st.x, v.n2, st.ln, None, B, d, ls.eps)
            ops.gemm(st.ln, v.Wgu, st.gu, B, 2 * ff, d, workspace=st.ws)
            ops.swiglu_fwd(st.gu, st.h, B, ff)
            ops.gemm(st.h, v.Wdn, st.x, B, d, ff, residual=st.x, workspace=st.ws)

    def _capture(self, st, with_head: bool, sampling: Optional[Sampling] = None, top2: bool = False):
        side = torch.cuda.Stream(device=st.arena.device)
        side.wait_stream(torch.cuda.current_stream())
        g = ops.Graph()
        with torch.cuda.stream(side):
            g.begin()
            self._step(st, with_head, sampling, top2)
            g.end()
        torch.cuda.current_stream().wait_stream(side)
        return g

    # ------------------------------------------------------------------------------------------------ public
    @torch.no_grad()
    def generate(self, images, prompt_ids: torch.Tensor, max_new_tokens: int, return_margins: bool = False,
                 use_graph: bool = True, sampling: Optional[Sampling] = None, return_dists: bool = False):
        """-> ids (B, P + max_new_tokens) [, margins (B, N) greedy only] [, dists (B, N, V) sampling only: the filtered,
        renormalised distribution every token was drawn from]."""
        eng = self.eng
        if not eng.dec.causal:      # nothing to cache under bidirectional attention: the reference's re-evaluation loop
            assert not return_margins and not return_dists, 'margins / distributions are recorded on the KV-cache path only'
            return generate_by_recompute(self.model, images, prompt_ids, max_new_tokens, sampling)
        a = eng.prepare(False)
        dc = eng.dec
        B, P = prompt_ids.shape
        total = P + max_new_tokens
        st = self._state
        if st is None or st.B != B or st.arena is not a or st.ids_ld < total or st.sparse_epoch != eng.sparse_epoch:
            st = self._state = self._build(B, max(total, dc.block))
        assert total <= st.tmax, f'prompt + new tokens ({total}) exceed the text window ({st.tmax})'
        assert not (return_margins and sampling is not None) and not (return_dists and sampling is None)
        # encoder + per-layer cross K/V (once per image)
        # (in slices of ENC_CHUNK images -- 4096: +1.4 % captions/s over 1024 -- every image is independent in the encoder, and its activations -- ~20 MB per
        # image in eval mode -- would otherwise set the memory footprint of a large caption batch)
        if B <= ENC_CHUNK:
            enc_out, _ = eng.encode(images, False)
        else:
            enc_out = None
            for i in range(0, B, ENC_CHUNK):
                part, _ = eng.encode(images[i:i + ENC_CHUNK], False)
                if enc_out is None:
                    enc_out = torch.empty(B, *part.shape[1:], dtype=part.dtype, device=part.device)
                enc_out[i:i + ENC_CHUNK].copy_(part)
                del part
        S = enc_out.shape[1]
        eng.prepare_lora_merged()                               # merged adapter weights follow the current parameters
        if st.cross_kv:
            assert S == next(iter(st.cross_kv.values()))[1]
            mem = eng._mem_bf16(enc_out)
            for l, (kv, _) in st.cross_kv.items():              # persistent buffers: captured graphs bake their pointers
                p = f'{eng.dp}transformer.h.{l}.cross_attn'
                ops.gemm(mem, self._w(l, 'xattn_c_attn', f'{p}.in_proj_weight', slice(dc.d, 3 * dc.d)), kv.view(B * S, 2 * dc.d), B * S,
                         2 * dc.d, dc.d, bias=a.P(f'{p}.in_proj_bias')[dc.d:])
        if st.prefix:       # the prompt rows' keys and values (one causal pass over the encoder outputs) open every caption's cache
            n_p = st.prefix
            if dc.llama is not None:          # row-major cache [B][slot][Hkv hd]; the saved keys already carry their rotation.  Each layer's
                ls = dc.llama                 # K / V go into the cache as the layer finishes (a 7-B model's 32 saves would not fit beside it)

                def take_kv(l, sv):
                    qkv = sv.qkv.view(B, n_p, -1)
                    st.kc[l][:, :n_p].copy_(qkv[..., ls.H * ls.hd:(ls.H + ls.Hkv) * ls.hd])
                    st.vc[l][:, :n_p].copy_(qkv[..., (ls.H + ls.Hkv) * ls.hd:])
                eng._layer_sink = take_kv
            try:
                _, _, pctx = eng.decode_segment(B, n_p, eng._mem_bf16(enc_out) if eng.cross_inputs else None, S, True,
                                                embeds=enc_out[:, :n_p].reshape(B * n_p, dc.d), pos_offset=0)
            finally:
                eng._layer_sink = None
            for l in range(dc.L if dc.llama is None else 0):
                qkv = pctx.saves[l].qkv.view(B, n_p, 3, dc.H, 64)
                st.kc[l].view(B, dc.H, st.clen, 64)[:, :, :n_p].copy_(qkv[:, :, 1].transpose(1, 2))
                st.vc[l].view(B, dc.H, st.clen, 64)[:, :, :n_p].copy_(qkv[:, :, 2].transpose(1, 2))
            del pctx
        if dc.fam is not None and dc.fam.moe is not None:       # the packed expert output weights follow the current parameters
            for l in range(dc.L):
                for part in ('c_fc', 'c_proj'):
                    mv = eng._moe_views(f'{eng.dp}transformer.h.{l}.mlp.{part}', dc.fam.moe)
                    ops.moe_pack_w2(mv.l2w, mv.l2b, mv.W2aug, mv.out_f, dc.fam.moe.E, dc.fam.moe.P)
        if sampling is not None:
            seed = sampling.seed if sampling.seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
            lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
            st.seed.copy_(torch.tensor([lo - (1 << 32) if lo >= (1 << 31) else lo, hi - (1 << 32) if hi >= (1 << 31) else hi],
                                       dtype=torch.int32))
            if return_dists != (st.dist is not None):         # captured sampling steps bake the dist pointer (or its absence)
                st.dist = torch.zeros(B, dc.V, dtype=F32, device=a.device) if return_dists else None
                st.graphs = {k: g for k, g in st.graphs.items() if k in (None, 'greedy', 'greedy_top2')}
        # greedy without margins: the head in its segment-maxima form (d % 128 == 0: the persistent GEMM kernel's K rule; I2T_DECODE_TOP2=0: the logits form)
        top2 = sampling is None and not return_margins and TOP2_HEAD and dc.d % 128 == 0
        if top2 and getattr(st, 'top2', None) is None:
            st.top2 = torch.zeros(B, (dc.V + 63) // 64, 4, dtype=F32, device=a.device)
        full_key = ('greedy_top2' if top2 else 'greedy') if sampling is None else sampling.key()

        def reset():
            st.ids.zero_()
            st.ids[:, :P] = prompt_ids
            st.counters.copy_(st.counters_init)                 # device-to-device: no host sync in the loop
        reset()
        margins = torch.zeros(max_new_tokens, B, dtype=F32, device=a.device) if return_margins else None
        dists = torch.zeros(max_new_tokens, B, dc.V, dtype=F32, device=a.device) if return_dists else None
        if use_graph and (full_key not in st.graphs or None not in st.graphs):
            # warm up eagerly once (code objects must be loaded before capture), then capture the step kinds that are missing
            self._step(st, True, sampling, top2)
            self._step(st, False)
            if full_key not in st.graphs:
                st.graphs[full_key] = self._capture(st, True, sampling, top2)
            if None not in st.graphs:
                st.graphs[None] = self._capture(st, False)
            reset()
        for _ in range(P - 1):                                  # prompt tokens before the last: fill the cache only
            st.graphs[None].launch() if use_graph else self._step(st, False)
        for i in range(max_new_tokens):
            st.graphs[full_key].launch() if use_graph else self._step(st, True, sampling, top2)
            if return_margins:
                margins[i].copy_(st.margin)
            if return_dists:
                dists[i].copy_(st.dist)
        out = st.ids[:, :total].clone()
        if return_margins:
            return out, margins.t().contiguous()
        if return_dists:
            return out, dists.transpose(0, 1).contiguous()
        return out


@torch.no_grad()
def generate_by_recompute(model, images, prompt_ids: torch.Tensor, max_new_tokens: int, sampling: Optional[Sampling] = None):
    """generate() for a NON-causal decoder, as the reference runs it (vision_encoder_decoder.py:143-180): the whole text segment is
    re-evaluated for every new token (with bidirectional attention a new token changes the state of all earlier ones, so there is
    nothing to cache); the encoder runs once, and the token choice -- n-gram ban + argmax or the sampling step -- stays on the device."""
    eng: HotPath = model._engine
    a = eng.prepare(False)
    dc, cfg = eng.dec, model.config
    dev = a.device
    B, P = prompt_ids.shape
    total = P + max_new_tokens
    enc_out, _ = eng.encode(images, False)
    ncls = enc_out.shape[1]
    mem = eng._mem_bf16(enc_out) if eng.cross_inputs else None
    off = ncls if cfg.use_soft_prompting else 0
    blk = dc.block - off
    ids = torch.zeros(B, total, dtype=torch.long, device=dev)
    ids[:, :P] = prompt_ids.to(dev)
    counters = torch.tensor([P - 1, P], dtype=torch.int32, device=dev)
    ngrams = torch.tensor(list(cfg.no_repeat_n_grams), dtype=torch.int32, device=dev)
    logits = torch.zeros(B, dc.Vp, dtype=F32, device=dev)
    margin = torch.zeros(B, dtype=F32, device=dev)
    seed = torch.zeros(2, dtype=torch.int32, device=dev)
    if sampling is not None:
        sd = sampling.seed if sampling.seed is not None else int(torch.randint(0, 2 ** 62, (1,)).item())
        lo, hi = sd & 0xFFFFFFFF, (sd >> 32) & 0xFFFFFFFF
        seed.copy_(torch.tensor([lo - (1 << 32) if lo >= (1 << 31) else lo, hi - (1 << 32) if hi >= (1 << 31) else hi], dtype=torch.int32))
    for t in range(P, total):
        cond = ids[:, :t] if t <= blk else ids[:, t - blk:t]                     # the reference crops the conditioning to the block
        Tc = cond.shape[1]
        _, hb, _ = eng.decode_segment(B, Tc, mem, ncls, False, ids=cond.contiguous(), pos_offset=off)
        last = hb.view(B, Tc, dc.d)[:, -1].contiguous()
        ops.gemm(last, a.W(eng.n_head), logits, B, dc.V, dc.d)
        if sampling is None:
            ops.ngram_ban_argmax(logits, dc.Vp, ids, total, counters[1:2], ngrams, ngrams.numel(), B, dc.V, margin)
        else:
            ops.sample_token(logits, dc.Vp, ids, total, counters[1:2], ngrams, ngrams.numel(), B, dc.V, sampling.temperature,
                             sampling.top_k, sampling.nucleus_p, seed)
        ops.advance(counters, 1)
    return ids


class ConcurrentGreedyDecoder:
    """Several independent caption batches decoded at the same time, one HIP stream + one captured graph each.

    A single decode stream is latency-bound (one token per caption per step: ~120 short dependent kernels, most of
    them on far fewer than 256 workgroups), so the chip is mostly idle; batches are independent (decode shards
    trivially), so their graphs are replayed on separate streams and overlap on the GPU."""

    def __init__(self, model, n_streams: int):
        self.model = model
        self.lanes = [(GreedyDecoder(model), torch.cuda.Stream()) for _ in range(n_streams)]

    @torch.no_grad()
    def generate(self, image_batches, prompt_batches, max_new_tokens: int):
        assert len(image_batches) == len(prompt_batches) <= len(self.lanes)
        cur = torch.cuda.current_stream()
        # shared state is brought up to date on the parent stream BEFORE fanning out: the bf16 weight shadow's re-cast (if a
        # torch-side optimizer touched the parameters) must not land on one lane's stream while the other lanes read it; the
        # lanes' own prepare() calls then find nothing to do.  Per-lane state (KV caches, graphs, the encoder's conv weight
        # workspace -- keyed by stream in the engine) is private.
        self.model._engine.prepare(False)
        self.model._engine.prepare_lora_merged()
        outs = []
        for (dec, stream), images, prompt in zip(self.lanes, image_batches, prompt_batches):
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                outs.append(dec.generate(images, prompt, max_new_tokens))
        for _, stream in self.lanes:
            cur.wait_stream(stream)
        return outs
