"""Llama-2 / Qwen2 decoder blocks on the HIP hot path (reference models/decoder.py:404-440 wraps transformers'
LlamaForCausalLM / Qwen2ForCausalLM -- the arithmetic followed here is transformers' modeling_llama.py / modeling_qwen2.py):

    h   = x + o_proj(attention(rope(q_proj(n1)), rope(k_proj(n1)), v_proj(n1))),     n1 = RMSNorm(x)       H query heads on Hkv K/V heads
    out = h + down_proj(silu(gate_proj(n2)) * up_proj(n2)),                           n2 = RMSNorm(h)

How it maps to the MI355X:
  * q | k | v and gate | up are ONE GEMM each: the arena lays the three (two) weight matrices next to each other (engine._arena_order),
    so the fused [N, K] operand is a view -- the persistent 256^2 MFMA kernel sees N = (H + 2 Hkv) hd and N = 2 ff;
  * the rotary embedding runs in place on the q and k columns of that GEMM's bf16 output (i2t_rope: table lookup of transformers' own
    cos / sin values, 16-byte accesses); its backward is the same kernel with the angle negated, in place on dq | dk;
  * attention = the grouped-query kernels of the nano-mini family (attention_g.hip) with Hkv > 1;
  * the residual stream stays fp32 ([M, d]); o_proj and down_proj add it in their GEMM epilogue; RMSNorm backward accumulates
    the branch gradient onto the fp32 stream gradient and emits the bf16 copy the next GEMM reads (as LayerNorm backward does);
  * no dropout (transformers' attention_dropout is 0 for these checkpoints), no gradient normaliser, no learned positions.
"""
from types import SimpleNamespace

import torch

from . import ops
from .engine_lora import LPAD

BF16, F32 = torch.bfloat16, torch.float32


def _f8pad(k: int) -> int:
    return (k + 255) // 256 * 256


class LlamaBlocks:
    """Mixin of engine.HotPath (uses its arena, ``_empty``, ``dec`` namespace and parameter prefix ``dp``)."""

    def _llama_views(self, l: int):
        key = ('llama', l, id(self.arena))
        v = self._sub_cache.get(key)
        if v is not None:
            return v
        a, ls = self.arena, self.dec.llama
        nq = ls.H * ls.hd + 2 * ls.Hkv * ls.hd
        if ls.arch == 'falcon':          # ONE fused query_key_value [q heads | k | v], ONE LayerNorm (weight + bias), a two-matrix GELU MLP
            p = f'{self.dp}backbone.transformer.h.{l}'
            nm = SimpleNamespace(qkv=[f'{p}.self_attention.query_key_value.weight'], qkv_b=[], o=f'{p}.self_attention.dense.weight',
                                 gu=[f'{p}.mlp.dense_h_to_4h.weight'], dn=f'{p}.mlp.dense_4h_to_h.weight',
                                 n1=f'{p}.input_layernorm.weight', b1=f'{p}.input_layernorm.bias', n2=None)
            v = SimpleNamespace(
                nq=nq, Wqkv=a.W(nm.qkv[0]), Gqkv=a.G(nm.qkv[0]), bqkv=None, gbqkv=None, Wo=a.W(nm.o), Go=a.G(nm.o),
                Wgu=a.W(nm.gu[0]), Ggu=a.G(nm.gu[0]), Wdn=a.W(nm.dn), Gdn=a.G(nm.dn), n1=a.P(nm.n1), gn1=a.G(nm.n1), b1=a.P(nm.b1),
                gb1=a.G(nm.b1), names=nm)
            self._sub_cache[key] = v
            return v
        p = f'{self.dp}backbone.model.layers.{l}'
        qkv_w = [f'{p}.self_attn.{x}_proj.weight' for x in 'qkv']
        qkv_b = [f'{p}.self_attn.{x}_proj.bias' for x in 'qkv']
        gu = [f'{p}.mlp.gate_proj.weight', f'{p}.mlp.up_proj.weight']
        v = SimpleNamespace(
            nq=nq,
            Wqkv=a.span('W', qkv_w, (nq, ls.d)), Gqkv=a.span('G', qkv_w, (nq, ls.d)),
            bqkv=a.span('P', qkv_b, (nq,)) if ls.qkv_bias else None, gbqkv=a.span('G', qkv_b, (nq,)) if ls.qkv_bias else None,
            Wo=a.W(f'{p}.self_attn.o_proj.weight'), Go=a.G(f'{p}.self_attn.o_proj.weight'),
            Wgu=a.span('W', gu, (2 * ls.ff, ls.d)), Ggu=a.span('G', gu, (2 * ls.ff, ls.d)),
            Wdn=a.W(f'{p}.mlp.down_proj.weight'), Gdn=a.G(f'{p}.mlp.down_proj.weight'),
            n1=a.P(f'{p}.input_layernorm.weight'), gn1=a.G(f'{p}.input_layernorm.weight'),
            n2=a.P(f'{p}.post_attention_layernorm.weight'), gn2=a.G(f'{p}.post_attention_layernorm.weight'),
            names=SimpleNamespace(qkv=qkv_w, qkv_b=qkv_b, o=f'{p}.self_attn.o_proj.weight', gu=gu, dn=f'{p}.mlp.down_proj.weight',
                                  n1=f'{p}.input_layernorm.weight', n2=f'{p}.post_attention_layernorm.weight'))
        self._sub_cache[key] = v
        return v

    # ---- fp8 operands for FROZEN weights (I2T_FP8=1; csrc/fp8.hip, BASELINE.json configs[4]): a frozen matrix has no dW, so both GEMMs
    # that touch it -- y = x W^T and dx = dy W -- run on the block-scaled e4m3 MFMA; W is quantised once per parameter version in
    # both orientations (per-output-row scales for the forward, per-input-row scales for the backward), activations per call
    def _fp8_on(self, names) -> bool:
        if not self.fp8:
            return False
        names = [names] if isinstance(names, str) else names
        if all(not self.arena.trainable(n) for n in names):
            return True
        # a weight that trains (again): the optimizer writes it through the arena without moving its version counter, so an e4m3 image
        # kept from an earlier frozen phase would be stale if the weight is frozen once more
        self._sub_cache.pop(('fp8w', tuple(names), id(self.arena)), None)
        return False

    def _fp8_weight(self, names, W):
        key = ('fp8w', tuple([names] if isinstance(names, str) else names), id(self.arena))
        ent = self._sub_cache.get(key)
        # a frozen parameter is skipped by the fused optimizers (arena.generation moves every step, these values do not): the image
        # is rebuilt only when torch-side code wrote the parameter (load_state_dict, a manual edit -> its version counter moves)
        version = tuple(self.arena.params[n]._version for n in key[1])
        if ent is None or ent.generation != version:
            self.arena.refresh_shadow()
            N, K = W.shape
            dev = W.device
            # rows zero-padded to a multiple of 256 bytes: the GEMM then runs K' = the padded length (zeros contribute nothing) and every
            # projection is eligible for the persistent fp8 kernel (K % 256 == 0; Falcon-7B: 4544 -> 4608)
            ent = SimpleNamespace(generation=version,
                                  w8=torch.empty(N, _f8pad(K), dtype=torch.uint8, device=dev), sw=torch.empty(N, dtype=F32, device=dev),
                                  wt8=torch.empty(K, _f8pad(N), dtype=torch.uint8, device=dev), swt=torch.empty(K, dtype=F32, device=dev))
            ops.quant_rows_fp8(W, ent.w8, ent.sw, N, K)
            ops.quant_cols_fp8(W, ent.wt8, ent.swt, N, K)
            self._sub_cache[key] = ent
        return ent

    def _fp8_rows(self, x_bf, M: int, K: int):
        x8 = torch.empty(M, _f8pad(K), dtype=torch.uint8, device=x_bf.device)
        sx = self._empty(M)
        ops.quant_rows_fp8(x_bf, x8, sx, M, K)
        return x8, sx

    def _lin(self, x_bf, W, names, out, M, N, K, bias=None, residual=None, act=0, xq=None):
        """out = act(x W^T (+ bias)) (+ residual): fp8 operands when the weight is frozen and I2T_FP8=1, else the bf16 GEMM.
        xq = (x8, scale): the producer already emitted the e4m3 operand (rmsnorm_fwd_fp8 / swiglu_fwd_fp8); x_bf may then be None."""
        if self._fp8_on(names):
            e = self._fp8_weight(names, W)
            x8, sx = xq if xq is not None else self._fp8_rows(x_bf, M, K)
            return ops.gemm_fp8(x8, sx, e.w8, e.sw, out, M, N, _f8pad(K), bias=bias, residual=residual, act=act)
        return ops.gemm(x_bf, W, out, M, N, K, bias=bias, residual=residual, act=act)

    def _lin_dx(self, dy_bf, W, names, out, M, N, K, residual=None, dq=None):
        """out [M, K] = dy [M, N] . W [N, K] (+ residual f32; may be ``out`` itself); dq = (dy8, scale) from a fused producer"""
        if self._fp8_on(names):
            e = self._fp8_weight(names, W)
            d8, sd = dq if dq is not None else self._fp8_rows(dy_bf, M, N)
            return ops.gemm_fp8(d8, sd, e.wt8, e.swt, out, M, K, _f8pad(N), residual=residual)
        return ops.gemm(dy_bf, W, out, M, K, N, b_kmajor=True, residual=residual)

    # ---- LoRA adapters on these blocks (reference models/utils.py:46-65 -> peft LoraModel over the transformers module; the targets of
    # training_configs/gpu/llama2-13b.yaml: q_proj, k_proj, v_proj, o_proj, up_proj, down_proj).  The fused projections keep ONE GEMM:
    # the adapters of q | k | v (gate | up) share a stacked lora_A -- u = dropout(x) [A_q; A_k; A_v]^T -- and their B matrices sit
    # block-diagonally in the K panel (engine_lora._lora_panel); everything else is engine_lora's machinery.
    _LLAMA_SITES = {'qkv': ('q', 'k', 'v'), 'o': ('o',), 'gu': ('gate', 'up'), 'dn': ('down',)}
    _FALCON_SITES = {'qkv': ('qkv',), 'o': ('o',), 'gu': ('fc',), 'dn': ('proj',)}      # query_key_value, dense, dense_h_to_4h, dense_4h_to_h

    def _llama_lora(self, l: int, site: str):
        lo = getattr(self.dec, 'lora', None)
        if lo is None or site not in lo.sites:
            return None
        key = ('llama_lora', l, site, id(self.arena))
        v = self._sub_cache.get(key)
        if v is None:
            a, ls = self.arena, self.dec.llama
            nA = f'{self.dp}lora_params.h{l}_{site}_A'
            K = a.entries[nA][2][1]
            rows = {'q': ls.H * ls.hd, 'k': ls.Hkv * ls.hd, 'v': ls.Hkv * ls.hd, 'o': ls.d, 'gate': ls.ff, 'up': ls.ff, 'down': ls.d,
                    'qkv': (ls.H + 2 * ls.Hkv) * ls.hd, 'fc': ls.ff, 'proj': ls.d}
            parts, row0, col0 = [], 0, 0
            for t in (self._FALCON_SITES if ls.arch == 'falcon' else self._LLAMA_SITES)[site]:
                nB = f'{self.dp}lora_params.h{l}_{t}_B'
                if nB in a.entries:
                    parts.append((row0, rows[t], col0, a.P(nB), a.G(nB)))
                    col0 += lo.r
                row0 += rows[t]
            names = [nA] + ([nA + '.<pad>'] if a.entries[nA][2][0] < LPAD else [])
            wn = getattr(self._llama_views(l).names, site)        # the adapted projection's base weight(s): frozen -> fp8 operands (I2T_FP8=1)
            v = self._sub_cache[key] = SimpleNamespace(K=K, N=row0, r=lo.r, scale=lo.scale, kind=f'lora_{site}', A=a.span('W', names, (LPAD, K)),
                                                       GA=a.span('G', names, (LPAD, K)), parts=parts, nA=nA,
                                                       wnames=[wn] if isinstance(wn, str) else list(wn))
        return v

    def rope_table(self):
        """fp32 [block, hd] = [cos | sin] per position, taken from the checkpoint's own rotary module (models/decoder.py)"""
        key = ('rope', str(self.arena.device))
        t = self._sub_cache.get(key)
        if t is None:
            t = self._sub_cache[key] = self.model.decoder.rope_table(self.dec.block).to(device=self.arena.device, dtype=F32).contiguous()
        return t

    # ------------------------------------------------------------------------------------------------ one block
    def llama_block_fwd(self, l: int, x, B: int, T: int, pos_offset: int, save: bool, vl=None):
        """vl: packed variable-length rows (cu, pos, total) -- x is [total, d], T the longest sequence"""
        ls, v = self.dec.llama, self._llama_views(l)
        M, d, H, G, hd, ff = (vl.total if vl is not None else B * T), ls.d, ls.H, ls.Hkv, ls.hd, ls.ff
        cu, rpos = (vl.cu, vl.pos) if vl is not None else (None, None)
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        cs = self.rope_table()
        plan = self.dec_drop if save else None
        ldrop = (lambda site: plan.get(l, f'lora_{site}') if plan is not None else None)
        lo = {site: self._llama_lora(l, site) for site in ('qkv', 'o', 'gu', 'dn')}
        svlo = {}
        # frozen projections on fp8 operands without adapters: the producing row kernel emits the e4m3 operand itself (csrc/fp8.hip) --
        # no bf16 copy of n1 / n2 / h is written and no quantisation pass reads it (nothing else wants them: a frozen weight has no dW)
        fuse = self.fp8 and self.fp8_fuse and d <= 8192 and ff <= 12288
        f_qkv = fuse and lo['qkv'] is None and self._fp8_on(v.names.qkv)
        f_gu = fuse and lo['gu'] is None and self._fp8_on(v.names.gu)
        f_dn = fuse and lo['dn'] is None and self._fp8_on(v.names.dn)
        r1 = self._empty(M)
        qkv = self._empty(M, v.nq, dtype=BF16)
        if f_qkv:
            n1 = None
            n1q = (torch.empty(M, _f8pad(d), dtype=torch.uint8, device=x.device), self._empty(M))
            ops.rmsnorm_fwd_fp8(x, v.n1, n1q[0], n1q[1], r1, M, d, ls.eps)
            self._lin(None, v.Wqkv, v.names.qkv, qkv, M, v.nq, d, bias=v.bqkv, xq=n1q)
        else:
            n1, n1q = self._empty(M, d, dtype=BF16), None
            if fuse and lo['qkv'] is not None and self._fp8_on(v.names.qkv):      # LoRA on a frozen fp8 base: the bf16 row (adapter) AND the e4m3 one
                n1q = (torch.empty(M, _f8pad(d), dtype=torch.uint8, device=x.device), self._empty(M))
                ops.rmsnorm_fwd_fp8(x, v.n1, n1q[0], n1q[1], r1, M, d, ls.eps, y_bf16=n1)
            else:
                ops.rmsnorm_fwd(x, v.n1, n1, r1, M, d, ls.eps)
        if f_qkv:
            pass
        elif lo['qkv'] is not None:
            svlo['qkv'] = self._lora_gemm(lo['qkv'], n1, v.Wqkv, qkv, M, ldrop('qkv'), save, bias=v.bqkv, xq=n1q)
        else:
            self._lin(n1, v.Wqkv, v.names.qkv, qkv, M, v.nq, d, bias=v.bqkv)
        ops.rope(qkv, v.nq, 0, H + G, hd, cs, M, pos=rpos, pos_offset=pos_offset, T=T)      # q heads and k heads are adjacent columns
        q3 = v3(qkv, v.nq)
        ao, lse = self._empty(M, H * hd, dtype=BF16), self._empty(H * M)
        ops.gq_attention_fwd(q3[..., :H * hd], q3[..., H * hd:(H + G) * hd], q3[..., (H + G) * hd:], v3(ao, H * hd), lse,
                             B, H, G, hd, T, T, True, cu_q=cu, cu_k=cu, total_q=M)
        x1 = self._empty(M, d)
        if lo['o'] is not None:
            svlo['o'] = self._lora_gemm(lo['o'], ao, v.Wo, x1, M, ldrop('o'), save, residual=x)
        else:
            self._lin(ao, v.Wo, v.names.o, x1, M, d, H * hd, residual=x)
        r2 = self._empty(M)
        gu = self._empty(M, 2 * ff, dtype=BF16)
        if f_gu:
            n2 = None
            n2q = (torch.empty(M, _f8pad(d), dtype=torch.uint8, device=x.device), self._empty(M))
            ops.rmsnorm_fwd_fp8(x1, v.n2, n2q[0], n2q[1], r2, M, d, ls.eps)
            self._lin(None, v.Wgu, v.names.gu, gu, M, 2 * ff, d, xq=n2q)
        else:
            n2, n2q = self._empty(M, d, dtype=BF16), None
            if fuse and lo['gu'] is not None and self._fp8_on(v.names.gu):
                n2q = (torch.empty(M, _f8pad(d), dtype=torch.uint8, device=x.device), self._empty(M))
                ops.rmsnorm_fwd_fp8(x1, v.n2, n2q[0], n2q[1], r2, M, d, ls.eps, y_bf16=n2)
            else:
                ops.rmsnorm_fwd(x1, v.n2, n2, r2, M, d, ls.eps)
            if lo['gu'] is not None:
                svlo['gu'] = self._lora_gemm(lo['gu'], n2, v.Wgu, gu, M, ldrop('gu'), save, xq=n2q)
            else:
                self._lin(n2, v.Wgu, v.names.gu, gu, M, 2 * ff, d)
        x2 = self._empty(M, d)
        if f_dn:
            h = None
            hq = (torch.empty(M, _f8pad(ff), dtype=torch.uint8, device=x.device), self._empty(M))
            ops.swiglu_fwd_fp8(gu, hq[0], hq[1], M, ff)
            self._lin(None, v.Wdn, v.names.dn, x2, M, d, ff, residual=x1, xq=hq)
        else:
            h, hq = self._empty(M, ff, dtype=BF16), None
            if fuse and lo['dn'] is not None and self._fp8_on(v.names.dn):
                hq = (torch.empty(M, _f8pad(ff), dtype=torch.uint8, device=x.device), self._empty(M))
                ops.swiglu_fwd_fp8(gu, hq[0], hq[1], M, ff, h_bf16=h)
            else:
                ops.swiglu_fwd(gu, h, M, ff)
            if lo['dn'] is not None:
                svlo['dn'] = self._lora_gemm(lo['dn'], h, v.Wdn, x2, M, ldrop('dn'), save, residual=x1, xq=hq)
            else:
                self._lin(h, v.Wdn, v.names.dn, x2, M, d, ff, residual=x1)
        return x2, (SimpleNamespace(x=x, n1=n1, r1=r1, qkv=qkv, ao=ao, lse=lse, x1=x1, n2=n2, r2=r2, gu=gu, h=h, lo=svlo,
                                    lo_drop={site: ldrop(site) for site in lo}) if save else None)

    def llama_block_bwd(self, l: int, sv, dx, dxb, B: int, T: int, pos_offset: int, vl=None):
        """dx fp32 / dxb bf16: gradient w.r.t. the block output; on return both hold the gradient w.r.t. the block input"""
        ls, v = self.dec.llama, self._llama_views(l)
        M, d, H, G, hd, ff = (vl.total if vl is not None else B * T), ls.d, ls.H, ls.Hkv, ls.hd, ls.ff
        cu, rpos = (vl.cu, vl.pos) if vl is not None else (None, None)
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        # frozen parameters (prepare_for_kbit_training, models/decoder.py): their gradient GEMMs are skipped, the input gradient is not
        tr = (lambda names: all(self.arena.trainable(n) for n in ([names] if isinstance(names, str) else names)))
        nm = v.names
        # ---- MLP
        svlo = getattr(sv, 'lo', None) or {}
        span_g = (lambda names, shape: self.arena.span('G', names, shape) if tr(names) else None)

        def lora_bwd(site, dY, x_in, W, names, shape, gb=None, dq=None):          # -> fp32 dx (engine_lora._lora_bwd), base dW only when trainable
            return self._lora_bwd(self._llama_lora(l, site), svlo[site], dY, x_in, W, span_g([names] if isinstance(names, str) else names, shape),
                                  gb, M, sv.lo_drop.get(site), dq=dq)
        dh = self._empty(M, ff, dtype=BF16)
        if 'dn' in svlo:
            ops.cast_f32_bf16(lora_bwd('dn', dxb, sv.h, v.Wdn, nm.dn, (d, ff)), dh)
        else:
            if tr(nm.dn):
                ops.gemm(dxb, sv.h, v.Gdn, d, ff, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dxb, v.Wdn, nm.dn, dh, M, d, ff)
        dn = self._empty(M, d, dtype=BF16)
        if 'gu' not in svlo and not tr(nm.gu) and self.fp8 and self.fp8_fuse and self._fp8_on(nm.gu) and ff <= 12288:
            dguq = (torch.empty(M, _f8pad(2 * ff), dtype=torch.uint8, device=dh.device), self._empty(M))
            ops.swiglu_bwd_fp8(dh, sv.gu, dguq[0], dguq[1], M, ff)             # [d gate | d up] straight to the e4m3 operand of dx = d(gu) . W
            self._lin_dx(None, v.Wgu, nm.gu, dn, M, 2 * ff, d, dq=dguq)
            dgu, dn2 = None, dn
        else:
            dgu, dguq = self._empty(M, 2 * ff, dtype=BF16), None
            if 'gu' in svlo and self.fp8 and self.fp8_fuse and self._fp8_on(nm.gu) and ff <= 12288:
                dguq = (torch.empty(M, _f8pad(2 * ff), dtype=torch.uint8, device=dh.device), self._empty(M))
                ops.swiglu_bwd_fp8(dh, sv.gu, dguq[0], dguq[1], M, ff, dgu_bf16=dgu)
            else:
                ops.swiglu_bwd(dh, sv.gu, dgu, M, ff)
        if dgu is None:
            pass
        elif 'gu' in svlo:
            dn2 = lora_bwd('gu', dgu, sv.n2, v.Wgu, nm.gu, (2 * ff, d), dq=dguq)      # fp32: rmsnorm_bwd takes either
        else:
            if tr(nm.gu):
                ops.gemm(dgu, sv.n2, v.Ggu, 2 * ff, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dgu, v.Wgu, nm.gu, dn, M, 2 * ff, d)
            dn2 = dn
        ops.rmsnorm_bwd(dn2, sv.x1, v.n2, sv.r2, dx, v.gn2 if tr(nm.n2) else None, M, d, dx_accumulate=True, dx_bf16=dxb)
        # ---- attention
        dao = self._empty(M, H * hd, dtype=BF16)
        if 'o' in svlo:
            ops.cast_f32_bf16(lora_bwd('o', dxb, sv.ao, v.Wo, nm.o, (d, H * hd)), dao)
        else:
            if tr(nm.o):
                ops.gemm(dxb, sv.ao, v.Go, d, H * hd, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dxb, v.Wo, nm.o, dao, M, d, H * hd)
        dqkv = self._empty(M, v.nq, dtype=BF16)
        q3, g3 = v3(sv.qkv, v.nq), v3(dqkv, v.nq)
        sl = (slice(0, H * hd), slice(H * hd, (H + G) * hd), slice((H + G) * hd, v.nq))
        ops.gq_attention_bwd(q3[..., sl[0]], q3[..., sl[1]], q3[..., sl[2]], v3(sv.ao, H * hd), v3(dao, H * hd), sv.lse,
                             self._empty(H * M), g3[..., sl[0]], g3[..., sl[1]], g3[..., sl[2]], B, H, G, hd, T, T, True,
                             cu_q=cu, cu_k=cu, total_q=M)
        ops.rope(dqkv, v.nq, 0, H + G, hd, self.rope_table(), M, pos=rpos, pos_offset=pos_offset, T=T, inverse=True)
        if 'qkv' in svlo:
            dn1 = lora_bwd('qkv', dqkv, sv.n1, v.Wqkv, nm.qkv, (v.nq, d), gb=v.gbqkv if (v.gbqkv is not None and tr(nm.qkv_b)) else None)
        else:
            if v.gbqkv is not None and tr(nm.qkv_b):
                ops.colsum(dqkv, v.gbqkv, M, v.nq, accumulate=True)
            if tr(nm.qkv):
                ops.gemm(dqkv, sv.n1, v.Gqkv, v.nq, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dqkv, v.Wqkv, nm.qkv, dn, M, v.nq, d)
            dn1 = dn
        ops.rmsnorm_bwd(dn1, sv.x, v.n1, sv.r1, dx, v.gn1 if tr(nm.n1) else None, M, d, dx_accumulate=True, dx_bf16=dxb)

    # ------------------------------------------------------------------------------------------------ one Falcon block
    # transformers' FalconDecoderLayer with parallel_attn (falcon-7b):  n = LN(x);  y = x + dense(attn(rope(qkv(n)))) + W2 gelu(W1 n)
    def falcon_block_fwd(self, l: int, x, B: int, T: int, pos_offset: int, save: bool, vl=None):
        ls, v = self.dec.llama, self._llama_views(l)
        M, d, H, G, hd, ff = (vl.total if vl is not None else B * T), ls.d, ls.H, ls.Hkv, ls.hd, ls.ff
        cu, rpos = (vl.cu, vl.pos) if vl is not None else (None, None)
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        n1, m1, r1 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x, v.n1, v.b1, n1, m1, r1, M, d, eps=ls.eps)
        plan = self.dec_drop if save else None
        ldrop = (lambda site: plan.get(l, f'lora_{site}') if plan is not None else None)
        lo = {site: self._llama_lora(l, site) for site in ('qkv', 'o', 'gu', 'dn')}
        svlo = {}
        qkv = self._empty(M, v.nq, dtype=BF16)
        if lo['qkv'] is not None:
            svlo['qkv'] = self._lora_gemm(lo['qkv'], n1, v.Wqkv, qkv, M, ldrop('qkv'), save)
        else:
            self._lin(n1, v.Wqkv, v.names.qkv, qkv, M, v.nq, d)
        ops.rope(qkv, v.nq, 0, H + G, hd, self.rope_table(), M, pos=rpos, pos_offset=pos_offset, T=T)
        q3 = v3(qkv, v.nq)
        ao, lse = self._empty(M, H * hd, dtype=BF16), self._empty(H * M)
        ops.gq_attention_fwd(q3[..., :H * hd], q3[..., H * hd:(H + G) * hd], q3[..., (H + G) * hd:], v3(ao, H * hd), lse,
                             B, H, G, hd, T, T, True, cu_q=cu, cu_k=cu, total_q=M)
        x1 = self._empty(M, d)
        if lo['o'] is not None:
            svlo['o'] = self._lora_gemm(lo['o'], ao, v.Wo, x1, M, ldrop('o'), save, residual=x)
        else:
            self._lin(ao, v.Wo, v.names.o, x1, M, d, H * hd, residual=x)
        h, pre = self._empty(M, ff, dtype=BF16), (self._empty(M, ff, dtype=BF16) if save else None)
        if lo['gu'] is not None:
            svlo['gu'] = self._lora_gemm(lo['gu'], n1, v.Wgu, h, M, ldrop('gu'), save, act=ops.ACT_GELU_ERF, aux_out=pre)
        elif self._fp8_on(v.names.gu):                    # frozen base on fp8 operands: product -> pre-activation, one more pass applies the GELU
            tgt = pre if pre is not None else h
            self._lin(n1, v.Wgu, v.names.gu, tgt, M, ff, d)
            ops.gelu_fwd(tgt, h, erf=True)
        else:
            ops.gemm(n1, v.Wgu, h, M, ff, d, act=ops.ACT_GELU_ERF, aux_out=pre)
        x2 = self._empty(M, d)
        if lo['dn'] is not None:
            svlo['dn'] = self._lora_gemm(lo['dn'], h, v.Wdn, x2, M, ldrop('dn'), save, residual=x1)
        else:
            self._lin(h, v.Wdn, v.names.dn, x2, M, d, ff, residual=x1)
        return x2, (SimpleNamespace(x=x, n1=n1, m1=m1, r1=r1, qkv=qkv, ao=ao, lse=lse, h=h, pre=pre, lo=svlo,
                                    lo_drop={site: ldrop(site) for site in lo}) if save else None)

    def falcon_block_bwd(self, l: int, sv, dx, dxb, B: int, T: int, pos_offset: int, vl=None):
        """dx fp32 / dxb bf16: gradient w.r.t. the block output; on return both hold the gradient w.r.t. the block input"""
        ls, v = self.dec.llama, self._llama_views(l)
        M, d, H, G, hd, ff = (vl.total if vl is not None else B * T), ls.d, ls.H, ls.Hkv, ls.hd, ls.ff
        cu, rpos = (vl.cu, vl.pos) if vl is not None else (None, None)
        v3 = (lambda t, w: t) if vl is not None else (lambda t, w: t.view(B, T, w))
        tr = (lambda names: all(self.arena.trainable(n) for n in ([names] if isinstance(names, str) else names)))
        nm, svlo = v.names, sv.lo
        gview = (lambda names, G_: G_ if tr(names) else None)

        def lora_bwd(site, dY, x_in, W, names, G_):
            return self._lora_bwd(self._llama_lora(l, site), svlo[site], dY, x_in, W, gview(names, G_), None, M, sv.lo_drop.get(site))
        # ---- MLP branch: dn1 (fp32) = (dy W2 * gelu'(pre)) W1
        dpre = self._empty(M, ff, dtype=BF16)
        if 'dn' in svlo:
            ops.dgelu_mul(lora_bwd('dn', dxb, sv.h, v.Wdn, nm.dn, v.Gdn), sv.pre, dpre, erf=True)
        else:
            if tr(nm.dn):
                ops.gemm(dxb, sv.h, v.Gdn, d, ff, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            if self._fp8_on(nm.dn):
                dh32 = self._empty(M, ff)
                self._lin_dx(dxb, v.Wdn, nm.dn, dh32, M, d, ff)
                ops.dgelu_mul(dh32, sv.pre, dpre, erf=True)
            else:
                ops.gemm(dxb, v.Wdn, dpre, M, ff, d, b_kmajor=True, act=ops.ACT_DGELU_ERF, aux_in=sv.pre)
        if 'gu' in svlo:
            dn1 = lora_bwd('gu', dpre, sv.n1, v.Wgu, nm.gu, v.Ggu)
        else:
            if tr(nm.gu):
                ops.gemm(dpre, sv.n1, v.Ggu, ff, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            dn1 = self._empty(M, d)
            self._lin_dx(dpre, v.Wgu, nm.gu, dn1, M, ff, d)
        # ---- attention branch: both branches read the same LayerNorm output, their input gradients add up in dn1
        dao = self._empty(M, H * hd, dtype=BF16)
        if 'o' in svlo:
            ops.cast_f32_bf16(lora_bwd('o', dxb, sv.ao, v.Wo, nm.o, v.Go), dao)
        else:
            if tr(nm.o):
                ops.gemm(dxb, sv.ao, v.Go, d, H * hd, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dxb, v.Wo, nm.o, dao, M, d, H * hd)
        dqkv = self._empty(M, v.nq, dtype=BF16)
        q3, g3 = v3(sv.qkv, v.nq), v3(dqkv, v.nq)
        sl = (slice(0, H * hd), slice(H * hd, (H + G) * hd), slice((H + G) * hd, v.nq))
        ops.gq_attention_bwd(q3[..., sl[0]], q3[..., sl[1]], q3[..., sl[2]], v3(sv.ao, H * hd), v3(dao, H * hd), sv.lse,
                             self._empty(H * M), g3[..., sl[0]], g3[..., sl[1]], g3[..., sl[2]], B, H, G, hd, T, T, True,
                             cu_q=cu, cu_k=cu, total_q=M)
        ops.rope(dqkv, v.nq, 0, H + G, hd, self.rope_table(), M, pos=rpos, pos_offset=pos_offset, T=T, inverse=True)
        if 'qkv' in svlo:
            dn1.add_(lora_bwd('qkv', dqkv, sv.n1, v.Wqkv, nm.qkv, v.Gqkv))
        else:
            if tr(nm.qkv):
                ops.gemm(dqkv, sv.n1, v.Gqkv, v.nq, d, M, a_kmajor=True, b_kmajor=True, accumulate=True)
            self._lin_dx(dqkv, v.Wqkv, nm.qkv, dn1, M, v.nq, d, residual=dn1)
        ops.layernorm_bwd(dn1, sv.x, v.n1, sv.m1, sv.r1, dx, v.gn1 if tr(nm.n1) else None, v.gb1 if tr(nm.b1) else None, M, d,
                          dx_accumulate=True, dx_bf16=dxb)

    # ------------------------------------------------------------------------------------------------ the decoder stack
    def llama_decode_fwd(self, B: int, T: int, save: bool, ids, embeds, pos_offset: int, vl):
        """decode_segment for these decoders: (hidden fp32 [M, d] after the final norm, its bf16 copy, ctx)"""
        a, dc, ls = self.arena, self.dec, self.dec.llama
        d, M = dc.d, (vl.total if vl is not None else B * T)
        if ids is not None:
            ids = ids.to(device=a.device, dtype=torch.long).contiguous()
            x = self._empty(M, d)
            if vl is not None:
                ops.embed_fwd(ids, a.P(self.n_wte), None, x, M, 1, d, 0, dc.V)
            else:
                ops.embed_fwd(ids, a.P(self.n_wte), None, x, B, T, d, 0, dc.V)
        else:
            x = embeds.to(device=a.device, dtype=F32).contiguous().view(M, d)
        saves, cur = [], x
        block = self.falcon_block_fwd if ls.arch == 'falcon' else self.llama_block_fwd
        sink = getattr(self, '_layer_sink', None)         # generation's prompt prefill: takes a layer's K / V and lets the rest go
        for l in range(dc.L):
            cur, sv = block(l, cur, B, T, pos_offset, save, vl)
            if sink is not None and sv is not None:
                sink(l, sv)
                sv = None
            saves.append(sv)
        wn = self.dp + ls.norm_f
        hid, hb, rf, mf = self._empty(M, d), self._empty(M, d, dtype=BF16), self._empty(M), None
        if ls.arch == 'falcon':          # ln_f: a LayerNorm; the fp32 output is what forward() returns, the bf16 copy feeds the head
            mf = self._empty(M)
            ops.layernorm_fwd(cur, a.P(wn + '.weight'), a.P(wn + '.bias'), hid, mf, rf, M, d, eps=ls.eps)
            ops.cast_f32_bf16(hid, hb)
        else:
            ops.rmsnorm_fwd(cur, a.P(wn + '.weight'), hb, rf, M, d, ls.eps, y_f32=hid)
        ctx = SimpleNamespace(ids=ids, saves=saves, xl=cur, rf=rf, mf=mf, hb=hb, B=B, T=T, S=0, pos_offset=pos_offset, vl=vl, M=M,
                              emb_drop=None, pos_ctx=None) if save else None
        return hid, hb, ctx

    def llama_decode_bwd(self, ctx, dh):
        """dh fp32 [M, d]: gradient w.r.t. the final norm's output.  Returns the gradient w.r.t. the block stack's input."""
        a, dc = self.arena, self.dec
        M, d = ctx.M, dc.d
        dx, dxb = self._empty(M, d), self._empty(M, d, dtype=BF16)
        ls = dc.llama
        wn = self.dp + ls.norm_f
        if ls.arch == 'falcon':
            ops.layernorm_bwd(dh, ctx.xl, a.P(wn + '.weight'), ctx.mf, ctx.rf, dx, a.Gt(wn + '.weight'), a.Gt(wn + '.bias'), M, d, dx_bf16=dxb)
        else:
            ops.rmsnorm_bwd(dh, ctx.xl, a.P(wn + '.weight'), ctx.rf, dx, a.Gt(wn + '.weight'), M, d, dx_bf16=dxb)
        block = self.falcon_block_bwd if ls.arch == 'falcon' else self.llama_block_bwd
        for l in reversed(range(dc.L)):
            block(l, ctx.saves[l], dx, dxb, ctx.B, ctx.T, ctx.pos_offset, ctx.vl)
        return dx
