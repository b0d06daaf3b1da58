"""LoRA adapters on the HIP hot path (reference models/utils.py:46-65 wraps the decoder in peft's LoraModel:
every targeted linear computes ``y = base(x) + lora_B(lora_A(dropout(x))) * lora_alpha / r``, the base parameters are frozen).

How it maps to the MI355X (dense GPT-2 blocks, engine.block_fwd / block_bwd):
  * forward: the adapter rides in the K panel of the layer's own GEMM --  ``[x | u] . [W | s B | 0]^T`` with ``u = dropout(x) . A^T`` --
    so the layer's fused epilogue (bias, GELU + saved pre-activation, residual + dropout) sees base + adapter as ONE accumulator and
    nothing is added afterwards.  The rank is padded to LPAD = 128 columns (zero rows behind lora_A in the arena, zero columns behind
    s B): K + 128 keeps the 256^2 persistent kernel's K % 128 rule, and u = x . A_pad^T is itself a plain GEMM (N = 128);
  * backward: the frozen base weight's dW GEMM -- a third of a layer's GEMM flops -- is skipped (ParamArena.trainable); the adapter
    costs four thin GEMMs: dB = s . dY^T u (N = 128), du = dY . (s B) (N = 128), dA = du^T dropout(x) (M = 128), and
    dx = dY . W + dropout(du . A) (K = 128, the adapter's input dropout applied by the GEMM's residual + dropout epilogue);
  * generation (no dropout): merged weights W + s B A in persistent bf16 buffers, refreshed per generate() call -- the decode graph
    runs the un-adapted step.
"""
from types import SimpleNamespace

import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
LPAD = 128


class LoraAdapters:
    """Mixin of engine.HotPath."""

    def _lora_site(self, l: int, site: str):
        lo = getattr(self.dec, 'lora', None)
        if lo is None or site not in lo.sites:
            return None
        key = ('lora', l, site, id(self.arena))
        v = self._sub_cache.get(key)
        if v is None:
            a = self.arena
            nA, nB = f'{self.dp}lora_params.h{l}_{site}_A', f'{self.dp}lora_params.h{l}_{site}_B'
            K, N = a.entries[nA][2][1], a.entries[nB][2][0]
            names = [nA] + ([nA + '.<pad>'] if lo.r < LPAD else [])
            v = self._sub_cache[key] = SimpleNamespace(
                K=K, N=N, r=lo.r, scale=lo.scale, kind=f'lora_{site}', A=a.span('W', names, (LPAD, K)), GA=a.span('G', names, (LPAD, K)),
                B=a.P(nB), GB=a.G(nB), parts=[(0, N, 0, a.P(nB), a.G(nB))], nA=nA)
        return v

    def _rank_gemm(self, a, b, out, M: int, K: int, b_kmajor: bool = False):
        """out bf16 [M, LPAD] = a [M, K] . b (b: [LPAD, K], or [K, LPAD] when b_kmajor) -- the adapters' u = dropout(x) A^T and
        du = dY (s B).  128 output columns are M / 256 half-empty tiles of the persistent kernel (12 820 rows: 51 workgroups, 1.4 TB/s);
        for a long K the fp32 split-K form (atomics into a zeroed plane, ~4 slices) + one cast runs 2.2x faster (tools/ab_thin_gemm.py)."""
        if K < 2048 or M < 2048:
            return ops.gemm(a, b, out, M, LPAD, K, b_kmajor=b_kmajor)
        plane = torch.zeros(M, LPAD, dtype=F32, device=out.device)
        ops.gemm(a, b, plane, M, LPAD, K, b_kmajor=b_kmajor, accumulate=True)
        return ops.cast_f32_bf16(plane, out)

    def _lora_panel(self, ls, dtype=BF16):
        """[N, LPAD] = the adapter's B matrices at their (row block, rank column block) -- ONE block for a plain linear, block-diagonal for
        a fused projection (q | k | v, gate | up: engine_llama) -- times the LoRA scale, zero elsewhere."""
        dev = ls.A.device
        panel = torch.zeros(ls.N, LPAD, dtype=dtype, device=dev)
        for row0, nrows, col0, B, _ in ls.parts:
            panel[row0:row0 + nrows, col0:col0 + B.shape[1]].copy_(B * ls.scale)
        return panel

    def _lora_gemm(self, ls, x, W, out, M: int, drop_l, save: bool, **epilogue):
        """out = epilogue([x | u] . [W | s B | 0]^T), u = dropout(x) . A^T.  x bf16 [M, K] contiguous, W bf16 [N, K].  Returns what
        the backward needs (u and s B, both [*, LPAD] bf16) when save."""
        K, N = ls.K, ls.N
        if getattr(ls, 'wnames', None) is not None and self._fp8_on(ls.wnames) and epilogue.get('act', 0) in (0, ops.ACT_GELU, ops.ACT_GELU_ERF):
            return self._lora_gemm_fp8(ls, x, W, out, M, drop_l, save, **epilogue)
        epilogue.pop('xq', None)
        xcat = torch.empty(M, K + LPAD, dtype=BF16, device=x.device)
        xd = torch.empty(M, K, dtype=BF16, device=x.device) if drop_l is not None else None
        ops.lora_stage(x, xcat, xd, M, K, drop_l)                    # one pass: x into the concatenated operand + its masked copy
        u = torch.empty(M, LPAD, dtype=BF16, device=x.device)
        self._rank_gemm(xd if xd is not None else x, ls.A, u, M, K)
        xcat[:, K:].copy_(u)
        wcat = torch.empty(N, K + LPAD, dtype=BF16, device=x.device)
        wcat[:, :K].copy_(W)
        wcat[:, K:].copy_(self._lora_panel(ls))
        ops.gemm(xcat, wcat, out, M, N, K + LPAD, **epilogue)
        # (the masked copy of x is kept for dA = du^T dropout(x): re-making it in backward cost two more passes over [M, K])
        return SimpleNamespace(u=u, sB=wcat[:, K:].contiguous(), xd=xd) if save else None

    def _lora_gemm_fp8(self, ls, x, W, out, M: int, drop_l, save: bool, bias=None, residual=None, act=0, aux_out=None, xq=None, **_):
        """The same layer with its FROZEN base weight on fp8 operands (I2T_FP8=1; engine_llama._fp8_*): the base product runs
        at the fp8 MFMA rate, so the adapter leaves the K panel -- out = fp8(x) . fp8(W)^T (+ bias) (+ residual) + u . (s B)^T, the
        rank-128 product added by a second, thin GEMM (in place on an fp32 output; ahead of the base GEMM, as its fp32 residual, when
        the output is bf16).  No K-concatenated copies of x and W; one quantisation pass over x instead."""
        K, N = ls.K, ls.N
        if act:          # GELU behind the layer (Falcon's dense_h_to_4h): the product goes to the pre-activation buffer, one more pass applies it
            pre = aux_out if aux_out is not None else torch.empty(M, N, dtype=BF16, device=x.device)
            sv = self._lora_gemm_fp8(ls, x, W, pre, M, drop_l, save, bias=bias, residual=residual, xq=xq)
            ops.gelu_fwd(pre, out, erf=(act == ops.ACT_GELU_ERF))
            return sv
        xd = None
        if drop_l is not None:
            xd = x.clone()
            ops.dropout_apply(xd, M, K, drop_l)                      # (the index space of lora_stage and of the backward's epilogue mask)
        u = torch.empty(M, LPAD, dtype=BF16, device=x.device)
        self._rank_gemm(xd if xd is not None else x, ls.A, u, M, K)
        panel = self._lora_panel(ls)
        e = self._fp8_weight(ls.wnames, W)
        x8, sx = xq if xq is not None else self._fp8_rows(x, M, K)          # (xq: the producer of x emitted the e4m3 row beside the bf16 one)
        if out.dtype == F32:
            ops.gemm_fp8(x8, sx, e.w8, e.sw, out, M, N, x8.shape[1], bias=bias, residual=residual)      # (K' = the zero-padded row length)
            ops.gemm(u, panel, out, M, N, LPAD, residual=out)
        else:
            tmp = torch.empty(M, N, dtype=F32, device=x.device)
            ops.gemm(u, panel, tmp, M, N, LPAD, residual=residual)
            ops.gemm_fp8(x8, sx, e.w8, e.sw, out, M, N, x8.shape[1], bias=bias, residual=tmp)
        return SimpleNamespace(u=u, sB=panel, xd=xd) if save else None

    def _lora_bwd(self, ls, sv_l, dY, x, W, gW, gb, M: int, drop_l, dq=None):
        """dY bf16 [M, N]: gradient w.r.t. the adapted linear's pre-epilogue output.  gW / gb: gradient views of the base weight /
        bias or None (frozen).  Accumulates every parameter gradient; returns dx fp32 [M, K] = dY . W + dropout(du . A)."""
        K, N = ls.K, ls.N
        if gb is not None:
            ops.colsum(dY, gb, M, N, accumulate=True)
        if gW is not None:
            ops.gemm(dY, x, gW, N, K, M, a_kmajor=True, b_kmajor=True, accumulate=True)
        tmp = torch.zeros(N, LPAD, dtype=F32, device=dY.device)
        ops.gemm(dY, sv_l.u, tmp, N, LPAD, M, a_kmajor=True, b_kmajor=True, accumulate=True)
        for row0, nrows, col0, B, GB in ls.parts:             # (a fused projection: only the diagonal blocks are parameters)
            GB.add_(tmp[row0:row0 + nrows, col0:col0 + B.shape[1]], alpha=ls.scale)
        du = torch.empty(M, LPAD, dtype=BF16, device=dY.device)
        self._rank_gemm(dY, sv_l.sB, du, M, N, b_kmajor=True)
        xd = sv_l.xd if sv_l.xd is not None else x
        ops.gemm(du, xd, ls.GA, LPAD, K, M, a_kmajor=True, b_kmajor=True, accumulate=True)
        dx = torch.empty(M, K, dtype=F32, device=dY.device)
        if getattr(ls, 'wnames', None) is not None and self._fp8_on(ls.wnames):      # frozen base weight: dx on fp8 operands too
            e = self._fp8_weight(ls.wnames, W)
            d8, sd = dq if dq is not None else self._fp8_rows(dY, M, N)
            ops.gemm_fp8(d8, sd, e.wt8, e.swt, dx, M, K, d8.shape[1])
        else:
            ops.gemm(dY, W, dx, M, K, N, b_kmajor=True)
        ops.gemm(du, ls.A, dx, M, K, LPAD, b_kmajor=True, residual=dx, drop=drop_l)
        return dx

    # ------------------------------------------------------------------------------------------------ generation: merged weights
    def lora_merged(self, l: int, site: str, W_name: str, rows=None):
        """bf16 W + s B A of one adapted linear in a persistent buffer (same address on every call: captured decode graphs read it);
        ``prepare_lora_merged`` recomputes the contents from the current parameters."""
        key = ('lora_merged', l, site, id(self.arena))
        buf = self._sub_cache.get(key)
        if buf is None:
            ls = self._lora_site(l, site) if self.dec.llama is None else self._llama_lora(l, site)
            buf = self._sub_cache[key] = torch.empty(ls.N, ls.K, dtype=BF16, device=self.arena.device)
            self._lora_merge_list.append((buf, l, site, W_name, rows))
        return buf

    def prepare_lora_merged(self):
        """Create every adapted linear's merged-weight buffer and bring the contents up to date with the parameters (skipped when
        nothing changed since the last call).  Called before a decode -- by ConcurrentGreedyDecoder on the parent stream, before it
        fans out: the buffers are shared by its lanes."""
        lo = getattr(self.dec, 'lora', None)
        if lo is None:
            return
        a, d = self.arena, self.dec.d
        for l in range(self.dec.L if self.dec.llama is not None else 0):          # Llama / Qwen2 blocks (engine_llama._llama_lora)
            v = self._llama_views(l)
            for site, names in (('qkv', v.names.qkv), ('o', v.names.o), ('gu', v.names.gu), ('dn', v.names.dn)):
                if self._llama_lora(l, site) is not None:
                    self.lora_merged(l, site, names, None)
        for l in range(self.dec.L if self.dec.llama is None else 0):
            p = f'{self.dp}transformer.h.{l}'
            for site, name, rows in (('attn_c_attn', f'{p}.attn.c_attn.weight', None), ('mlp_c_fc', f'{p}.mlp.c_fc.weight', None),
                                     ('mlp_c_proj', f'{p}.mlp.c_proj.weight', None),
                                     ('xattn_c_attn', f'{p}.cross_attn.in_proj_weight', slice(d, 3 * d))):
                if site in lo.sites and name in a.entries:
                    self.lora_merged(l, site, name, rows)
        ver = (id(a), len(self._lora_merge_list), a.generation)       # (prepare() has run refresh_shadow: torch-side writes are counted)
        if ver == getattr(self, '_lora_merged_ver', None):
            return
        self._lora_merged_ver = ver
        for buf, l, site, W_name, rows in self._lora_merge_list:
            ls = self._lora_site(l, site) if self.dec.llama is None else self._llama_lora(l, site)
            W = a.P(W_name) if isinstance(W_name, str) else a.span('P', W_name, (ls.N, ls.K))
            W = W if rows is None else W[rows]
            # buf = bf16(W + s B A): one GEMM on the padded rank (the [N, LPAD] panel of _lora_panel; A's pad rows are zero in the arena),
            # the fp32 base weight as the epilogue's residual
            ops.gemm(self._lora_panel(ls), ls.A, buf, ls.N, ls.K, LPAD, b_kmajor=True, residual=W.contiguous())
