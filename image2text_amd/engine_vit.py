"""The ``PretrainedViT`` encoder on the HIP path (reference models/encoder.py:56-127): torchvision's ViT-B/16 backbone and the three
heads that turn its 768-wide class-token feature into ``n_cls`` encoder outputs.  A mixin of ``engine.HotPath``.

Backbone (torchvision ``VisionTransformer.forward`` restated; the module tree ``models/encoder.py::TorchvisionViT`` holds the
parameters under torchvision's own names):
    conv_proj (16 x 16, stride 16)  -> ``i2t_patchify`` (im2col rows, bf16) + one GEMM (K = 768)
    [class_token | patches] + pos   -> ``i2t_vit_tokens``
    12 x EncoderBlock               -> LayerNorm (eps 1e-6) -> packed in_proj GEMM -> ``i2t_attention_fwd`` (12 heads of 64, T = 197,
                                       no mask, no dropout) -> out_proj GEMM + residual -> LayerNorm -> mlp.0 GEMM with the exact-erf
                                       GELU epilogue (``I2T_ACT_GELU_ERF``) -> mlp.3 GEMM + residual
    encoder.ln on row 0             -> the feature, fp32 [B, 768]
``refine_base_model: False`` (5 of the 7 shipped yamls with this encoder): the backbone runs without saving anything and its
parameters never receive a gradient (the reference wraps it in ``torch.no_grad``, encoder.py:110-112; the arena marks them
``skip_grad`` so that optimizers leave them alone exactly as torch does with ``grad is None``).  ``refine_base_model: True``: the
hand-written backward below (no gradient normaliser, no dropout in these blocks).

Heads
    slot MLPs  (encoder.py:118-119)  normalize -> one private MLP per slot -> normalize.  Every slot sees the same input, its weights sit
               a constant stride apart in the arena -> ``i2t_grouped_gemm`` with group = slot on slot-major rows (the per-position MLP
               machinery of the decoder's ``use_advanced_pos_emb``), the Linear residual connector as one more grouped GEMM.
    PEER       (encoder.py:115-116, layers.py:37-109)  (768, 768, n_cls) expansion = one GEMM on the [768, 768 n_cls] view of
               ``peer_proj_wt``, the four linear maps as GEMMs, then ``i2t_peer_lookup_fwd`` (top-k / softmax / expert gathers).
    LSH        (encoder.py:117, layers.py:112-143,190-219)  fp32 projections (``i2t_gemm_f32``: bucket boundaries are discontinuities,
               bf16 operands would move rows across them), ``i2t_lsh_embed_fwd``; gradients reach the embedding tables only.
"""
from types import SimpleNamespace

import numpy as np
import torch

from . import ops

BF16, F32 = torch.bfloat16, torch.float32
VIT_EPS = 1e-6


class ViTEncoder:
    # ------------------------------------------------------------------------------------------------ set-up
    def vit_setup(self, model, ecfg):
        enc_mod = model.encoder[0] if model.has_bridge else model.encoder
        spec = enc_mod.model.spec
        d, p, img = spec['hidden_dim'], spec['patch_size'], spec['image_size']
        head = 'peer' if enc_mod.use_peer else ('lsh' if enc_mod.use_lsh else 'mlp')
        self.enc = SimpleNamespace(kind='vit', ncls=ecfg.n_cls, d=d, out=ecfg.n_embd_out_vit, L=spec['num_layers'], H=spec['num_heads'],
                                   ff=spec['mlp_dim'], p=p, img=img, P2=(img // p) ** 2, T=(img // p) ** 2 + 1, dropout=0.0,
                                   attn_dropout=0.0, fam=None, causal=False, refine=bool(enc_mod.refine), head=head, module=enc_mod)
        if head == 'peer':
            pc = ecfg.peer_config
            self.enc.peer = SimpleNamespace(nq=pc.num_units_sqrt, topk=pc.topk, nh=pc.nhead, qd=enc_mod.peer.query_dim)
        if head == 'lsh':
            lc = ecfg.lsh_config
            self.enc.lsh = SimpleNamespace(bins=tuple(lc.num_bins), n_proj=lc.num_proj, cache=None)
        self.conv, self.conv_mfma, self.cls_only_last = [], False, False

    def vit_skip_grad(self):
        """Names of the backbone's parameters when it is frozen by no_grad (refine off): no gradient is ever produced for them."""
        if self.enc.refine:
            return set()
        return {n for n in self.arena.entries if n.startswith(f'{self.ep}model.')}

    # ------------------------------------------------------------------------------------------------ backbone
    def _vit_lin(self, x_bf, W, names, out, M, N, K, **kw):
        """A backbone GEMM: e4m3 operands only under the backbone's own switch (``fp8_vit``, I2T_FP8_VIT=1) and only for frozen weights;
        the decoder's switch (``fp8``: I2T_FP8 / a load_in_4bit request) never reaches the encoder."""
        if self.fp8_vit and all(not self.arena.trainable(n) for n in ([names] if isinstance(names, str) else names)):
            fp8, self.fp8 = self.fp8, True
            try:
                return self._lin(x_bf, W, names, out, M, N, K, **kw)
            finally:
                self.fp8 = fp8
        return ops.gemm(x_bf, W, out, M, N, K, **kw)

    def _vit_block_fwd(self, q: str, x, B, T, d, H, ff, save: bool):
        a, M = self.arena, B * T
        ln1, m1, r1 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x, a.P(q + 'ln_1.weight'), a.P(q + 'ln_1.bias'), ln1, m1, r1, M, d, eps=VIT_EPS)
        qkv = self._empty(M, 3 * d, dtype=BF16)
        # (self._vit_lin: e4m3 operands when the backbone is frozen and I2T_FP8_VIT=1 -- forward-only GEMMs of weights that never change)
        lin = (lambda x_, name, out, N_, K_, **kw: self._vit_lin(x_, a.W(name), name, out, M, N_, K_, **kw))
        lin(ln1, q + 'self_attention.in_proj_weight', qkv, 3 * d, d, bias=a.P(q + 'self_attention.in_proj_bias'))
        q3 = qkv.view(B, T, 3 * d)
        ao, lse = self._empty(B, T, d, dtype=BF16), self._empty(H * M)
        ops.attention_fwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], ao, lse, B, H, T, T, False)
        x1 = self._empty(M, d)
        lin(ao.view(M, d), q + 'self_attention.out_proj.weight', x1, d, d, bias=a.P(q + 'self_attention.out_proj.bias'), residual=x)
        ln2, m2, r2 = self._empty(M, d, dtype=BF16), self._empty(M), self._empty(M)
        ops.layernorm_fwd(x1, a.P(q + 'ln_2.weight'), a.P(q + 'ln_2.bias'), ln2, m2, r2, M, d, eps=VIT_EPS)
        h = self._empty(M, ff, dtype=BF16)
        pre = self._empty(M, ff, dtype=BF16) if save else None
        if save:
            ops.gemm(ln2, a.W(q + 'mlp.0.weight'), h, M, ff, d, bias=a.P(q + 'mlp.0.bias'), act=ops.ACT_GELU_ERF, aux_out=pre)
        else:
            lin(ln2, q + 'mlp.0.weight', h, ff, d, bias=a.P(q + 'mlp.0.bias'), act=ops.ACT_GELU_ERF)
        x2 = self._empty(M, d)
        lin(h, q + 'mlp.3.weight', x2, d, ff, bias=a.P(q + 'mlp.3.bias'), residual=x1)
        sv = SimpleNamespace(x=x, ln1=ln1, m1=m1, r1=r1, qkv=qkv, ao=ao, lse=lse, x1=x1, ln2=ln2, m2=m2, r2=r2, h=h, pre=pre) if save else None
        return x2, sv

    def _vit_block_bwd(self, q: str, sv, dx, dxb, B, T, d, H, ff):
        """dx fp32 [M, d] / dxb its bf16 copy: gradient w.r.t. the block output; on return both hold the gradient w.r.t. its input."""
        a, M = self.arena, B * T
        dpre = self._empty(M, ff, dtype=BF16)
        self._linear_bwd(dxb, M, d, ff, sv.h, q + 'mlp.3.weight', q + 'mlp.3.bias', dx_out=dpre, act=ops.ACT_DGELU_ERF, aux_in=sv.pre)
        dln = self._empty(M, d, dtype=BF16)
        self._linear_bwd(dpre, M, ff, d, sv.ln2, q + 'mlp.0.weight', q + 'mlp.0.bias', dx_out=dln)
        ops.layernorm_bwd(dln, sv.x1, a.P(q + 'ln_2.weight'), sv.m2, sv.r2, dx, a.G(q + 'ln_2.weight'), a.G(q + 'ln_2.bias'), M, d,
                          dx_accumulate=True, dx_bf16=dxb)
        dao = self._empty(B, T, d, dtype=BF16)
        self._linear_bwd(dxb, M, d, d, sv.ao.view(M, d), q + 'self_attention.out_proj.weight', q + 'self_attention.out_proj.bias',
                         dx_out=dao.view(M, d))
        dqkv = self._empty(B, T, 3 * d, dtype=BF16)
        q3 = sv.qkv.view(B, T, 3 * d)
        ops.attention_bwd(q3[..., :d], q3[..., d:2 * d], q3[..., 2 * d:], sv.ao, dao, sv.lse, self._empty(H * M), dqkv[..., :d],
                          dqkv[..., d:2 * d], dqkv[..., 2 * d:], B, H, T, T, False)
        self._linear_bwd(dqkv.view(M, 3 * d), M, 3 * d, d, sv.ln1, q + 'self_attention.in_proj_weight', q + 'self_attention.in_proj_bias',
                         dx_out=dln)
        ops.layernorm_bwd(dln, sv.x, a.P(q + 'ln_1.weight'), sv.m1, sv.r1, dx, a.G(q + 'ln_1.weight'), a.G(q + 'ln_1.bias'), M, d,
                          dx_accumulate=True, dx_bf16=dxb)

    def vit_backbone_fwd(self, images, save: bool):
        a, e = self.arena, self.enc
        B = images.shape[0]
        if tuple(images.shape[1:]) != (3, e.img, e.img):
            raise AssertionError(f'Wrong image size! Expected (3, {e.img}, {e.img}) but got {tuple(images.shape[1:])}')      # torchvision's check
        m, d, T, K0 = f'{self.ep}model.', e.d, e.T, 3 * e.p * e.p
        patches = self._empty(B * e.P2, K0, dtype=BF16)
        ops.patchify(images, patches, B, 3, e.img, e.img, e.p)
        proj = self._empty(B * e.P2, d)
        self._vit_lin(patches, a.W(m + 'conv_proj.weight').view(d, K0), m + 'conv_proj.weight', proj, B * e.P2, d, K0, bias=a.P(m + 'conv_proj.bias'))
        x = self._empty(B * T, d)
        ops.vit_tokens(proj, a.P(m + 'class_token'), a.P(m + 'encoder.pos_embedding'), x, B, T, d)
        saves = []
        for l in range(e.L):
            x, sv = self._vit_block_fwd(f'{m}encoder.layers.encoder_layer_{l}.', x, B, T, d, e.H, e.ff, save)
            saves.append(sv)
        cls = self._empty(B, d)
        ops.copy_rows(x, T * d, cls, d, B, 1, d)                      # x[:, 0]: the final LayerNorm is row-wise, only row 0 is read
        feat, mf, rf = self._empty(B, d), self._empty(B), self._empty(B)
        ops.layernorm_fwd(cls, a.P(m + 'encoder.ln.weight'), a.P(m + 'encoder.ln.bias'), feat, mf, rf, B, d, eps=VIT_EPS)
        ctx = SimpleNamespace(patches=patches, saves=saves, cls=cls, mf=mf, rf=rf) if save else None
        return feat, ctx

    def vit_backbone_bwd(self, ctx, dfeat, B: int):
        a, e = self.arena, self.enc
        m, d, T, K0 = f'{self.ep}model.', e.d, e.T, 3 * e.p * e.p
        dcls = self._empty(B, d)
        ops.layernorm_bwd(dfeat, ctx.cls, a.P(m + 'encoder.ln.weight'), ctx.mf, ctx.rf, dcls, a.G(m + 'encoder.ln.weight'),
                          a.G(m + 'encoder.ln.bias'), B, d)
        dx = torch.zeros(B, T, d, dtype=F32, device=a.device)
        ops.copy_rows(dcls, d, dx, T * d, B, 1, d)
        dx = dx.view(B * T, d)
        dxb = self._empty(B * T, d, dtype=BF16)
        ops.cast_f32_bf16(dx, dxb)
        for l in reversed(range(e.L)):
            self._vit_block_bwd(f'{m}encoder.layers.encoder_layer_{l}.', ctx.saves[l], dx, dxb, B, T, d, e.H, e.ff)
        # tokens: pos_embedding and class_token see the batch sum; the patch rows feed conv_proj's GEMM
        ssum = self._empty(T, d)
        ops.sum_over_batch(dx, T * d, ssum, B, T, d)
        ops.add_(a.G(m + 'encoder.pos_embedding').view(T, d), ssum)
        ops.add_(a.G(m + 'class_token').view(1, d), ssum[:1])
        dproj = self._empty(B * e.P2, d)
        ops.copy_rows(dx.view(B, T, d)[:, 1:], T * d, dproj, e.P2 * d, B, e.P2, d)
        dpb = self._empty(B * e.P2, d, dtype=BF16)
        ops.cast_f32_bf16(dproj, dpb)
        ops.colsum(dpb, a.G(m + 'conv_proj.bias'), B * e.P2, d, accumulate=True)
        ops.gemm(dpb, ctx.patches, a.G(m + 'conv_proj.weight').view(d, K0), d, K0, B * e.P2, a_kmajor=True, b_kmajor=True, accumulate=True)

    # ------------------------------------------------------------------------------------------------ slot-MLP head
    def _slot_views(self):
        a = self.arena
        key = ('vit_slots', a.p32.data_ptr())
        pv = self._moe_cache.get(key)
        if pv is not None:
            return pv
        pfx, n = f'{self.ep}proj.models', self.enc.ncls

        def layer(w, b):
            ow, _, shp = a.entries[w]
            ob = a.entries[b][0]
            return SimpleNamespace(N=shp[0], K=shp[1], W=a.pbf[ow:], gW=a.g32[ow:], b=a.p32[ob:], gb=a.g32[ob:], w=w, bn=b)
        layers, i = [], 0
        while f'{pfx}.0.model.{i}.weight' in a.entries:
            layers.append(layer(f'{pfx}.0.model.{i}.weight', f'{pfx}.0.model.{i}.bias'))
            i += 2
        res = layer(f'{pfx}.0.residual_connector.weight', f'{pfx}.0.residual_connector.bias') \
            if f'{pfx}.0.residual_connector.weight' in a.entries else None
        stride = a.entries[f'{pfx}.1.model.0.weight'][0] - a.entries[f'{pfx}.0.model.0.weight'][0] if n > 1 else 0
        for t in range(1, n):                                  # the constant-stride layout the grouped GEMMs rely on
            for L in layers + ([res] if res is not None else []):
                for nm in (L.w, L.bn):
                    assert a.entries[nm.replace(f'{pfx}.0.', f'{pfx}.{t}.')][0] - a.entries[nm][0] == t * stride
        pv = SimpleNamespace(layers=layers, res=res, stride=stride)
        self._moe_cache[key] = pv
        return pv

    def _vit_head_mlp_fwd(self, feat, B: int, save: bool):
        e, pv = self.enc, self._slot_views()
        d, n, M = e.d, e.ncls, e.ncls * B
        plan = self._pos_plan(B, n, None)                      # slot-major row order of a [B, n_cls] batch
        xn, inv0 = self._empty(B, d), self._empty(B)
        ops.l2norm_fwd(feat, xn, None, inv0, B, d)
        rep32 = self._empty(M, d)
        ops.bcast_rows(xn, rep32, B * d, n, B, d)              # every slot's rows = the same B normalised features
        repb = self._empty(M, d, dtype=BF16)
        ops.cast_f32_bf16(rep32, repb)
        kw = dict(seg=plan.seg, n_groups=n, max_rows=B, group0=0, b_group_stride=pv.stride, bias_group_stride=pv.stride)
        if pv.res is not None:
            r32 = self._empty(M, e.out)
            ops.grouped_gemm(0, repb, pv.res.W[:pv.res.N * pv.res.K].view(pv.res.N, pv.res.K), r32, pv.res.N, pv.res.K, bias=pv.res.b, **kw)
        else:
            r32 = rep32
        h, hs, pres = repb, [repb], []
        for i, L in enumerate(pv.layers):
            last = i == len(pv.layers) - 1
            out = self._empty(M, L.N, dtype=F32 if last else BF16)
            pre = self._empty(M, L.N, dtype=BF16) if (save and not last) else None
            ops.grouped_gemm(0, h, L.W[:L.N * L.K].view(L.N, L.K), out, L.N, L.K, bias=L.b, act=0 if last else 1, aux_out=pre,
                             residual=r32 if last else None, **kw)
            if not last:
                hs.append(out)
                pres.append(pre)
            h = out
        yn, inv1 = self._empty(M, e.out), self._empty(M)
        ops.l2norm_fwd(h, yn, None, inv1, M, e.out)
        enc_out = self._empty(M, e.out)
        ops.scatter_rows(yn, plan.rows, enc_out, M, e.out)     # slot-major -> (image, slot)
        ctx = SimpleNamespace(plan=plan, feat=feat, inv0=inv0, y=h, inv1=inv1, hs=hs, pres=pres, repb=repb) if save else None
        return enc_out, ctx

    def _vit_head_mlp_bwd(self, ctx, denc, B: int):
        """denc fp32 [B * n_cls, out] -> gradients of every slot MLP; returns d/d(feature) fp32 [B, d] (None unless refining)."""
        e, pv = self.enc, self._slot_views()
        d, n, M, plan = e.d, e.ncls, e.ncls * B, ctx.plan
        g32 = self._empty(M, e.out)
        ops.gather_rows(denc, plan.rows, M, e.out, out_f32=g32)
        dy32 = self._empty(M, e.out)
        ops.l2norm_bwd(g32, ctx.y, ctx.inv1, dy32, M, e.out)
        g = self._empty(M, e.out, dtype=BF16)
        ops.cast_f32_bf16(dy32, g)
        kw = dict(seg=plan.seg, n_groups=n, max_rows=B, group0=0)
        if pv.res is not None:
            R = pv.res
            ops.grouped_gemm(2, g, ctx.repb, R.gW[:R.N * R.K].view(R.N, R.K), R.N, R.K, c_group_stride=pv.stride, accumulate=True, **kw)
            ops.grouped_colsum(g, plan.seg, n, R.gb, pv.stride, 0, R.N)
            drep = self._empty(M, d)
            ops.grouped_gemm(1, g, R.W[:R.N * R.K].view(R.N, R.K), drep, R.N, R.K, b_group_stride=pv.stride, **kw)
        else:
            drep = dy32                                        # identity residual connector (widths equal)
        de = None
        for i in reversed(range(len(pv.layers))):
            L = pv.layers[i]
            ops.grouped_gemm(2, g, ctx.hs[i], L.gW[:L.N * L.K].view(L.N, L.K), L.N, L.K, c_group_stride=pv.stride, accumulate=True, **kw)
            ops.grouped_colsum(g, plan.seg, n, L.gb, pv.stride, 0, L.N)
            W = L.W[:L.N * L.K].view(L.N, L.K)
            if i > 0:
                nxt = self._empty(M, L.K, dtype=BF16)
                ops.grouped_gemm(1, g, W, nxt, L.N, L.K, b_group_stride=pv.stride, act=2, aux_in=ctx.pres[i - 1], **kw)
                g = nxt
            elif e.refine:
                de = self._empty(M, d)
                ops.grouped_gemm(1, g, W, de, L.N, L.K, b_group_stride=pv.stride, residual=drep, **kw)
        if not e.refine:
            return None
        dxn = self._empty(B, d)
        ops.sum_over_batch(de, B * d, dxn, n, B, d)            # every slot read the same normalised feature
        dfeat = self._empty(B, d)
        ops.l2norm_bwd(dxn, ctx.feat, ctx.inv0, dfeat, B, d)
        return dfeat

    # ------------------------------------------------------------------------------------------------ PEER head
    def _vit_head_peer_fwd(self, feat, B: int, save: bool):
        a, e, pc = self.arena, self.enc, self.enc.peer
        d, n, M, q = e.d, e.ncls, e.ncls * B, f'{self.ep}peer.'
        featb = self._empty(B, d, dtype=BF16)
        ops.cast_f32_bf16(feat, featb)
        Wm = a.W(f'{self.ep}peer_proj_wt').view(d, d * n)      # [d_in][(e, s)]: a k-major B operand
        Y = self._empty(B, d * n)
        ops.gemm(featb, Wm, Y, B, d * n, d, b_kmajor=True)
        inpb = self._empty(M, d, dtype=BF16)
        ops.transpose_last2(Y, None, inpb, B, d, n)            # (b, e, s) -> rows (b, s) of width e
        xq = self._empty(M, pc.nh * pc.qd, dtype=BF16)
        ops.gemm(inpb, a.W(q + 'query_linear.weight'), xq, M, pc.nh * pc.qd, d)
        Wlr = a.span('W', [q + 'query_left.linear.weight', q + 'query_right.linear.weight'], (2 * pc.nq, pc.qd))
        S = self._empty(M * pc.nh, 2 * pc.nq)
        ops.gemm(xq.view(M * pc.nh, pc.qd), Wlr, S, M * pc.nh, 2 * pc.nq, pc.qd)
        ipb = self._empty(M, pc.nh * d, dtype=BF16)
        ops.gemm(inpb, a.W(q + 'key_linear.weight'), ipb, M, pc.nh * d, d)
        res = self._empty(M, e.out)
        ops.gemm(inpb, a.W(q + 'residual.weight'), res, M, e.out, d)
        dev = a.device
        sv = SimpleNamespace(unit=torch.empty(M, pc.nh, pc.topk, dtype=torch.int32, device=dev),
                             lr=torch.empty(M, pc.nh, pc.topk, 2, dtype=torch.int32, device=dev),
                             score=self._empty(M, pc.nh, pc.topk), dot=self._empty(M, pc.nh, pc.topk))
        out = self._empty(M, e.out)
        ops.peer_lookup_fwd(S, ipb, res, a.W(q + 'emb_in.weight'), a.W(q + 'emb_out.weight'), out, sv, M, pc.nh, pc.nq, pc.topk, d, e.out)
        self.peer_trace = sv                                   # tests read the routing decisions
        ctx = SimpleNamespace(featb=featb, inpb=inpb, xq=xq, ipb=ipb, sv=sv) if save else None
        return out, ctx

    def _vit_head_peer_bwd(self, ctx, denc, B: int):
        a, e, pc = self.arena, self.enc, self.enc.peer
        d, n, M, q = e.d, e.ncls, e.ncls * B, f'{self.ep}peer.'
        dS = torch.zeros(M * pc.nh, 2 * pc.nq, dtype=F32, device=a.device)
        dipb = self._empty(M, pc.nh * d, dtype=BF16)
        ops.peer_lookup_bwd(denc, ctx.ipb, a.W(q + 'emb_in.weight'), a.W(q + 'emb_out.weight'), ctx.sv, dS, dipb,
                            a.Gt(q + 'emb_in.weight'), a.Gt(q + 'emb_out.weight'), M, pc.nh, pc.nq, pc.topk, d, e.out)
        doutb = self._empty(M, e.out, dtype=BF16)
        ops.cast_f32_bf16(denc, doutb)
        dinp = self._empty(M, d)
        self._linear_bwd(doutb, M, e.out, d, ctx.inpb, q + 'residual.weight', None, dx_out=dinp)
        self._linear_bwd(dipb, M, pc.nh * d, d, ctx.inpb, q + 'key_linear.weight', None, dx_out=dinp, accumulate=True)
        dSb = self._empty(M * pc.nh, 2 * pc.nq, dtype=BF16)
        ops.cast_f32_bf16(dS, dSb)
        Wlr = a.span('W', [q + 'query_left.linear.weight', q + 'query_right.linear.weight'], (2 * pc.nq, pc.qd))
        Glr = a.span('G', [q + 'query_left.linear.weight', q + 'query_right.linear.weight'], (2 * pc.nq, pc.qd))
        xqv = ctx.xq.view(M * pc.nh, pc.qd)
        ops.gemm(dSb, xqv, Glr, 2 * pc.nq, pc.qd, M * pc.nh, a_kmajor=True, b_kmajor=True, accumulate=True)
        dxq = self._empty(M * pc.nh, pc.qd, dtype=BF16)
        ops.gemm(dSb, Wlr, dxq, M * pc.nh, pc.qd, 2 * pc.nq, b_kmajor=True)
        self._linear_bwd(dxq.view(M, pc.nh * pc.qd), M, pc.nh * pc.qd, d, ctx.inpb, q + 'query_linear.weight', None, dx_out=dinp, accumulate=True)
        dY = self._empty(B, d * n, dtype=BF16)
        ops.transpose_last2(dinp, None, dY, B, n, d)           # rows (b, s) of width e -> (b, e, s)
        ops.gemm(ctx.featb, dY, a.G(f'{self.ep}peer_proj_wt').view(d, d * n), d, d * n, B, a_kmajor=True, b_kmajor=True, accumulate=True)
        if not e.refine:
            return None
        dfeat = self._empty(B, d)
        ops.gemm(dY, a.W(f'{self.ep}peer_proj_wt').view(d, d * n), dfeat, B, d, d * n)
        return dfeat

    # ------------------------------------------------------------------------------------------------ LSH head
    def _lsh_tables(self):
        """Device-side description of the slots' CosineVectorEmbeddings: the concatenated projection buffers (they are persistent
        buffers of the state dict: rebuilt when a checkpoint replaced them), the bucket grids, and where each table sits in the arena."""
        a, e, lc = self.arena, self.enc, self.enc.lsh
        mods = [[cv for cv in comp.emb] for comp in e.module.lsh_emb]
        version = tuple((cv.projection_mat.data_ptr(), cv.projection_mat._version, cv.grid._version) for row in mods for cv in row) \
            + (a.p32.data_ptr(),)
        if lc.cache is not None and lc.cache.version == version:
            return lc.cache
        dev, nK = a.device, len(lc.bins)
        P = torch.cat([cv.projection_mat.to(device=dev, dtype=F32) for row in mods for cv in row], dim=1).contiguous()
        grids = [cv.grid.detach().to(dtype=F32).cpu() for cv in mods[0]]
        for row in mods[1:]:
            for g0, cv in zip(grids, row):
                if not torch.equal(g0, cv.grid.detach().float().cpu()):
                    raise NotImplementedError('LSH slots with different bucket grids are outside the HIP hot path')
        goff = np.concatenate(([0], np.cumsum([g.numel() for g in grids])[:-1])).astype(np.int32)
        name = lambda s, k: f'{self.ep}lsh_emb.{s}.emb.{k}.emb.weight'
        base = a.entries[name(0, 0)][0]
        toff = np.asarray([a.entries[name(0, k)][0] - base for k in range(nK)], dtype=np.int64)
        stride = a.entries[name(1, 0)][0] - base if e.ncls > 1 else 0
        for s in range(e.ncls):
            for k in range(nK):
                assert a.entries[name(s, k)][0] == base + s * stride + toff[k]
                assert tuple(a.entries[name(s, k)][2]) == ((lc.bins[k] + 1) * lc.n_proj, e.out)
        lc.cache = SimpleNamespace(version=version, P=P, grids=torch.cat(grids).to(dev), goff=torch.from_numpy(goff).to(dev),
                                   toff=torch.from_numpy(toff).to(dev), nbins=torch.tensor(lc.bins, dtype=torch.int32, device=dev),
                                   base=base, stride=stride, nK=nK)
        return lc.cache

    def _vit_head_lsh_fwd(self, feat, B: int, save: bool):
        a, e, lc = self.arena, self.enc, self.enc.lsh
        t = self._lsh_tables()
        xn = self._empty(B, e.d)
        ops.l2norm_fwd(feat, xn, None, None, B, e.d)
        ncol = e.ncls * t.nK * lc.n_proj
        z = self._empty(B, ncol)
        ops.gemm_f32(xn, t.P, z, B, ncol, e.d)
        out = self._empty(B * e.ncls, e.out)
        rows = torch.empty(B * e.ncls, t.nK, lc.n_proj, dtype=torch.int32, device=a.device)
        ops.lsh_embed_fwd(z, a.p32[t.base:], t.stride, t.toff, t.nbins, t.grids, t.goff, out, rows, B, e.ncls, t.nK, lc.n_proj, e.out)
        self.lsh_trace = SimpleNamespace(z=z, rows=rows)
        return out, (SimpleNamespace(rows=rows) if save else None)

    def _vit_head_lsh_bwd(self, ctx, denc, B: int):
        a, e, lc = self.arena, self.enc, self.enc.lsh
        t = self._lsh_tables()
        ops.lsh_embed_bwd(denc, ctx.rows, a.g32[t.base:], t.stride, t.toff, B, e.ncls, t.nK, lc.n_proj, e.out)
        return None                                            # bucketize passes no gradient (and LSH forces refine off)

    # ------------------------------------------------------------------------------------------------ encoder entry points
    def vit_encode(self, images: torch.Tensor, save: bool):
        a, e = self.arena, self.enc
        images = images.to(device=a.device, dtype=F32).contiguous()
        B = images.shape[0]
        feat, bctx = self.vit_backbone_fwd(images, save and e.refine)
        head_fwd = {'mlp': self._vit_head_mlp_fwd, 'peer': self._vit_head_peer_fwd, 'lsh': self._vit_head_lsh_fwd}[e.head]
        y, hctx = head_fwd(feat, B, save)
        yb = None
        if self.has_bridge:
            yb = self._empty(B * e.ncls, e.out, dtype=BF16)
            ops.cast_f32_bf16(y, yb)
            enc_out = self._empty(B * e.ncls, self.dec.d)
            ops.gemm(yb, a.W('encoder.1.weight'), enc_out, B * e.ncls, self.dec.d, e.out)
        else:
            enc_out = y
        ctx = SimpleNamespace(B=B, bctx=bctx, hctx=hctx, yb=yb) if save else None
        return enc_out.view(B, e.ncls, -1), ctx

    def vit_encode_backward(self, ctx, denc: torch.Tensor):
        """denc fp32 [B * n_cls, d_out]: gradient w.r.t. the encoder output (the bridge's when there is one)."""
        a, e, B = self.arena, self.enc, ctx.B
        M = B * e.ncls
        denc = denc.contiguous()
        if self.has_bridge:
            dencb = self._empty(M, self.dec.d, dtype=BF16)
            ops.cast_f32_bf16(denc, dencb)
            dy = self._empty(M, e.out)
            self._linear_bwd(dencb, M, self.dec.d, e.out, ctx.yb, 'encoder.1.weight', None, dx_out=dy)
            denc = dy
        head_bwd = {'mlp': self._vit_head_mlp_bwd, 'peer': self._vit_head_peer_bwd, 'lsh': self._vit_head_lsh_bwd}[e.head]
        dfeat = head_bwd(ctx.hctx, denc, B)
        if e.refine and dfeat is not None:
            self.vit_backbone_bwd(ctx.bctx, dfeat, B)
