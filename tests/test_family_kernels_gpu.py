"""Kernel-level parity of the nano-mini family kernels (csrc/attention_g.hip, csrc/family.hip) against fp64 torch references of the
same operations, through the C ABI.  bf16 operands: tolerances as in test_kernels_gpu.py."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from test_kernels_gpu import check, dev, rnd

pytestmark = pytest.mark.gpu
BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def ops():
    from image2text_amd import ops as _ops
    from image2text_amd.build import build_library
    build_library()
    return _ops


def ref_gq_attention(q, k, v, H, Hkv, hd, causal, mask=None, sc=1.0):
    """q [B, Tq, H hd], k / v [B, Tk, Hkv hd] (double, requires_grad) -> o [B, Tq, H hd], lse [B, H, Tq]"""
    B, Tq, Tk = q.shape[0], q.shape[1], k.shape[1]
    G = H // Hkv
    qh = q.view(B, Tq, H, hd).permute(0, 2, 1, 3)
    kh = k.view(B, Tk, Hkv, hd).permute(0, 2, 1, 3).repeat_interleave(G, dim=1)
    vh = v.view(B, Tk, Hkv, hd).permute(0, 2, 1, 3).repeat_interleave(G, dim=1)
    s = qh @ kh.transpose(-1, -2) / math.sqrt(hd)
    if causal:
        i, j = torch.arange(Tq, device=q.device)[:, None], torch.arange(Tk, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Tk - Tq), float('-inf'))
    p = torch.softmax(s, dim=-1)
    if mask is not None:
        p = p * mask * sc
    return (p @ vh).permute(0, 2, 1, 3).reshape(B, Tq, H * hd), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize('B,H,Hkv,hd,Tq,Tk,causal,drop', [
    (2, 8, 1, 128, 160, 160, False, False),       # nano-mini encoder block (sparse subset of 320)
    (3, 8, 1, 128, 96, 96, True, True),           # nano-mini decoder self-attention
    (2, 8, 8, 128, 70, 64, False, True),          # nano-mini cross-attention: multi-head, 128-wide heads
    (2, 4, 1, 16, 100, 100, False, False),        # the reference unit test's heads (16 wide)
    (2, 4, 2, 32, 37, 37, True, True),
    (1, 2, 1, 64, 200, 200, True, False),
    (2, 2, 1, 128, 1, 50, True, False),           # a decode-shaped call
    (2, 3, 3, 16, 33, 131, False, True),
    (2, 12, 2, 128, 300, 300, True, False),       # Qwen2-1.5B heads: 12 query heads of 128 on 2 K/V heads (engine_llama.py)
    (1, 8, 4, 64, 257, 257, True, False)])        # grouped 64-wide heads, a ragged last tile
def test_gq_attention_fwd_bwd(ops, B, H, Hkv, hd, Tq, Tk, causal, drop):
    from image2text_amd import rng
    dq_, dk_ = H * hd, Hkv * hd
    q = rnd(B, Tq, dq_, dtype=BF16, seed=401)
    kv = rnd(B, Tk, 2 * dk_, dtype=BF16, seed=402)
    k, v = kv[..., :dk_], kv[..., dk_:]
    dp, mask, sc = None, None, 1.0
    if drop:
        key, thr = rng.site_key(7, 31), rng.threshold(0.2)
        sc = rng.scale(thr)
        dp = (1, key, thr, sc)
        mask = rng.keep_mask(key, B * H * Tq * Tk, thr).view(B, H, Tq, Tk).to(dev())
    qr, kr, vr = (t.double().contiguous().requires_grad_(True) for t in (q, k, v))
    o_ref, lse_ref = ref_gq_attention(qr, kr, vr, H, Hkv, hd, causal, mask, sc)
    o, lse = torch.empty(B, Tq, dq_, dtype=BF16, device=dev()), torch.empty(B, H, Tq, device=dev())
    ops.gq_attention_fwd(q, k, v, o, lse, B, H, Hkv, hd, Tq, Tk, causal, drop=dp)
    check('gq out', o, o_ref, 1e-2, 1 / 128)
    check('gq lse', lse, lse_ref, 2e-3, 2e-3)
    do = rnd(B, Tq, dq_, dtype=BF16, seed=403)
    o_ref.backward(do.double())
    dq = torch.zeros(B, Tq, dq_, dtype=BF16, device=dev())
    dkv = torch.zeros(B, Tk, 2 * dk_, dtype=BF16, device=dev())
    ops.gq_attention_bwd(q, k, v, o, do, lse, torch.empty(B, H, Tq, device=dev()), dq, dkv[..., :dk_], dkv[..., dk_:], B, H, Hkv, hd, Tq,
                         Tk, causal, drop=dp)
    s = float(do.float().abs().max()) * math.sqrt(H // Hkv)
    for name, got, ref in (('dq', dq, qr.grad), ('dk', dkv[..., :dk_], kr.grad), ('dv', dkv[..., dk_:], vr.grad)):
        check(f'gq {name}', got, ref, 3e-2 * s, 1 / 32)


def test_gq_attention_matches_head64_kernels(ops):
    """hd = 64, Hkv = H: the grouped kernels and the tuned 64-wide ones are two implementations of the same function"""
    B, H, T = 2, 3, 130
    d = 64 * H
    qkv = rnd(B, T, 3 * d, dtype=BF16, seed=410)
    q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    o0, l0 = torch.empty(B, T, d, dtype=BF16, device=dev()), torch.empty(B, H, T, device=dev())
    o1, l1 = torch.empty_like(o0), torch.empty_like(l0)
    ops.attention_fwd(q, k, v, o0, l0, B, H, T, T, True)
    ops.gq_attention_fwd(q, k, v, o1, l1, B, H, H, 64, T, T, True)
    assert torch.equal(o0, o1)
    check('lse', l1, l0, 1e-6, 1e-6)


def test_gq_attention_packed_rows_and_token_multipliers(ops):
    """packed variable-length multi-query self-attention == per-sequence dense calls; out_drop multipliers on dq / dk / dv"""
    from image2text_amd import rng
    H, hd = 4, 128
    lens = [5, 64, 0, 37, 70]
    B, total, Tmax = len(lens), sum(lens), 70
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    q = rnd(total, H * hd, dtype=BF16, seed=420)
    kv = rnd(total, 2 * hd, dtype=BF16, seed=421)
    do = rnd(total, H * hd, dtype=BF16, seed=422)
    o, lse = torch.zeros(total, H * hd, dtype=BF16, device=dev()), torch.zeros(H * total, device=dev())
    ops.gq_attention_fwd(q, kv[:, :hd], kv[:, hd:], o, lse, B, H, 1, hd, Tmax, Tmax, True, cu_q=cu, cu_k=cu, total_q=total)
    key, thr = rng.site_key(5, 9), rng.threshold(0.25)
    sc = rng.scale(thr)
    outs = []
    for od in (None, (2, key, thr, sc)):
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        ops.gq_attention_bwd(q, kv[:, :hd], kv[:, hd:], o, do, lse, torch.empty(H * total, device=dev()), dq, dkv[:, :hd], dkv[:, hd:],
                             B, H, 1, hd, Tmax, Tmax, True, cu_q=cu, cu_k=cu, total_q=total, out_drop=od)
        outs.append((dq, dkv))
    (dq0, dkv0), (dq1, dkv1) = outs
    mult = torch.stack([rng.keep_mask((key + t) & 0xFFFFFFFF, total, thr) for t in range(3)], 1).float().to(dev()) * sc
    check('dq multipliers', dq1, dq0.float() * mult[:, 0:1], 1e-6, 1 / 128)
    check('dk multipliers', dkv1[:, :hd], dkv0[:, :hd].float() * mult[:, 1:2], 1e-6, 1 / 128)
    check('dv multipliers', dkv1[:, hd:], dkv0[:, hd:].float() * mult[:, 2:3], 1e-6, 1 / 128)
    for b, n in enumerate(lens):
        if n == 0:
            continue
        s0 = int(cu[b])
        qs, kvs, dos = (t[s0:s0 + n].unsqueeze(0).contiguous() for t in (q, kv, do))
        o1, l1 = torch.empty(1, n, H * hd, dtype=BF16, device=dev()), torch.empty(1, H, n, device=dev())
        ops.gq_attention_fwd(qs, kvs[..., :hd], kvs[..., hd:], o1, l1, 1, H, 1, hd, n, n, True)
        assert torch.equal(o1[0], o[s0:s0 + n]), f'fwd seq {b}'
        dq1_, dkv1_ = torch.zeros_like(qs), torch.zeros_like(kvs)
        ops.gq_attention_bwd(qs, kvs[..., :hd], kvs[..., hd:], o1, dos, l1, torch.empty(1, H, n, device=dev()), dq1_, dkv1_[..., :hd],
                             dkv1_[..., hd:], 1, H, 1, hd, n, n, True)
        assert torch.equal(dq1_[0], dq0[s0:s0 + n]) and torch.equal(dkv1_[0], dkv0[s0:s0 + n]), f'bwd seq {b}'


def test_row_sections_dropout(ops):
    from image2text_amd import rng
    M, d, hd = 77, 256, 32
    key, thr = rng.site_key(3, 4), rng.threshold(0.3)
    sc = rng.scale(thr)
    q, kv = rnd(M, d, dtype=BF16, seed=430), rnd(M, 2 * hd, dtype=BF16, seed=431)
    q0, kv0 = q.clone(), kv.clone()
    ops.row_sections_dropout(q, M, d, d, (2, key, thr, sc), 0)
    ops.row_sections_dropout(kv, M, 2 * hd, hd, (2, key, thr, sc), 1)
    m = [rng.keep_mask((key + t) & 0xFFFFFFFF, M, thr).float().to(dev()) * sc for t in range(3)]
    check('q', q, q0.float() * m[0][:, None], 1e-6, 1 / 128)
    check('k', kv[:, :hd], kv0[:, :hd].float() * m[1][:, None], 1e-6, 1 / 128)
    check('v', kv[:, hd:], kv0[:, hd:].float() * m[2][:, None], 1e-6, 1 / 128)


def test_gather_scatter_rows(ops):
    n_src, d = 500, 256
    src = rnd(n_src, d, seed=440)
    perm = torch.randperm(n_src, generator=torch.Generator().manual_seed(1))
    a, b = perm[:180].sort().values.to(torch.int32).to(dev()), perm[180:].sort().values.to(torch.int32).to(dev())
    of, ob = torch.empty(180, d, device=dev()), torch.empty(180, d, dtype=BF16, device=dev())
    ops.gather_rows(src, a, 180, d, out_f32=of, out_bf16=ob)
    assert torch.equal(of, src[a.long()]) and torch.equal(ob, src[a.long()].to(BF16))
    ob2 = torch.empty(320, d, dtype=BF16, device=dev())
    ops.gather_rows(src, b, 320, d, out_bf16=ob2)
    assert torch.equal(ob2, src[b.long()].to(BF16))
    dst = torch.full((n_src, d), float('nan'), device=dev())
    ops.scatter_rows(of, a, dst, 180, d)
    ops.scatter_rows(src[b.long()].contiguous(), b, dst, 320, d)
    assert torch.equal(dst, src)


def ref_moe_linear(x, W1, b1, Wg, bg, wg2, bg2, W2, b2, top_k):
    """x [M, in] double; W1 [E, P, in], b1 [E, P], gate first layer Wg [G | E, in] (+bg), optional second layer wg2 [E, G], bg2 [E];
    W2 [E, out, P], b2 [E, out] -> y [M, out], (gates, idx)"""
    E = W1.shape[0]
    g = x @ Wg.t() + (bg if bg is not None else 0)
    if wg2 is not None:
        g = F.gelu(g, approximate='tanh') @ wg2.t() + (bg2 if bg2 is not None else 0)
    gates = (g / math.sqrt(x.shape[1])).softmax(-1)
    w, idx = torch.topk(gates, top_k, dim=-1)
    outs = torch.stack([F.gelu(x @ W1[e].t() + b1[e], approximate='tanh') @ W2[e].t() + b2[e] for e in range(E)], 1)
    sel = outs.gather(1, idx[..., None].expand(-1, -1, outs.shape[-1]))
    return (sel * w[..., None]).sum(1), gates, idx


@pytest.mark.parametrize('M,inf,outf,E,P,G,top_k', [(300, 256, 512, 4, 16, 32, 2), (1000, 1024, 2048, 4, 16, 32, 1), (257, 64, 160, 4, 8, 0, 2),
                                                    (64, 128, 128, 8, 8, 16, 3)])
def test_moe_linear_two_gemm_form(ops, M, inf, outf, E, P, G, top_k):
    """MoELinear as GEMM -> i2t_moe_gate_fwd -> GEMM and its backward (GEMM, i2t_moe_gate_bwd, GEMMs) against the reference
    formulation (gather / scatter over the chosen experts) in fp64 on the same bf16-rounded operands.  Gates are sharpened so that
    the top-k choice is decisive (near-ties are a separate, model-level question)."""
    EP, NG = E * P, (G if G else E)
    N1 = EP + NG
    Kp = -(-(EP + E) // 64) * 64
    x = rnd(M, inf, dtype=BF16, seed=450)
    W1 = rnd(E, P, inf, scale=inf ** -0.5, dtype=BF16, seed=451)
    b1 = rnd(E, P, scale=0.1, seed=452)
    Wg = rnd(NG, inf, scale=(4.0 if G == 0 else 1.0) * (inf ** -0.5) * (inf ** 0.5 if G == 0 else 1.0), dtype=BF16, seed=453)
    bg = rnd(NG, scale=0.1, seed=454)
    wg2 = rnd(E, G, scale=4.0 * inf ** 0.5 * G ** -0.5, seed=455) if G else None
    bg2 = rnd(E, scale=0.1, seed=456) if G else None
    W2 = rnd(E, outf, P, scale=P ** -0.5, dtype=BF16, seed=457)
    b2 = rnd(E, outf, scale=0.1, seed=458)
    # ---- device: forward
    Wcat = torch.cat((W1.view(EP, inf), Wg), 0).contiguous()
    bcat = torch.cat((b1.view(EP), bg), 0).contiguous()
    U = torch.empty(M, N1, device=dev())
    ops.gemm(x, Wcat, U, M, N1, inf, bias=bcat)
    A = torch.empty(M, Kp, dtype=BF16, device=dev())
    gates, wsel = torch.empty(M, E, device=dev()), torch.empty(M, E, device=dev())
    ops.moe_gate_fwd(U, wg2, bg2, A, gates, wsel, M, E, P, G, top_k, inf ** -0.5)
    W2aug = torch.empty(outf, Kp, dtype=BF16, device=dev())
    ops.moe_pack_w2(W2, b2, W2aug, outf, E, P)
    y = torch.empty(M, outf, device=dev())
    ops.gemm(A, W2aug, y, M, outf, Kp)
    # ---- reference (fp64 on the same operands; the bf16 roundings of b2 / A are the device's)
    leaves = [t.double().requires_grad_(True) for t in (x, W1, b1, Wg, bg, W2, b2)]
    xr, W1r, b1r, Wgr, bgr, W2r, b2r = leaves
    wg2r = wg2.double().requires_grad_(True) if G else None
    bg2r = bg2.double().requires_grad_(True) if G else None
    y_ref, g_ref, idx_ref = ref_moe_linear(xr, W1r, b1r, Wgr, bgr, wg2r, bg2r, W2r, b2r, top_k)
    chosen = (wsel > 0)
    want = torch.zeros_like(chosen).scatter_(1, idx_ref, True)
    srt = g_ref.detach().sort(dim=1, descending=True).values
    margin = srt[:, top_k - 1] - srt[:, top_k] if top_k < E else torch.ones(M, device=dev())
    flip = (chosen != want).any(1)
    assert not bool((flip & (margin > 2e-3)).any()), 'expert choice differs away from a tie'
    ok = ~flip
    assert int(ok.sum()) >= 0.97 * M
    check('gates', gates, g_ref, 2e-3, 1e-2)
    check('moe y', y[ok], y_ref[ok], 2e-2, 2e-2)
    # ---- device: backward
    dy = rnd(M, outf, dtype=BF16, seed=459)
    dy = dy * ok[:, None].to(BF16)                         # rows with a flipped choice carry no gradient in the comparison
    (y_ref * dy.double()).sum().backward()
    dA = torch.empty(M, Kp, dtype=BF16, device=dev())
    ops.gemm(dy, W2aug, dA, M, Kp, outf, b_kmajor=True)
    dW2aug = torch.zeros(outf, Kp, device=dev())
    ops.gemm(dy, A, dW2aug, outf, Kp, M, a_kmajor=True, b_kmajor=True, accumulate=True)
    gw2, gb2 = torch.zeros(E, outf, P, device=dev()), torch.zeros(E, outf, device=dev())
    ops.moe_unpack_dw2(dW2aug, gw2, gb2, outf, E, P)
    N1p = -(-N1 // 8) * 8
    D1 = torch.full((M, N1p), float('nan'), dtype=BF16, device=dev())
    dwg2 = torch.zeros(E, G, device=dev()) if G else None
    dbg2 = torch.zeros(E, device=dev()) if G else None
    part = torch.empty(ops.moe_gate_bwd_blocks(M), E * G + E, device=dev()) if G else None
    ops.moe_gate_bwd(dA, U, gates, wsel, wg2, D1, dwg2, dbg2, part, M, E, P, G, top_k, inf ** -0.5)
    assert torch.isfinite(D1.float()).all()
    dWcat = torch.zeros(N1, inf, device=dev())
    ops.gemm(D1, x, dWcat, N1, inf, M, a_kmajor=True, b_kmajor=True, accumulate=True, lda=N1p)
    dbcat = torch.zeros(N1, device=dev())
    ops.colsum(D1, dbcat, M, N1, accumulate=True)
    dx = torch.empty(M, inf, device=dev())
    ops.gemm(D1, Wcat, dx, M, inf, N1, b_kmajor=True, lda=N1p)
    sc = float(dy.float().abs().max())
    check('dW2', gw2, W2r.grad, 3e-2 * sc * M ** 0.5 * 0.2, 3e-2)
    check('db2', gb2, b2r.grad, 3e-2 * sc * M ** 0.5 * 0.2, 3e-2)
    check('dW1', dWcat[:EP].view(E, P, inf), W1r.grad, 3e-2 * float(W1r.grad.abs().max()), 3e-2)
    check('db1', dbcat[:EP].view(E, P), b1r.grad, 3e-2 * float(b1r.grad.abs().max()), 3e-2)
    check('dWg', dWcat[EP:], Wgr.grad, 4e-2 * float(Wgr.grad.abs().max()), 4e-2)
    check('dbg', dbcat[EP:], bgr.grad, 4e-2 * float(bgr.grad.abs().max()), 4e-2)
    if G:
        check('dwg2', dwg2, wg2r.grad, 4e-2 * float(wg2r.grad.abs().max()), 4e-2)
        check('dbg2', dbg2, bg2r.grad, 4e-2 * float(bg2r.grad.abs().max()), 4e-2)
    check('dx', dx, xr.grad, 4e-2 * float(xr.grad.abs().max()), 4e-2)


@pytest.mark.parametrize('G,N,K,lens', [(5, 32, 256, [7, 0, 64, 130, 1]), (3, 256, 32, [65, 64, 3]), (4, 128, 64, [10, 20, 30, 40]),
                                        (2, 1024, 32, [300, 5])])
def test_grouped_gemm_three_modes(ops, G, N, K, lens):
    """i2t_grouped_gemm (csrc/grouped.hip): forward with bias / GELU / residual, dX with the GELU derivative, dW accumulation and the
    grouped bias gradient against per-group fp64 products"""
    group0, n_pos = 2, G + 3
    seg_h = np.concatenate(([0], np.cumsum(lens))).astype(np.int32)
    M = int(seg_h[-1])
    seg = torch.from_numpy(seg_h).to(dev())
    stride = N * K + 24                                   # weights of consecutive positions a constant stride apart (with a gap)
    Wall = rnd(n_pos * stride, scale=K ** -0.5, dtype=BF16, seed=470)
    ball = rnd(n_pos * stride, scale=0.3, seed=471)
    X = rnd(M, K, dtype=BF16, seed=472)
    res = rnd(M, N, seed=473)
    Wg = lambda g: Wall[(g + group0) * stride:(g + group0) * stride + N * K].view(N, K)
    bg = lambda g: ball[(g + group0) * stride:(g + group0) * stride + N]
    W0, b0 = Wall[:N * K].view(N, K), ball
    # mode 0: gelu + pre-activation
    Y, pre = torch.full((M, N), float('nan'), dtype=BF16, device=dev()), torch.full((M, N), float('nan'), dtype=BF16, device=dev())
    ops.grouped_gemm(0, X, W0, Y, N, K, b_group_stride=stride, bias=b0, bias_group_stride=stride, act=1, aux_out=pre, seg=seg, n_groups=G,
                     max_rows=max(lens), group0=group0)
    Yf = torch.full((M, N), float('nan'), device=dev())
    ops.grouped_gemm(0, X, W0, Yf, N, K, b_group_stride=stride, bias=b0, bias_group_stride=stride, residual=res, seg=seg, n_groups=G,
                     max_rows=max(lens), group0=group0)
    dY = rnd(M, N, dtype=BF16, seed=474)
    dX = torch.full((M, K), float('nan'), dtype=BF16, device=dev())
    aux = rnd(M, K, dtype=BF16, seed=475)
    ops.grouped_gemm(1, dY, W0, dX, N, K, b_group_stride=stride, act=2, aux_in=aux, seg=seg, n_groups=G, max_rows=max(lens), group0=group0)
    dW = rnd(n_pos * stride, seed=476)
    dW_before = dW.clone()
    ops.grouped_gemm(2, dY, X, dW[:N * K].view(N, K), N, K, c_group_stride=stride, accumulate=True, seg=seg, n_groups=G, max_rows=max(lens),
                     group0=group0)
    db = torch.zeros(n_pos * stride, device=dev())
    ops.grouped_colsum(dY, seg, G, db, stride, group0, N)
    from test_kernels_gpu import gelu_grad
    for g in range(G):
        lo, hi = int(seg_h[g]), int(seg_h[g + 1])
        if hi == lo:
            continue
        z = X[lo:hi].double() @ Wg(g).double().t() + bg(g).double()
        check(f'pre g{g}', pre[lo:hi], z, 2e-2, 1e-2)
        check(f'gelu g{g}', Y[lo:hi], F.gelu(z, approximate='tanh'), 2e-2, 1e-2)
        check(f'f32+res g{g}', Yf[lo:hi], z + res[lo:hi].double(), 1e-3, 1e-3)
        want_dx = (dY[lo:hi].double() @ Wg(g).double()) * gelu_grad(aux[lo:hi].double())
        check(f'dx g{g}', dX[lo:hi], want_dx, 3e-2, 2e-2)
        o = (g + group0) * stride
        check(f'dW g{g}', dW[o:o + N * K].view(N, K), dW_before[o:o + N * K].view(N, K).double() + dY[lo:hi].double().t() @ X[lo:hi].double(),
              2e-3 * (hi - lo) ** 0.5, 2e-3)
        check(f'db g{g}', db[o:o + N], dY[lo:hi].double().sum(0), 1e-3, 1e-3)
    # untouched: other positions' weights gradients and the gaps
    o0, o1 = group0 * stride, (group0 + G) * stride
    assert torch.equal(dW[:o0], dW_before[:o0]) and torch.equal(dW[o1:], dW_before[o1:])
    # device-side group index (the decode step): every row uses group *ptr + group0
    ptr = torch.tensor([1], dtype=torch.int32, device=dev())
    Yd = torch.empty(9, N, dtype=BF16, device=dev())
    ops.grouped_gemm(0, X[:9].contiguous(), W0, Yd, N, K, b_group_stride=stride, bias=b0, bias_group_stride=stride, act=1, n_groups=1, max_rows=9,
                     group_ptr=ptr, group0=group0)
    check('decode group', Yd, F.gelu(X[:9].double() @ Wg(1).double().t() + bg(1).double(), approximate='tanh'), 2e-2, 1e-2)
