"""Per-kernel numerics on the MI355X: every C-ABI entry point against a plain PyTorch fp32/fp64 statement of the same
op, computed from the SAME bf16-rounded inputs (so the tolerance only has to cover fp32 accumulation order and the
final bf16 store: rtol 2^-7 for bf16 outputs, 2e-3 for f32 outputs)."""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

BF16, F32 = torch.bfloat16, torch.float32


@pytest.fixture(scope='module')
def ops():
    from image2text_amd import ops as _ops
    from image2text_amd.build import build_library
    build_library()
    return _ops


def dev():
    return torch.device('cuda:0')


def rnd(*shape, scale=1.0, seed=0, dtype=F32):
    g = torch.Generator().manual_seed(seed + sum(shape))
    return (torch.randn(*shape, generator=g) * scale).to(dtype).to(dev())


def check(name, got, ref, atol, rtol):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, f'{name}: shape {tuple(got.shape)} vs {tuple(ref.shape)}'
    assert torch.isfinite(got).all(), f'{name}: non-finite output'
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = err > tol
    if bad.any():
        idx = np.unravel_index(int((err - tol).argmax()), tuple(got.shape))
        raise AssertionError(f'{name}: {int(bad.sum())}/{bad.numel()} out of tolerance; worst at {idx}: got {got[idx].item():.6g} '
                             f'ref {ref[idx].item():.6g} (max abs err {err.max().item():.3g}, ref absmax {ref.abs().max().item():.3g})')


def gelu_grad(x):
    x = x.detach().clone().requires_grad_(True)
    F.gelu(x, approximate='tanh').sum().backward()
    return x.grad


# ------------------------------------------------------------------------------------------------------ GEMM
@pytest.mark.parametrize('M,N,K', [(128, 128, 64), (200, 136, 192), (1024, 768, 512), (64, 2304, 768), (300, 72, 8192)])
@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_layouts(ops, M, N, K, a_km, b_km):
    a = rnd(M, K, dtype=BF16, seed=1)
    b = rnd(N, K, dtype=BF16, seed=2)
    ref = a.float() @ b.float().t()
    a_in = a.t().contiguous() if a_km else a
    b_in = b.t().contiguous() if b_km else b
    if (a_km and M % 8) or (b_km and N % 8):
        pytest.skip('k-major operand needs a leading dimension padded to 8')
    out = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a_in, b_in, out, M, N, K, a_kmajor=a_km, b_kmajor=b_km)
    check(f'gemm f32 {M}x{N}x{K} a_km={a_km} b_km={b_km}', out, ref, 2e-3 * math.sqrt(K) / 8, 2e-3)
    outb = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a_in, b_in, outb, M, N, K, a_kmajor=a_km, b_kmajor=b_km)
    check('gemm bf16 out', outb, ref, 2e-2 * math.sqrt(K) / 8, 1 / 128)


def test_gemm_epilogues(ops):
    M, N, K = 260, 512, 256
    a, b = rnd(M, K, dtype=BF16, seed=3), rnd(N, K, dtype=BF16, seed=4, scale=0.1)
    bias, res = rnd(N, seed=5), rnd(M, N, seed=6)
    base = a.float() @ b.float().t()
    # bias + gelu with pre-activation side output, bf16 result
    out = torch.empty(M, N, dtype=BF16, device=dev())
    pre = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, out, M, N, K, bias=bias, act=1, aux_out=pre)
    check('pre-activation', pre, base + bias, 1e-2, 1 / 128)
    check('gelu(bias + ab)', out, F.gelu(base + bias, approximate='tanh'), 1e-2, 1 / 128)
    # residual + bias, f32 result, alpha
    o32 = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a, b, o32, M, N, K, alpha=0.5, bias=bias, residual=res)
    check('alpha/bias/residual', o32, 0.5 * base + bias + res, 2e-3, 2e-3)
    # accumulate into f32
    acc = res.clone()
    ops.gemm(a, b, acc, M, N, K, accumulate=True)
    check('accumulate', acc, base + res, 2e-3, 2e-3)
    # dgelu: out = (a b^T) * gelu'(aux)
    aux = rnd(M, N, dtype=BF16, seed=7)
    od = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, od, M, N, K, act=2, aux_in=aux)
    check('dgelu', od, base * gelu_grad(aux.float()), 2e-2, 1 / 128)


def test_gemm_vocab_shapes(ops):
    """lm_head shapes: N = 50257 (odd) with ldc padded to 50264; K = 50257 reduction with zero pads; dW over M."""
    V, Vp, d, M = 50257, 50264, 768, 192
    h = rnd(M, d, dtype=BF16, seed=8)
    wte = rnd(V, d, dtype=BF16, seed=9, scale=0.05)
    logits = torch.zeros(M, Vp, dtype=BF16, device=dev())
    ops.gemm(h, wte, logits, M, V, d)
    ref = h.float() @ wte.float().t()
    check('logits', logits[:, :V], ref, 2e-2, 1 / 128)
    assert float(logits[:, V:].abs().max()) == 0.0, 'pad columns must stay untouched'
    logits32 = torch.zeros(M, Vp, dtype=F32, device=dev())          # the decode path keeps fp32 logits for the argmax
    ops.gemm(h, wte, logits32, M, V, d)
    check('logits f32', logits32[:, :V], ref, 2e-3, 2e-3)
    assert float(logits32[:, V:].abs().max()) == 0.0, 'pad columns must stay untouched (f32)'
    dl = torch.zeros(M, Vp, dtype=BF16, device=dev())
    dl[:, :V] = rnd(M, V, dtype=BF16, seed=10, scale=0.01)
    dh = torch.empty(M, d, dtype=F32, device=dev())
    ops.gemm(dl, wte, dh, M, d, V, b_kmajor=True)                       # dH = dL . W       (K = V, pads are zero)
    check('dH', dh, dl[:, :V].float() @ wte.float(), 5e-3, 3e-3)
    dw = torch.empty(V, d, dtype=F32, device=dev())
    ops.gemm(dl, h, dw, V, d, M, a_kmajor=True, b_kmajor=True)          # dW = dL^T . H     (reduction over M)
    check('dW', dw, dl[:, :V].float().t() @ h.float(), 2e-3, 3e-3)


@pytest.mark.parametrize('M,N,K', [(2560, 2560, 512), (2500, 2440, 768), (3000, 2304, 128), (2304, 4104, 1536)])
@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 0), (1, 1)])
def test_gemm_large_tile_layouts(ops, M, N, K, a_km, b_km):
    """Shapes with >= 96 tiles of 256 x 256: the 8-wave large-tile kernel (ragged M / N edges, K = 2..24 K-tiles)."""
    if (a_km and M % 8) or (b_km and N % 8):
        pytest.skip('k-major operand needs a leading dimension padded to 8')
    a = rnd(M, K, dtype=BF16, seed=21)
    b = rnd(N, K, dtype=BF16, seed=22)
    ref = a.float() @ b.float().t()
    a_in = a.t().contiguous() if a_km else a
    b_in = b.t().contiguous() if b_km else b
    out = torch.full((M, N), float('nan'), dtype=F32, device=dev())
    ops.gemm(a_in, b_in, out, M, N, K, a_kmajor=a_km, b_kmajor=b_km)
    check(f'large gemm f32 {M}x{N}x{K} a_km={a_km} b_km={b_km}', out, ref, 2e-3 * math.sqrt(K) / 8, 2e-3)
    for _ in range(3):          # the DMA pipeline has no data-dependent path: repeated launches must agree bit for bit
        again = torch.empty(M, N, dtype=F32, device=dev())
        ops.gemm(a_in, b_in, again, M, N, K, a_kmajor=a_km, b_kmajor=b_km)
        assert torch.equal(again, out)


def test_gemm_dw_reduction_longer_than_a_32bit_panel(ops):
    """dW = dY^T . X whose k-major panel exceeds the 4 GiB a buffer descriptor can address ((K + 512) rows x ld x 2 bytes:
    the tied lm_head at B = 2048 reads 74 k rows of 50 264 logits): the call runs as K chunks accumulating into C."""
    M, N, K, lda = 512, 768, 43037, 50264                     # (K + 512) * lda * 2 = 4.38 GB
    assert (K + 512) * lda * 2 > 2 ** 32
    dy = torch.empty(K, lda, dtype=BF16, device=dev())        # only the first M columns take part
    dy[:, :M] = rnd(K, M, dtype=BF16, seed=41, scale=0.05)
    x = rnd(K, N, dtype=BF16, seed=42, scale=0.05)
    c0 = rnd(M, N, seed=43)
    out = c0.clone()
    ops.gemm(dy, x, out, M, N, K, a_kmajor=True, b_kmajor=True, accumulate=True)
    ref = c0 + dy[:, :M].float().t() @ x.float()
    check('chunked dW', out, ref, 2e-3 * math.sqrt(K) / 8, 3e-3)


@pytest.mark.parametrize('K', [1000, 1050, 50257])
def test_gemm_large_tile_ragged_k_with_kmajor_b(ops, K, monkeypatch):
    """dX = dY . W with a reduction length that is no multiple of 64 (the lm_head's 50257): the large-tile kernel relies
    on the k-major B panel reading as zeros past K; A's pad columns are zero, A's next row is not."""
    monkeypatch.setenv('I2T_G256_MIN_TILES', '1')
    M, N = 600, 768
    ld = (K + 7) // 8 * 8
    a = torch.zeros(M, ld, dtype=BF16, device=dev())
    a[:, :K] = rnd(M, K, dtype=BF16, seed=31, scale=0.1)
    w = rnd(K, N, dtype=BF16, seed=32, scale=0.1)
    out = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a, w, out, M, N, K, b_kmajor=True)
    check(f'ragged-K dX K={K}', out, a[:, :K].float() @ w.float(), 2e-3 * math.sqrt(K) / 8, 3e-3)


def test_gemm_large_tile_random_shapes(ops, monkeypatch):
    """Seeded random problems forced through the persistent large-tile kernels (every size, not only the ones the
    heuristic would send there): ragged M / N down to a single partial tile, K from 2 K-tiles up, forward (B^T), dX (B) and
    dW (A^T.B, K slices + atomics) forms, with the epilogue options each form meets in the model."""
    monkeypatch.setenv('I2T_G256_MIN_TILES', '1')
    rs = np.random.RandomState(1234)
    for case in range(36):
        form = ('fwd', 'dx', 'dw')[case % 3]
        M = int(rs.choice([8, 100, 256, 257, 700, 1500]))
        N = int(rs.choice([8, 64, 256, 264, 520, 1000]))
        if form == 'dw':
            M, N = max(256, M // 8 * 8), max(256, N // 8 * 8)          # dW route: M, N >= 256, multiples of 8 (k-major leading dims)
            K = int(rs.choice([1024, 1111 * 8, 4096 + 8, 20000]))
        elif form == 'dx':
            N = N // 8 * 8 if N >= 8 else 8
            K = int(rs.choice([128, 256, 1000, 3072]))
        else:
            K = int(rs.choice([128, 256, 768, 2048]))
        a = rnd(M, K, dtype=BF16, seed=1000 + case, scale=0.5)
        b = rnd(N, K, dtype=BF16, seed=2000 + case, scale=0.5)
        ref = a.float() @ b.float().t()
        tag = f'case {case} {form} M={M} N={N} K={K}'
        tol = 2e-3 * math.sqrt(K) / 8
        if form == 'fwd':
            bias = rnd(N, seed=3000 + case)
            if N % 4 == 0:
                res = rnd(M, N, seed=4000 + case)
                out = torch.empty(M, N, dtype=F32, device=dev())
                ops.gemm(a, b, out, M, N, K, bias=bias, residual=res)
                check(tag + ' f32+bias+res', out, ref + bias + res, tol, 2e-3)
            outb = torch.empty(M, (N + 7) // 8 * 8, dtype=BF16, device=dev())
            ops.gemm(a, b, outb, M, N, K, bias=bias, act=1)
            check(tag + ' gelu', outb[:, :N], F.gelu(ref + bias, approximate='tanh'), 10 * tol + 1e-2, 1 / 128)
        elif form == 'dx':
            bt = b.t().contiguous()                                   # [K][N], N contiguous
            out = rnd(M, N, seed=5000 + case)
            base = out.clone()
            ops.gemm(a, bt, out, M, N, K, b_kmajor=True, accumulate=True)
            check(tag + ' f32 accumulate', out, base + ref, tol, 2e-3)
            outb = torch.empty(M, N, dtype=BF16, device=dev())
            ops.gemm(a, bt, outb, M, N, K, b_kmajor=True)
            check(tag + ' bf16', outb, ref, 10 * tol + 1e-2, 1 / 128)
        else:
            at, bt = a.t().contiguous(), b.t().contiguous()           # [K][M], [K][N]
            out = rnd(M, N, seed=6000 + case)
            base = out.clone()
            ops.gemm(at, bt, out, M, N, K, a_kmajor=True, b_kmajor=True, accumulate=True)
            check(tag + ' dW accumulate', out, base + ref, 2 * tol, 3e-3)


def test_gemm_reserved_cus(ops):
    """i2t_gemm_reserve_cus: the persistent kernels run on fewer workgroups (each walks more tiles) with the same result."""
    M, N, K = 4000, 2304, 512
    a, b = rnd(M, K, dtype=BF16, seed=41), rnd(N, K, dtype=BF16, seed=42, scale=0.1)
    full = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, full, M, N, K)
    try:
        for r in (16, 100, 1000):          # 1000 > #CUs: ignored
            ops.gemm_reserve_cus(r)
            out = torch.empty(M, N, dtype=BF16, device=dev())
            ops.gemm(a, b, out, M, N, K)
            assert torch.equal(out, full), r
    finally:
        ops.gemm_reserve_cus(0)


def test_gemm_large_tile_epilogues(ops):
    from image2text_amd import rng
    M, N, K = 2900, 2304, 512
    a, b = rnd(M, K, dtype=BF16, seed=23), rnd(N, K, dtype=BF16, seed=24, scale=0.1)
    bias, res = rnd(N, seed=25), rnd(M, N, seed=26)
    base = a.float() @ b.float().t()
    out = torch.empty(M, N, dtype=BF16, device=dev())
    pre = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, out, M, N, K, bias=bias, act=1, aux_out=pre)
    check('pre-activation', pre, base + bias, 1e-2, 1 / 128)
    check('gelu(bias + ab)', out, F.gelu(base + bias, approximate='tanh'), 1e-2, 1 / 128)
    key, thr = rng.site_key(5, 9), rng.threshold(0.1)
    sc = rng.scale(thr)
    o32 = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a, b, o32, M, N, K, alpha=0.5, bias=bias, residual=res, drop=(1, key, thr, sc))
    m = rng.keep_mask(key, M * N, thr).view(M, N).to(dev())
    check('alpha/bias/dropout/residual', o32, res + (0.5 * base + bias) * m * sc, 2e-3, 2e-3)
    acc = res.clone()
    ops.gemm(a, b, acc, M, N, K, accumulate=True)
    check('accumulate', acc, base + res, 2e-3, 2e-3)
    aux = rnd(M, N, dtype=BF16, seed=27)
    od = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, od, M, N, K, act=2, aux_in=aux)
    check('dgelu', od, base * gelu_grad(aux.float()), 2e-2, 1 / 128)
    # classes that exist only for the B^T (forward) form must fall back to the generic epilogue with a k-major B
    bt = b.t().contiguous()
    o32k = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a, bt, o32k, M, N, K, b_kmajor=True, bias=bias, residual=res)
    check('k-major B: bias/residual (generic class)', o32k, base + bias + res, 2e-3, 2e-3)
    ogk = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, bt, ogk, M, N, K, b_kmajor=True, bias=bias, act=1)
    check('k-major B: gelu (generic class)', ogk, F.gelu(base + bias, approximate='tanh'), 1e-2, 1 / 128)
    # forward form without bias / pre-activation output / residual (run-time options of classes 2, 3, 5)
    og = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, og, M, N, K, act=1)
    check('gelu without bias or aux', og, F.gelu(base, approximate='tanh'), 1e-2, 1 / 128)
    o32p = torch.empty(M, N, dtype=F32, device=dev())
    ops.gemm(a, b, o32p, M, N, K, bias=bias)
    check('f32 + bias, no residual', o32p, base + bias, 2e-3, 2e-3)
    ops.gemm(a, b, o32p, M, N, K)
    check('f32 plain', o32p, base, 2e-3, 2e-3)
    outq = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, outq, M, N, K, bias=bias, drop=(2, key, thr, sc))
    mult = torch.stack([rng.keep_mask((key + t) & 0xFFFFFFFF, M, thr) for t in range(3)], 1).float().to(dev()) * sc
    check('qkv multipliers', outq, (base + bias) * mult.repeat_interleave(N // 3, dim=1), 2e-2, 1 / 128)


@pytest.mark.parametrize('M,N,K', [(260, 512, 256), (2900, 2304, 512), (2560, 2048, 768)])
def test_gemm_gelu_derivative_output_and_multiply_epilogues(ops, M, N, K):
    """I2T_ACT_GELU_DOUT (forward: C = gelu(v), aux_out = gelu'(v)) and I2T_ACT_MUL_AUX (backward: v *= aux_in): the pair the dense MLP
    uses instead of (GELU + pre-activation, GELU' re-evaluated) -- small shape = the 128^2 kernel's generic epilogue, large = the
    persistent kernel's classes 10 / 11 (and the generic class for the layouts those are not built for)."""
    a, b = rnd(M, K, dtype=BF16, seed=23), rnd(N, K, dtype=BF16, seed=24, scale=0.1)
    bias = rnd(N, seed=25)
    base = a.float() @ b.float().t()
    out, dout = torch.empty(M, N, dtype=BF16, device=dev()), torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, out, M, N, K, bias=bias, act=ops.ACT_GELU_DOUT, aux_out=dout)
    check('gelu(bias + ab)', out, F.gelu(base + bias, approximate='tanh'), 1e-2, 1 / 128)
    check("gelu'(bias + ab)", dout, gelu_grad(base + bias), 1e-2, 1 / 128)
    ref_out = torch.empty_like(out)
    ops.gemm(a, b, ref_out, M, N, K, bias=bias, act=1)
    assert torch.equal(out, ref_out), 'the activated output must not depend on what the second output holds'
    ops.gemm(a, b, out, M, N, K, act=ops.ACT_GELU_DOUT, aux_out=dout)           # no bias
    check("gelu'(ab)", dout, gelu_grad(base), 1e-2, 1 / 128)
    aux = rnd(M, N, dtype=BF16, seed=27)
    bt = b.t().contiguous()
    for kw, what in ((dict(), 'B^T form'), (dict(b_kmajor=True), 'k-major B (the dX form)')):
        od = torch.empty(M, N, dtype=BF16, device=dev())
        ops.gemm(a, bt if kw else b, od, M, N, K, act=ops.ACT_MUL_AUX, aux_in=aux, **kw)
        check(f'multiply by aux, {what}', od, base * aux.float(), 2e-2, 1 / 128)
    # the pair is the old pair: (ab^T) * gelu'(pre) either way
    pre = rnd(M, N, dtype=BF16, seed=28)
    old_, new_ = torch.empty(M, N, dtype=BF16, device=dev()), torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, bt, old_, M, N, K, b_kmajor=True, act=2, aux_in=pre)
    ops.gemm(a, bt, new_, M, N, K, b_kmajor=True, act=ops.ACT_MUL_AUX, aux_in=gelu_grad(pre.float()).to(BF16))
    check('dgelu == multiply by the stored derivative', new_, old_.float(), 1e-2, 1 / 64)


def test_colsum(ops):
    x = rnd(5000, 200, dtype=BF16, seed=11)
    out = torch.zeros(200, device=dev())
    ops.colsum(x, out, 5000, 200)
    check('colsum', out, x.float().sum(0), 2e-2, 2e-3)
    ops.colsum(x[:100], out, 100, 200, accumulate=True)
    check('colsum acc', out, x.float().sum(0) + x[:100].float().sum(0), 2e-2, 2e-3)


# ------------------------------------------------------------------------------------------------------ LayerNorm
@pytest.mark.parametrize('M,d,bias', [(1000, 768, True), (777, 512, False), (64, 64, False), (33, 128, True), (16, 1024, True),
                                         (77, 4544, True), (19, 8192, False)])      # > 1024: the wide-row form (Falcon-7B's d = 4544)
def test_layernorm(ops, M, d, bias):
    x = rnd(M, d, seed=12) * 2 + 0.5
    g = 1 + 0.1 * rnd(d, seed=13)
    b = 0.1 * rnd(d, seed=14) if bias else None
    ref = F.layer_norm(x, (d,), g, b, 1e-5)
    y = torch.empty(M, d, dtype=BF16, device=dev())
    mean, rstd = torch.empty(M, device=dev()), torch.empty(M, device=dev())
    ops.layernorm_fwd(x, g, b, y, mean, rstd, M, d)
    check('ln fwd bf16', y, ref, 1e-2, 1 / 128)
    y32 = torch.empty(M, d, dtype=F32, device=dev())
    ops.layernorm_fwd(x, g, b, y32, mean, rstd, M, d)
    check('ln fwd f32', y32, ref, 1e-5, 1e-5)
    check('ln mean', mean, x.mean(-1), 1e-5, 1e-5)
    # backward
    xr = x.clone().requires_grad_(True)
    gr = g.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    dy = rnd(M, d, seed=15)
    F.layer_norm(xr, (d,), gr, br, 1e-5).backward(dy)
    for dy_in, tol in ((dy, 1e-4), (dy.to(BF16), 2e-2)):
        dx = torch.full((M, d), 0.25, device=dev())
        dg, db = torch.zeros(d, device=dev()), torch.zeros(d, device=dev())
        dxb = torch.empty(M, d, dtype=BF16, device=dev())
        ops.layernorm_bwd(dy_in, x, g, mean, rstd, dx, dg, db if bias else None, M, d, dx_accumulate=True, dx_bf16=dxb)
        check('ln dx (+accumulate)', dx, xr.grad + 0.25, tol, tol)
        check('ln dx bf16 copy', dxb, dx, 0, 1 / 128)
        check('ln dgamma', dg, gr.grad, tol * math.sqrt(M), tol)
        if bias:
            check('ln dbeta', db, br.grad, tol * math.sqrt(M), tol)


@pytest.mark.parametrize('B,rows,d,bias', [(3, 16, 64, False), (2, 196, 512, False), (2, 16, 64, True),
                                           (130, 16, 64, True)])       # 130 images: the backward runs in 4 ragged batch slices
def test_layernorm_nd(ops, B, rows, d, bias):
    x = rnd(B, rows, d, seed=16) * 1.5 + 0.3
    add = rnd(rows, d, seed=17)
    g = 1 + 0.1 * rnd(rows, d, seed=18)
    b = 0.1 * rnd(rows, d, seed=19) if bias else None
    ncls = 8
    for use_add in (False, True):
        xin = x + add if use_add else x
        ref = F.layer_norm(xin, (rows, d), g, b, 1e-5)
        ybuf = torch.zeros(B, ncls + rows, d, device=dev())
        stats = torch.zeros(B, ops.LNND_STATS_STRIDE, device=dev())
        ops.layernorm_nd_fwd(x, add if use_add else None, g, b, ybuf[:, ncls:], (ncls + rows) * d, stats, B, rows, d)
        check('lnnd fwd', ybuf[:, ncls:], ref, 2e-5, 2e-5)
        assert float(ybuf[:, :ncls].abs().max()) == 0.0
        xr = x.clone().requires_grad_(True)
        ar = add.clone().requires_grad_(True)
        gr = g.clone().requires_grad_(True)
        br = b.clone().requires_grad_(True) if bias else None
        dyb = torch.zeros(B, ncls + rows, d, device=dev())
        dyb[:, ncls:] = rnd(B, rows, d, seed=20)
        F.layer_norm(xr + ar if use_add else xr, (rows, d), gr, br, 1e-5).backward(dyb[:, ncls:])
        dx = torch.empty(B, rows, d, device=dev())
        dg, db, da = torch.zeros(rows, d, device=dev()), torch.zeros(rows, d, device=dev()), torch.zeros(rows, d, device=dev())
        ops.layernorm_nd_bwd(dyb[:, ncls:], (ncls + rows) * d, x, add if use_add else None, g, stats, dx, dg,
                             db if bias else None, da if use_add else None, B, rows, d)
        check('lnnd dx', dx, xr.grad, 1e-4, 1e-4)
        check('lnnd dgamma', dg, gr.grad, 1e-4, 1e-4)
        if bias:
            check('lnnd dbeta', db, br.grad, 1e-4, 1e-4)
        if use_add:
            check('lnnd dadd', da, ar.grad, 1e-4, 1e-4)


# ------------------------------------------------------------------------------------------------------ attention
def ref_attention(q, k, v, causal):
    B, Tq, H, _ = q.shape
    Tk = k.shape[1]
    qh, kh, vh = (t.permute(0, 2, 1, 3).double() for t in (q, k, v))
    s = qh @ kh.transpose(-1, -2) / 8.0
    if causal:
        i = torch.arange(Tq, device=q.device)[:, None]
        j = torch.arange(Tk, device=q.device)[None, :]
        s = s.masked_fill(j > i + (Tk - Tq), float('-inf'))
    p = torch.softmax(s, dim=-1)
    return (p @ vh).permute(0, 2, 1, 3), torch.logsumexp(s, dim=-1)


@pytest.mark.parametrize('B,H,Tq,Tk,causal,packed', [
    (2, 3, 64, 64, True, True), (2, 8, 260, 260, False, True), (3, 12, 128, 64, False, False), (2, 2, 37, 37, True, True),
    (1, 4, 200, 200, True, True), (2, 2, 16, 16, True, True), (2, 1, 24, 24, False, True), (2, 2, 1, 50, True, False),
    # the resident-operand (v2) kernels: ViT-B/16 rows, the encoder's CLS-only last block, their size limits, a single short block
    (2, 12, 197, 197, False, True), (2, 2, 64, 260, False, False), (1, 2, 304, 288, False, False), (2, 1, 5, 9, False, False)])
def test_attention_fwd_bwd(ops, B, H, Tq, Tk, causal, packed):
    d = 64 * H
    if packed:
        qkv = rnd(B, Tq, 3 * d, dtype=BF16, seed=21)
        q, k, v = qkv[..., :d], qkv[..., d:2 * d], qkv[..., 2 * d:]
    else:
        q = rnd(B, Tq, d, dtype=BF16, seed=22)
        kv = rnd(B, Tk, 2 * d, dtype=BF16, seed=23)
        k, v = kv[..., :d], kv[..., d:]
    qr, kr, vr = (t.float().reshape(B, -1, H, 64).requires_grad_(True) for t in (q, k, v))
    o_ref, lse_ref = ref_attention(qr, kr, vr, causal)
    o = torch.empty(B, Tq, d, dtype=BF16, device=dev())
    lse = torch.empty(B, H, Tq, device=dev())
    ops.attention_fwd(q, k, v, o, lse, B, H, Tq, Tk, causal)
    check('attn out', o.reshape(B, Tq, H, 64), o_ref, 1e-2, 1 / 128)
    check('attn lse', lse, lse_ref, 2e-3, 2e-3)
    do = rnd(B, Tq, d, dtype=BF16, seed=24)
    o_ref.backward(do.double().reshape(B, Tq, H, 64))
    if packed:
        dqkv = torch.zeros(B, Tq, 3 * d, dtype=BF16, device=dev())
        dq, dk, dv = dqkv[..., :d], dqkv[..., d:2 * d], dqkv[..., 2 * d:]
    else:
        dq = torch.zeros(B, Tq, d, dtype=BF16, device=dev())
        dkv = torch.zeros(B, Tk, 2 * d, dtype=BF16, device=dev())
        dk, dv = dkv[..., :d], dkv[..., d:]
    ws = torch.empty(B, H, Tq, device=dev())
    ops.attention_bwd(q, k, v, o, do, lse, ws, dq, dk, dv, B, H, Tq, Tk, causal)
    scale = float(do.float().abs().max())
    check('attn dq', dq.reshape(B, Tq, H, 64), qr.grad, 3e-2 * scale, 1 / 32)
    check('attn dk', dk.reshape(B, Tk, H, 64), kr.grad, 3e-2 * scale, 1 / 32)
    check('attn dv', dv.reshape(B, Tk, H, 64), vr.grad, 3e-2 * scale, 1 / 32)


# ------------------------------------------------------------------------------------------------------ embed / CE / norm
def test_embed(ops):
    B, T, d, V, off = 3, 20, 128, 384, 8
    ids = torch.randint(0, V, (B, T), device=dev())
    ids[0, :3] = 7                                             # repeated ids -> atomics must accumulate
    wte, wpe = rnd(V, d, seed=25), rnd(48, d, seed=26)
    x = torch.empty(B, T, d, device=dev())
    ops.embed_fwd(ids, wte, wpe, x, B, T, d, off, V)
    check('embed fwd', x, wte[ids] + wpe[off:off + T], 0, 0)
    dx = rnd(B, T, d, seed=27)
    dwte, dwpe = torch.ones(V, d, device=dev()), torch.ones(48, d, device=dev())
    ops.embed_bwd(ids, dx, dwte, dwpe, B, T, d, off, V)
    ref_te = torch.ones(V, d, device=dev()).index_add_(0, ids.reshape(-1), dx.reshape(-1, d))
    ref_pe = torch.ones(48, d, device=dev())
    ref_pe[off:off + T] += dx.sum(0)
    check('embed dwte', dwte, ref_te, 1e-5, 1e-5)
    check('embed dwpe', dwpe, ref_pe, 1e-5, 1e-5)


@pytest.mark.parametrize('M,V,ld,temp', [(64, 50257, 50264, 1.0), (48, 384, 384, 0.7), (5, 1000, 1008, 1.0)])
def test_cross_entropy(ops, M, V, ld, temp):
    logits = torch.zeros(M, ld, dtype=BF16, device=dev())
    logits[:, :V] = rnd(M, V, dtype=BF16, seed=28, scale=2.0)
    labels = torch.randint(0, V, (M,), device=dev())
    labels[::5] = -100
    w = torch.rand(M, device=dev())
    w[labels == -100] = 0
    lr = logits[:, :V].float().requires_grad_(True)
    ce = F.cross_entropy(lr / temp, labels, ignore_index=-100, reduction='none')
    ref_loss = (ce * w).sum()
    (ref_loss * 0.5).backward()
    lse, loss = torch.empty(M, device=dev()), torch.zeros(1, device=dev())
    ops.ce_fwd(logits, ld, labels, w, 1.0 / temp, -100, lse, loss, M, V)
    check('ce loss', loss[0], ref_loss, 1e-4, 1e-4)
    gscale = torch.full((1,), 0.5, device=dev())
    ops.ce_bwd(logits, ld, labels, w, 1.0 / temp, -100, lse, gscale, M, V)
    check('ce dlogits', logits[:, :V], lr.grad, 1e-6, 1 / 128)
    if ld > V:
        assert float(logits[:, V:].float().abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize('M,V,ld,temp', [(37, 1000, 1000, 1.0), (19, 50257, 50264, 0.7), (5, 65536, 65536, 1.0), (9, 13, 16, 2.0)])
def test_cross_entropy_one_pass(ops, M, V, ld, temp):
    """i2t_ce_fwd_bwd (the training step's form): loss and lse of i2t_ce_fwd, rows overwritten with the gradient at upstream scale 1
    (dead rows zero, pad columns untouched); i2t_scale_bf16 then applies the upstream scalar -- and leaves the buffer bit for bit alone at 1."""
    logits = torch.zeros(M, ld, dtype=BF16, device=dev())
    logits[:, :V] = rnd(M, V, dtype=BF16, seed=28, scale=2.0)
    labels = torch.randint(0, V, (M,), device=dev())
    labels[::5] = -100
    labels[1] = V - 1                                  # a label in the V % 8 tail
    w = torch.rand(M, device=dev())
    w[labels == -100] = 0
    lr = logits[:, :V].float().requires_grad_(True)
    ce = F.cross_entropy(lr / temp, labels, ignore_index=-100, reduction='none')
    ref_loss = (ce * w).sum()
    ref_loss.backward()
    two = logits.clone()
    lse2, loss2 = torch.empty(M, device=dev()), torch.zeros(1, device=dev())
    ops.ce_fwd(two, ld, labels, w, 1.0 / temp, -100, lse2, loss2, M, V)
    ops.ce_bwd(two, ld, labels, w, 1.0 / temp, -100, lse2, torch.ones(1, device=dev()), M, V)
    lse, loss = torch.empty(M, device=dev()), torch.zeros(1, device=dev())
    ops.ce_fwd_bwd(logits, ld, labels, w, 1.0 / temp, -100, lse, loss, M, V)
    check('ce one-pass loss', loss[0], ref_loss, 1e-4, 1e-4)
    check('ce one-pass lse', lse, lse2, 1e-5, 1e-5)
    check('ce one-pass dlogits', logits[:, :V], lr.grad, 1e-6, 1 / 128)
    check('ce one-pass vs two-pass', logits[:, :V], two[:, :V].float(), 1e-6, 1 / 128)
    if ld > V:
        assert float(logits[:, V:].float().abs().max()) == 0.0
    keep = logits.clone()
    ops.scale_bf16(logits, M * ld, torch.ones(1, device=dev()))
    assert torch.equal(keep, logits)
    ops.scale_bf16(logits, M * ld, torch.full((1,), 0.25, device=dev()))
    assert torch.equal(logits.float(), keep.float() * 0.25)          # a power of two: exact in bf16


def test_grad_normalize(ops):
    for n in (1000, 64 * 128 * 768 + 3):
        g = rnd(n, seed=29) * 3
        ref = g / (torch.linalg.vector_norm(g.double()).float() + 1e-6)
        ws = torch.empty(1, device=dev())
        gb = torch.empty(n, dtype=BF16, device=dev())
        ops.grad_normalize(g, ws, gb)
        check('grad_normalize', g, ref, 1e-7, 1e-4)
        check('grad_normalize bf16 copy', gb, g, 0, 1 / 128)
    # sum of squares handed over by the producer (layernorm_bwd's sumsq_out) instead of a reduction pass
    M, d = 500, 512
    x, gam, dy = rnd(M, d, seed=30), rnd(d, seed=31), rnd(M, d, seed=32)
    mean, rstd = x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    dx = rnd(M, d, seed=33)
    ws2 = torch.zeros(2, device=dev())
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx, None, None, M, d, dx_accumulate=True, sumsq_out=ws2[0:1])
    check('ln_bwd sumsq_out', ws2[0:1], (dx.double() ** 2).sum().float().view(1), 1e-3, 1e-5)
    ws2[1] = 123.0
    ref2 = dx / (torch.linalg.vector_norm(dx.double()).float() + 1e-6)
    ops.grad_normalize(dx, ws2[0:1], None, presummed=True, clear_after=ws2[1:2])
    check('grad_normalize presummed', dx, ref2, 1e-7, 1e-4)
    assert float(ws2[1]) == 0.0
    # deferred form: the fp32 tensor is left as it is, only the normalised bf16 copy is written; the next LayerNorm backward
    # that accumulates onto it applies the factor -> same result as normalise-then-accumulate
    g = rnd(M, d, seed=34) * 2
    g0 = g.clone()
    ws3 = torch.zeros(1, device=dev())
    gb = torch.empty(M, d, dtype=BF16, device=dev())
    ops.grad_normalize(g, ws3, gb, keep_f32=True)
    assert torch.equal(g, g0)
    inv = 1.0 / (torch.linalg.vector_norm(g0.double()).float() + 1e-6)
    check('grad_normalize keep_f32 bf16 copy', gb, g0 * inv, 1e-7, 1 / 128)
    want = g0 * inv
    ops.layernorm_bwd(dy, x, gam, mean, rstd, want, None, None, M, d, dx_accumulate=True)          # reference: already scaled
    ops.layernorm_bwd(dy, x, gam, mean, rstd, g, None, None, M, d, dx_accumulate=True, dx_pre_sumsq=ws3)
    check('ln_bwd dx_pre_sumsq', g, want, 1e-6, 1e-5)


# ------------------------------------------------------------------------------------------------------ conv stack
def test_dw_gemm_between_half_and_one_round_of_tiles(ops):
    """dW = dY^T X with 129 .. 255 output tiles of 256^2 (a Qwen2-1.5B down_proj: 1536 x 8960 = 210 tiles): too many to split K,
    too few for a full round -- one tile per workgroup of the persistent kernel (it used to fall back to the 128^2 kernel)"""
    M, N, K = 4096 + 40, 1536, 8960                      # rows (the reduction), output rows, output columns
    dy = rnd(M, N, seed=301, dtype=BF16)
    x = rnd(M, K, seed=302, dtype=BF16)
    g = torch.full((N, K), 0.25, device=dev())
    ops.gemm(dy, x, g, N, K, M, a_kmajor=True, b_kmajor=True, accumulate=True)
    ref = 0.25 + dy.double().t() @ x.double()
    check('dW 210 tiles', g, ref, 2e-2, 2e-3)


@pytest.mark.parametrize('M,N,K,kind', [(256, 1536, 8960, 'res'), (200, 768, 3072, 'gelu'), (1000, 130, 1024, 'bias_f32'), (96, 2304, 768, 'bias'),
                                        (2048, 768, 3072, 'res'), (64, 768, 3072, 'res'), (300, 50264, 768, 'bias_f32'), (4096, 768, 3072, 'res')])
def test_gemm_deterministic_splitk(ops, M, N, K, kind):
    """i2t_gemm_bf16_ws (decode steps at mid-sized caption batches): K slices to private fp32 planes + an ordered reduce with the
    fused epilogue -- against fp64, bit-identical from launch to launch, and falling through to the ordinary kernels where
    splitting does not pay (M <= 64, M > 2048, many tiles)"""
    x = rnd(M, K, seed=201, dtype=BF16)
    w = (rnd(N, K, seed=202) / math.sqrt(K)).to(BF16)
    bias = 0.1 * rnd(N, seed=203)
    ws = torch.empty(16 << 20, dtype=F32, device=dev())
    ref = x.double() @ w.double().t()
    outs = []
    for rep in range(3):
        ws.fill_(float('nan'))                     # a plane that is read without having been written would show
        if kind == 'res':
            res0 = rnd(M, N, seed=204)
            out = res0.clone()
            ops.gemm(x, w, out, M, N, K, bias=bias, residual=out, workspace=ws)          # in place on the fp32 residual stream
            want = ref + bias.double() + res0.double()
        elif kind == 'gelu':
            out = torch.empty(M, N, dtype=BF16, device=dev())
            ops.gemm(x, w, out, M, N, K, bias=bias, act=1, workspace=ws)
            want = F.gelu((ref + bias.double()).float(), approximate='tanh').double()
        else:
            out = torch.empty(M, (N + 7) // 8 * 8, dtype=F32 if kind == 'bias_f32' else BF16, device=dev()).zero_()
            ops.gemm(x, w, out, M, N, K, bias=bias, workspace=ws)
            want = ref + bias.double()
            out = out[:, :N]
        outs.append(out.clone())
        check(f'splitk {kind} {M}x{N}x{K}', out, want, 2e-2 if out.dtype == BF16 else 3e-3, 1e-2 if out.dtype == BF16 else 2e-3)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), 'the split-K form must be bit-reproducible'


def _conv2d(x, w, b):
    """the reference convolution (and, through autograd, its gradients) evaluated on the HOST: keeps MIOpen -- whose first use of a
    configuration on a fresh box builds kernels and once aborted the process -- out of the GPU suite; the device transfers are
    differentiable, so leaf gradients still land on the device tensors"""
    return F.conv2d(x.cpu(), w.cpu(), None if b is None else b.cpu()).to(x.device)


@pytest.mark.parametrize('B,H,W,chans,k', [(2, 32, 32, (3, 4, 8, 8), 6), (2, 64, 96, (3, 8, 16, 32), 6), (1, 40, 40, (3, 8, 16), 4)])
def test_conv_stack(ops, B, H, W, chans, k):
    x0 = rnd(B, chans[0], H, W, seed=30)
    ws = [rnd(chans[i + 1], chans[i], k, k, seed=31 + i) / math.sqrt(chans[i] * k * k) for i in range(len(chans) - 1)]
    bs = [0.1 * rnd(chans[i + 1], seed=41 + i) for i in range(len(chans) - 1)]
    # reference chain with the same bf16 rounding of stored pre-activations
    pres, inp = [], x0
    wr = [w.clone().requires_grad_(True) for w in ws]
    br = [b.clone().requires_grad_(True) for b in bs]
    ref_in = []
    for i, (w, b) in enumerate(zip(wr, br)):
        a = inp if i == 0 else F.gelu(inp, approximate='tanh')
        ref_in.append(a)
        pre = _conv2d(F.pad(a, ((k - 1) // 2, k // 2, (k - 1) // 2, k // 2)), w, b)
        pres.append(pre)
        inp = pre.detach().to(BF16).float().requires_grad_(True)     # what the kernel chain stores / re-reads
        pres[-1] = (pre, inp)
    w_ws = torch.empty(max(w.numel() for w in ws), device=dev())
    ys, cur = [], x0
    for i, (w, b) in enumerate(zip(ws, bs)):
        y = torch.empty(B, chans[i + 1], H, W, dtype=BF16, device=dev())
        ops.conv_fwd(cur, i > 0, w, b, y, w_ws, B, chans[i], chans[i + 1], H, W, k)
        check(f'conv{i} fwd', y, pres[i][0], 2e-2, 1 / 64)
        ys.append(y)
        cur = y
    # backward through the chain, layer by layer against autograd on the reference layer
    dy = rnd(B, chans[-1], H, W, dtype=BF16, seed=50)
    for i in reversed(range(len(ws))):
        pre, _ = pres[i]
        xin = x0 if i == 0 else ys[i - 1]
        a_ref = ref_in[i] if i == 0 else F.gelu(ys[i - 1].float(), approximate='tanh')
        a_ref = a_ref.detach().requires_grad_(True)
        wi, bi = ws[i].clone().requires_grad_(True), bs[i].clone().requires_grad_(True)
        _conv2d(F.pad(a_ref, ((k - 1) // 2, k // 2, (k - 1) // 2, k // 2)), wi, bi).backward(dy.float())
        dw, db = torch.zeros_like(ws[i]), torch.zeros_like(bs[i])
        ops.conv_bwd_weight(dy, xin, i > 0, dw, db, B, chans[i], chans[i + 1], H, W, k)
        sc = float(wi.grad.abs().max())
        check(f'conv{i} dW', dw, wi.grad, 2e-3 * sc, 5e-3)
        check(f'conv{i} db', db, bi.grad, 2e-3 * float(bi.grad.abs().max()), 5e-3)
        if i > 0:
            dx = torch.empty(B, chans[i], H, W, dtype=BF16, device=dev())
            ops.conv_bwd_data(dy, ws[i], ys[i - 1], True, dx, w_ws, B, chans[i], chans[i + 1], H, W, k)
            ref_dx = a_ref.grad * gelu_grad(ys[i - 1].float())
            check(f'conv{i} dX', dx, ref_dx, 2e-2 * float(ref_dx.abs().max()), 1 / 64)
            dy = dx


# ------------------------------------------------------------------------------------------------------ optimiser / misc
def test_adamw_matches_torch(ops):
    n1, n2 = 1000, 2048
    p = rnd(n1 + n2, seed=60)
    ref_p = [p[:n1].clone().requires_grad_(True), p[n1:].clone().requires_grad_(True)]
    opt = torch.optim.AdamW([{'params': [ref_p[0]], 'lr': 1e-2, 'weight_decay': 0.1},
                             {'params': [ref_p[1]], 'lr': 3e-3, 'weight_decay': 0.0}], betas=(0.9, 0.95), eps=1e-8)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    pb = torch.empty(n1 + n2, dtype=BF16, device=dev())
    seg_end = torch.tensor([n1, n1 + n2], dtype=torch.long, device=dev())
    seg_lr = torch.tensor([1e-2, 3e-3], device=dev())
    seg_wd = torch.tensor([0.1, 0.0], device=dev())
    for step in range(1, 4):
        g = rnd(n1 + n2, seed=61 + step)
        ref_p[0].grad, ref_p[1].grad = g[:n1].clone(), g[n1:].clone()
        opt.step()
        ops.adamw_step(p, g, m, v, pb, n1 + n2, seg_end, seg_lr, seg_wd, 2, 0.9, 0.95, 1e-8, step)
        check(f'adamw step {step}', p, torch.cat([ref_p[0], ref_p[1]]).detach(), 1e-6, 1e-5)
    check('adamw bf16 shadow', pb, p, 0, 1 / 128)


def test_small_helpers(ops):
    B, rows, d, ncls = 3, 5, 64, 2
    src = rnd(rows, d, seed=70)
    y = torch.zeros(B, ncls + rows, d, device=dev())
    ops.bcast_rows(src, y[:, ncls:], (ncls + rows) * d, B, rows, d)
    check('bcast_rows', y[:, ncls:], src.expand(B, rows, d), 0, 0)
    dst = torch.ones(rows, d, device=dev())
    ops.sum_over_batch(y[:, ncls:], (ncls + rows) * d, dst, B, rows, d, accumulate=True)
    check('sum_over_batch', dst, 1 + B * src, 1e-6, 1e-6)
    # a batch large enough for the sliced form (131 entries -> 4 ragged slices, float atomics), with and without accumulate
    Bb = 131
    xb = rnd(Bb, ncls + rows, d, seed=73)
    want = xb[:, ncls:].double().sum(0).float()
    for acc in (True, False):
        dst = torch.full((rows, d), 2.0, device=dev())
        ops.sum_over_batch(xb[:, ncls:], (ncls + rows) * d, dst, Bb, rows, d, accumulate=acc)
        check(f'sum_over_batch sliced acc={acc}', dst, want + (2.0 if acc else 0.0), 1e-4, 1e-5)
    out = torch.zeros(B, rows, d, dtype=BF16, device=dev())
    ops.copy_rows(y[:, ncls:], (ncls + rows) * d, out, rows * d, B, rows, d)
    check('copy_rows', out, src.expand(B, rows, d), 0, 1 / 128)
    a, b = rnd(1003, seed=71), rnd(1003, seed=72)
    ref = a + b
    ops.add_(a, b)
    check('add', a, ref, 0, 0)
    c = torch.empty(1003, dtype=BF16, device=dev())
    ops.cast_f32_bf16(b, c)
    check('cast', c, b, 0, 1 / 128)


# ------------------------------------------------------------------------------------------------------ decode pieces
def test_ngram_ban_argmax_vs_oracle(ops):
    from oracle import reference_model as orc
    B, V, L = 6, 384, 40
    g = torch.Generator().manual_seed(5)
    ids = torch.randint(0, 6, (B, L), generator=g)            # tiny alphabet -> many repeated n-grams
    sizes = (2, 3, 4, 5)
    for cur in (1, 2, 5, 17, 39):
        logits = torch.randn(B, V, generator=g)
        ref = orc.apply_ngram_ban(ids[:, :cur], logits.clone(), sizes)
        t2 = torch.topk(ref, 2, dim=-1).values
        dev_ids = torch.zeros(B, L, dtype=torch.long, device=dev())
        dev_ids[:, :cur] = ids[:, :cur].to(dev())
        len_ptr = torch.tensor([cur], dtype=torch.int32, device=dev())
        margin = torch.zeros(B, device=dev())
        ops.ngram_ban_argmax(logits.to(dev()), V, dev_ids, L, len_ptr, torch.tensor(sizes, dtype=torch.int32, device=dev()),
                             len(sizes), B, V, margin)
        assert torch.equal(dev_ids[:, cur].cpu(), ref.argmax(-1)), f'cur={cur}'
        check('margin', margin, t2[:, 0] - t2[:, 1], 1e-6, 1e-6)
        ops.advance(len_ptr, 1)
        assert int(len_ptr.item()) == cur + 1


def test_decode_attention_and_kv_append(ops):
    B, H, Tmax = 3, 2, 32
    d = 64 * H
    kc, vc = rnd(B, Tmax, d, dtype=BF16, seed=80), rnd(B, Tmax, d, dtype=BF16, seed=81)
    qkv = rnd(B, 3 * d, dtype=BF16, seed=82)
    pos = torch.tensor([9], dtype=torch.int32, device=dev())
    ops.kv_append(qkv, 3 * d, kc, vc, Tmax * d, d, pos, B, d)
    assert torch.equal(kc[:, 9], qkv[:, d:2 * d]) and torch.equal(vc[:, 9], qkv[:, 2 * d:])
    o = torch.empty(B, d, dtype=BF16, device=dev())
    ops.decode_attention(qkv, 3 * d, kc, vc, Tmax * d, d, o, d, pos, 0, B, H)
    q = qkv[:, :d].float().reshape(B, 1, H, 64)
    ref, _ = ref_attention(q, kc[:, :10].float().reshape(B, 10, H, 64), vc[:, :10].float().reshape(B, 10, H, 64), False)
    check('decode attention (cache)', o.reshape(B, 1, H, 64), ref, 1e-2, 1 / 128)
    ops.decode_attention(qkv, 3 * d, kc, vc, Tmax * d, d, o, d, None, 20, B, H)
    ref, _ = ref_attention(q, kc[:, :20].float().reshape(B, 20, H, 64), vc[:, :20].float().reshape(B, 20, H, 64), False)
    check('decode attention (fixed keys)', o.reshape(B, 1, H, 64), ref, 1e-2, 1 / 128)
    # fused append: position 10 is written by the attention launch itself and attended to
    pos.fill_(10)
    qkv2 = rnd(B, 3 * d, dtype=BF16, seed=83)
    ops.decode_attention(qkv2, 3 * d, kc, vc, Tmax * d, d, o, d, pos, 0, B, H, append_dm=d)
    assert torch.equal(kc[:, 10], qkv2[:, d:2 * d]) and torch.equal(vc[:, 10], qkv2[:, 2 * d:])
    q2 = qkv2[:, :d].float().reshape(B, 1, H, 64)
    ref, _ = ref_attention(q2, kc[:, :11].float().reshape(B, 11, H, 64), vc[:, :11].float().reshape(B, 11, H, 64), False)
    check('decode attention (fused append)', o.reshape(B, 1, H, 64), ref, 1e-2, 1 / 128)
    # head-major cache [B][H][Tmax][64] (the self-attention cache of the decode step): same keys, same result, and the appended
    # row lands in its head's run
    for Hh in (2, 4):                                       # 1 and 4 heads per workgroup
        dd = 64 * Hh
        kt, vt = rnd(B, Tmax, dd, dtype=BF16, seed=84), rnd(B, Tmax, dd, dtype=BF16, seed=85)
        kh = kt.view(B, Tmax, Hh, 64).permute(0, 2, 1, 3).contiguous()
        vh = vt.view(B, Tmax, Hh, 64).permute(0, 2, 1, 3).contiguous()
        q3 = rnd(B, 3 * dd, dtype=BF16, seed=86)
        o_t, o_h = torch.empty(B, dd, dtype=BF16, device=dev()), torch.empty(B, dd, dtype=BF16, device=dev())
        ops.decode_attention(q3, 3 * dd, kt, vt, Tmax * dd, dd, o_t, dd, pos, 0, B, Hh, append_dm=dd)
        ops.decode_attention(q3, 3 * dd, kh, vh, Tmax * dd, 64, o_h, dd, pos, 0, B, Hh, append_dm=dd, cache_hs=Tmax * 64)
        assert torch.equal(o_t, o_h)
        assert torch.equal(kh.permute(0, 2, 1, 3).reshape(B, Tmax, dd), kt) and torch.equal(vh.permute(0, 2, 1, 3).reshape(B, Tmax, dd), vt)


# ------------------------------------------------------------------------------------------------------ split-K / skinny paths
@pytest.mark.parametrize('a_km,b_km', [(0, 0), (0, 1), (1, 1), (1, 0)])
def test_gemm_splitk_accumulate(ops, a_km, b_km):
    """dW-shaped problems (small M x N, long K) take the split-K + float-atomics path: C += op(A).op(B)."""
    M, N, K = 512, 768, 8192
    a, b = rnd(M, K, dtype=BF16, seed=90, scale=0.2), rnd(N, K, dtype=BF16, seed=91, scale=0.2)
    base = rnd(M, N, seed=92)
    ref = base.double() + a.double() @ b.double().t()
    a_in = a.t().contiguous() if a_km else a
    b_in = b.t().contiguous() if b_km else b
    c = base.clone()
    ops.gemm(a_in, b_in, c, M, N, K, a_kmajor=a_km, b_kmajor=b_km, accumulate=True)
    check(f'splitk a_km={a_km} b_km={b_km}', c, ref, 3e-3, 2e-3)
    # odd sizes + K tail
    M2, N2, K2 = 200, 136, 4104
    a2, b2 = rnd(M2, K2, dtype=BF16, seed=93, scale=0.2), rnd(N2, K2, dtype=BF16, seed=94, scale=0.2)
    c2 = torch.zeros(M2, N2, device=dev())
    ops.gemm(a2.t().contiguous() if a_km else a2, b2.t().contiguous() if b_km else b2, c2, M2, N2, K2, a_kmajor=a_km,
             b_kmajor=b_km, accumulate=True)
    check('splitk odd', c2, a2.double() @ b2.double().t(), 3e-3, 2e-3)


@pytest.mark.parametrize('M', [1, 8, 33, 64])
def test_gemm_skinny_decode_shapes(ops, M):
    d, ff, V = 768, 3072, 50257
    x = rnd(M, d, dtype=BF16, seed=95)
    w1, b1 = rnd(ff, d, dtype=BF16, seed=96, scale=0.05), rnd(ff, seed=97)
    h = torch.empty(M, ff, dtype=BF16, device=dev())
    ops.gemm(x, w1, h, M, ff, d, bias=b1, act=1)                                   # c_fc + GELU, bf16 out
    ref_h = F.gelu(x.float() @ w1.float().t() + b1, approximate='tanh')
    check('skinny fc+gelu', h, ref_h, 1e-2, 1 / 128)
    w2, b2 = rnd(d, ff, dtype=BF16, seed=98, scale=0.05), rnd(d, seed=99)
    res = rnd(M, d, seed=100)
    want = res.double() + (h.double() @ w2.double().t() + b2.double())
    ops.gemm(h, w2, res, M, d, ff, bias=b2, residual=res)                          # in-place residual (decode form)
    check('skinny proj in-place residual', res, want, 5e-3, 2e-3)
    wte = rnd(V, d, dtype=BF16, seed=101, scale=0.05)
    logits = torch.empty(M, V, device=dev())
    ops.gemm(x, wte, logits, M, V, d)                                              # lm_head, fp32 logits, odd N
    check('skinny lm_head', logits, x.float() @ wte.float().t(), 3e-3, 2e-3)
    out = torch.empty(M, d, device=dev())
    ops.gemm(h, w2, out, M, d, ff, bias=b2, residual=rnd(M, d, seed=102))          # out-of-place residual (no split)
    check('skinny proj residual', out, rnd(M, d, seed=102).double() + h.double() @ w2.double().t() + b2.double(), 5e-3, 2e-3)


# ------------------------------------------------------------------------------------------------------ MFMA convolutions
@pytest.mark.parametrize('B,H,W,chans', [(2, 32, 48, (3, 8, 16, 32)), (1, 40, 24, (3, 8, 8)), (2, 224, 224, (3, 8, 16, 32))])
def test_conv6_mfma_stack(ops, B, H, W, chans):
    """Channels-last implicit-GEMM convolutions: forward chain, then backward-data / backward-weight layer by layer against
    autograd on the same bf16-rounded operands (weights are bf16-rounded inside the kernels: tolerance 2^-7)."""
    k, L = 6, len(chans) - 1
    x0 = rnd(B, chans[0], H, W, seed=130)
    ws = [rnd(chans[i + 1], chans[i], k, k, seed=131 + i) / math.sqrt(chans[i] * k * k) for i in range(L)]
    bs = [0.1 * rnd(chans[i + 1], seed=141 + i) for i in range(L)]
    w_ws = torch.empty(32 * 36 * 32, dtype=BF16, device=dev())
    scratch = torch.empty(32 * 36 * 16, dtype=F32, device=dev())
    pad = lambda t: F.pad(t, (2, 3, 2, 3))
    ys, cur, refs = [], x0, []
    for i in range(L):
        last = i == L - 1
        y = torch.empty((B, chans[i + 1], H, W) if last else (B, H, W, chans[i + 1]), dtype=BF16, device=dev())
        ops.conv6_fwd(cur, 0 if i == 0 else 2, i > 0, ws[i], bs[i], y, last, w_ws, B, chans[i], chans[i + 1], H, W)
        a = x0 if i == 0 else F.gelu(ys[i - 1].float().permute(0, 3, 1, 2), approximate='tanh')      # kernel's own stored input
        ref = _conv2d(pad(a), ws[i].to(BF16).float(), bs[i])
        check(f'conv6 fwd layer {i}', y.float() if last else y.float().permute(0, 3, 1, 2), ref, 2e-2, 1 / 64)
        ys.append(y)
        refs.append(a)
        cur = y
    dy = rnd(B, chans[-1], H, W, dtype=BF16, seed=150)           # NCHW, as the projector's dX GEMM produces it
    dy_layout = 1
    for i in reversed(range(L)):
        a = refs[i].detach().requires_grad_(True)
        wi, bi = ws[i].clone().requires_grad_(True), bs[i].clone().requires_grad_(True)
        dyn = dy.float() if dy_layout == 1 else dy.float().permute(0, 3, 1, 2)
        _conv2d(pad(a), wi, bi).backward(dyn)
        dw, db = torch.ones_like(ws[i]), torch.ones_like(bs[i])
        ops.conv6_bwd_weight(dy, dy_layout, x0 if i == 0 else ys[i - 1], 0 if i == 0 else 2, i > 0, dw, db, scratch, B, chans[i],
                             chans[i + 1], H, W)
        check(f'conv6 dW layer {i}', dw, wi.grad + 1, 1e-2 * float(wi.grad.abs().max()), 1 / 64)
        check(f'conv6 db layer {i}', db, bi.grad + 1, 1e-2 * float(bi.grad.abs().max()), 1 / 64)
        if i > 0:
            dx = torch.empty(B, H, W, chans[i], dtype=BF16, device=dev())
            ops.conv6_bwd_data(dy, dy_layout, ws[i], ys[i - 1], dx, w_ws, B, chans[i], chans[i + 1], H, W)
            a2 = refs[i].detach().requires_grad_(True)
            _conv2d(pad(a2), ws[i].to(BF16).float(), None).backward(dyn)
            ref_dx = a2.grad * gelu_grad(ys[i - 1].float().permute(0, 3, 1, 2))
            check(f'conv6 dX layer {i}', dx.float().permute(0, 3, 1, 2), ref_dx, 2e-2 * float(ref_dx.abs().max()), 1 / 32)
            dy, dy_layout = dx, 2


# ------------------------------------------------------------------------------------------------------ dropout
def test_dropout_rule_matches_host_replica(ops):
    from image2text_amd import rng
    key, thr = rng.site_key(123456789012345, 4099), rng.threshold(0.1)
    x = torch.ones(300, 768, device=dev())
    ops.dropout_apply(x, 300, 768, (1, key, thr, 1 / 0.9))
    m = rng.keep_mask(key, 300 * 768, thr).view(300, 768)
    assert torch.equal(x.cpu() != 0, m) and abs(float(m.float().mean()) - 0.9) < 5e-3
    check('scale', x.cpu()[m], torch.full((int(m.sum()),), 1 / 0.9), 1e-6, 0)
    y = torch.ones(64, 3 * 128, dtype=BF16, device=dev())
    ops.dropout_apply(y, 64, 384, (2, key, thr, 1 / 0.9))
    for t in range(3):
        mt = rng.keep_mask((key + t) & 0xFFFFFFFF, 64, thr)
        assert torch.equal((y[:, t * 128:(t + 1) * 128].float().cpu() != 0), mt[:, None].expand(64, 128))


def test_gemm_epilogue_dropout(ops):
    from image2text_amd import rng
    M, N, K = 200, 384, 128
    a, b = rnd(M, K, dtype=BF16, seed=160), rnd(N, K, dtype=BF16, seed=161, scale=0.2)
    bias, res = rnd(N, seed=162), rnd(M, N, seed=163)
    key, thr, sc = rng.site_key(77, 5), rng.threshold(0.25), 1 / 0.75
    base = a.float() @ b.float().t() + bias
    out = torch.empty(M, N, device=dev())
    ops.gemm(a, b, out, M, N, K, bias=bias, residual=res, drop=(1, key, thr, sc))
    m = rng.keep_mask(key, M * N, thr).view(M, N).to(dev())
    check('resid dropout', out, res + base * m * sc, 2e-3, 2e-3)
    outb = torch.empty(M, N, dtype=BF16, device=dev())
    ops.gemm(a, b, outb, M, N, K, bias=bias, drop=(2, key, thr, sc))
    mult = torch.stack([rng.keep_mask((key + t) & 0xFFFFFFFF, M, thr) for t in range(3)], 1).float().to(dev()) * sc     # (M, 3)
    check('qkv multipliers', outb, base * mult.repeat_interleave(N // 3, dim=1), 2e-2, 1 / 128)


@pytest.mark.parametrize('B,H,Tq,Tk,causal', [(2, 2, 64, 64, True), (2, 3, 100, 40, False), (1, 2, 70, 37, False),
                                                (2, 1, 33, 131, True),       # Tk % 4 != 0: the per-lane alignment variant
                                                (2, 2, 260, 260, False), (1, 2, 200, 200, True),     # 64 n + r rows: 5-wave tail workgroups
                                                (1, 3, 197, 197, False), (2, 2, 64, 260, False)])     # v2 kernels, odd / even key counts
def test_attention_dropout_fwd_bwd(ops, B, H, Tq, Tk, causal):
    from image2text_amd import rng
    d = 64 * H
    q, k, v = rnd(B, Tq, d, dtype=BF16, seed=170), rnd(B, Tk, d, dtype=BF16, seed=171), rnd(B, Tk, d, dtype=BF16, seed=172)
    key, thr = rng.site_key(99, 18), rng.threshold(0.2)
    sc = rng.scale(thr)
    mask = rng.keep_mask(key, B * H * Tq * Tk, thr).view(B, H, Tq, Tk).to(dev())
    qr, kr, vr = (t.float().reshape(B, -1, H, 64).permute(0, 2, 1, 3).double().requires_grad_(True) for t in (q, k, v))
    s = qr @ kr.transpose(-1, -2) / 8.0
    if causal:
        i, j = torch.arange(Tq, device=dev())[:, None], torch.arange(Tk, device=dev())[None, :]
        s = s.masked_fill(j > i + (Tk - Tq), float('-inf'))
    p = torch.softmax(s, -1) * mask * sc
    o_ref = (p @ vr).permute(0, 2, 1, 3)
    o, lse = torch.empty(B, Tq, d, dtype=BF16, device=dev()), torch.empty(B, H, Tq, device=dev())
    ops.attention_fwd(q, k, v, o, lse, B, H, Tq, Tk, causal, drop=(1, key, thr, sc))
    check('attn dropout out', o.reshape(B, Tq, H, 64), o_ref, 1e-2, 1 / 128)
    do = rnd(B, Tq, d, dtype=BF16, seed=173)
    o_ref.backward(do.double().reshape(B, Tq, H, 64))
    dq, dk, dv = (torch.zeros_like(t) for t in (q, k, v))
    ops.attention_bwd(q, k, v, o, do, lse, torch.empty(B, H, Tq, device=dev()), dq, dk, dv, B, H, Tq, Tk, causal, drop=(1, key, thr, sc))
    sc_ = float(do.float().abs().max())
    for name, got, ref in (('dq', dq, qr.grad), ('dk', dk, kr.grad), ('dv', dv, vr.grad)):
        check(f'attn dropout {name}', got.reshape(B, -1, H, 64), ref.permute(0, 2, 1, 3), 3e-2 * sc_, 1 / 32)


def test_fused_backward_dropout_outputs(ops):
    """The dropout re-application fused into the producers of a gradient copy: grad_normalize / layernorm_bwd (elementwise
    mask on the bf16 copy only) and attention_bwd (per-token q/k/v multipliers on dq / dk / dv)."""
    from image2text_amd import rng
    key, thr = rng.site_key(4242, 77), rng.threshold(0.1)
    sc = rng.scale(thr)
    M, d = 300, 768
    g = rnd(M * d, seed=301) * 2
    ws, gb = torch.empty(1, device=dev()), torch.empty(M * d, dtype=BF16, device=dev())
    ops.grad_normalize(g, ws, gb, bf16_drop=(1, key, thr, sc))
    m = rng.keep_mask(key, M * d, thr).to(dev())
    check('grad_normalize masked copy', gb, g * m * sc, 1e-7, 1 / 128)
    x, gam, dy = rnd(M, d, seed=302), rnd(d, seed=303), rnd(M, d, seed=304)
    mean, rstd = x.mean(-1), (x.var(-1, unbiased=False) + 1e-5).rsqrt()
    dx0, dx1 = torch.zeros(M, d, device=dev()), torch.zeros(M, d, device=dev())
    b0, b1 = torch.empty(M, d, dtype=BF16, device=dev()), torch.empty(M, d, dtype=BF16, device=dev())
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx0, None, None, M, d, dx_bf16=b0)
    ops.layernorm_bwd(dy, x, gam, mean, rstd, dx1, None, None, M, d, dx_bf16=b1, bf16_drop=(1, key, thr, sc))
    assert torch.equal(dx0, dx1), 'the f32 gradient must not see the mask'
    check('ln_bwd masked copy', b1, dx0 * m.view(M, d) * sc, 1e-7, 1 / 128)
    B, H, T = 3, 2, 40
    dd = 64 * H
    qkv = rnd(B, T, 3 * dd, dtype=BF16, seed=305)
    do = rnd(B, T, dd, dtype=BF16, seed=306)
    o, lse = torch.empty(B, T, dd, dtype=BF16, device=dev()), torch.empty(B, H, T, device=dev())
    q, k, v = qkv[..., :dd], qkv[..., dd:2 * dd], qkv[..., 2 * dd:]
    ops.attention_fwd(q, k, v, o, lse, B, H, T, T, True)
    g0, g1 = torch.zeros_like(qkv), torch.zeros_like(qkv)
    for gq, od in ((g0, None), (g1, (2, key, thr, sc))):
        ops.attention_bwd(q, k, v, o, do, lse, torch.empty(B, H, T, device=dev()), gq[..., :dd], gq[..., dd:2 * dd], gq[..., 2 * dd:],
                          B, H, T, T, True, out_drop=od)
    mult = torch.stack([rng.keep_mask((key + t) & 0xFFFFFFFF, B * T, thr) for t in range(3)], 1).float().to(dev()) * sc      # (B*T, 3)
    ref = g0.float().view(B * T, 3, dd) * mult[:, :, None]
    check('attention_bwd token multipliers', g1.view(B * T, 3, dd), ref, 1e-6, 1 / 128)


def test_top2_head_chooses_the_tokens_of_the_logits_form(ops):
    """Greedy decode's lm_head in its segment-maxima form (i2t_gemm_bf16_top2 + i2t_top2_ngram_argmax) against the logits form
    (i2t_gemm_bf16 into fp32 logits + i2t_ngram_ban_argmax): the same token for every caption -- with repeated n-grams in the history
    (bans), with the best AND the second-best column of one segment banned (the exact re-evaluation path), with duplicated head rows
    (ties: the lower column wins) and with a vocabulary that ends inside a segment."""
    B, d, V, L = 300, 256, 1000 + 37, 24
    Vp = (V + 63) // 64 * 64
    hid = rnd(B, d, dtype=BF16, seed=501)
    W = (rnd(V, d, seed=502) * 0.2).to(BF16)
    W[700] = W[70]                                            # exact ties between columns 70 and 700
    g = torch.Generator().manual_seed(503)
    ids = torch.randint(0, V, (B, L + 1), generator=g).to(dev())
    ln = 17
    logits = torch.zeros(B, Vp, device=dev())
    ops.gemm(hid, W, logits, B, V, d)
    ngr = torch.tensor([2, 3], dtype=torch.int32, device=dev())
    top1 = logits[:, :V].argmax(-1)
    # rows 0..99: the un-banned winner follows an earlier copy of the current last token -> banned by the 2-gram rule
    for b in range(100):
        ids[b, 5] = ids[b, ln - 1]
        ids[b, 6] = top1[b]
    # rows 100..149: the segment's runner-up is banned too (a second earlier copy of the last token, followed by it)
    for b in range(100, 150):
        seg = int(top1[b]) // 64
        l2 = logits[b, seg * 64:min(V, seg * 64 + 64)].clone()
        l2[int(top1[b]) - seg * 64] = float('-inf')
        ids[b, 5], ids[b, 6] = ids[b, ln - 1], top1[b]
        ids[b, 9], ids[b, 10] = ids[b, ln - 1], seg * 64 + int(l2.argmax())
    lens = torch.tensor([ln], dtype=torch.int32, device=dev())
    ref_ids, got_ids = ids.clone(), ids.clone()
    ops.ngram_ban_argmax(logits, Vp, ref_ids, L + 1, lens, ngr, 2, B, V, torch.empty(B, device=dev()))
    top2 = torch.zeros(B, (V + 63) // 64, 4, device=dev())
    ops.gemm_top2(hid, W, top2, B, V, d)
    # the segments against the logits themselves
    lg = torch.full((B, Vp), float('-inf'), device=dev())
    lg[:, :V] = logits[:, :V]
    segs = lg.view(B, -1, 64)
    v, i = segs.sort(dim=-1, descending=True, stable=True)
    assert torch.equal(top2[..., 0], v[..., 0]) and torch.equal(top2[..., 2], v[..., 1])
    cols = i + (torch.arange(segs.shape[1], device=dev()) * 64)[None, :, None]
    assert torch.equal(top2[..., 1].contiguous().view(torch.int32), cols[..., 0].int())
    ops.top2_ngram_argmax(top2, hid, W, got_ids, L + 1, lens, ngr, 2, B, V, d)
    same = got_ids[:, ln] == ref_ids[:, ln]
    if not bool(same.all()):      # the re-evaluated segments sum in another order: a different token only at a (near-)tie
        bad = (~same).nonzero().flatten()
        assert bool((bad >= 100).all() and (bad < 150).all())
        gap = (logits[bad, got_ids[bad, ln]] - logits[bad, ref_ids[bad, ln]]).abs()
        assert float(gap.max()) <= 1e-4 * float(logits.abs().max())
    assert torch.equal(got_ids[:, :ln], ids[:, :ln]) and bool((ref_ids[:100, ln] != top1[:100]).all())
    assert int((got_ids[:, ln] == 700).sum()) == 0 or int((ref_ids[:, ln] == 700).sum()) > 0


def test_round4_fused_dropout_sites(ops):
    """The dropout passes folded into their neighbours in round 4, each against the pass it replaces: (a) attention_bwd's per-token
    multipliers when the queries are the FIRST rows of longer sequences (the encoder's CLS-only last block: out_drop_q_seq), (b) the
    embedding dropout applied by layernorm_nd_fwd / bcast_rows while they write their slabs of x, (c) its backward as a mask on the
    f32 dx that layernorm_bwd stores (sum of squares and bf16 copy unmasked)."""
    from image2text_amd import rng
    key, thr = rng.site_key(777, 5), rng.threshold(0.1)
    sc = rng.scale(thr)
    # (a) B x T rows of q|k|v, queries = rows [0, ncls) of every sequence
    B, H, T, ncls = 3, 4, 260, 64
    dd = 64 * H
    qkv = rnd(B, T, 3 * dd, dtype=BF16, seed=401)
    do = rnd(B, ncls, dd, dtype=BF16, seed=402)
    q, k, v = qkv[:, :ncls, :dd], qkv[..., dd:2 * dd], qkv[..., 2 * dd:]
    o, lse = torch.empty(B, ncls, dd, dtype=BF16, device=dev()), torch.empty(B, H, ncls, device=dev())
    ops.attention_fwd(q, k, v, o, lse, B, H, ncls, T, False)
    assert ops.attention_bwd_takes_q_seq(ncls, T, None)
    g0, g1 = torch.zeros_like(qkv), torch.zeros_like(qkv)
    for gq, od in ((g0, None), (g1, (2, key, thr, sc))):
        ops.attention_bwd(q, k, v, o, do, lse, torch.empty(B, H, ncls, device=dev()), gq[:, :ncls, :dd], gq[..., dd:2 * dd], gq[..., 2 * dd:],
                          B, H, ncls, T, False, out_drop=od, out_drop_q_seq=T)
    ref = g0.clone()
    ops.dropout_apply(ref, B * T, 3 * dd, (2, key, thr, sc))                  # the pass it replaces: token row b * T + t, thirds q | k | v
    check('attention_bwd multipliers, queries as the first rows', g1, ref, 1e-6, 1 / 128)
    assert torch.equal(g1 == 0, ref == 0)
    # (b) x[B, T, d] = [cls rows | LayerNormND(patch rows)] with the elementwise mask over the whole tensor
    Bx, P2, d = 5, 12, 64
    Tx = 4 + P2
    src, add, gam, bet, cls = rnd(Bx, P2, d, seed=403), rnd(P2, d, seed=404), rnd(P2, d, seed=405), rnd(P2, d, seed=406), rnd(4, d, seed=407)
    x0, x1 = torch.empty(Bx, Tx, d, device=dev()), torch.empty(Bx, Tx, d, device=dev())
    st0, st1 = torch.empty(Bx, ops.LNND_STATS_STRIDE, device=dev()), torch.empty(Bx, ops.LNND_STATS_STRIDE, device=dev())
    ops.layernorm_nd_fwd(src, add, gam, bet, x0[:, 4:], Tx * d, st0, Bx, P2, d)
    ops.bcast_rows(cls, x0, Tx * d, Bx, 4, d)
    ops.dropout_apply(x0, Bx * Tx, d, (1, key, thr, sc))
    ops.layernorm_nd_fwd(src, add, gam, bet, x1[:, 4:], Tx * d, st1, Bx, P2, d, drop=(1, key, thr, sc), drop_base=4 * d)
    ops.bcast_rows(cls, x1, Tx * d, Bx, 4, d, drop=(1, key, thr, sc))
    assert torch.equal(x0, x1)
    # (c) the mask on the stored f32 dx
    M, dl = 300, 768
    xx, gm, dy, prev = rnd(M, dl, seed=408), rnd(dl, seed=409), rnd(M, dl, seed=410), rnd(M, dl, seed=411)
    mean, rstd = xx.mean(-1), (xx.var(-1, unbiased=False) + 1e-5).rsqrt()
    d0, d1 = prev.clone(), prev.clone()
    b0, b1 = torch.empty(M, dl, dtype=BF16, device=dev()), torch.empty(M, dl, dtype=BF16, device=dev())
    s0, s1 = torch.zeros(1, device=dev()), torch.zeros(1, device=dev())
    ops.layernorm_bwd(dy, xx, gm, mean, rstd, d0, None, None, M, dl, dx_accumulate=True, dx_bf16=b0, sumsq_out=s0)
    ops.dropout_apply(d0, M, dl, (1, key, thr, sc))
    ops.layernorm_bwd(dy, xx, gm, mean, rstd, d1, None, None, M, dl, dx_accumulate=True, dx_bf16=b1, sumsq_out=s1, dx_mask=(1, key, thr, sc))
    assert torch.equal(d0, d1) and torch.equal(b0, b1)
    assert abs(float(s0) - float(s1)) <= 1e-5 * float(s0)


def test_attention_packed_varlen(ops, monkeypatch):
    """Packed variable-length self-attention (causal) and packed-query cross-attention == per-sequence dense calls (bit for bit:
    both sides on the tiled kernels -- a dense 64-row call would otherwise take the resident-operand kernels, whose summation
    order differs)."""
    monkeypatch.setenv('I2T_ATTN_V2', '0')
    H, d = 2, 128
    lens = [5, 64, 0, 37, 16]
    B, total, Tmax = len(lens), sum(lens), 64
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    qkv = rnd(total, 3 * d, dtype=BF16, seed=180)
    do = rnd(total, d, dtype=BF16, seed=181)
    o, lse = torch.zeros(total, d, dtype=BF16, device=dev()), torch.zeros(H * total, device=dev())
    ops.attention_fwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, lse, B, H, Tmax, Tmax, True, cu_q=cu, cu_k=cu, total_q=total)
    dqkv = torch.zeros(total, 3 * d, dtype=BF16, device=dev())
    ops.attention_bwd(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], o, do, lse, torch.empty(H * total, device=dev()),
                      dqkv[:, :d], dqkv[:, d:2 * d], dqkv[:, 2 * d:], B, H, Tmax, Tmax, True, cu_q=cu, cu_k=cu, total_q=total)
    mem = rnd(B, 24, 2 * d, dtype=BF16, seed=182)
    oc, lsec = torch.zeros(total, d, dtype=BF16, device=dev()), torch.zeros(H * total, device=dev())
    ops.attention_fwd(qkv[:, :d], mem[..., :d], mem[..., d:], oc, lsec, B, H, Tmax, 24, False, cu_q=cu, total_q=total)
    # (dmem starts as garbage: a sequence without query rows must still have zeros WRITTEN to its memory rows' gradient)
    dqc, dmem = torch.zeros(total, d, dtype=BF16, device=dev()), torch.full((B, 24, 2 * d), 3.0, dtype=BF16, device=dev())
    ops.attention_bwd(qkv[:, :d], mem[..., :d], mem[..., d:], oc, do, lsec, torch.empty(H * total, device=dev()), dqc,
                      dmem[..., :d], dmem[..., d:], B, H, Tmax, 24, False, cu_q=cu, total_q=total)
    for b, n in enumerate(lens):
        s0 = int(cu[b])
        if n == 0:
            assert float(dmem[b].float().abs().max()) == 0.0
            continue
        seg = qkv[s0:s0 + n].unsqueeze(0).contiguous()
        o1, l1 = torch.empty(1, n, d, dtype=BF16, device=dev()), torch.empty(1, H, n, device=dev())
        ops.attention_fwd(seg[..., :d], seg[..., d:2 * d], seg[..., 2 * d:], o1, l1, 1, H, n, n, True)
        assert torch.equal(o1[0], o[s0:s0 + n]), f'self fwd seq {b}'
        check(f'lse seq {b}', lse.view(H, total)[:, s0:s0 + n], l1[0], 1e-6, 1e-6)
        g1 = torch.zeros(1, n, 3 * d, dtype=BF16, device=dev())
        dseg = do[s0:s0 + n].unsqueeze(0).contiguous()
        ops.attention_bwd(seg[..., :d], seg[..., d:2 * d], seg[..., 2 * d:], o1, dseg, l1, torch.empty(1, H, n, device=dev()),
                          g1[..., :d], g1[..., d:2 * d], g1[..., 2 * d:], 1, H, n, n, True)
        assert torch.equal(g1[0], dqkv[s0:s0 + n]), f'self bwd seq {b}'
        qd = seg[..., :d].contiguous()
        oc1, lc1 = torch.empty(1, n, d, dtype=BF16, device=dev()), torch.empty(1, H, n, device=dev())
        ops.attention_fwd(qd, mem[b:b + 1, :, :d], mem[b:b + 1, :, d:], oc1, lc1, 1, H, n, 24, False)
        assert torch.equal(oc1[0], oc[s0:s0 + n]), f'cross fwd seq {b}'
        dq1, dm1 = torch.zeros(1, n, d, dtype=BF16, device=dev()), torch.zeros(1, 24, 2 * d, dtype=BF16, device=dev())
        ops.attention_bwd(qd, mem[b:b + 1, :, :d], mem[b:b + 1, :, d:], oc1, dseg, lc1, torch.empty(1, H, n, device=dev()), dq1,
                          dm1[..., :d], dm1[..., d:], 1, H, n, 24, False)
        assert torch.equal(dq1[0], dqc[s0:s0 + n]) and torch.equal(dm1[0], dmem[b]), f'cross bwd seq {b}'


def test_embed_packed_positions(ops):
    rows, d, V, off = 37, 128, 384, 8
    ids = torch.randint(0, V, (rows,), device=dev())
    pos = torch.randint(0, 16, (rows,), dtype=torch.int32, device=dev())
    wte, wpe = rnd(V, d, seed=190), rnd(48, d, seed=191)
    x = torch.empty(rows, d, device=dev())
    ops.embed_fwd(ids, wte, wpe, x, rows, 1, d, off, V, pos=pos)
    check('packed embed fwd', x, wte[ids] + wpe[pos.long() + off], 0, 0)
    dx = rnd(rows, d, seed=192)
    dwte, dwpe = torch.zeros(V, d, device=dev()), torch.zeros(48, d, device=dev())
    ops.embed_bwd(ids, dx, dwte, dwpe, rows, 1, d, off, V, pos=pos)
    check('packed dwte', dwte, torch.zeros(V, d, device=dev()).index_add_(0, ids, dx), 1e-5, 1e-5)
    check('packed dwpe', dwpe, torch.zeros(48, d, device=dev()).index_add_(0, pos.long() + off, dx), 1e-5, 1e-5)


@pytest.mark.parametrize('C,H,W', [(32, 20, 70), (32, 9, 224), (32, 5, 72), (16, 7, 64)])
def test_nchw_to_nhwc(ops, C, H, W):
    """W % 8 == 0 with C = 32 takes the 16-byte-vector kernel (full and ragged 64-pixel tiles), the rest the scalar one."""
    x = rnd(2, C, H, W, dtype=BF16, seed=200)
    y = torch.empty(2, H, W, C, dtype=BF16, device=dev())
    ops.nchw_to_nhwc(x, y, 2, C, H, W)
    assert torch.equal(y, x.permute(0, 2, 3, 1).contiguous())


# ------------------------------------------------------------------------------------------------ fused cross-attention
def _xattn_reference(mem, w_kv, bias, q_rows, lens, H, mask=None, sc=1.0):
    """fp64 reference on the bf16-rounded K / V the kernel stores: per image b, queries q_rows[b] (n_b x d)."""
    B, S, d = mem.shape
    kv = (mem.double().reshape(B * S, d) @ w_kv.double().t() + bias.double()).reshape(B, S, 2 * d)
    kvb = kv.to(BF16).double()
    outs, lses = [], []
    for b in range(B):
        n = lens[b]
        qh = q_rows[b].double().reshape(n, H, 64).permute(1, 0, 2)                 # (H, n, 64)
        kh = kvb[b, :, :d].reshape(S, H, 64).permute(1, 0, 2)
        vh = kvb[b, :, d:].reshape(S, H, 64).permute(1, 0, 2)
        s = qh @ kh.transpose(-1, -2) / 8.0
        lses.append(torch.logsumexp(s, -1))                                        # (H, n)
        p = torch.softmax(s, -1)
        if mask is not None:
            p = p * mask[b][:, :n].double() * sc
        outs.append((p @ vh).permute(1, 0, 2).reshape(n, d))
    return kv, outs, lses


@pytest.mark.parametrize('B,H,packed,drop', [(6, 2, True, False), (9, 12, True, True), (5, 4, False, False), (4, 2, False, True)])
def test_xattn_kv_fused(ops, B, H, packed, drop):
    """K/V projection + attention in one launch (i2t_xattn_kv_fused) against an fp64 reference: stored K/V, attention output,
    log-sum-exp; partial workgroup tiles (B % 4 != 0), ragged packed queries including an image with no query row and one with
    64, dense [B, T] queries with T > 64 (7 query blocks), probability dropout on the index space of i2t_attention_fwd -- and
    the unfused backward kernels run on the fused forward's outputs."""
    from image2text_amd import rng
    S, d = 64, 64 * H
    mem = rnd(B, S, d, dtype=BF16, seed=400)
    w_in = rnd(3 * d, d, dtype=BF16, seed=401, scale=d ** -0.5)
    b_in = rnd(3 * d, seed=402, scale=0.3)
    T = 64 if packed else 100
    if packed:
        lens = [int(x) for x in torch.randint(1, 65, (B,), generator=torch.Generator().manual_seed(7))]
        lens[0], lens[1 % B] = 64, 0
        if B > 2:
            lens[2] = 17
    else:
        lens = [T] * B
    total = sum(lens)
    cu = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32, device=dev())
    q = rnd(total, d, dtype=BF16, seed=403) if packed else rnd(B, T, d, dtype=BF16, seed=403)
    key, thr = rng.site_key(31337, 4100), rng.threshold(0.1)
    sc = rng.scale(thr)
    dr = (1, key, thr, sc) if drop else None
    kv = torch.zeros(B, S, 2 * d, dtype=BF16, device=dev())
    o = torch.full_like(q, 7.0)
    lse = torch.zeros(H * total if packed else B * H * T, device=dev())
    ops.xattn_kv_fused(mem.view(B * S, d), w_in[d:], b_in[d:], q, kv, o, lse, B, S, H, T, drop=dr,
                       cu_q=cu if packed else None, total_q=total if packed else 0)
    torch.cuda.synchronize()
    q_rows = [q[int(cu[b]):int(cu[b + 1])] for b in range(B)] if packed else [q[b] for b in range(B)]
    mask = None
    if drop:
        mask = rng.keep_mask(key, B * H * T * S, thr).view(B, H, T, S).to(dev())
    kv_ref, o_ref, lse_ref = _xattn_reference(mem, w_in[d:], b_in[d:], q_rows, lens, H, mask, sc)
    check('fused kv', kv, kv_ref, 2e-2, 1 / 128)
    for b in range(B):
        n = lens[b]
        if n == 0:
            continue
        ob = o[int(cu[b]):int(cu[b + 1])] if packed else o[b]
        check(f'fused o[{b}]', ob, o_ref[b], 1e-2, 1 / 64)
        for h in range(H):
            got = lse[h * total + int(cu[b]):h * total + int(cu[b]) + n] if packed else lse.view(B, H, T)[b, h]
            check(f'fused lse[{b},{h}]', got, lse_ref[b][h], 2e-3, 1e-3)
    # the same call through the unfused kernels (GEMM + attention_fwd): stored K/V bit-identical?  outputs within bf16 rounding
    kv2 = torch.empty_like(kv)
    ops.gemm(mem.view(B * S, d), w_in[d:], kv2.view(B * S, 2 * d), B * S, 2 * d, d, bias=b_in[d:])
    o2, lse2 = torch.zeros_like(q), torch.zeros_like(lse)
    ops.attention_fwd(q, kv2[..., :d], kv2[..., d:], o2, lse2, B, H, T, S, False, drop=dr, cu_q=cu if packed else None, total_q=total if packed else 0)
    assert float((kv.float() - kv2.float()).abs().max()) <= 2.0 ** -7 * float(kv2.float().abs().max())
    live = torch.ones(q.shape[0] if packed else B * T, dtype=torch.bool, device=dev())
    check('fused vs unfused o', o.reshape(-1, d)[live], o2.reshape(-1, d)[live], 2e-2, 1 / 32)
    # backward of the fused forward = the unfused backward kernels on its saved tensors
    do = rnd(*q.shape, dtype=BF16, seed=404)
    dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
    ops.attention_bwd(q, kv[..., :d], kv[..., d:], o, do, lse, torch.empty_like(lse), dq, dkv[..., :d], dkv[..., d:], B, H, T, S, False,
                      drop=dr, cu_q=cu if packed else None, total_q=total if packed else 0)
    dq2, dkv2 = torch.zeros_like(q), torch.zeros_like(kv)
    ops.attention_bwd(q, kv2[..., :d], kv2[..., d:], o2, do, lse2, torch.empty_like(lse), dq2, dkv2[..., :d], dkv2[..., d:], B, H, T, S, False,
                      drop=dr, cu_q=cu if packed else None, total_q=total if packed else 0)
    check('bwd dq on fused outputs', dq, dq2, 3e-2 * float(do.float().abs().max()), 1 / 16)
    check('bwd dkv on fused outputs', dkv, dkv2, 3e-2 * float(do.float().abs().max()), 1 / 16)


@pytest.mark.parametrize('M,N,K,drop', [(70001, 1160, 512, False), (33280, 1536, 512, True), (12345, 768, 768, False), (8200, 2304, 384, True)])
def test_gemm3_overlapped_epilogue_is_bit_equal(ops, monkeypatch, M, N, K, drop):
    """gemm3 (256 x 128 tiles, two accumulator sets: a tile's epilogue runs inside the next tile's K loop) against the 256^2
    kernel on class-1 problems (bf16 C, bias, optional per-(row, third) dropout multipliers): same K order per output, so the
    results must be BIT-equal -- in both of its modes, on ragged M / N edges, and across many tiles per workgroup."""
    from image2text_amd import rng
    a, w = rnd(M, K, dtype=BF16, seed=500, scale=0.5), rnd(N, K, dtype=BF16, seed=501, scale=0.05)
    bias = rnd(N, seed=502)
    dr = (2, rng.site_key(9, 9), rng.threshold(0.1), rng.scale(rng.threshold(0.1))) if drop else None
    outs = {}
    monkeypatch.setenv('I2T_G256_MIN_TILES', '1')
    for mode in ('0', '1', '2'):
        monkeypatch.setenv('I2T_GEMM3', mode)
        c = torch.full((M, N), 7.0, dtype=BF16, device=dev())
        ops.gemm(a, w, c, M, N, K, bias=bias, drop=dr)
        torch.cuda.synchronize()
        outs[mode] = c
    ref = a[:512].float() @ w.float().t() + bias
    if not drop:
        check('gemm3 vs fp32', outs['2'][:512], ref, 2e-2, 1 / 128)
    assert torch.equal(outs['1'], outs['0']), f'mode 1: {(outs["1"] != outs["0"]).sum().item()} elements differ'
    assert torch.equal(outs['2'], outs['0']), f'mode 2: {(outs["2"] != outs["0"]).sum().item()} elements differ'
