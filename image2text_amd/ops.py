"""Tensor-level wrappers over the C ABI: pointer/stride marshalling only, one function per entry point.

PyTorch supplies device memory and the current HIP stream; every op below launches hand-written gfx950 kernels
through ``libi2t_hip.so`` and raises ``I2TError`` on any failure (no fallback).
"""
import os
import weakref
from typing import Optional

import torch

from . import lib as _l

BF16 = torch.bfloat16
F32 = torch.float32


def _lib():
    return _l.load()


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _need_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise _l.I2TError('image2text_amd ops need tensors on the MI355X (cuda) device; there is no CPU path')


def _drop(drop):
    """drop = None | (mode, key, thr, scale) -> the four trailing dropout arguments of the C ABI."""
    return (0, 0, 0, 1.0) if drop is None else (int(drop[0]), int(drop[1]), int(drop[2]), float(drop[3]))


# ------------------------------------------------------------------------------------------------------------------------------
# I2T_PRECISE=1 -- the opt-in PARITY mode of the teacher-forced forward (never part of a measured step).  Every
# bf16 GEMM operand this module produces from fp32 data (LayerNorm rows, fp32 -> bf16 casts, the parameter shadow, activated GEMM
# outputs) is written as TWO bf16 terms, hi = bf16(x) and lo = bf16(x - hi); ``lo`` rides on the hi tensor object (``_i2t_lo``), so a
# tensor that lost it (a view, an attention output, the conv stack's bf16 patches) simply counts as exact: never wrong, only one
# term short.  gemm() then runs hi.hi + lo.hi + hi.lo through the SAME MFMA kernels (the second and third product through the
# accumulate class) into fp32, and applies bias / GELU / the bf16 rounding of its output afterwards.  Attention keeps bf16 q, k, v, P
# and the conv stack its bf16 operands: the oracle with exactly those two roundings left deviates by 5e-3 on the nano-224 logits
# (tools/diag_precision.py), the shipped one-term pipeline by 1.7e-2.
# ------------------------------------------------------------------------------------------------------------------------------
PRECISE_CALLS = {'gemm': 0, 'products': 0, 'splits': 0}
_SHADOWS = []            # weak references to bf16 parameter shadows that carry a low-order term (engine.ParamArena.refresh_shadow)


def precise() -> bool:
    return os.environ.get('I2T_PRECISE', '0') not in ('', '0')


def split_f32(src: torch.Tensor, hi: torch.Tensor, lo: Optional[torch.Tensor], n: Optional[int] = None, act: int = 0):
    """hi = bf16(f(src)), lo = bf16(f(src) - hi) over the first n elements (include/i2t.h::i2t_split_f32_bf16)"""
    _need_cuda(src, hi, lo)
    assert src.dtype == F32 and hi.dtype == BF16 and (lo is None or lo.dtype == BF16)
    _l.check(_lib().i2t_split_f32_bf16(_stream(), _p(src), _p(hi), _p(lo), src.numel() if n is None else n, int(act)), 'i2t_split_f32_bf16')
    PRECISE_CALLS['splits'] += 1
    return hi


def register_shadow(shadow: torch.Tensor):
    """the bf16 parameter shadow (whose ``_i2t_lo`` a precise cast just wrote): weight VIEWS of it find their low-order term by address"""
    _SHADOWS[:] = [r for r in _SHADOWS if r() is not None and r() is not shadow]
    _SHADOWS.append(weakref.ref(shadow))


def _lo_of(t: torch.Tensor):
    lo = getattr(t, '_i2t_lo', None)
    if lo is not None:
        return lo
    for r in _SHADOWS:
        sh = r()
        if sh is None or getattr(sh, '_i2t_lo', None) is None:
            continue
        off = t.data_ptr() - sh.data_ptr()
        if 0 <= off < 2 * sh.numel():
            return torch.as_strided(sh._i2t_lo, t.size(), t.stride(), off // 2)
    return None


def _gemm_precise(a, b, out, M, N, K, *, a_kmajor, b_kmajor, lda, ldb, ldc, alpha, bias, act, aux_in, aux_out, residual, ldr, accumulate,
                  drop, alpha_sumsq):
    if drop is not None or aux_in is not None or aux_out is not None or alpha_sumsq is not None or act not in (0, ACT_GELU, ACT_GELU_ERF):
        raise _l.I2TError('I2T_PRECISE=1 covers inference forwards only (no dropout, no saved pre-activations, no backward epilogues)')
    direct = out.dtype == F32 and act == 0
    if not direct and (residual is not None or accumulate or not out.is_contiguous() or (ldc is not None and ldc != out.stride(0))):
        raise _l.I2TError('I2T_PRECISE=1: a bf16 / activated GEMM output must be a plain contiguous buffer')
    acc = out if direct else torch.zeros(M, out.shape[-1], dtype=F32, device=out.device)          # (pad columns stay zero)
    kw = dict(a_kmajor=a_kmajor, b_kmajor=b_kmajor, alpha=alpha)
    ldacc = ldc if direct else None
    _gemm(a, b, acc, M, N, K, lda=lda, ldb=ldb, ldc=ldacc, bias=bias, residual=residual, ldr=ldr, accumulate=accumulate, **kw)
    n = 1
    a_lo, b_lo = _lo_of(a), _lo_of(b)
    if a_lo is not None:
        _gemm(a_lo, b, acc, M, N, K, lda=lda, ldb=ldb, ldc=ldacc, accumulate=True, **kw)
        n += 1
    if b_lo is not None:
        _gemm(a, b_lo, acc, M, N, K, lda=lda, ldb=ldb, ldc=ldacc, accumulate=True, **kw)
        n += 1
    PRECISE_CALLS['gemm'] += 1
    PRECISE_CALLS['products'] += n
    if not direct:
        if out.dtype == F32:
            raise _l.I2TError('I2T_PRECISE=1: an activated GEMM writes bf16 operands')
        lo = torch.empty_like(out)
        split_f32(acc, out, lo, act=act)
        out._i2t_lo = lo
    return out


def gemm(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, M: int, N: int, K: int, *, a_kmajor=False, b_kmajor=False,
         lda=None, ldb=None, ldc=None, alpha=1.0, bias=None, act=0, aux_in=None, aux_out=None, residual=None,
         ldr=None, accumulate=False, drop=None, workspace=None, alpha_sumsq=None):
    """out[M,N] = epilogue(alpha * op(a) . op(b)); see include/i2t.h::i2t_gemm_bf16.  workspace (fp32, decode steps): the
    deterministic split-K form i2t_gemm_bf16_ws when the problem has few tiles and a long K.  alpha_sumsq (1-float device tensor):
    alpha is further divided by sqrt(alpha_sumsq) + 1e-6 on the device (i2t_gemm_bf16_ex: a folded gradient normaliser).
    Under I2T_PRECISE=1: the three-product form of the parity mode (above)."""
    if precise():
        return _gemm_precise(a, b, out, M, N, K, a_kmajor=a_kmajor, b_kmajor=b_kmajor, lda=lda, ldb=ldb, ldc=ldc, alpha=alpha, bias=bias,
                             act=act, aux_in=aux_in, aux_out=aux_out, residual=residual, ldr=ldr, accumulate=accumulate, drop=drop,
                             alpha_sumsq=alpha_sumsq)
    return _gemm(a, b, out, M, N, K, a_kmajor=a_kmajor, b_kmajor=b_kmajor, lda=lda, ldb=ldb, ldc=ldc, alpha=alpha, bias=bias, act=act,
                 aux_in=aux_in, aux_out=aux_out, residual=residual, ldr=ldr, accumulate=accumulate, drop=drop, workspace=workspace,
                 alpha_sumsq=alpha_sumsq)


def _gemm(a, b, out, M, N, K, *, a_kmajor=False, b_kmajor=False, lda=None, ldb=None, ldc=None, alpha=1.0, bias=None, act=0, aux_in=None,
          aux_out=None, residual=None, ldr=None, accumulate=False, drop=None, workspace=None, alpha_sumsq=None):
    _need_cuda(a, b, out)
    assert a.dtype == BF16 and b.dtype == BF16 and out.dtype in (BF16, F32)
    lda = a.stride(0) if lda is None else lda
    ldb = b.stride(0) if ldb is None else ldb
    ldc = out.stride(0) if ldc is None else ldc
    if workspace is not None and not (a_kmajor or b_kmajor or accumulate or drop is not None or aux_in is not None
                                      or aux_out is not None or alpha != 1.0 or act not in (0, 1)):
        ldr_ = (residual.stride(0) if residual is not None else 0) if ldr is None else ldr
        _l.check(_lib().i2t_gemm_bf16_ws(_stream(), _p(a), lda, _p(b), ldb, _p(out), ldc, int(out.dtype == F32), M, N, K, _p(bias), int(act),
                                         _p(residual), ldr_, _p(workspace), workspace.numel()), 'i2t_gemm_bf16_ws')
        return out
    ld_ai = aux_in.stride(0) if aux_in is not None else 0
    ld_ao = aux_out.stride(0) if aux_out is not None else 0
    ldr = (residual.stride(0) if residual is not None else 0) if ldr is None else ldr
    if alpha_sumsq is not None:
        _l.check(_lib().i2t_gemm_bf16_ex(_stream(), _p(a), lda, int(a_kmajor), _p(b), ldb, int(b_kmajor), _p(out), ldc,
                                         int(out.dtype == F32), M, N, K, float(alpha), _p(bias), int(act), _p(aux_in), ld_ai,
                                         _p(aux_out), ld_ao, _p(residual), ldr, int(accumulate), *_drop(drop), _p(alpha_sumsq)), 'i2t_gemm_bf16_ex')
        return out
    _l.check(_lib().i2t_gemm_bf16(_stream(), _p(a), lda, int(a_kmajor), _p(b), ldb, int(b_kmajor), _p(out), ldc,
                                  int(out.dtype == F32), M, N, K, float(alpha), _p(bias), int(act), _p(aux_in), ld_ai,
                                  _p(aux_out), ld_ao, _p(residual), ldr, int(accumulate), *_drop(drop)), 'i2t_gemm_bf16')
    return out


def set_deterministic(on: bool):
    """Fixed-order reductions on the gradient path (include/i2t.h::i2t_set_deterministic): slow, bit-reproducible backward passes."""
    _l.check(_lib().i2t_set_deterministic(int(bool(on))), 'i2t_set_deterministic')


def deterministic() -> bool:
    return bool(_lib().i2t_deterministic())


def gemm_reserve_cus(n: int):
    """Leave n CUs free in later persistent-GEMM launches (0 = use all); see include/i2t.h::i2t_gemm_reserve_cus."""
    _l.check(_lib().i2t_gemm_reserve_cus(int(n)), 'i2t_gemm_reserve_cus')


def gemm_reserved_cus() -> int:
    return int(_lib().i2t_gemm_reserved_cus())


def colsum(x: torch.Tensor, out: torch.Tensor, M: int, N: int, ld=None, accumulate=False, alpha_sumsq=None):
    _need_cuda(x, out)
    _l.check(_lib().i2t_colsum_bf16_ex(_stream(), _p(x), x.stride(0) if ld is None else ld, M, N, _p(out), int(accumulate), _p(alpha_sumsq)),
             'i2t_colsum_bf16_ex')
    return out


def layernorm_fwd(x, gamma, beta, y, mean, rstd, M, d, eps=None):
    """eps None: the reference's LayerNorm (1e-5); torchvision's ViT blocks pass 1e-6."""
    _need_cuda(x, y)
    if y.dtype == BF16 and precise():          # parity mode: the row in fp32, then its two bf16 terms
        y32 = torch.empty(y.shape, dtype=F32, device=y.device)
        layernorm_fwd(x, gamma, beta, y32, mean, rstd, M, d, eps)
        y._i2t_lo = torch.empty_like(y)
        return split_f32(y32, y, y._i2t_lo)
    if eps is not None:
        _l.check(_lib().i2t_layernorm_fwd_eps(_stream(), _p(x), _p(gamma), _p(beta), _p(y), int(y.dtype == F32), _p(mean), _p(rstd),
                                              M, d, float(eps)), 'i2t_layernorm_fwd_eps')
        return y
    _l.check(_lib().i2t_layernorm_fwd(_stream(), _p(x), _p(gamma), _p(beta), _p(y), int(y.dtype == F32), _p(mean), _p(rstd),
                                      M, d), 'i2t_layernorm_fwd')
    return y


def layernorm_bwd(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, M, d, dx_accumulate=False, dx_bf16=None, bf16_drop=None,
                  sumsq_out=None, dx_pre_sumsq=None, dx_mask=None, acc_period=0, acc_rows=0):
    """bf16_drop = (1, key, thr, scale): elementwise dropout mask applied to the bf16 copy only; dx_mask = (1, key, thr, scale): the
    same kind of mask on the f32 dx this call stores; acc_period / acc_rows: accumulate onto the first acc_rows rows of every period only
    (see include/i2t.h::i2t_layernorm_bwd_ex)."""
    _need_cuda(dy, x, dx)
    assert (bf16_drop is None or int(bf16_drop[0]) == 1) and (dx_mask is None or int(dx_mask[0]) == 1)
    _l.check(_lib().i2t_layernorm_bwd_ex(_stream(), _p(dy), int(dy.dtype == F32), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dx),
                                         int(dx_accumulate), _p(dx_bf16), _p(dgamma), _p(dbeta), M, d, *_drop(bf16_drop)[1:],
                                         _p(sumsq_out), _p(dx_pre_sumsq), *_drop(dx_mask)[1:], int(acc_period), int(acc_rows)), 'i2t_layernorm_bwd_ex')
    return dx


LNND_STATS_STRIDE = 34


def layernorm_nd_fwd(x, add, gamma, beta, y, y_batch_stride, stats, B, rows, d, drop=None, drop_base=0):
    """drop = (1, key, thr, scale) + drop_base: the elementwise dropout of the tensor y is a slab of, applied while writing
    (include/i2t.h::i2t_layernorm_nd_fwd_drop)"""
    _need_cuda(x, y, stats)
    assert drop is None or int(drop[0]) == 1
    _l.check(_lib().i2t_layernorm_nd_fwd_drop(_stream(), _p(x), _p(add), _p(gamma), _p(beta), _p(y), y_batch_stride, _p(stats), B,
                                              rows, d, *_drop(drop)[1:], int(drop_base)), 'i2t_layernorm_nd_fwd_drop')
    return y


def layernorm_nd_bwd(dy, dy_batch_stride, x, add, gamma, stats, dx, dgamma, dbeta, dadd, B, rows, d):
    _need_cuda(dy, x, dx)
    _l.check(_lib().i2t_layernorm_nd_bwd(_stream(), _p(dy), dy_batch_stride, _p(x), _p(add), _p(gamma), _p(stats), _p(dx),
                                         _p(dgamma), _p(dbeta), _p(dadd), B, rows, d), 'i2t_layernorm_nd_bwd')
    return dx


def _bs_rs(t: torch.Tensor):
    """(batch stride, row stride) in elements of a [B, T, *] view whose last dim is contiguous; a packed [rows, *] view
    has no batch stride (0)."""
    assert t.stride(-1) == 1
    return (0, t.stride(0)) if t.dim() == 2 else (t.stride(0), t.stride(1))


def attention_fwd(q, k, v, o, lse, B, H, Tq, Tk, causal, drop=None, cu_q=None, cu_k=None, total_q=0):
    """q,k,v,o: bf16 [B, T, >=64H] views (last dim contiguous; heads at 64-column steps), or packed [rows, >=64H] views
    together with cu_q / cu_k (device int32 [B+1])."""
    _need_cuda(q, k, v, o)
    qb, qr = _bs_rs(q); kb, kr = _bs_rs(k); vb, vr = _bs_rs(v); ob, orr = _bs_rs(o)
    _l.check(_lib().i2t_attention_fwd(_stream(), _p(q), qb, qr, _p(k), kb, kr, _p(v), vb, vr, _p(o), ob, orr, _p(lse), B, H, Tq,
                                      Tk, int(causal), *_drop(drop)[1:], _p(cu_q), _p(cu_k), int(total_q)), 'i2t_attention_fwd')
    return o


def attention_bwd(q, k, v, o, do, lse, delta_ws, dq, dk, dv, B, H, Tq, Tk, causal, drop=None, cu_q=None, cu_k=None, total_q=0,
                  out_drop=None, out_drop_q_seq=0):
    """out_drop = (2, key, thr, scale): the forward's per-token q/k/v multipliers applied to dq/dk/dv on the way out;
    out_drop_q_seq: the queries are the first Tq rows of sequences of that many rows (i2t_attention_bwd_ex)."""
    _need_cuda(q, k, v, o, do, dq, dk, dv)
    qb, qr = _bs_rs(q); kb, kr = _bs_rs(k); vb, vr = _bs_rs(v); ob, orr = _bs_rs(o); gb, gr = _bs_rs(do)
    dqb, dqr = _bs_rs(dq); dkb, dkr = _bs_rs(dk); dvb, dvr = _bs_rs(dv)
    _l.check(_lib().i2t_attention_bwd_ex(_stream(), _p(q), qb, qr, _p(k), kb, kr, _p(v), vb, vr, _p(o), ob, orr, _p(do), gb, gr,
                                         _p(lse), _p(delta_ws), _p(dq), dqb, dqr, _p(dk), dkb, dkr, _p(dv), dvb, dvr, B, H, Tq, Tk,
                                         int(causal), *_drop(drop)[1:], _p(cu_q), _p(cu_k), int(total_q), *_drop(out_drop)[1:],
                                         int(out_drop_q_seq)), 'i2t_attention_bwd_ex')


def attention_bwd_takes_q_seq(Tq: int, Tk: int, drop) -> bool:
    """whether a dense, non-causal attention_bwd of these sizes runs on the one-pass resident-operand kernel, the only one that takes
    out_drop_q_seq (the rule of csrc/attention.hip::v2_applies + the default I2T_ATTN_BWD mode; the C side refuses loudly otherwise)"""
    if os.environ.get('I2T_ATTN_V2', '1')[:1] == '0' or os.environ.get('I2T_ATTN_BWD2', '1')[:1] == '0' or os.environ.get('I2T_ATTN_BWD', '3') != '3':
        return False
    return 64 <= Tq <= 288 and Tk <= 288 and (drop is None or Tk % 4 == 0)


def gq_attention_fwd(q, k, v, o, lse, B, H, Hkv, hd, Tq, Tk, causal, drop=None, cu_q=None, cu_k=None, total_q=0, split=0):
    """Grouped-query attention (include/i2t.h::i2t_gq_attention_fwd): H query heads of width hd on Hkv shared key/value heads."""
    _need_cuda(q, k, v, o, lse)
    qb, qr = _bs_rs(q); kb, kr = _bs_rs(k); vb, vr = _bs_rs(v); ob, orr = _bs_rs(o)
    _l.check(_lib().i2t_gq_attention_fwd(_stream(), _p(q), qb, qr, _p(k), kb, kr, _p(v), vb, vr, _p(o), ob, orr, _p(lse), B, H, Hkv, hd,
                                         Tq, Tk, int(causal), *_drop(drop)[1:], _p(cu_q), _p(cu_k), int(total_q), int(split)), 'i2t_gq_attention_fwd')
    return o


def gq_attention_bwd(q, k, v, o, do, lse, delta_ws, dq, dk, dv, B, H, Hkv, hd, Tq, Tk, causal, drop=None, cu_q=None, cu_k=None,
                     total_q=0, out_drop=None):
    _need_cuda(q, k, v, o, do, dq, dk, dv)
    qb, qr = _bs_rs(q); kb, kr = _bs_rs(k); vb, vr = _bs_rs(v); ob, orr = _bs_rs(o); gb, gr = _bs_rs(do)
    dqb, dqr = _bs_rs(dq); dkb, dkr = _bs_rs(dk); dvb, dvr = _bs_rs(dv)
    _l.check(_lib().i2t_gq_attention_bwd(_stream(), _p(q), qb, qr, _p(k), kb, kr, _p(v), vb, vr, _p(o), ob, orr, _p(do), gb, gr,
                                         _p(lse), _p(delta_ws), _p(dq), dqb, dqr, _p(dk), dkb, dkr, _p(dv), dvb, dvr, B, H, Hkv, hd, Tq,
                                         Tk, int(causal), *_drop(drop)[1:], _p(cu_q), _p(cu_k), int(total_q), *_drop(out_drop)[1:]),
             'i2t_gq_attention_bwd')


def row_sections_dropout(x, rows, cols, section, drop, key_offset=0):
    """drop = (2, key, thr, scale): x[m, n] *= multiplier(key + key_offset + n // section, m)  (bf16 [rows, ld], in place)."""
    if drop is None:
        return x
    _need_cuda(x)
    assert x.dtype == BF16 and x.stride(-1) == 1
    _l.check(_lib().i2t_row_sections_dropout(_stream(), _p(x), x.stride(-2), rows, cols, section, (int(drop[1]) + key_offset) & 0xFFFFFFFF,
                                             int(drop[2]), float(drop[3])), 'i2t_row_sections_dropout')
    return x


def gather_rows(src, idx, n, d, out_f32=None, out_bf16=None):
    _need_cuda(src, idx, out_f32, out_bf16)
    assert src.dtype == F32 and idx.dtype == torch.int32
    _l.check(_lib().i2t_gather_rows(_stream(), _p(src), _p(idx), _p(out_f32), _p(out_bf16), n, d), 'i2t_gather_rows')


def scatter_rows(src, idx, dst, n, d):
    _need_cuda(src, idx, dst)
    assert src.dtype == F32 and dst.dtype == F32 and idx.dtype == torch.int32
    _l.check(_lib().i2t_scatter_rows(_stream(), _p(src), _p(idx), _p(dst), n, d), 'i2t_scatter_rows')


def moe_gate_fwd(U, wg2, bg2, A, gates, wsel, M, E, P, G, top_k, inv_sqrt_in):
    _need_cuda(U, A, gates, wsel)
    assert U.dtype == F32 and A.dtype == BF16
    _l.check(_lib().i2t_moe_gate_fwd(_stream(), _p(U), U.stride(0), _p(wg2), _p(bg2), _p(A), A.stride(0), _p(gates), _p(wsel), M, E, P, G,
                                     top_k, float(inv_sqrt_in)), 'i2t_moe_gate_fwd')


def moe_gate_bwd_blocks(M: int) -> int:
    return _lib().i2t_moe_gate_bwd_blocks(M)


def moe_gate_bwd(dA, U, gates, wsel, wg2, D1, dwg2, dbg2, part_ws, M, E, P, G, top_k, inv_sqrt_in):
    _need_cuda(dA, U, D1)
    assert dA.dtype == BF16 and D1.dtype == BF16 and U.dtype == F32
    _l.check(_lib().i2t_moe_gate_bwd(_stream(), _p(dA), dA.stride(0), _p(U), U.stride(0), _p(gates), _p(wsel), _p(wg2), _p(D1), D1.stride(0),
                                     _p(dwg2), _p(dbg2), _p(part_ws), M, E, P, G, top_k, float(inv_sqrt_in)), 'i2t_moe_gate_bwd')


def moe_pack_w2(l2w, l2b, W, out, E, P):
    _need_cuda(l2w, l2b, W)
    assert l2w.dtype == BF16 and l2b.dtype == F32 and W.dtype == BF16
    _l.check(_lib().i2t_moe_pack_w2(_stream(), _p(l2w), _p(l2b), _p(W), out, E, P, W.stride(0)), 'i2t_moe_pack_w2')


def moe_unpack_dw2(dW, gw, gb, out, E, P):
    _need_cuda(dW, gw, gb)
    assert dW.dtype == F32 and gw.dtype == F32
    _l.check(_lib().i2t_moe_unpack_dw2(_stream(), _p(dW), _p(gw), _p(gb), out, E, P, dW.stride(0)), 'i2t_moe_unpack_dw2')


def gq_decode_attention(q, k_new, v_new, kcache, vcache, cache_bs, cache_rs, out, pos_ptr, n_keys_fixed, max_keys, B, H, Hkv, hd):
    """Decode-step attention of the family (include/i2t.h::i2t_gq_decode_attention); q / k_new / v_new / out are row-major 2-D views."""
    _need_cuda(q, kcache, vcache, out)
    kv_rs = k_new.stride(0) if k_new is not None else 0
    _l.check(_lib().i2t_gq_decode_attention(_stream(), _p(q), q.stride(0), _p(k_new), _p(v_new), kv_rs, _p(kcache), _p(vcache), cache_bs,
                                            cache_rs, _p(out), out.stride(0), _p(pos_ptr), n_keys_fixed, max_keys, B, H, Hkv, hd),
             'i2t_gq_decode_attention')
    return out


def sparse_step_setup(pos_ptr, rank, member, lpos, lmem, L, tmax):
    _need_cuda(pos_ptr, rank, member, lpos, lmem)
    _l.check(_lib().i2t_sparse_step_setup(_stream(), _p(pos_ptr), _p(rank), _p(member), _p(lpos), _p(lmem), L, tmax), 'i2t_sparse_step_setup')


def select_rows(flag, a, b, out, n):
    _need_cuda(flag, a, b, out)
    _l.check(_lib().i2t_select_rows(_stream(), _p(flag), _p(a), _p(b), _p(out), n), 'i2t_select_rows')


def grouped_gemm(mode, a, b, c, N, K, *, b_group_stride=0, c_group_stride=0, bias=None, bias_group_stride=0, act=0, aux_in=None,
                 aux_out=None, residual=None, accumulate=False, seg=None, n_groups=1, max_rows=0, group_ptr=None, group0=0):
    """One GEMM per group (position) in one launch; see include/i2t.h::i2t_grouped_gemm for the three modes."""
    _need_cuda(a, b, c)
    assert a.dtype == BF16 and b.dtype == BF16 and c.dtype in (BF16, F32)
    aux = aux_in if aux_in is not None else aux_out
    _l.check(_lib().i2t_grouped_gemm(_stream(), mode, _p(a), a.stride(0), _p(b), b.stride(0), b_group_stride, _p(c), c.stride(0),
                                     c_group_stride, int(c.dtype == F32), _p(bias), bias_group_stride, int(act), _p(aux_in), _p(aux_out),
                                     aux.stride(0) if aux is not None else 0, _p(residual), residual.stride(0) if residual is not None else 0,
                                     int(accumulate), _p(seg), n_groups, max_rows, _p(group_ptr), group0, N, K), 'i2t_grouped_gemm')
    return c


def grouped_colsum(x, seg, n_groups, out, out_group_stride, group0, N):
    _need_cuda(x, seg, out)
    _l.check(_lib().i2t_grouped_colsum(_stream(), _p(x), x.stride(0), _p(seg), n_groups, _p(out), out_group_stride, group0, N),
             'i2t_grouped_colsum')


def xattn_kv_fused(mem, w_kv, bias_kv, q, kv, o, lse, B, S, H, Tq, drop=None, cu_q=None, total_q=0):
    """Fused cross-attention forward (include/i2t.h::i2t_xattn_kv_fused): kv = mem . w_kv^T + bias written once, attention of every
    (image, head) out of the projection's accumulators.  q / o: [B, Tq, >= 64 H] views or packed [rows, >= 64 H] with cu_q."""
    _need_cuda(mem, w_kv, q, kv, o)
    qb, qr = _bs_rs(q); ob, orr = _bs_rs(o)
    assert mem.dtype == BF16 and w_kv.dtype == BF16 and q.dtype == BF16 and kv.dtype == BF16 and o.dtype == BF16
    _l.check(_lib().i2t_xattn_kv_fused(_stream(), _p(mem), mem.stride(0), _p(w_kv), w_kv.stride(0), _p(bias_kv), _p(q), qb, qr,
                                       _p(cu_q), int(total_q), _p(kv), kv.stride(-2), _p(o), ob, orr, _p(lse), B, S, H, Tq,
                                       *_drop(drop)[1:]), 'i2t_xattn_kv_fused')
    return o


def embed_fwd(ids, wte, wpe, x, B, T, d, pos_offset, vocab, pos=None):
    _need_cuda(ids, wte, x)
    _l.check(_lib().i2t_embed_fwd(_stream(), _p(ids), _p(wte), _p(wpe), _p(x), B, T, d, pos_offset, vocab, _p(pos)), 'i2t_embed_fwd')
    return x


def embed_bwd(ids, dx, dwte, dwpe, B, T, d, pos_offset, vocab, pos=None):
    _need_cuda(ids, dx)
    _l.check(_lib().i2t_embed_bwd(_stream(), _p(ids), _p(dx), _p(dwte), _p(dwpe), B, T, d, pos_offset, vocab, _p(pos)), 'i2t_embed_bwd')


def ce_fwd(logits, ld, labels, w, inv_temp, ignore_index, lse, loss, M, V):
    _need_cuda(logits, labels, w, lse, loss)
    _l.check(_lib().i2t_ce_fwd(_stream(), _p(logits), ld, _p(labels), _p(w), float(inv_temp), int(ignore_index), _p(lse),
                               _p(loss), M, V), 'i2t_ce_fwd')


def ce_bwd(logits, ld, labels, w, inv_temp, ignore_index, lse, gscale, M, V):
    _need_cuda(logits, labels, w, lse, gscale)
    _l.check(_lib().i2t_ce_bwd(_stream(), _p(logits), ld, _p(labels), _p(w), float(inv_temp), int(ignore_index), _p(lse),
                               _p(gscale), M, V), 'i2t_ce_bwd')


def ce_fwd_bwd(logits, ld, labels, w, inv_temp, ignore_index, lse, loss, M, V):
    """One pass: lse / loss as ce_fwd, and the rows overwritten with the un-scaled gradient (include/i2t.h::i2t_ce_fwd_bwd)."""
    _need_cuda(logits, labels, w, lse, loss)
    _l.check(_lib().i2t_ce_fwd_bwd(_stream(), _p(logits), ld, _p(labels), _p(w), float(inv_temp), int(ignore_index), _p(lse),
                                   _p(loss), M, V), 'i2t_ce_fwd_bwd')


CE_ONE_PASS_MAX_V = 8 * 16 * 512


def scale_bf16(x, n, scale):
    """x[0 .. n) *= scale[0] in place (bf16 x, device scalar); a no-op launch when the scale is exactly 1."""
    _need_cuda(x, scale)
    assert x.dtype == BF16 and scale.dtype == F32
    _l.check(_lib().i2t_scale_bf16(_stream(), _p(x), int(n), _p(scale)), 'i2t_scale_bf16')


def ce_distill_fwd(logits, ld, teacher, ld_t, alpha, labels, w, inv_temp, ignore_index, lse, lse_t, loss, M, V):
    _need_cuda(logits, teacher, labels, w, lse, lse_t, loss)
    _l.check(_lib().i2t_ce_distill_fwd(_stream(), _p(logits), ld, _p(teacher), ld_t, float(alpha), _p(labels), _p(w), float(inv_temp),
                                       int(ignore_index), _p(lse), _p(lse_t), _p(loss), M, V), 'i2t_ce_distill_fwd')


def ce_distill_bwd(logits, ld, teacher, ld_t, alpha, labels, w, inv_temp, ignore_index, lse, lse_t, gscale, M, V):
    _need_cuda(logits, teacher, labels, w, lse, lse_t, gscale)
    _l.check(_lib().i2t_ce_distill_bwd(_stream(), _p(logits), ld, _p(teacher), ld_t, float(alpha), _p(labels), _p(w), float(inv_temp),
                                       int(ignore_index), _p(lse), _p(lse_t), _p(gscale), M, V), 'i2t_ce_distill_bwd')


def ema_update(pm, p, pm_bf16, n, momentum):
    _need_cuda(pm, p)
    _l.check(_lib().i2t_ema_update(_stream(), _p(pm), _p(p), _p(pm_bf16), int(n), float(momentum)), 'i2t_ema_update')


def lm_inputs(labels, ids, B, L, bos, eos, mask_id, vocab, ignore_index, mask_fraction=0.0, random_fraction=0.0, seed=0):
    """ids = [BOS, labels[:-1]] with ignored labels -> EOS and the optional MLM corruption (include/i2t.h::i2t_lm_inputs)."""
    _need_cuda(labels, ids)
    assert labels.dtype == torch.long and ids.dtype == torch.long and labels.is_contiguous() and ids.is_contiguous()
    _l.check(_lib().i2t_lm_inputs(_stream(), _p(labels), _p(ids), B, L, int(bos), int(eos), int(-1 if mask_id is None else mask_id), int(vocab),
                                  int(ignore_index), float(mask_fraction), float(random_fraction), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF),
             'i2t_lm_inputs')
    return ids


def grad_normalize(g: torch.Tensor, ws: torch.Tensor, g_bf16=None, bf16_drop=None, presummed=False, clear_after=None,
                   keep_f32=False):
    """presummed: ws already holds sum(g^2) (layernorm_bwd's sumsq_out); clear_after: 1-float tensor zeroed afterwards;
    keep_f32: g stays un-normalised, only the bf16 copy is written (the first layernorm_bwd onto g passes dx_pre_sumsq=ws)."""
    _need_cuda(g, ws)
    assert bf16_drop is None or int(bf16_drop[0]) == 1
    _l.check(_lib().i2t_grad_normalize(_stream(), _p(g), g.numel(), _p(ws), _p(g_bf16), *_drop(bf16_drop)[1:],
                                       int(bool(presummed)) | (2 if keep_f32 else 0), _p(clear_after)), 'i2t_grad_normalize')
    return g


def rmsnorm_fwd(x, w, y, rstd, M, d, eps, y_f32=None):
    _need_cuda(x, w)
    _l.check(_lib().i2t_rmsnorm_fwd(_stream(), _p(x), _p(w), _p(y), _p(y_f32), _p(rstd), M, d, float(eps)), 'i2t_rmsnorm_fwd')
    return y


def rmsnorm_bwd(dy, x, w, rstd, dx, dw, M, d, dx_accumulate=False, dx_bf16=None):
    _need_cuda(dy, x, w, rstd, dx)
    _l.check(_lib().i2t_rmsnorm_bwd(_stream(), _p(dy), int(dy.dtype == F32), _p(x), _p(w), _p(rstd), _p(dx), int(dx_accumulate),
                                    _p(dx_bf16), _p(dw), M, d), 'i2t_rmsnorm_bwd')
    return dx


def rope(x, rs, col0, n_heads, hd, cos_sin, M, pos=None, pos_ptr=None, pos_offset=0, T=0, inverse=False):
    """in place on bf16 rows (include/i2t.h::i2t_rope)"""
    _need_cuda(x, cos_sin)
    _l.check(_lib().i2t_rope(_stream(), _p(x), rs, col0, n_heads, hd, _p(cos_sin), cos_sin.shape[0], _p(pos), _p(pos_ptr), pos_offset,
                             T, M, int(inverse)), 'i2t_rope')
    return x


def swiglu_fwd(gate_up, h, M, ff):
    _need_cuda(gate_up, h)
    _l.check(_lib().i2t_swiglu_fwd(_stream(), _p(gate_up), gate_up.stride(0), _p(h), M, ff), 'i2t_swiglu_fwd')
    return h


def swiglu_bwd(dh, gate_up, d_gate_up, M, ff):
    _need_cuda(dh, gate_up, d_gate_up)
    _l.check(_lib().i2t_swiglu_bwd(_stream(), _p(dh), _p(gate_up), gate_up.stride(0), _p(d_gate_up), M, ff), 'i2t_swiglu_bwd')
    return d_gate_up


def dgelu_mul(dh, pre, out, erf=False):
    """out (bf16) = dh (fp32) * gelu'(pre (bf16)), contiguous; tanh form (include/i2t.h::i2t_dgelu_mul) or exact (i2t_dgelu_erf_mul)"""
    _need_cuda(dh, pre, out)
    assert dh.dtype == F32 and pre.dtype == BF16 and out.dtype == BF16 and dh.is_contiguous() and pre.is_contiguous() and out.is_contiguous()
    if erf:
        _l.check(_lib().i2t_dgelu_erf_mul(_stream(), _p(dh), _p(pre), _p(out), dh.numel()), 'i2t_dgelu_erf_mul')
    else:
        _l.check(_lib().i2t_dgelu_mul(_stream(), _p(dh), _p(pre), _p(out), dh.numel()), 'i2t_dgelu_mul')
    return out


def gelu_fwd(pre, out, erf=False):
    """out (bf16) = gelu(pre (bf16)), contiguous (include/i2t.h::i2t_gelu_fwd)"""
    _need_cuda(pre, out)
    assert pre.dtype == BF16 and out.dtype == BF16 and pre.is_contiguous() and out.is_contiguous()
    _l.check(_lib().i2t_gelu_fwd(_stream(), _p(pre), _p(out), pre.numel(), int(bool(erf))), 'i2t_gelu_fwd')
    return out


def lora_stage(x, xcat, xd, M, K, drop):
    """xcat[:, :K] = x and (xd not None) xd = dropout(x): include/i2t.h::i2t_lora_stage; drop = (1, key, thr, scale) | None"""
    _need_cuda(x, xcat)
    assert x.is_contiguous() and (drop is None) == (xd is None) and (drop is None or int(drop[0]) == 1)
    _l.check(_lib().i2t_lora_stage(_stream(), _p(x), _p(xcat), xcat.stride(0), _p(xd), M, K, *_drop(drop)[1:]), 'i2t_lora_stage')


def sumsq(g: torch.Tensor, ws: torch.Tensor, accumulate=False):
    """ws[0] (+)= sum(g^2)  (include/i2t.h::i2t_sumsq)"""
    _need_cuda(g, ws)
    _l.check(_lib().i2t_sumsq(_stream(), _p(g), g.numel(), _p(ws), int(accumulate)), 'i2t_sumsq')
    return ws


def conv_fwd(x, in_gelu, w, bias, y, w_ws, B, Cin, Cout, H, W, k):
    _need_cuda(x, w, y, w_ws)
    _l.check(_lib().i2t_conv_fwd(_stream(), _p(x), int(x.dtype == F32), int(in_gelu), _p(w), _p(bias), _p(y), _p(w_ws), B, Cin,
                                 Cout, H, W, k), 'i2t_conv_fwd')
    return y


def conv_bwd_data(dy, w, x_pre, in_gelu, dx, w_ws, B, Cin, Cout, H, W, k):
    _need_cuda(dy, w, dx, w_ws)
    _l.check(_lib().i2t_conv_bwd_data(_stream(), _p(dy), _p(w), _p(x_pre), int(in_gelu), _p(dx), _p(w_ws), B, Cin, Cout, H, W,
                                      k), 'i2t_conv_bwd_data')
    return dx


def conv_bwd_weight(dy, x, in_gelu, dw, db, B, Cin, Cout, H, W, k):
    _need_cuda(dy, x, dw)
    _l.check(_lib().i2t_conv_bwd_weight(_stream(), _p(dy), _p(x), int(x.dtype == F32), int(in_gelu), _p(dw), _p(db), B, Cin,
                                        Cout, H, W, k), 'i2t_conv_bwd_weight')


LAYOUT_NCHW_F32, LAYOUT_NCHW_BF16, LAYOUT_NHWC_BF16 = 0, 1, 2


def conv6_fwd(x, x_layout, in_gelu, w, bias, y, y_nchw, w_ws, B, Cin, Cout, H, W):
    _need_cuda(x, w, y, w_ws)
    _l.check(_lib().i2t_conv6_fwd(_stream(), _p(x), x_layout, int(in_gelu), _p(w), _p(bias), _p(y), int(y_nchw), _p(w_ws), B, Cin,
                                  Cout, H, W), 'i2t_conv6_fwd')
    return y


def conv6_bwd_data(dy, dy_layout, w, x_pre, dx, w_ws, B, Cin, Cout, H, W):
    _need_cuda(dy, w, x_pre, dx, w_ws)
    _l.check(_lib().i2t_conv6_bwd_data(_stream(), _p(dy), dy_layout, _p(w), _p(x_pre), _p(dx), _p(w_ws), B, Cin, Cout, H, W),
             'i2t_conv6_bwd_data')
    return dx


def conv6_bwd_weight(dy, dy_layout, x, x_layout, in_gelu, dw, db, scratch, B, Cin, Cout, H, W):
    _need_cuda(dy, x, dw, scratch)
    _l.check(_lib().i2t_conv6_bwd_weight(_stream(), _p(dy), dy_layout, _p(x), x_layout, int(in_gelu), _p(dw), _p(db), _p(scratch),
                                         B, Cin, Cout, H, W), 'i2t_conv6_bwd_weight')


def dropout_apply(x, rows, cols, drop):
    """in place; drop = (mode, key, thr, scale)"""
    _need_cuda(x)
    _l.check(_lib().i2t_dropout_apply(_stream(), _p(x), int(x.dtype == F32), rows, cols, *_drop(drop)), 'i2t_dropout_apply')
    return x


def nchw_to_nhwc(src, dst, B, C, H, W):
    _need_cuda(src, dst)
    _l.check(_lib().i2t_nchw_to_nhwc_bf16(_stream(), _p(src), _p(dst), B, C, H, W), 'i2t_nchw_to_nhwc_bf16')
    return dst


def cast_f32_bf16(src, dst, n=None):
    _need_cuda(src, dst)
    if precise():
        dst._i2t_lo = torch.zeros_like(dst)
        return split_f32(src, dst, dst._i2t_lo, n)
    _l.check(_lib().i2t_cast_f32_bf16(_stream(), _p(src), _p(dst), src.numel() if n is None else n), 'i2t_cast_f32_bf16')
    return dst


def adamw_step(p, g, m, v, p_bf16, n, seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, step, grad_scale=1.0):
    _need_cuda(p, g, m, v)
    _l.check(_lib().i2t_adamw_step(_stream(), _p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(seg_end), _p(seg_lr), _p(seg_wd),
                                   nseg, float(beta1), float(beta2), float(eps), int(step), float(grad_scale)),
             'i2t_adamw_step')


def snradam_step(p, g, m, v, p_bf16, n, seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, step, grad_scale=1.0):
    _need_cuda(p, g, m, v)
    _l.check(_lib().i2t_snradam_step(_stream(), _p(p), _p(g), _p(m), _p(v), _p(p_bf16), n, _p(seg_end), _p(seg_lr), _p(seg_wd),
                                     nseg, float(beta1), float(beta2), float(eps), int(step), float(grad_scale)),
             'i2t_snradam_step')


def bcast_rows(src, y, y_batch_stride, B, rows, d, drop=None):
    assert drop is None or int(drop[0]) == 1
    _l.check(_lib().i2t_bcast_rows_drop(_stream(), _p(src), _p(y), y_batch_stride, B, rows, d, *_drop(drop)[1:]), 'i2t_bcast_rows_drop')


def sum_over_batch(x, x_batch_stride, dst, B, rows, d, accumulate=False):
    _l.check(_lib().i2t_sum_over_batch(_stream(), _p(x), x_batch_stride, _p(dst), B, rows, d, int(accumulate)),
             'i2t_sum_over_batch')


def copy_rows(x, x_bs, y, y_bs, B, rows, d):
    _l.check(_lib().i2t_copy_rows(_stream(), _p(x), x_bs, _p(y), y_bs, int(y.dtype == BF16), B, rows, d), 'i2t_copy_rows')


def add_(dst, src):
    _l.check(_lib().i2t_add_f32(_stream(), _p(dst), _p(src), dst.numel()), 'i2t_add_f32')
    return dst


def decode_attention(q, q_rs, kcache, vcache, cache_bs, cache_rs, o, o_rs, pos, n_keys_fixed, B, H, append_dm=0, cache_hs=64):
    """cache_hs = 64: token-major cache rows [t][H][64]; cache_hs = tmax * 64 with cache_rs = 64: head-major [H][t][64]"""
    _l.check(_lib().i2t_decode_attention(_stream(), _p(q), q_rs, _p(kcache), _p(vcache), cache_bs, cache_rs, cache_hs, _p(o), o_rs,
                                         _p(pos), n_keys_fixed, append_dm, B, H), 'i2t_decode_attention')


def kv_append(qkv, qkv_rs, kcache, vcache, cache_bs, cache_rs, pos, B, d):
    _l.check(_lib().i2t_kv_append(_stream(), _p(qkv), qkv_rs, _p(kcache), _p(vcache), cache_bs, cache_rs, _p(pos), B, d),
             'i2t_kv_append')


def ngram_ban_argmax(logits, ld, ids, ids_ld, len_ptr, ngram_sizes, n_sizes, B, V, margin_out=None):
    _l.check(_lib().i2t_ngram_ban_argmax(_stream(), _p(logits), ld, int(logits.dtype == F32), _p(ids), ids_ld, _p(len_ptr),
                                         _p(ngram_sizes), n_sizes, B, V, _p(margin_out)), 'i2t_ngram_ban_argmax')


def gemm_top2(a, b, top2, M, N, K):
    """the two largest of every 64-column segment of a . b^T, not the product (include/i2t.h::i2t_gemm_bf16_top2); top2 f32 [M, ceil(N/64), 4]"""
    _need_cuda(a, b, top2)
    assert a.dtype == BF16 and b.dtype == BF16 and top2.dtype == F32 and top2.is_contiguous() and top2.shape[-2] == (N + 63) // 64
    _l.check(_lib().i2t_gemm_bf16_top2(_stream(), _p(a), a.stride(0), _p(b), b.stride(0), M, N, K, _p(top2), (N + 63) // 64), 'i2t_gemm_bf16_top2')
    return top2


def top2_ngram_argmax(top2, hidden, w_head, ids, ids_ld, len_ptr, ngram_sizes, n_sizes, B, V, d):
    """n-gram ban + argmax over gemm_top2's segments (include/i2t.h::i2t_top2_ngram_argmax)"""
    _need_cuda(top2, hidden, w_head, ids)
    _l.check(_lib().i2t_top2_ngram_argmax(_stream(), _p(top2), (V + 63) // 64, _p(hidden), hidden.stride(0), _p(w_head), w_head.stride(0), d,
                                          _p(ids), ids_ld, _p(len_ptr), _p(ngram_sizes), n_sizes, B, V), 'i2t_top2_ngram_argmax')


def sample_token(logits, ld, ids, ids_ld, len_ptr, ngram_sizes, n_sizes, B, V, temperature, top_k, nucleus_p, seed, dist_out=None):
    """One sampling step (include/i2t.h::i2t_sample_token); top_k None/0 = no crop, nucleus_p None = no cut."""
    _need_cuda(logits, ids, seed)
    assert logits.dtype == F32 and seed.dtype == torch.int32 and seed.numel() >= 2
    _l.check(_lib().i2t_sample_token(_stream(), _p(logits), ld, _p(ids), ids_ld, _p(len_ptr), _p(ngram_sizes), n_sizes, B, V,
                                     float(temperature), int(top_k or 0), float(-1.0 if nucleus_p is None else nucleus_p),
                                     _p(seed), _p(dist_out), 0 if dist_out is None else dist_out.stride(0)), 'i2t_sample_token')


def embed_step(ids, ids_ld, len_ptr, wte, wpe, x, B, d, pos_offset, vocab):
    _l.check(_lib().i2t_embed_step(_stream(), _p(ids), ids_ld, _p(len_ptr), _p(wte), _p(wpe), _p(x), B, d, pos_offset, vocab),
             'i2t_embed_step')


def advance(counters, delta=1):
    _l.check(_lib().i2t_advance(_stream(), _p(counters), counters.numel(), delta), 'i2t_advance')


class Graph:
    """hipGraph captured from the launches issued between ``begin()`` and ``end()`` on the current stream."""

    def __init__(self):
        self._exec = None
        self._stream = None

    def begin(self):
        self._stream = torch.cuda.current_stream().cuda_stream
        _l.check(_lib().i2t_graph_capture_begin(self._stream), 'i2t_graph_capture_begin')

    def end(self):
        import ctypes as C
        h = C.c_void_p()
        _l.check(_lib().i2t_graph_capture_end(self._stream, C.byref(h)), 'i2t_graph_capture_end')
        self._exec = h

    def launch(self):
        _l.check(_lib().i2t_graph_launch(self._exec, torch.cuda.current_stream().cuda_stream), 'i2t_graph_launch')

    def __del__(self):
        if self._exec is not None:
            try:
                _lib().i2t_graph_destroy(self._exec)
            except Exception:
                pass


# ---- PretrainedViT pieces (csrc/vit.hip)
ACT_NONE, ACT_GELU, ACT_DGELU, ACT_GELU_ERF, ACT_DGELU_ERF, ACT_GELU_DOUT, ACT_MUL_AUX = 0, 1, 2, 3, 4, 5, 6


def patchify(images, out, B, C, H, W, p):
    _need_cuda(images, out)
    assert images.dtype == F32 and out.dtype == BF16 and images.is_contiguous()
    _l.check(_lib().i2t_patchify(_stream(), _p(images), _p(out), B, C, H, W, p), 'i2t_patchify')
    return out


def vit_tokens(proj, cls, pos, x, B, T, d):
    _need_cuda(proj, cls, pos, x)
    _l.check(_lib().i2t_vit_tokens(_stream(), _p(proj), _p(cls), _p(pos), _p(x), B, T, d), 'i2t_vit_tokens')
    return x


def l2norm_fwd(x, y, y_bf16, inv_norm, M, d):
    _need_cuda(x, y, y_bf16, inv_norm)
    _l.check(_lib().i2t_l2norm_fwd(_stream(), _p(x), _p(y), _p(y_bf16), _p(inv_norm), M, d), 'i2t_l2norm_fwd')


def l2norm_bwd(dy, x, inv_norm, dx, M, d, accumulate=False):
    _need_cuda(dy, x, inv_norm, dx)
    _l.check(_lib().i2t_l2norm_bwd(_stream(), _p(dy), _p(x), _p(inv_norm), _p(dx), int(accumulate), M, d), 'i2t_l2norm_bwd')
    return dx


def transpose_last2(src, dst, dst_bf16, B, R, C):
    _need_cuda(src, dst, dst_bf16)
    assert src.dtype == F32 and src.is_contiguous()
    _l.check(_lib().i2t_transpose_last2(_stream(), _p(src), _p(dst), _p(dst_bf16), B, R, C), 'i2t_transpose_last2')


def peer_lookup_fwd(scores, inp_proj, residual, emb_in, emb_out, out, sv, M, nhead, nq, topk, din, dout):
    """sv: namespace(unit int32 [M, nhead, topk], lr int32 [M, nhead, topk, 2], score f32, dot f32)."""
    _need_cuda(scores, inp_proj, residual, emb_in, emb_out, out)
    assert scores.dtype == F32 and inp_proj.dtype == BF16 and emb_in.dtype == BF16 and emb_out.dtype == BF16 and residual.dtype == F32
    _l.check(_lib().i2t_peer_lookup_fwd(_stream(), _p(scores), _p(inp_proj), _p(residual), _p(emb_in), _p(emb_out), _p(out), _p(sv.unit),
                                        _p(sv.lr), _p(sv.score), _p(sv.dot), M, nhead, nq, topk, din, dout), 'i2t_peer_lookup_fwd')
    return out


def peer_lookup_bwd(dout, inp_proj, emb_in, emb_out, sv, dscores, dinp_proj, g_emb_in, g_emb_out, M, nhead, nq, topk, din, dout_w):
    _need_cuda(dout, inp_proj, emb_in, emb_out, dscores, dinp_proj)
    _l.check(_lib().i2t_peer_lookup_bwd(_stream(), _p(dout), _p(inp_proj), _p(emb_in), _p(emb_out), _p(sv.unit), _p(sv.lr), _p(sv.score),
                                        _p(sv.dot), _p(dscores), _p(dinp_proj), _p(g_emb_in), _p(g_emb_out), M, nhead, nq, topk, din, dout_w),
             'i2t_peer_lookup_bwd')


def gemm_f32(x, P, z, M, N, K):
    _need_cuda(x, P, z)
    assert x.dtype == F32 and P.dtype == F32 and z.dtype == F32 and x.is_contiguous() and P.is_contiguous() and z.is_contiguous()
    _l.check(_lib().i2t_gemm_f32(_stream(), _p(x), _p(P), _p(z), M, N, K), 'i2t_gemm_f32')
    return z


def lsh_embed_fwd(z, tables, slot_stride, tab_off, nbins, grids, grid_off, out, rows, B, n_cls, nK, n_proj, dout):
    _need_cuda(z, tables, tab_off, nbins, grids, grid_off, out, rows)
    assert tab_off.dtype == torch.int64 and nbins.dtype == torch.int32 and grid_off.dtype == torch.int32 and rows.dtype == torch.int32
    _l.check(_lib().i2t_lsh_embed_fwd(_stream(), _p(z), _p(tables), int(slot_stride), _p(tab_off), _p(nbins), _p(grids), _p(grid_off), _p(out),
                                      _p(rows), B, n_cls, nK, n_proj, dout), 'i2t_lsh_embed_fwd')
    return out


def lsh_embed_bwd(dy, rows, g_tables, slot_stride, tab_off, B, n_cls, nK, n_proj, dout):
    _need_cuda(dy, rows, g_tables, tab_off)
    _l.check(_lib().i2t_lsh_embed_bwd(_stream(), _p(dy), _p(rows), _p(g_tables), int(slot_stride), _p(tab_off), B, n_cls, nK, n_proj, dout),
             'i2t_lsh_embed_bwd')


# ---- fp8 (e4m3) operand path for frozen weights (csrc/fp8.hip)
def quant_rows_fp8(x, out, scale, M, K):
    """x bf16 / f32 [M, >= K] -> out uint8 [M, ld_out] (e4m3 bytes, zero pad), scale f32 [M] (amax / 448)."""
    _need_cuda(x, out, scale)
    assert out.dtype == torch.uint8 and scale.dtype == F32 and x.dtype in (BF16, F32) and x.stride(-1) == 1
    _l.check(_lib().i2t_quant_rows_fp8(_stream(), _p(x), int(x.dtype == F32), x.stride(0), _p(out), out.stride(0), _p(scale), M, K), 'i2t_quant_rows_fp8')
    return out


def quant_cols_fp8(w, out, scale, N, K):
    """W bf16 [N, K] -> out uint8 [K, ld_out] = W^T quantised per k, scale f32 [K]."""
    _need_cuda(w, out, scale)
    assert w.dtype == BF16 and out.dtype == torch.uint8
    _l.check(_lib().i2t_quant_cols_fp8(_stream(), _p(w), w.stride(0), _p(out), out.stride(0), _p(scale), N, K), 'i2t_quant_cols_fp8')
    return out


def rmsnorm_fwd_fp8(x, w, y8, scale, rstd, M, d, eps, y_bf16=None):
    """y8 (e4m3 [M, ld8]) , scale [M] = quantised RMSNorm(x); rstd [M] or None; y_bf16: optional contiguous bf16 copy  (include/i2t.h)"""
    _need_cuda(x, w, y8, scale)
    assert y_bf16 is None or (y_bf16.dtype == BF16 and y_bf16.is_contiguous())
    _l.check(_lib().i2t_rmsnorm_fwd_fp8(_stream(), _p(x), _p(w), _p(y8), y8.stride(0), _p(scale), _p(rstd), M, d, float(eps), _p(y_bf16)),
             'i2t_rmsnorm_fwd_fp8')
    return y8


def swiglu_fwd_fp8(gate_up, h8, scale, M, ff, h_bf16=None):
    _need_cuda(gate_up, h8, scale)
    assert h_bf16 is None or (h_bf16.dtype == BF16 and h_bf16.is_contiguous())
    _l.check(_lib().i2t_swiglu_fwd_fp8(_stream(), _p(gate_up), gate_up.stride(0), _p(h8), h8.stride(0), _p(scale), M, ff, _p(h_bf16)), 'i2t_swiglu_fwd_fp8')
    return h8


def swiglu_bwd_fp8(dh, gate_up, dgu8, scale, M, ff, dgu_bf16=None):
    _need_cuda(dh, gate_up, dgu8, scale)
    assert dh.is_contiguous() and (dgu_bf16 is None or (dgu_bf16.dtype == BF16 and dgu_bf16.is_contiguous()))
    _l.check(_lib().i2t_swiglu_bwd_fp8(_stream(), _p(dh), _p(gate_up), gate_up.stride(0), _p(dgu8), dgu8.stride(0), _p(scale), M, ff, _p(dgu_bf16)),
             'i2t_swiglu_bwd_fp8')
    return dgu8


def gemm_fp8(a8, sa, b8, sb, out, M, N, K, bias=None, residual=None, act=0):
    """out[M, N] = act((a8[M, K] . b8[N, K]^T) * sa[m] * sb[n] (+ bias)) (+ residual f32); include/i2t.h::i2t_gemm_fp8."""
    _need_cuda(a8, b8, sa, sb, out)
    assert a8.dtype == torch.uint8 and b8.dtype == torch.uint8 and out.dtype in (BF16, F32)
    _l.check(_lib().i2t_gemm_fp8(_stream(), _p(a8), a8.stride(0), _p(sa), _p(b8), b8.stride(0), _p(sb), _p(out), out.stride(0), int(out.dtype == F32),
                                 M, N, K, _p(bias), int(act), _p(residual), residual.stride(0) if residual is not None else 0), 'i2t_gemm_fp8')
    return out
