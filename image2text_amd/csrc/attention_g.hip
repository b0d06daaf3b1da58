// Grouped-query attention forward / backward for gfx950: head_dim 16 / 32 / 64 / 128, G = H / H_kv query heads per key/value
// head (multi-query: H_kv = 1; G = 1 is plain multi-head attention at a head width attention.hip does not cover).
// Reference: models/layers.py:391-430 (MultiQueryAttention) and nn.MultiheadAttention at 128-wide heads (layers.py:537-542).
//
// Same structure as attention.hip (read its header first): a workgroup = 4 waves = one 64-row tile of one (sequence, head);
// 64-row operand tiles staged in LDS with a (2 D + 32)-byte row stride (conflict-free for row reads and transposed reads),
// transposed scores so that a lane owns one query column, accumulators fed back as MFMA operands.  D is the LDS tile width
// (32 / 64 / 128); a 16-wide head is a 32-wide one whose upper half is zero-filled on load and never stored.
//
// Grouped K/V in the backward pass: dK and dV of a shared head are the SUM over its G query heads, so the dK/dV workgroup of
// (key tile, kv head) walks the query tiles of all G heads back to back and accumulates in registers -- no atomics, no second
// pass, and K / V are loaded once per workgroup.
#include "attention_common.h"

namespace {

template <int D>
struct GTile {
    static constexpr int TS = 2 * D + 32;           // row stride in bytes
    static constexpr int BYTES = 64 * TS;
    static constexpr int KS = D / 32;               // MFMA k-steps along the head dim
    static constexpr int DT = D / 16;               // 16-column output subtiles
    static constexpr int CH = D / 8;                // 16-byte chunks per row
    static constexpr int NU = (64 * CH) / 256;      // chunks per thread per tile (D = 32: 1, 64: 2, 128: 4)

    struct Regs {
        u32x4 v[NU];
    };
    // rows [r0, r0 + 64) x hd columns of one (sequence, head) slice; rows >= nrows and columns >= hd read as zero
    static __device__ __forceinline__ void load(Regs& t, const bf16_t* base, int rs, int r0, int nrows, int hd, int tid) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = tid + 256 * u, r = c / CH, kc = c % CH;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (r0 + r < nrows && kc * 8 < hd) v = *reinterpret_cast<const u32x4*>(base + (size_t)(r0 + r) * rs + kc * 8);
            t.v[u] = v;
        }
    }
    static __device__ __forceinline__ void store(const Regs& t, unsigned char* lds, int tid) {
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int c = tid + 256 * u, r = c / CH, kc = c % CH;
            *reinterpret_cast<u32x4*>(lds + r * TS + kc * 16) = t.v[u];
        }
    }
    // row fragment: lane (g, i) <- tile[r0 + i][32 ks + 8 g .. +7]
    static __device__ __forceinline__ bf16x8 row_frag(const unsigned char* lds, int r0, int ks, int lane) {
        const int g = lane >> 4, i = lane & 15;
        return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + (r0 + i) * TS + (ks * 4 + g) * 16));
    }
    // transposed fragment for k-step s2 (32 rows) and column subtile c0: lane (g, i) <- tile[row(g, j)][c0 + i],
    // row(g, j) = 32 s2 + 16 (j >> 2) + 4 g + (j & 3) -- the row order of a transposed-score accumulator
    static __device__ __forceinline__ bf16x8 tr_frag(const unsigned char* lds, int s2, int c0, int lane) {
        const int g = lane >> 4, i = lane & 15;
        const unsigned char* a = lds + (32 * s2 + 4 * g + (i >> 2)) * TS + (c0 + 4 * (i & 3)) * 2;
        s16x4 lo = lds_read_tr16(a);
        s16x4 hi = lds_read_tr16(a + 16 * TS);
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
    // global row fragment: lane (g, i) <- M[row0 + i][32 ks + 8 g .. +7], zeros beyond nrows / hd
    static __device__ __forceinline__ bf16x8 global_frag(const bf16_t* base, int rs, int row0, int nrows, int hd, int ks, int lane) {
        const int g = lane >> 4, i = lane & 15;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + i < nrows && ks * 32 + g * 8 < hd) v = *reinterpret_cast<const u32x4*>(base + (size_t)(row0 + i) * rs + ks * 32 + g * 8);
        return __builtin_bit_cast(bf16x8, v);
    }
};

struct GShape {
    int H, G, hd;             // query heads, query heads per kv head, true head width
    int TqMax, TkMax, causal;
    int split;                // forward only, > 0: rows >= split do not see keys < split (a non-causal decoder's text rows never attend
                              // the soft-prompt columns while the prompt rows see everything: vision_encoder_decoder.py:93-99,106-113)
    float scale;              // 1 / sqrt(hd)
    unsigned drop_key, drop_thr;
    float drop_scale;
};

// ================================================================================================== forward
template <int D, bool DROP>
__global__ __launch_bounds__(256) void gattn_fwd_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, bf16_t* __restrict__ O, long o_bs, int o_rs,
                                                        float* __restrict__ lse, GShape sh, VarLen vl) {
    using T = GTile<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * T::BYTES];
    unsigned char* kt_lds = smem;
    unsigned char* vt_lds = smem + T::BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = sh.H, hd = sh.hd, TqMax = sh.TqMax, TkMax = sh.TkMax, causal = sh.causal;
    int qt, h, b;
    attn_block_coords((TqMax + 63) / 64, H, vl.nseq, qt, h, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, ooff = (size_t)b * o_bs;
    size_t stat_base = ((size_t)b * H + h) * TqMax;
    if (vl.cu_q) {
        const int s0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - s0;
        qoff = (size_t)s0 * Q.rs; ooff = (size_t)s0 * o_rs;
        stat_base = (size_t)h * vl.total_q + s0;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs;
    }
    if (qt * 64 >= Tq) return;                      // workgroup-uniform
    const bf16_t* qb = Q.p + qoff + h * hd;
    const bf16_t* kb = K.p + koff + (h / sh.G) * hd;
    const bf16_t* vb = V.p + voff + (h / sh.G) * hd;
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + li;
    const int shift = Tk - Tq;                      // causal: key j visible iff j <= q + shift
    bf16x8 qf[T::KS];
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) qf[ks] = T::global_frag(qb, Q.rs, q0, Tq, hd, ks, lane);
    int last_key = Tk - 1;
    if (causal) last_key = min(last_key, qt * 64 + 63 + shift);
    const int nkt = last_key / 64 + 1;
    const float sl2 = sh.scale * LOG2E;

    f32x4 o[T::DT];
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;
    const int qlim = causal ? (min(qrow, Tq - 1) + shift) : (Tk - 1);
    const unsigned drow = (((unsigned)b * H + h) * TqMax + min(qrow, Tq - 1)) * (unsigned)TkMax;

    const bool wave_on = q0 < Tq;
    typename T::Regs kr, vr;
    T::load(kr, kb, K.rs, 0, Tk, hd, tid);
    T::load(vr, vb, V.rs, 0, Tk, hd, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        T::store(kr, kt_lds, tid);
        T::store(vr, vt_lds, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            T::load(kr, kb, K.rs, (kt + 1) * 64, Tk, hd, tid);
            T::load(vr, vb, V.rs, (kt + 1) * 64, Tk, hd, tid);
        }
        if (!wave_on) continue;
        const int nkj = min(4, (last_key - kt * 64) / 16 + 1);
        const bool full = kt * 64 + 63 < Tk && (!causal || kt * 64 + 63 <= q0 + shift) &&
                          (sh.split == 0 || kt * 64 >= sh.split || q0 + 15 < sh.split);
        f32x4 s[4];
        float mx = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            f32x4 a = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kj < nkj) {
                a = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < T::KS; ++ks)
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::row_frag(kt_lds, kj * 16, ks, lane), qf[ks], a, 0, 0, 0);
                if (!full) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + kj * 16 + 4 * g + r;
                        if (!(key <= qlim && key < Tk) || (sh.split && qrow >= sh.split && key < sh.split)) a[r] = -INFINITY;
                    }
                }
                mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
            }
            s[kj] = a;
        }
        mx = quad_max(mx) * sl2;
        const float m_new = fmaxf(m, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m - m_safe);
        float rs = 0.f;
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (kj >= nkj) {
                s[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) dropout_keep4(sh.drop_key, drow + kt * 64 + kj * 16 + 4 * g, sh.drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s[kj][r] * sl2 - m_safe);
                rs += p;
                if constexpr (DROP) p = keep[r] ? p : 0.f;
                s[kj][r] = p;
            }
        }
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) o[dt] *= alpha;
        const bf16x8 p0 = pack_frag(s[0], s[1]);
        const bf16x8 p1 = pack_frag(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(vt_lds, 0, dt * 16, lane), p0, o[dt], 0, 0, 0);
            if (nkj > 2) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(vt_lds, 1, dt * 16, lane), p1, o[dt], 0, 0, 0);
        }
    }
    l = quad_sum(l);
    const float inv = l > 0.f ? (DROP ? sh.drop_scale : 1.0f) / l : 0.f;
    if (qrow < Tq) {
        bf16_t* op = O + ooff + (size_t)qrow * o_rs + h * hd;
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
            if (dt * 16 + 4 * g < hd) {
                u32x2 pk = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = pk;
            }
        }
        if (g == 0 && lse) lse[stat_base + qrow] = (m + log2f(l)) * LN2;
    }
}

// ================================================================================================== backward: dQ
struct GOutDrop {
    unsigned key, thr;        // per-token multipliers that scaled q / k / v in the forward (sections 0 / 1 / 2), thr 0 = none
    float scale;
};

template <int D, bool DROP>
__global__ __launch_bounds__(256) void gattn_bwd_dq_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O,
                                                           const float* __restrict__ lse, float* __restrict__ delta,
                                                           bf16_t* __restrict__ dQ, long dq_bs, int dq_rs, GShape sh, VarLen vl,
                                                           GOutDrop od) {
    using T = GTile<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * T::BYTES];
    unsigned char* kt_lds = smem;
    unsigned char* vt_lds = smem + T::BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = sh.H, hd = sh.hd, TqMax = sh.TqMax, TkMax = sh.TkMax, causal = sh.causal;
    int qt, h, b;
    attn_block_coords((TqMax + 63) / 64, H, vl.nseq, qt, h, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, dooff = (size_t)b * dO.bs, dqoff = (size_t)b * dq_bs;
    size_t ooff = (size_t)b * O.bs;
    size_t stat_base = ((size_t)b * H + h) * TqMax;
    if (vl.cu_q) {
        const int s0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - s0;
        qoff = (size_t)s0 * Q.rs; dooff = (size_t)s0 * dO.rs; dqoff = (size_t)s0 * dq_rs; ooff = (size_t)s0 * O.rs;
        stat_base = (size_t)h * vl.total_q + s0;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs;
    }
    if (qt * 64 >= Tq) return;
    const bf16_t* qb = Q.p + qoff + h * hd;
    const bf16_t* kb = K.p + koff + (h / sh.G) * hd;
    const bf16_t* vb = V.p + voff + (h / sh.G) * hd;
    const bf16_t* dob = dO.p + dooff + h * hd;
    const bf16_t* ob = O.p + ooff + h * hd;
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + li;
    const int shift = Tk - Tq;
    bf16x8 qf[T::KS], df[T::KS];
    float sdl = 0.f;
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) {
        qf[ks] = T::global_frag(qb, Q.rs, q0, Tq, hd, ks, lane);
        df[ks] = T::global_frag(dob, dO.rs, q0, Tq, hd, ks, lane);
        const bf16x8 of = T::global_frag(ob, O.rs, q0, Tq, hd, ks, lane);
#pragma unroll
        for (int e = 0; e < 8; ++e) sdl += (float)df[ks][e] * (float)of[e];
    }
    const int qc = min(qrow, Tq - 1);
    const float lse2 = lse[stat_base + qc] * LOG2E;
    const float dl = quad_sum(sdl);                  // delta[q] = sum_d dO[q][d] O[q][d], kept for the dK/dV kernel
    if (g == 0 && qrow < Tq) delta[stat_base + qrow] = dl;
    int last_key = Tk - 1;
    if (causal) last_key = min(last_key, qt * 64 + 63 + shift);
    const int nkt = last_key / 64 + 1;
    const int qlim = causal ? (qc + shift) : (Tk - 1);
    const unsigned drow = (((unsigned)b * H + h) * TqMax + qc) * (unsigned)TkMax;
    const float sl2 = sh.scale * LOG2E;

    f32x4 acc[T::DT];
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dscale = DROP ? sh.drop_scale : 1.f;

    const bool wave_on = q0 < Tq;
    typename T::Regs kr, vr;
    T::load(kr, kb, K.rs, 0, Tk, hd, tid);
    T::load(vr, vb, V.rs, 0, Tk, hd, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        T::store(kr, kt_lds, tid);
        T::store(vr, vt_lds, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            T::load(kr, kb, K.rs, (kt + 1) * 64, Tk, hd, tid);
            T::load(vr, vb, V.rs, (kt + 1) * 64, Tk, hd, tid);
        }
        if (!wave_on) continue;
        const int nkj = min(4, (last_key - kt * 64) / 16 + 1);
        const bool full = kt * 64 + 63 < Tk && (!causal || kt * 64 + 63 <= q0 + shift);
        f32x4 ds[4];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (kj >= nkj) {
                ds[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < T::KS; ++ks) {
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::row_frag(kt_lds, kj * 16, ks, lane), qf[ks], a, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::row_frag(vt_lds, kj * 16, ks, lane), df[ks], dp, 0, 0, 0);
            }
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) dropout_keep4(sh.drop_key, drow + kt * 64 + kj * 16 + 4 * g, sh.drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kt * 64 + kj * 16 + 4 * g + r;
                float p = __builtin_amdgcn_exp2f(a[r] * sl2 - lse2);
                if (!full && !(key <= qlim && key < Tk)) p = 0.f;
                float dpr = dp[r];
                if constexpr (DROP) dpr = keep[r] ? dpr : 0.f;
                ds[kj][r] = p * fmaf(dpr, dscale, -dl);              // x 1/sqrt(hd): once, on dQ
            }
        }
        const bf16x8 s0 = pack_frag(ds[0], ds[1]);
        const bf16x8 s1 = pack_frag(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {   // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(kt_lds, 0, dt * 16, lane), s0, acc[dt], 0, 0, 0);
            if (nkj > 2) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(kt_lds, 1, dt * 16, lane), s1, acc[dt], 0, 0, 0);
        }
    }
    if (qrow < Tq) {
        bf16_t* op = dQ + dqoff + (size_t)qrow * dq_rs + h * hd;
        const unsigned grow = (unsigned)((vl.cu_q ? vl.cu_q[b] : b * TqMax) + qrow);
        const float f = (od.thr ? (dropout_keep(od.key, grow, od.thr) ? od.scale : 0.f) : 1.f) * sh.scale;
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
            if (dt * 16 + 4 * g < hd) {
                acc[dt] *= f;
                u32x2 pk = {pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3])};
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = pk;
            }
        }
    }
}

// ================================================================================================== backward: dK, dV
// One workgroup per (key tile, kv head, sequence); each wave owns 16 keys and walks the 64-row query tiles of the G query
// heads that share this kv head (iteration it -> head hk G + it / nq, tile qt0 + it % nq).
template <int D, bool DROP>
__global__ __launch_bounds__(256) void gattn_bwd_dkv_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, const float* __restrict__ lse,
                                                            const float* __restrict__ delta, bf16_t* __restrict__ dK, long dk_bs,
                                                            int dk_rs, bf16_t* __restrict__ dV, long dv_bs, int dv_rs, GShape sh,
                                                            VarLen vl, GOutDrop od) {
    using T = GTile<D>;
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * T::BYTES + 2 * 64 * 4];
    unsigned char* q_lds = smem;
    unsigned char* do_lds = smem + T::BYTES;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * T::BYTES);
    float* dl_lds = lse_lds + 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int H = sh.H, G = sh.G, hd = sh.hd, TqMax = sh.TqMax, TkMax = sh.TkMax, causal = sh.causal;
    const int Hkv = H / G;
    int kt, hk, b;
    attn_block_coords((TkMax + 63) / 64, Hkv, vl.nseq, kt, hk, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, dooff = (size_t)b * dO.bs;
    size_t dkoff = (size_t)b * dk_bs, dvoff = (size_t)b * dv_bs;
    int qs0 = 0;
    if (vl.cu_q) {
        qs0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - qs0;
        qoff = (size_t)qs0 * Q.rs; dooff = (size_t)qs0 * dO.rs;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs; dkoff = (size_t)s0 * dk_rs; dvoff = (size_t)s0 * dv_rs;
    }
    if (kt * 64 >= Tk) return;
    const bf16_t* kb = K.p + koff + hk * hd;
    const bf16_t* vb = V.p + voff + hk * hd;
    const int k0 = kt * 64 + w * 16;
    const int key = k0 + li;
    const int shift = Tk - Tq;
    bf16x8 kf[T::KS], vf[T::KS];
#pragma unroll
    for (int ks = 0; ks < T::KS; ++ks) {
        kf[ks] = T::global_frag(kb, K.rs, k0, Tk, hd, ks, lane);
        vf[ks] = T::global_frag(vb, V.rs, k0, Tk, hd, ks, lane);
    }
    int first_q = 0;
    if (causal) first_q = max(0, kt * 64 - shift);
    const int qt0 = first_q / 64, nqt = (Tq + 63) / 64;
    const int nq = max(nqt - qt0, 0), nit = nq * G;
    const float sl2 = sh.scale * LOG2E;

    f32x4 adk[T::DT], adv[T::DT];
#pragma unroll
    for (int dt = 0; dt < T::DT; ++dt) adk[dt] = adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool wave_on = k0 < Tk;
    const float dscale = DROP ? sh.drop_scale : 1.f;
    typename T::Regs qr, dr;
    float lse_r = 0.f, dl_r = 0.f;
    auto fetch = [&](int it) {
        const int h = hk * G + it / nq, qt = qt0 + it % nq;
        T::load(qr, Q.p + qoff + h * hd, Q.rs, qt * 64, Tq, hd, tid);
        T::load(dr, dO.p + dooff + h * hd, dO.rs, qt * 64, Tq, hd, tid);
        if (tid < 64) {
            const size_t sb = vl.cu_q ? (size_t)h * vl.total_q + qs0 : ((size_t)b * H + h) * TqMax;
            const int q = min(qt * 64 + tid, Tq - 1);
            lse_r = lse[sb + q] * LOG2E;
            dl_r = delta[sb + q];
        }
    };
    if (nit > 0) fetch(0);
    for (int it = 0; it < nit; ++it) {
        const int h = hk * G + it / nq, qt = qt0 + it % nq;
        __syncthreads();
        T::store(qr, q_lds, tid);
        T::store(dr, do_lds, tid);
        if (tid < 64) {
            lse_lds[tid] = lse_r;
            dl_lds[tid] = dl_r;
        }
        __syncthreads();
        if (it + 1 < nit) fetch(it + 1);
        if (!wave_on) continue;
        const int nqj = min(4, (Tq - 1 - qt * 64) / 16 + 1);
        const bool full = qt * 64 + 63 < Tq && k0 + 15 < Tk && (!causal || k0 + 15 <= qt * 64 + shift);
        f32x4 p[4], ds[4];
#pragma unroll
        for (int qj = 0; qj < 4; ++qj) {
            if (qj >= nqj) {
                p[qj] = ds[qj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int ks = 0; ks < T::KS; ++ks) {     // D[q][key]: rows = q (LDS tile), cols = key (this lane's register fragment)
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::row_frag(q_lds, qj * 16, ks, lane), kf[ks], a, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::row_frag(do_lds, qj * 16, ks, lane), vf[ks], dp, 0, 0, 0);
            }
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_lds + qj * 16 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_lds + qj * 16 + 4 * g);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int q = qt * 64 + qj * 16 + 4 * g + r;
                float pv = __builtin_amdgcn_exp2f(a[r] * sl2 - l4[r]);
                if (!full && !((q < Tq) && (key < Tk) && (!causal || key <= q + shift))) pv = 0.f;
                float pd = pv, dpr = dp[r];
                if constexpr (DROP) {
                    const unsigned rb = ((unsigned)b * H + h) * TqMax;
                    const bool keep = dropout_keep(sh.drop_key, (rb + min(q, Tq - 1)) * (unsigned)TkMax + key, sh.drop_thr);
                    pd = keep ? pv : 0.f;
                    dpr = keep ? dpr : 0.f;
                }
                p[qj][r] = pd;
                ds[qj][r] = pv * fmaf(dpr, dscale, -d4[r]);
            }
        }
        const bf16x8 p0 = pack_frag(p[0], p[1]), p1 = pack_frag(p[2], p[3]);
        const bf16x8 s0 = pack_frag(ds[0], ds[1]), s1 = pack_frag(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
            adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(do_lds, 0, dt * 16, lane), p0, adv[dt], 0, 0, 0);
            adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(q_lds, 0, dt * 16, lane), s0, adk[dt], 0, 0, 0);
            if (nqj > 2) {
                adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(do_lds, 1, dt * 16, lane), p1, adv[dt], 0, 0, 0);
                adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(T::tr_frag(q_lds, 1, dt * 16, lane), s1, adk[dt], 0, 0, 0);
            }
        }
    }
    if (key < Tk) {
        bf16_t* pk_ = dK + dkoff + (size_t)key * dk_rs + hk * hd;
        bf16_t* pv_ = dV + dvoff + (size_t)key * dv_rs + hk * hd;
        const unsigned grow = (unsigned)((vl.cu_k ? vl.cu_k[b] : b * TkMax) + key);
        const float fk = (od.thr ? (dropout_keep(od.key + 1u, grow, od.thr) ? od.scale : 0.f) : 1.f) * sh.scale;
        const float fv = (od.thr ? (dropout_keep(od.key + 2u, grow, od.thr) ? od.scale : 0.f) : 1.f) * dscale;
#pragma unroll
        for (int dt = 0; dt < T::DT; ++dt) {
            if (dt * 16 + 4 * g < hd) {
                adk[dt] *= fk;
                adv[dt] *= fv;
                u32x2 a = {pack_bf16x2(adk[dt][0], adk[dt][1]), pack_bf16x2(adk[dt][2], adk[dt][3])};
                u32x2 c = {pack_bf16x2(adv[dt][0], adv[dt][1]), pack_bf16x2(adv[dt][2], adv[dt][3])};
                *reinterpret_cast<u32x2*>(pk_ + dt * 16 + 4 * g) = a;
                *reinterpret_cast<u32x2*>(pv_ + dt * 16 + 4 * g) = c;
            }
        }
    }
}

#define GATTN_DISPATCH(KERNEL, hd, drop_thr, ...)                                                    \
    do {                                                                                             \
        if ((hd) <= 32) {                                                                            \
            if (drop_thr) hipLaunchKernelGGL((KERNEL<32, true>), __VA_ARGS__);                       \
            else hipLaunchKernelGGL((KERNEL<32, false>), __VA_ARGS__);                               \
        } else if ((hd) == 64) {                                                                     \
            if (drop_thr) hipLaunchKernelGGL((KERNEL<64, true>), __VA_ARGS__);                       \
            else hipLaunchKernelGGL((KERNEL<64, false>), __VA_ARGS__);                               \
        } else {                                                                                     \
            if (drop_thr) hipLaunchKernelGGL((KERNEL<128, true>), __VA_ARGS__);                      \
            else hipLaunchKernelGGL((KERNEL<128, false>), __VA_ARGS__);                              \
        }                                                                                            \
    } while (0)

bool g_strides_ok(const void* p, long bs, int rs, int width) { return p && ALIGNED16(p) && (bs % 8 == 0) && (rs % 8 == 0) && rs >= width; }

int g_check(const char* who, int B, int H, int Hkv, int hd, int Tq, int Tk, int causal, unsigned drop_thr, const int* cu_q, int total_q) {
    I2T_REQUIRE(B > 0 && H > 0 && Hkv > 0 && Tq > 0 && Tk > 0, "%s: empty problem", who);
    I2T_REQUIRE(H % Hkv == 0, "%s: H = %d is not a multiple of H_kv = %d", who, H, Hkv);
    I2T_REQUIRE(hd == 16 || hd == 32 || hd == 64 || hd == 128, "%s: head_dim %d (16, 32, 64 or 128)", who, hd);
    I2T_REQUIRE(!cu_q || total_q > 0, "%s: packed queries need total_q", who);
    I2T_REQUIRE(drop_thr == 0 || (double)B * H * Tq * Tk < 4294967296.0, "%s: dropout index overflows 32 bits", who);
    I2T_REQUIRE(!causal || Tk >= Tq, "%s: causal needs Tk >= Tq", who);
    I2T_REQUIRE((double)((Tq + 63) / 64 + (Tk + 63) / 64) * H * B < 2147483647.0, "%s: grid too large", who);
    return I2T_OK;
}

}  // namespace

extern "C" int i2t_gq_attention_fwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs, const void* v,
                                    long v_bs, int v_rs, void* o, long o_bs, int o_rs, float* lse, int B, int H, int Hkv, int hd, int Tq,
                                    int Tk, int causal, unsigned drop_key, unsigned drop_thr, float drop_scale, const int* cu_q,
                                    const int* cu_k, int total_q, int split) {
    I2T_REQUIRE(split == 0 || (!causal && !cu_q && !cu_k && Tq == Tk && split < Tq), "i2t_gq_attention_fwd: split needs dense non-causal self-attention");
    if (int rc = g_check("i2t_gq_attention_fwd", B, H, Hkv, hd, Tq, Tk, causal, drop_thr, cu_q, total_q)) return rc;
    I2T_REQUIRE(g_strides_ok(q, q_bs, q_rs, H * hd) && g_strides_ok(k, k_bs, k_rs, Hkv * hd) && g_strides_ok(v, v_bs, v_rs, Hkv * hd) &&
                    g_strides_ok(o, o_bs, o_rs, H * hd),
                "i2t_gq_attention_fwd: operands must be 16-byte aligned with strides that are multiples of 8 and cover every head");
    AttnPtr Q{(const bf16_t*)q, q_bs, q_rs}, K{(const bf16_t*)k, k_bs, k_rs}, V{(const bf16_t*)v, v_bs, v_rs};
    const GShape sh{H, H / Hkv, hd, Tq, Tk, causal, split, 1.0f / sqrtf((float)hd), drop_key, drop_thr, drop_scale};
    GATTN_DISPATCH(gattn_fwd_kernel, hd, drop_thr, dim3(((Tq + 63) / 64) * H * B), dim3(256), 0, (hipStream_t)stream, Q, K, V, (bf16_t*)o,
                   o_bs, o_rs, lse, sh, VarLen{cu_q, cu_k, total_q, B});
    I2T_CHECK_LAUNCH("i2t_gq_attention_fwd");
    return I2T_OK;
}

extern "C" int i2t_gq_attention_bwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs, const void* v,
                                    long v_bs, int v_rs, const void* o, long o_bs, int o_rs, const void* d_o, long do_bs, int do_rs,
                                    const float* lse, float* delta_ws, void* dq, long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs,
                                    void* dv, long dv_bs, int dv_rs, int B, int H, int Hkv, int hd, int Tq, int Tk, int causal,
                                    unsigned drop_key, unsigned drop_thr, float drop_scale, const int* cu_q, const int* cu_k, int total_q,
                                    unsigned out_drop_key, unsigned out_drop_thr, float out_drop_scale) {
    if (int rc = g_check("i2t_gq_attention_bwd", B, H, Hkv, hd, Tq, Tk, causal, drop_thr, cu_q, total_q)) return rc;
    I2T_REQUIRE(lse && delta_ws, "i2t_gq_attention_bwd: lse and the delta workspace are required");
    I2T_REQUIRE(g_strides_ok(q, q_bs, q_rs, H * hd) && g_strides_ok(k, k_bs, k_rs, Hkv * hd) && g_strides_ok(v, v_bs, v_rs, Hkv * hd) &&
                    g_strides_ok(o, o_bs, o_rs, H * hd) && g_strides_ok(d_o, do_bs, do_rs, H * hd) && g_strides_ok(dq, dq_bs, dq_rs, H * hd) &&
                    g_strides_ok(dk, dk_bs, dk_rs, Hkv * hd) && g_strides_ok(dv, dv_bs, dv_rs, Hkv * hd),
                "i2t_gq_attention_bwd: operands must be 16-byte aligned with strides that are multiples of 8 and cover every head");
    hipStream_t s = (hipStream_t)stream;
    AttnPtr Q{(const bf16_t*)q, q_bs, q_rs}, K{(const bf16_t*)k, k_bs, k_rs}, V{(const bf16_t*)v, v_bs, v_rs};
    AttnPtr DO{(const bf16_t*)d_o, do_bs, do_rs}, Ow{(const bf16_t*)o, o_bs, o_rs};
    const VarLen vl{cu_q, cu_k, total_q, B};
    const GShape sh{H, H / Hkv, hd, Tq, Tk, causal, 0, 1.0f / sqrtf((float)hd), drop_key, drop_thr, drop_scale};
    const GOutDrop od{out_drop_key, out_drop_thr, out_drop_scale};
    GATTN_DISPATCH(gattn_bwd_dq_kernel, hd, drop_thr, dim3(((Tq + 63) / 64) * H * B), dim3(256), 0, s, Q, K, V, DO, Ow, lse, delta_ws,
                   (bf16_t*)dq, dq_bs, dq_rs, sh, vl, od);
    GATTN_DISPATCH(gattn_bwd_dkv_kernel, hd, drop_thr, dim3(((Tk + 63) / 64) * Hkv * B), dim3(256), 0, s, Q, K, V, DO, lse, delta_ws,
                   (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, sh, vl, od);
    I2T_CHECK_LAUNCH("i2t_gq_attention_bwd");
    return I2T_OK;
}
