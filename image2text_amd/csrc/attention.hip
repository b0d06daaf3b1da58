// Fused attention forward / backward for gfx950, head_dim 64, bf16 operands, fp32 softmax and accumulation.
//
// Problem sizes on this path are short (Tk <= a few hundred), so the structure optimises for few, fully coalesced
// passes rather than deep pipelines: a workgroup = 4 waves = one 64-row tile of one (batch, head); 64-key tiles are
// staged into LDS once per workgroup and shared by the 4 waves.
//
// LDS image ("P160"): every 64 x 64 bf16 tile is stored [row][64] with a 160-byte row stride (128 B data + 32 B pad).
// With that stride BOTH access kinds this kernel needs are bank-conflict free on the 64-bank LDS:
//   * row reads  (ds_read_b128: 16 lanes read 16 consecutive rows at one 16-B column) for operands whose MFMA
//     reduction index runs along the row (Q.K^T over head_dim, dO.V^T over head_dim);
//   * transposed reads (ds_read_b64_tr_b16: 4 rows x 16 columns per 16 lanes) for operands whose reduction index is
//     the ROW (P.V over keys, dS.K over keys, dS^T.Q / P^T.dO over queries).
//
// MFMA orientation: scores are produced TRANSPOSED (D[key][q] = K . Q^T) so that a lane owns one query column
// (q = lane & 15) and 4 consecutive keys per 16-key subtile (rows 4*(lane>>4) + r).  Row max / row sum then need
// only 2 xor-shuffles (16, 32), and the exponentiated tile is directly the B operand of the next product
// (O^T[d][q] = V^T[d][key] . P^T[key][q]) with the key order permuted identically on the V^T side (the transposed
// LDS read takes any 4-row set) -- cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand".
#include "attention_common.h"
#include <stdlib.h>

namespace {

constexpr int TS = 160;                         // tile row stride in bytes
constexpr int TILE_BYTES = 64 * TS;             // 10240
constexpr float SCALE = 0.125f;                 // 1/sqrt(64)

// stage rows [r0, r0+64) x 64 columns of one (b, h) slice into a P160 tile; rows >= nrows are zero-filled
__device__ __forceinline__ void stage_tile(unsigned char* lds, const bf16_t* base, int rs, int r0, int nrows, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        int c = tid + 256 * u;
        int r = c >> 3, kc = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)(r0 + r) * rs + kc * 8);
        *reinterpret_cast<u32x4*>(lds + r * TS + kc * 16) = v;
    }
}

// The same staging split in two (issue the global loads of tile t+1 before computing on tile t, write them to LDS after
// the barrier that ends tile t): the load latency, which the one-piece form exposed once per tile, hides behind the MFMAs.
struct TileRegs {
    u32x4 v[2];
};
__device__ __forceinline__ void tile_load(TileRegs& t, const bf16_t* base, int rs, int r0, int nrows, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + 256 * u, r = c >> 3, kc = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r0 + r < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)(r0 + r) * rs + kc * 8);
        t.v[u] = v;
    }
}
__device__ __forceinline__ void tile_store(const TileRegs& t, unsigned char* lds, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + 256 * u, r = c >> 3, kc = c & 7;
        *reinterpret_cast<u32x4*>(lds + r * TS + kc * 16) = t.v[u];
    }
}

// row fragment: lane (g, i) <- tile[r0 + i][32 ks + 8 g .. +7]
__device__ __forceinline__ bf16x8 tile_row_frag(const unsigned char* lds, int r0, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    u32x4 v = *reinterpret_cast<const u32x4*>(lds + (r0 + i) * TS + (ks * 4 + g) * 16);
    return __builtin_bit_cast(bf16x8, v);
}

// transposed fragment for k-step s2 (32 rows) and column subtile c0 (16 cols): lane (g, i) <- tile[row(g,j)][c0 + i],
// row(g, j) = 32 s2 + 16 (j >> 2) + 4 g + (j & 3)   -- the row order produced by a transposed-score accumulator
__device__ __forceinline__ bf16x8 tile_tr_frag(const unsigned char* lds, int s2, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* a = lds + (32 * s2 + 4 * g + (i >> 2)) * TS + (c0 + 4 * (i & 3)) * 2;
    s16x4 lo = lds_read_tr16(a);
    s16x4 hi = lds_read_tr16(a + 16 * TS);
    s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// global row fragment: lane (g, i) <- M[row0 + i][32 ks + 8 g .. +7], zeros beyond nrows
__device__ __forceinline__ bf16x8 global_row_frag(const bf16_t* base, int rs, int row0, int nrows, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (row0 + i < nrows) v = *reinterpret_cast<const u32x4*>(base + (size_t)(row0 + i) * rs + ks * 32 + g * 8);
    return __builtin_bit_cast(bf16x8, v);
}

// ================================================================================================== forward
// All three kernels are VALU-bound (softmax + the dropout hash; ~4 VALU issue slots per MFMA slot), so the variants are
// compile-time: DROP (probability dropout on) and EVEN (TkMax % 4 == 0: a lane's 4 consecutive keys are the 4 bytes of exactly
// one hash, no per-lane alignment case).  The dropout scale 1/(1-p) and the 1/sqrt(d) factor of dS are applied once to
// the accumulators at the end instead of per probability.
template <bool DROP, bool EVEN>
__global__ __launch_bounds__(256) void attn_fwd_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, bf16_t* __restrict__ O,
                                                       long o_bs, int o_rs, float* __restrict__ lse, int H, int TqMax,
                                                       int TkMax, int causal, unsigned drop_key, unsigned drop_thr,
                                                       float drop_scale, VarLen vl) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* kt_lds = smem;
    unsigned char* vt_lds = smem + TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
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
    const bf16_t* qb = Q.p + qoff + h * 64;
    const bf16_t* kb = K.p + koff + h * 64;
    const bf16_t* vb = V.p + voff + h * 64;
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + li;                       // this lane's query row
    const int shift = Tk - Tq;                      // causal: key j visible iff j <= q + shift
    const bf16x8 qf0 = global_row_frag(qb, Q.rs, q0, Tq, 0, lane);
    const bf16x8 qf1 = global_row_frag(qb, Q.rs, q0, Tq, 1, lane);

    int last_key = Tk - 1;
    if (causal) last_key = min(last_key, qt * 64 + 63 + shift);
    const int nkt = last_key / 64 + 1;

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    float m = -INFINITY, l = 0.f;                   // running max (log2 domain) and this lane's partial row sum
    const int qlim = causal ? (min(qrow, Tq - 1) + shift) : (Tk - 1);
    const unsigned drow = (((unsigned)b * H + h) * TqMax + min(qrow, Tq - 1)) * (unsigned)TkMax;   // dropout index of (b, h, q, key 0)

    // A wave whose 16 query rows are all past Tq only helps staging; 16-key blocks past the last visible key of a tile are
    // skipped (T = 260 runs 5 key tiles, the last with 4 keys: whole-tile work there was ~30 % of the encoder's attention).
    const bool wave_on = q0 < Tq;
    TileRegs kr, vr;
    tile_load(kr, kb, K.rs, 0, Tk, tid);
    tile_load(vr, vb, V.rs, 0, Tk, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_store(kr, kt_lds, tid);
        tile_store(vr, vt_lds, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load(kr, kb, K.rs, (kt + 1) * 64, Tk, tid);
            tile_load(vr, vb, V.rs, (kt + 1) * 64, Tk, tid);
        }
        if (!wave_on) continue;
        const int nkj = min(4, (last_key - kt * 64) / 16 + 1);          // 16-key blocks of this tile that hold a visible key
        // every key of the tile visible to every row of this wave (wave-uniform): no per-element compare / select
        const bool full = kt * 64 + 63 < Tk && (!causal || kt * 64 + 63 <= q0 + shift);
        f32x4 s[4];                                                      // RAW scores; the 1/sqrt(d) log2(e) factor rides in the exp's fma
        float mx = -INFINITY;
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            f32x4 a = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
            if (kj < nkj) {
                a = f32x4{0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(kt_lds, kj * 16, 0, lane), qf0, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(kt_lds, kj * 16, 1, lane), qf1, a, 0, 0, 0);
                if (!full) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int key = kt * 64 + kj * 16 + 4 * g + r;
                        if (!(key <= qlim && key < Tk)) a[r] = -INFINITY;
                    }
                }
                mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
            }
            s[kj] = a;
        }
        mx = quad_max(mx) * (SCALE * LOG2E);                             // (-inf stays -inf; the factor is positive)
        const float m_new = fmaxf(m, mx);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m - m_safe);      // m = -inf -> 0
        float rs = 0.f;
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (kj >= nkj) {
                s[kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) {
                if constexpr (EVEN) dropout_keep4_even(drop_key, drow + kt * 64 + kj * 16 + 4 * g, drop_thr, keep);
                else dropout_keep4(drop_key, drow + kt * 64 + kj * 16 + 4 * g, drop_thr, keep);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float p = __builtin_amdgcn_exp2f(s[kj][r] * (SCALE * LOG2E) - m_safe);      // one fma + raw v_exp_f32
                rs += p;                               // the softmax denominator is dropout-free
                if constexpr (DROP) p = keep[r] ? p : 0.f;                                   // x drop_scale: once, on O
                s[kj][r] = p;
            }
        }
        l = l * alpha + rs;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] *= alpha;
        const bf16x8 p0 = pack_frag(s[0], s[1]);
        const bf16x8 p1 = pack_frag(s[2], s[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(vt_lds, 0, dt * 16, lane), p0, o[dt], 0, 0, 0);
            if (nkj > 2) o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(vt_lds, 1, dt * 16, lane), p1, o[dt], 0, 0, 0);
        }
    }
    l = quad_sum(l);
    const float inv = l > 0.f ? (DROP ? drop_scale : 1.0f) / l : 0.f;
    if (qrow < Tq) {
        bf16_t* op = O + ooff + (size_t)qrow * o_rs + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            u32x2 pk = {pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv), pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv)};
            *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = pk;
        }
        if (g == 0 && lse) lse[stat_base + qrow] = (m + log2f(l)) * LN2;
    }
}

// ================================================================================================== backward
// dQ: one workgroup per (q tile, h, b); loops over key tiles.  Scores transposed (lane owns a query column).
template <bool DROP, bool EVEN>
__global__ __launch_bounds__(256) void attn_bwd_dq_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O,
                                                          const float* __restrict__ lse, float* __restrict__ delta,
                                                          bf16_t* __restrict__ dQ, long dq_bs, int dq_rs, int H, int TqMax,
                                                          int TkMax, int causal, unsigned drop_key, unsigned drop_thr,
                                                          float drop_scale, VarLen vl, unsigned od_key, unsigned od_thr,
                                                          float od_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* kt_lds = smem;
    unsigned char* vt_lds = smem + TILE_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int qt, h, b;
    attn_block_coords((TqMax + 63) / 64, H, vl.nseq, qt, h, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, dooff = (size_t)b * dO.bs, dqoff = (size_t)b * dq_bs;
    size_t stat_base = ((size_t)b * H + h) * TqMax;
    if (vl.cu_q) {
        const int s0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - s0;
        qoff = (size_t)s0 * Q.rs; dooff = (size_t)s0 * dO.rs; dqoff = (size_t)s0 * dq_rs;
        stat_base = (size_t)h * vl.total_q + s0;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs;
    }
    if (qt * 64 >= Tq) return;
    const bf16_t* qb = Q.p + qoff + h * 64;
    const bf16_t* kb = K.p + koff + h * 64;
    const bf16_t* vb = V.p + voff + h * 64;
    const bf16_t* dob = dO.p + dooff + h * 64;
    const int q0 = qt * 64 + w * 16;
    const int qrow = q0 + li;
    const int shift = Tk - Tq;
    const bf16x8 qf0 = global_row_frag(qb, Q.rs, q0, Tq, 0, lane);
    const bf16x8 qf1 = global_row_frag(qb, Q.rs, q0, Tq, 1, lane);
    const bf16x8 df0 = global_row_frag(dob, dO.rs, q0, Tq, 0, lane);
    const bf16x8 df1 = global_row_frag(dob, dO.rs, q0, Tq, 1, lane);
    const int qc = min(qrow, Tq - 1);
    const float lse2 = lse[stat_base + qc] * LOG2E;
    // delta[q] = sum_d dO[q][d] O[q][d], computed here (the 4 lanes of a query column hold 16 dims each of dO already) and
    // stored for the dK/dV kernel that runs next on the stream -- the separate pass over O and dO is gone
    float dl;
    {
        const bf16_t* ob = O.p + (vl.cu_q ? (size_t)vl.cu_q[b] * O.rs : (size_t)b * O.bs) + h * 64;
        const bf16x8 of0 = global_row_frag(ob, O.rs, q0, Tq, 0, lane), of1 = global_row_frag(ob, O.rs, q0, Tq, 1, lane);
        float sdl = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sdl += (float)df0[e] * (float)of0[e] + (float)df1[e] * (float)of1[e];
        dl = quad_sum(sdl);
        if (g == 0 && qrow < Tq) delta[stat_base + qrow] = dl;
    }
    int last_key = Tk - 1;
    if (causal) last_key = min(last_key, qt * 64 + 63 + shift);
    const int nkt = last_key / 64 + 1;
    const int qlim = causal ? (qc + shift) : (Tk - 1);
    const unsigned drow = (((unsigned)b * H + h) * TqMax + qc) * (unsigned)TkMax;

    f32x4 acc[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dscale = DROP ? drop_scale : 1.f;

    const bool wave_on = q0 < Tq;
    TileRegs kr, vr;
    tile_load(kr, kb, K.rs, 0, Tk, tid);
    tile_load(vr, vb, V.rs, 0, Tk, tid);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();
        tile_store(kr, kt_lds, tid);
        tile_store(vr, vt_lds, tid);
        __syncthreads();
        if (kt + 1 < nkt) {
            tile_load(kr, kb, K.rs, (kt + 1) * 64, Tk, tid);
            tile_load(vr, vb, V.rs, (kt + 1) * 64, Tk, tid);
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
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(kt_lds, kj * 16, 0, lane), qf0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(kt_lds, kj * 16, 1, lane), qf1, a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(vt_lds, kj * 16, 0, lane), df0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(vt_lds, kj * 16, 1, lane), df1, dp, 0, 0, 0);
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) {
                if constexpr (EVEN) dropout_keep4_even(drop_key, drow + kt * 64 + kj * 16 + 4 * g, drop_thr, keep);
                else dropout_keep4(drop_key, drow + kt * 64 + kj * 16 + 4 * g, drop_thr, keep);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int key = kt * 64 + kj * 16 + 4 * g + r;
                float p = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - lse2);
                if (!full && !(key <= qlim && key < Tk)) p = 0.f;
                float dpr = dp[r];                     // gradient w.r.t. the dropped probabilities -> undo the mask
                if constexpr (DROP) dpr = keep[r] ? dpr : 0.f;
                ds[kj][r] = p * fmaf(dpr, dscale, -dl);              // x 1/sqrt(d): once, on dQ (a power of two: exact)
            }
        }
        const bf16x8 s0 = pack_frag(ds[0], ds[1]);
        const bf16x8 s1 = pack_frag(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {   // dQ^T[d][q] += K^T[d][key] . dS^T[key][q]
            acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(kt_lds, 0, dt * 16, lane), s0, acc[dt], 0, 0, 0);
            if (nkj > 2) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(kt_lds, 1, dt * 16, lane), s1, acc[dt], 0, 0, 0);
        }
    }
    if (qrow < Tq) {
        bf16_t* op = dQ + dqoff + (size_t)qrow * dq_rs + h * 64;
        // od_*: the per-token multiplier that scaled q in the forward (GEMM drop_mode 2, third 0), row = token index
        const unsigned grow = (unsigned)((vl.cu_q ? vl.cu_q[b] : b * TqMax) + qrow);
        const float f = (od_thr ? (dropout_keep(od_key, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            acc[dt] *= f;
            u32x2 pk = {pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3])};
            *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = pk;
        }
    }
}

// dK, dV: one workgroup per (key tile, h, b); each wave owns 16 keys and loops over 64-row query tiles.
// Scores NOT transposed here (D[q][key]: lane owns a key column), so P / dS are the B operands of
// dV^T[d][key] = dO^T[d][q] . P[q][key] and dK^T[d][key] = Q^T[d][q] . dS[q][key].
template <bool DROP, bool EVEN>
__global__ __launch_bounds__(256) void attn_bwd_dkv_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO,
                                                           const float* __restrict__ lse, const float* __restrict__ delta,
                                                           bf16_t* __restrict__ dK, long dk_bs, int dk_rs,
                                                           bf16_t* __restrict__ dV, long dv_bs, int dv_rs, int H, int TqMax,
                                                           int TkMax, int causal, unsigned drop_key, unsigned drop_thr,
                                                           float drop_scale, VarLen vl, unsigned od_key, unsigned od_thr,
                                                           float od_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES + 2 * 64 * 4];
    unsigned char* q_lds = smem;
    unsigned char* do_lds = smem + TILE_BYTES;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
    float* dl_lds = lse_lds + 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int kt, h, b;
    attn_block_coords((TkMax + 63) / 64, H, vl.nseq, kt, h, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, dooff = (size_t)b * dO.bs;
    size_t dkoff = (size_t)b * dk_bs, dvoff = (size_t)b * dv_bs;
    size_t stat_base = ((size_t)b * H + h) * TqMax;
    if (vl.cu_q) {
        const int s0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - s0;
        qoff = (size_t)s0 * Q.rs; dooff = (size_t)s0 * dO.rs;
        stat_base = (size_t)h * vl.total_q + s0;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs; dkoff = (size_t)s0 * dk_rs; dvoff = (size_t)s0 * dv_rs;
    }
    if (kt * 64 >= Tk) return;
    const bf16_t* qb = Q.p + qoff + h * 64;
    const bf16_t* kb = K.p + koff + h * 64;
    const bf16_t* vb = V.p + voff + h * 64;
    const bf16_t* dob = dO.p + dooff + h * 64;
    const int k0 = kt * 64 + w * 16;
    const int key = k0 + li;                         // this lane's key column
    const int shift = Tk - Tq;
    const bf16x8 kf0 = global_row_frag(kb, K.rs, k0, Tk, 0, lane);
    const bf16x8 kf1 = global_row_frag(kb, K.rs, k0, Tk, 1, lane);
    const bf16x8 vf0 = global_row_frag(vb, V.rs, k0, Tk, 0, lane);
    const bf16x8 vf1 = global_row_frag(vb, V.rs, k0, Tk, 1, lane);
    int first_q = 0;
    if (causal) first_q = max(0, kt * 64 - shift);   // first query row that can see any key of this tile
    const int qt0 = first_q / 64, nqt = (Tq + 63) / 64;

    f32x4 adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) adk[dt] = adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    const bool wave_on = k0 < Tk;
    const float dscale = DROP ? drop_scale : 1.f;
    TileRegs qr, dr;
    float lse_r = 0.f, dl_r = 0.f;
    if (qt0 < nqt) {
        tile_load(qr, qb, Q.rs, qt0 * 64, Tq, tid);
        tile_load(dr, dob, dO.rs, qt0 * 64, Tq, tid);
        if (tid < 64) {
            const int q = min(qt0 * 64 + tid, Tq - 1);
            lse_r = lse[stat_base + q] * LOG2E;
            dl_r = delta[stat_base + q];
        }
    }
    for (int qt = qt0; qt < nqt; ++qt) {
        __syncthreads();
        tile_store(qr, q_lds, tid);
        tile_store(dr, do_lds, tid);
        if (tid < 64) {
            lse_lds[tid] = lse_r;
            dl_lds[tid] = dl_r;
        }
        __syncthreads();
        if (qt + 1 < nqt) {
            tile_load(qr, qb, Q.rs, (qt + 1) * 64, Tq, tid);
            tile_load(dr, dob, dO.rs, (qt + 1) * 64, Tq, tid);
            if (tid < 64) {
                const int q = min((qt + 1) * 64 + tid, Tq - 1);
                lse_r = lse[stat_base + q] * LOG2E;
                dl_r = delta[stat_base + q];
            }
        }
        if (!wave_on) continue;
        const int nqj = min(4, (Tq - 1 - qt * 64) / 16 + 1);          // 16-row query blocks of this tile that exist
        // every (query, key) pair of this wave's 64 x 16 block valid and visible (wave-uniform)
        const bool full = qt * 64 + 63 < Tq && k0 + 15 < Tk && (!causal || k0 + 15 <= qt * 64 + shift);
        f32x4 p[4], ds[4];
#pragma unroll
        for (int qj = 0; qj < 4; ++qj) {
            if (qj >= nqj) {
                p[qj] = ds[qj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
            // swapped issue: D[q][key] with rows = q (from the LDS tile), cols = key (this lane's register fragment)
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(q_lds, qj * 16, 0, lane), kf0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(q_lds, qj * 16, 1, lane), kf1, a, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(do_lds, qj * 16, 0, lane), vf0, dp, 0, 0, 0);
            dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(do_lds, qj * 16, 1, lane), vf1, dp, 0, 0, 0);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_lds + qj * 16 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_lds + qj * 16 + 4 * g);
            bool keep4[4] = {true, true, true, true};
            if constexpr (DROP) {
                // This lane's 4 elements sit in 4 different query rows (one hash word each).  With TkMax % 4 == 0 the four lanes
                // of keys 4c .. 4c+3 (a lane quad: k0 is a multiple of 16) read the four BYTES of the same words: lane j of the
                // quad hashes row j, the words are broadcast inside the quad (DPP quad_perm: a VALU move each) and every lane
                // picks its own byte -- 1 hash + 4 moves instead of 4 hashes per lane.
                const unsigned rb = ((unsigned)b * H + h) * TqMax;
                if constexpr (EVEN) {
                    const int j = li & 3;
                    const int qm = min(qt * 64 + qj * 16 + 4 * g + j, Tq - 1);
                    const int mine = (int)dropout_hash(drop_key, ((rb + qm) * (unsigned)TkMax + (unsigned)key) >> 2);
                    const unsigned sh = 8u * (unsigned)j;
                    keep4[0] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;   // quad_perm [0,0,0,0]
                    keep4[1] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;   // [1,1,1,1]
                    keep4[2] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xaa, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;   // [2,2,2,2]
                    keep4[3] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xff, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;   // [3,3,3,3]
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        keep4[r] = dropout_keep(drop_key, (rb + min(qt * 64 + qj * 16 + 4 * g + r, Tq - 1)) * (unsigned)TkMax + key, drop_thr);
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                int q = qt * 64 + qj * 16 + 4 * g + r;
                float pv = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - l4[r]);
                if (!full && !((q < Tq) && (key < Tk) && (!causal || key <= q + shift))) pv = 0.f;
                float pd = pv, dpr = dp[r];
                if constexpr (DROP) {
                    const bool keep = keep4[r];
                    pd = keep ? pv : 0.f;                  // dV sees the dropped probabilities (x drop_scale: once, on dV)
                    dpr = keep ? dpr : 0.f;
                }
                p[qj][r] = pd;
                ds[qj][r] = pv * fmaf(dpr, dscale, -d4[r]);          // x 1/sqrt(d): once, on dK
            }
        }
        const bf16x8 p0 = pack_frag(p[0], p[1]), p1 = pack_frag(p[2], p[3]);
        const bf16x8 s0 = pack_frag(ds[0], ds[1]), s1 = pack_frag(ds[2], ds[3]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(do_lds, 0, dt * 16, lane), p0, adv[dt], 0, 0, 0);
            adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(q_lds, 0, dt * 16, lane), s0, adk[dt], 0, 0, 0);
            if (nqj > 2) {
                adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(do_lds, 1, dt * 16, lane), p1, adv[dt], 0, 0, 0);
                adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(q_lds, 1, dt * 16, lane), s1, adk[dt], 0, 0, 0);
            }
        }
    }
    if (key < Tk) {
        bf16_t* pk_ = dK + dkoff + (size_t)key * dk_rs + h * 64;
        bf16_t* pv_ = dV + dvoff + (size_t)key * dv_rs + h * 64;
        const unsigned grow = (unsigned)((vl.cu_k ? vl.cu_k[b] : b * TkMax) + key);     // thirds 1 (k) and 2 (v) of the fused c_attn
        const float fk = (od_thr ? (dropout_keep(od_key + 1u, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
        const float fv = (od_thr ? (dropout_keep(od_key + 2u, grow, od_thr) ? od_scale : 0.f) : 1.f) * dscale;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            adk[dt] *= fk;
            adv[dt] *= fv;
            u32x2 a = {pack_bf16x2(adk[dt][0], adk[dt][1]), pack_bf16x2(adk[dt][2], adk[dt][3])};
            u32x2 c = {pack_bf16x2(adv[dt][0], adv[dt][1]), pack_bf16x2(adv[dt][2], adv[dt][3])};
            *reinterpret_cast<u32x2*>(pk_ + dt * 16 + 4 * g) = a;
            *reinterpret_cast<u32x2*>(pv_ + dt * 16 + 4 * g) = c;
        }
    }
}

// ================================================================================================== single-tile backward
// Tq <= 64 and Tk <= 64 (the decoder: causal self-attention over a caption, cross-attention onto 64 memory rows): ONE workgroup per
// (sequence, head) produces dK, dV AND dQ.  The tiled pair above evaluates every (query, key) pair twice -- S, dP, the exponential and
// the dropout hash once in the dQ kernel and once in the dK/dV kernel -- and hands delta over through HBM.  Here the dK/dV
// orientation's pass is the only one: its dS tile goes to LDS as bf16 [key][query] (one ds_write_b64 per lane and 16 x 16 block), and
// after ONE barrier wave w contracts dQ^T[d][its 16 queries] = K^T[d][key] . dS^T[key][q] over the keys, both operands as transposed
// LDS reads in the same permuted key order.  16 x 16 blocks wholly above the causal diagonal are skipped (6 of 16 at T = 64).
template <bool DROP, bool EVEN>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(6, 8))) void attn_bwd1_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O, const float* __restrict__ lse,
                                                        bf16_t* __restrict__ dQ, long dq_bs, int dq_rs, bf16_t* __restrict__ dK, long dk_bs,
                                                        int dk_rs, bf16_t* __restrict__ dV, long dv_bs, int dv_rs, int H, int TqMax, int TkMax,
                                                        int causal, unsigned drop_key, unsigned drop_thr, float drop_scale, VarLen vl,
                                                        unsigned od_key, unsigned od_thr, float od_scale) {
    // two tiles: Q and dO for the first pass; after it K takes Q's place and dS^T takes dO's (20.5 KiB: 7 workgroups per CU by LDS --
    // with four tiles resident only 3 fit, and a workgroup's life is a chain of dependent memory round trips that needs company)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES + 2 * 64 * 4];
    unsigned char* q_lds = smem;
    unsigned char* do_lds = smem + TILE_BYTES;
    unsigned char* k_lds = smem;
    unsigned char* ds_lds = smem + TILE_BYTES;
    float* lse_lds = reinterpret_cast<float*>(smem + 2 * TILE_BYTES);
    float* dl_lds = lse_lds + 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    int tile_, h, b;
    attn_block_coords(1, H, vl.nseq, tile_, h, b);
    int Tq = TqMax, Tk = TkMax;
    size_t qoff = (size_t)b * Q.bs, koff = (size_t)b * K.bs, voff = (size_t)b * V.bs, dooff = (size_t)b * dO.bs, ooff = (size_t)b * O.bs;
    size_t dqoff = (size_t)b * dq_bs, dkoff = (size_t)b * dk_bs, dvoff = (size_t)b * dv_bs;
    size_t stat_base = ((size_t)b * H + h) * TqMax;
    if (vl.cu_q) {
        const int s0 = vl.cu_q[b];
        Tq = vl.cu_q[b + 1] - s0;
        qoff = (size_t)s0 * Q.rs; dooff = (size_t)s0 * dO.rs; ooff = (size_t)s0 * O.rs; dqoff = (size_t)s0 * dq_rs;
        stat_base = (size_t)h * vl.total_q + s0;
    }
    if (vl.cu_k) {
        const int s0 = vl.cu_k[b];
        Tk = vl.cu_k[b + 1] - s0;
        koff = (size_t)s0 * K.rs; voff = (size_t)s0 * V.rs; dkoff = (size_t)s0 * dk_rs; dvoff = (size_t)s0 * dv_rs;
    }
    if (Tk <= 0) return;                             // workgroup-uniform
    const int r0 = w * 16;                           // this wave's 16 keys (first pass) and 16 queries (second pass)
    if (Tq <= 0) {                                   // a packed sequence without a live query row (a caption with no label): its keys get zeros
        if (r0 + li < Tk) {
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                *reinterpret_cast<u32x2*>(dK + dkoff + (size_t)(r0 + li) * dk_rs + h * 64 + dt * 16 + 4 * g) = u32x2{0u, 0u};
                *reinterpret_cast<u32x2*>(dV + dvoff + (size_t)(r0 + li) * dv_rs + h * 64 + dt * 16 + 4 * g) = u32x2{0u, 0u};
            }
        }
        return;
    }
    const bf16_t* qb = Q.p + qoff + h * 64;
    const bf16_t* kb = K.p + koff + h * 64;
    const bf16_t* vb = V.p + voff + h * 64;
    const bf16_t* dob = dO.p + dooff + h * 64;
    const bf16_t* ob = O.p + ooff + h * 64;
    const int shift = Tk - Tq;
    stage_tile(q_lds, qb, Q.rs, 0, Tq, tid);
    stage_tile(do_lds, dob, dO.rs, 0, Tq, tid);
    TileRegs kr;                                     // the whole K tile for the second pass: in flight during the first
    tile_load(kr, kb, K.rs, 0, Tk, tid);
    {   // delta[q] = sum_d dO[q][d] O[q][d] and lse for the wave's 16 query rows (zeros past Tq)
        const bf16x8 df0 = global_row_frag(dob, dO.rs, r0, Tq, 0, lane), df1 = global_row_frag(dob, dO.rs, r0, Tq, 1, lane);
        const bf16x8 of0 = global_row_frag(ob, O.rs, r0, Tq, 0, lane), of1 = global_row_frag(ob, O.rs, r0, Tq, 1, lane);
        float sdl = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) sdl += (float)df0[e] * (float)of0[e] + (float)df1[e] * (float)of1[e];
        sdl = quad_sum(sdl);
        if (g == 0) {
            dl_lds[r0 + li] = sdl;
            lse_lds[r0 + li] = lse[stat_base + min(r0 + li, Tq - 1)] * LOG2E;
        }
    }
    const int key = r0 + li;                         // this lane's key column
    const bf16x8 kf0 = global_row_frag(kb, K.rs, r0, Tk, 0, lane), kf1 = global_row_frag(kb, K.rs, r0, Tk, 1, lane);
    const bf16x8 vf0 = global_row_frag(vb, V.rs, r0, Tk, 0, lane), vf1 = global_row_frag(vb, V.rs, r0, Tk, 1, lane);
    f32x4 adk[4], adv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) adk[dt] = adv[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float dscale = DROP ? drop_scale : 1.f;
    __syncthreads();
    {
        const bool wave_on = r0 < Tk;
        const int nqj = min(4, (Tq - 1) / 16 + 1);
        const bool full = 63 < Tq && r0 + 15 < Tk && (!causal || r0 + 15 <= shift);
        f32x4 p[4], ds[4];
        u32x2 dsp[4];
#pragma unroll
        for (int qj = 0; qj < 4; ++qj) {
            p[qj] = ds[qj] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (block past the last query, or wholly above the causal diagonal: its lowest key is invisible to its highest query)
            if (wave_on && qj < nqj && !(causal && r0 > qj * 16 + 15 + shift)) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(q_lds, qj * 16, 0, lane), kf0, a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(q_lds, qj * 16, 1, lane), kf1, a, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(do_lds, qj * 16, 0, lane), vf0, dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_row_frag(do_lds, qj * 16, 1, lane), vf1, dp, 0, 0, 0);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_lds + qj * 16 + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_lds + qj * 16 + 4 * g);
                bool keep4[4] = {true, true, true, true};
                if constexpr (DROP) {      // (as in attn_bwd_dkv_kernel: one hash per lane, its words shared inside the lane quad)
                    const unsigned rb = ((unsigned)b * H + h) * TqMax;
                    if constexpr (EVEN) {
                        const int j = li & 3;
                        const int qm = min(qj * 16 + 4 * g + j, Tq - 1);
                        const int mine = (int)dropout_hash(drop_key, ((rb + qm) * (unsigned)TkMax + (unsigned)key) >> 2);
                        const unsigned sh = 8u * (unsigned)j;
                        keep4[0] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[1] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[2] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xaa, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[3] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xff, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            keep4[r] = dropout_keep(drop_key, (rb + min(qj * 16 + 4 * g + r, Tq - 1)) * (unsigned)TkMax + key, drop_thr);
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int q = qj * 16 + 4 * g + r;
                    float pv = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - l4[r]);
                    if (!full && !((q < Tq) && (key < Tk) && (!causal || key <= q + shift))) pv = 0.f;
                    float pd = pv, dpr = dp[r];
                    if constexpr (DROP) {
                        pd = keep4[r] ? pv : 0.f;
                        dpr = keep4[r] ? dpr : 0.f;
                    }
                    p[qj][r] = pd;
                    ds[qj][r] = pv * fmaf(dpr, dscale, -d4[r]);
                }
            }
            dsp[qj] = u32x2{pack_bf16x2(ds[qj][0], ds[qj][1]), pack_bf16x2(ds[qj][2], ds[qj][3])};
        }
        if (wave_on) {
            const bf16x8 p0 = pack_frag(p[0], p[1]), p1 = pack_frag(p[2], p[3]);
            const bf16x8 s0 = pack_frag(ds[0], ds[1]), s1 = pack_frag(ds[2], ds[3]);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(do_lds, 0, dt * 16, lane), p0, adv[dt], 0, 0, 0);
                adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(q_lds, 0, dt * 16, lane), s0, adk[dt], 0, 0, 0);
                if (nqj > 2) {
                    adv[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(do_lds, 1, dt * 16, lane), p1, adv[dt], 0, 0, 0);
                    adk[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(q_lds, 1, dt * 16, lane), s1, adk[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();                             // every wave is done with Q and dO
        tile_store(kr, k_lds, tid);
#pragma unroll
        for (int qj = 0; qj < 4; ++qj)               // dS^T[key][query]: this lane's 4 consecutive queries of its key row
            *reinterpret_cast<u32x2*>(ds_lds + (r0 + li) * TS + (qj * 16 + 4 * g) * 2) = dsp[qj];
    }
    __syncthreads();
    if (r0 < Tq) {      // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] for the wave's 16 queries
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int last_key = causal ? min(Tk - 1, r0 + 15 + shift) : Tk - 1;      // (keys beyond it hold zeros in dS anyway)
        const bf16x8 b0 = tile_tr_frag(ds_lds, 0, r0, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(k_lds, 0, dt * 16, lane), b0, acc[dt], 0, 0, 0);
        if (last_key >= 32) {
            const bf16x8 b1 = tile_tr_frag(ds_lds, 1, r0, lane);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tile_tr_frag(k_lds, 1, dt * 16, lane), b1, acc[dt], 0, 0, 0);
        }
        const int qrow = r0 + li;
        if (qrow < Tq) {
            bf16_t* op = dQ + dqoff + (size_t)qrow * dq_rs + h * 64;
            const unsigned grow = (unsigned)((vl.cu_q ? vl.cu_q[b] : b * TqMax) + qrow);
            const float f = (od_thr ? (dropout_keep(od_key, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                acc[dt] *= f;
                const u32x2 pk = {pack_bf16x2(acc[dt][0], acc[dt][1]), pack_bf16x2(acc[dt][2], acc[dt][3])};
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = pk;
            }
        }
    }
    if (key < Tk) {
        bf16_t* pk_ = dK + dkoff + (size_t)key * dk_rs + h * 64;
        bf16_t* pv_ = dV + dvoff + (size_t)key * dv_rs + h * 64;
        const unsigned grow = (unsigned)((vl.cu_k ? vl.cu_k[b] : b * TkMax) + key);
        const float fk = (od_thr ? (dropout_keep(od_key + 1u, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
        const float fv = (od_thr ? (dropout_keep(od_key + 2u, grow, od_thr) ? od_scale : 0.f) : 1.f) * dscale;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            adk[dt] *= fk;
            adv[dt] *= fv;
            const u32x2 a = {pack_bf16x2(adk[dt][0], adk[dt][1]), pack_bf16x2(adk[dt][2], adk[dt][3])};
            const u32x2 c = {pack_bf16x2(adv[dt][0], adv[dt][1]), pack_bf16x2(adv[dt][2], adv[dt][3])};
            *reinterpret_cast<u32x2*>(pk_ + dt * 16 + 4 * g) = a;
            *reinterpret_cast<u32x2*>(pv_ + dt * 16 + 4 * g) = c;
        }
    }
}

// ================================================================================================== v2: resident operands
// The kernels above pay per workgroup ITERATION (two barriers + a staged tile each): at T = 260 a (sequence, head) pair costs 5 x 5
// of them for 3 % more work than T = 256, every wave re-reads the whole K and V tile from LDS for only 16 query rows (the LDS
// array, not the VALU, sets the pace), and each pair's K / V are fetched 5 times.  v2 (dense, non-causal, up to 288 keys / 304
// queries: the encoders): ONE workgroup per pair, its K and V (or Q and dO) staged ONCE into LDS and resident -- 128-byte rows,
// 16-byte chunks XOR-swizzled by (row & 7): conflict-free for the row reads and for the transposed reads, 2 x 288 rows = 72 KiB,
// two workgroups per CU -- no barrier after the staging one.  A wave owns up to 4 sixteen-row blocks of the other operand and
// walks the resident one in 64-row steps, so an LDS fragment serves 4 MFMAs instead of 1.  The row blocks that do not divide by
// the 4 waves (T = 260: block 17 holds 4 rows) are shared by all waves ACROSS the resident operand -- every wave takes a quarter
// of its rows -- and combined through LDS (split softmax: (m, l, O) partials), so no wave carries a fifth block.
constexpr int V2_MAXROWS = 288, V2_RB = 128, V2_NQB = 4;
constexpr int V2_MAX_OTHER = 16 * (4 * V2_NQB + 3);                    // 304 rows of the per-wave operand

__device__ __forceinline__ void v2_stage2(unsigned char* la, unsigned char* lb, const bf16_t* a, int a_rs, const bf16_t* b, int b_rs,
                                          int nrows, int rows_pad, int tid) {
    const int total = rows_pad * 8;
    for (int c0 = tid; c0 < total; c0 += 1024) {
        u32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 256 * u, r = c >> 3, kc = c & 7;
            va[u] = vb[u] = u32x4{0u, 0u, 0u, 0u};
            if (c < total && r < nrows) {
                va[u] = *reinterpret_cast<const u32x4*>(a + (size_t)r * a_rs + kc * 8);
                vb[u] = *reinterpret_cast<const u32x4*>(b + (size_t)r * b_rs + kc * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 256 * u, r = c >> 3, kc = c & 7;
            if (c < total) {
                const int off = r * V2_RB + ((kc ^ (r & 7)) << 4);
                *reinterpret_cast<u32x4*>(la + off) = va[u];
                *reinterpret_cast<u32x4*>(lb + off) = vb[u];
            }
        }
    }
}
// row fragment: lane (g, i) <- rows r0 + i, elements 32 ks + 8 g .. + 7
__device__ __forceinline__ bf16x8 v2_row_frag(const unsigned char* lds, int r0, int ks, int lane) {
    const int g = lane >> 4, r = r0 + (lane & 15);
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(lds + r * V2_RB + (((ks * 4 + g) ^ (r & 7)) << 4)));
}
// transposed fragment: lane (g, i) <- column c0 + i of rows rbase + 16 (j >> 2) + 4 g + (j & 3), j = 0 .. 7
__device__ __forceinline__ bf16x8 v2_tr_frag(const unsigned char* lds, int rbase, int c0, int lane) {
    const int g = lane >> 4, i = lane & 15;
    const int r = rbase + 4 * g + (i >> 2), cb = (c0 + 4 * (i & 3)) * 2;
    const unsigned char* a = lds + r * V2_RB + (((cb >> 4) ^ (r & 7)) << 4) + (cb & 15);
    const s16x4 lo = lds_read_tr16(a), hi = lds_read_tr16(a + 16 * V2_RB);       // (r + 16) & 7 == r & 7
    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    return __builtin_bit_cast(bf16x8, v);
}

// One step of CNT query blocks against NKJ (2 or 4) 16-key blocks starting at block cb, everything a compile-time constant: one
// basic block the scheduler can interleave (runtime `qb < cnt` / `kj < nkj` tests compiled into ~160 scalar branches and cost 40 %).
// MASK: keys >= Tk get -inf (the last step of a range; an odd block count is rounded up to the next even one: rows exist to a
// multiple of 32 and V's are zero there).
template <bool DROP, int CNT, int NKJ, bool MASK, int NA>
__device__ __forceinline__ void v2_fwd_step(const unsigned char* k_lds, const unsigned char* v_lds, const bf16x8 (&qf)[NA][2], int cb, int Tk,
                                            const unsigned (&drow)[NA], unsigned drop_key, unsigned drop_thr, f32x4 (&o)[NA][4],
                                            float (&m)[NA], float (&l)[NA], int lane) {
    const int g = lane >> 4;
    f32x4 s[CNT][NKJ];
#pragma unroll
    for (int kj = 0; kj < NKJ; ++kj) {
        const bf16x8 k0 = v2_row_frag(k_lds, (cb + kj) * 16, 0, lane), k1 = v2_row_frag(k_lds, (cb + kj) * 16, 1, lane);
#pragma unroll
        for (int qb = 0; qb < CNT; ++qb) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[qb][0], a, 0, 0, 0);
            s[qb][kj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[qb][1], a, 0, 0, 0);
        }
    }
    bf16x8 pf[CNT][NKJ / 2];
#pragma unroll
    for (int qb = 0; qb < CNT; ++qb) {
        if constexpr (MASK) {
#pragma unroll
            for (int kj = 0; kj < NKJ; ++kj)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if ((cb + kj) * 16 + 4 * g + r >= Tk) s[qb][kj][r] = -INFINITY;
        }
        float mx = fmaxf(fmaxf(s[qb][0][0], s[qb][0][1]), fmaxf(s[qb][0][2], s[qb][0][3]));
#pragma unroll
        for (int kj = 1; kj < NKJ; ++kj) mx = fmaxf(fmaxf(mx, fmaxf(s[qb][kj][0], s[qb][kj][1])), fmaxf(s[qb][kj][2], s[qb][kj][3]));
        mx = quad_max(mx) * (SCALE * LOG2E);
        float m_new = fmaxf(m[qb], mx);
        const float m_ref = (MASK && m_new == -INFINITY) ? 0.f : m_new;   // (only a fully masked step with no history can be -inf)
        const float alpha = __builtin_amdgcn_exp2f(m[qb] - m_ref);
        float rs = 0.f;
#pragma unroll
        for (int kj = 0; kj < NKJ; ++kj) {
            bool keep[4] = {true, true, true, true};
            if constexpr (DROP) dropout_keep4_even(drop_key, drow[qb] + (cb + kj) * 16 + 4 * g, drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = __builtin_amdgcn_exp2f(s[qb][kj][r] * (SCALE * LOG2E) - m_ref);
                rs += pr;
                if constexpr (DROP) pr = keep[r] ? pr : 0.f;
                s[qb][kj][r] = pr;
            }
        }
        l[qb] = l[qb] * alpha + rs;
        m[qb] = m_new;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qb][dt] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < NKJ / 2; ++s2) pf[qb][s2] = pack_frag(s[qb][2 * s2], s[qb][2 * s2 + 1]);
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
#pragma unroll
        for (int s2 = 0; s2 < NKJ / 2; ++s2) {
            const bf16x8 vf = v2_tr_frag(v_lds, cb * 16 + 32 * s2, dt * 16, lane);
#pragma unroll
            for (int qb = 0; qb < CNT; ++qb) o[qb][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf[qb][s2], o[qb][dt], 0, 0, 0);
        }
    }
}

// CNT query blocks against the resident keys of blocks [kb0, kb1) (kb0 even): unmasked 64-key steps while every key is valid, then
// at most two masked ones.  FAST = false (the remainder phase): masked steps only, fewer instantiations.
template <bool DROP, int CNT, bool FAST, int NA>
__device__ __forceinline__ void v2_fwd_walk(const unsigned char* k_lds, const unsigned char* v_lds, const bf16x8 (&qf)[NA][2], int kb0, int kb1,
                                            int Tk, const unsigned (&drow)[NA], unsigned drop_key, unsigned drop_thr, f32x4 (&o)[NA][4],
                                            float (&m)[NA], float (&l)[NA], int lane) {
    int cb = kb0;
    if constexpr (FAST)
        for (; cb + 4 <= kb1 && (cb + 4) * 16 <= Tk; cb += 4)
            v2_fwd_step<DROP, CNT, 4, false, NA>(k_lds, v_lds, qf, cb, Tk, drow, drop_key, drop_thr, o, m, l, lane);
    for (; cb + 2 < kb1; cb += 4) v2_fwd_step<DROP, CNT, 4, true, NA>(k_lds, v_lds, qf, cb, Tk, drow, drop_key, drop_thr, o, m, l, lane);
    if (cb < kb1) v2_fwd_step<DROP, CNT, 2, true, NA>(k_lds, v_lds, qf, cb, Tk, drow, drop_key, drop_thr, o, m, l, lane);
}

template <bool DROP, int PER>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, bf16_t* __restrict__ O, long o_bs, int o_rs,
                                                           float* __restrict__ lse, int H, int Tq, int Tk, unsigned drop_key,
                                                           unsigned drop_thr, float drop_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * V2_MAXROWS * V2_RB];
    unsigned char* k_lds = smem;
    unsigned char* v_lds = smem + V2_MAXROWS * V2_RB;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const bf16_t* qb_ = Q.p + (size_t)b * Q.bs + h * 64;
    bf16_t* ob = O + (size_t)b * o_bs + h * 64;
    const size_t stat_base = ((size_t)b * H + h) * Tq;
    const int rows_pad = (Tk + 31) & ~31, nkb = (Tk + 15) >> 4;
    const int nqb = (Tq + 15) >> 4, rem = nqb & 3;
    constexpr int per = PER;                                             // == nqb >> 2 (the host picks the instantiation)
    // this wave's query fragments travel while the workgroup stages K and V
    bf16x8 qf[V2_NQB][2];
    unsigned drow[V2_NQB];
#pragma unroll
    for (int qb = 0; qb < V2_NQB; ++qb) {
        const int q0 = (w * per + qb) * 16;
        if (qb < per) {
            qf[qb][0] = global_row_frag(qb_, Q.rs, q0, Tq, 0, lane);
            qf[qb][1] = global_row_frag(qb_, Q.rs, q0, Tq, 1, lane);
        }
        drow[qb] = (((unsigned)b * H + h) * Tq + min(q0 + li, Tq - 1)) * (unsigned)Tk;
    }
    // ... and so do the fragments of the remaining 1-3 blocks every wave shares (requested after the main walk they cost each pair one
    // exposed global-load latency: T = 260 ran 40 % over T = 256 for 1.5 % more work)
    bf16x8 rq[3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        rq[r][0] = rq[r][1] = bf16x8{0, 0, 0, 0, 0, 0, 0, 0};
        if (r < rem) {
            rq[r][0] = global_row_frag(qb_, Q.rs, (4 * per + r) * 16, Tq, 0, lane);
            rq[r][1] = global_row_frag(qb_, Q.rs, (4 * per + r) * 16, Tq, 1, lane);
        }
    }
    v2_stage2(k_lds, v_lds, K.p + (size_t)b * K.bs + h * 64, K.rs, V.p + (size_t)b * V.bs + h * 64, V.rs, Tk, rows_pad, tid);
    __syncthreads();
    f32x4 o[V2_NQB][4];
    float m[V2_NQB], l[V2_NQB];
#pragma unroll
    for (int qb = 0; qb < V2_NQB; ++qb) {
        m[qb] = -INFINITY; l[qb] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[qb][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    v2_fwd_walk<DROP, PER, true, V2_NQB>(k_lds, v_lds, qf, 0, nkb, Tk, drow, drop_key, drop_thr, o, m, l, lane);
#pragma unroll
    for (int qb = 0; qb < V2_NQB; ++qb) {
        if (qb >= per) continue;
        const int qrow = (w * per + qb) * 16 + li;
        const float lt = quad_sum(l[qb]);
        const float inv = lt > 0.f ? (DROP ? drop_scale : 1.0f) / lt : 0.f;
        if (qrow < Tq) {
            bf16_t* op = ob + (size_t)qrow * o_rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = u32x2{pack_bf16x2(o[qb][dt][0] * inv, o[qb][dt][1] * inv),
                                                                       pack_bf16x2(o[qb][dt][2] * inv, o[qb][dt][3] * inv)};
            if (g == 0 && lse) lse[stat_base + qrow] = (m[qb] + log2f(lt)) * LN2;
        }
    }
    if (rem == 0) return;                                               // workgroup-uniform
    // ---- the remaining 1-3 query blocks: every wave runs them against its quarter of the keys, partials combined through LDS
    unsigned rdrow[3];
    f32x4 ro[3][4];
    float rm[3], rl[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int q0 = (4 * per + r) * 16;
        rdrow[r] = (((unsigned)b * H + h) * Tq + min(q0 + li, Tq - 1)) * (unsigned)Tk;
        rm[r] = -INFINITY; rl[r] = 0.f;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) ro[r][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int kq = 2 * ((nkb + 7) >> 3);       // quarters in units of 32 rows: the transposed V reads take 32-row groups from the step's base
    {
        const int k0 = min(w * kq, nkb), k1 = min((w + 1) * kq, nkb);
        if (rem == 1) v2_fwd_walk<DROP, 1, false, 3>(k_lds, v_lds, rq, k0, k1, Tk, rdrow, drop_key, drop_thr, ro, rm, rl, lane);
        else if (rem == 2) v2_fwd_walk<DROP, 2, false, 3>(k_lds, v_lds, rq, k0, k1, Tk, rdrow, drop_key, drop_thr, ro, rm, rl, lane);
        else v2_fwd_walk<DROP, 3, false, 3>(k_lds, v_lds, rq, k0, k1, Tk, rdrow, drop_key, drop_thr, ro, rm, rl, lane);
    }
    __syncthreads();                                                    // every wave is done with K and V: the region becomes scratch
    float* part = reinterpret_cast<float*>(smem);                       // [rem][4 waves][16 q x 64 d | m[16] | l[16]]
    constexpr int PSZ = 16 * 64 + 32;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (r >= rem) continue;
        float* pp = part + (size_t)(r * 4 + w) * PSZ;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(pp + li * 64 + dt * 16 + 4 * g) = ro[r][dt];
        const float lt = quad_sum(rl[r]);
        if (g == 0) {
            pp[1024 + li] = rm[r];
            pp[1040 + li] = lt;
        }
    }
    __syncthreads();
    if (w < rem) {
        const int r = w, qrow = (4 * per + r) * 16 + li;
        float mw[4], M = -INFINITY;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            mw[u] = part[(size_t)(r * 4 + u) * PSZ + 1024 + li];
            M = fmaxf(M, mw[u]);
        }
        const float Ms = (M == -INFINITY) ? 0.f : M;
        float L = 0.f;
        f32x4 acc[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float* pp = part + (size_t)(r * 4 + u) * PSZ;
            const float f = __builtin_amdgcn_exp2f(mw[u] - Ms);
            L += pp[1040 + li] * f;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) acc[dt] += *reinterpret_cast<const f32x4*>(pp + li * 64 + dt * 16 + 4 * g) * f;
        }
        const float inv = L > 0.f ? (DROP ? drop_scale : 1.0f) / L : 0.f;
        if (qrow < Tq) {
            bf16_t* op = ob + (size_t)qrow * o_rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt)
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = u32x2{pack_bf16x2(acc[dt][0] * inv, acc[dt][1] * inv),
                                                                       pack_bf16x2(acc[dt][2] * inv, acc[dt][3] * inv)};
            if (g == 0 && lse) lse[stat_base + qrow] = (M + log2f(L)) * LN2;
        }
    }
}

// ================================================================================================== v2 backward (fused)
// One workgroup (8 waves) per (sequence, head) with Q, K, V and dO ALL resident in LDS (4 x 288 rows x 128 B = 144 KiB, one workgroup
// per CU): the tiled pair above reads every operand 5 times per pair at T = 260 and stages it behind two barriers per tile; here each
// operand is read ONCE, delta = rowsum(dO . O) is computed while staging, and after the staging barrier no wave waits for another:
//   phase A (dQ):      a wave owns up to 3 sixteen-query blocks (Q / dO fragments in registers) and walks the keys in 64-row steps --
//                      S^T = K Q^T, dP^T = V dO^T (row fragments of K and V serve all its blocks), dS^T, dQ^T += K^T dS^T;
//   phase B (dK, dV):  a wave owns up to 3 sixteen-key blocks (K / V fragments in registers) and walks the queries in 32-row steps --
//                      S = Q K^T, dP = dO V^T, dV^T += dO^T P, dK^T += Q^T dS (lane owns a key column).
// Blocks are dealt round-robin, phase B in the opposite direction, so the wave that got the extra query block (T = 260: 17 blocks
// on 8 waves) is not the one that gets the extra key block; both phases only READ the resident operands, so a wave moves on to
// phase B without a barrier.
constexpr int B2_MAXBLK = 3;                                            // blocks per wave and phase: 8 x 3 x 16 = 384 rows >= V2_MAXROWS

template <bool DROP, int CNT>
__device__ __forceinline__ void b2_phase_dq(const unsigned char* q_lds, const unsigned char* k_lds, const unsigned char* v_lds,
                                            const unsigned char* do_lds, const float* lse_l, const float* dl_l, int w, int Tq, int Tk,
                                            unsigned drow_base, unsigned drop_key, unsigned drop_thr, float dscale, bf16_t* dqb, int dq_rs,
                                            unsigned od_key, unsigned od_thr, float od_scale, unsigned grow_base, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const int nkb = (Tk + 15) >> 4;
    bf16x8 qf[CNT][2], df[CNT][2];
    float lse2[CNT], dl[CNT];
    unsigned drow[CNT];
    f32x4 acc[CNT][4];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int q0 = (w + 8 * c) * 16, qrow = min(q0 + li, Tq - 1);
        qf[c][0] = v2_row_frag(q_lds, q0, 0, lane); qf[c][1] = v2_row_frag(q_lds, q0, 1, lane);
        df[c][0] = v2_row_frag(do_lds, q0, 0, lane); df[c][1] = v2_row_frag(do_lds, q0, 1, lane);
        lse2[c] = lse_l[q0 + li]; dl[c] = dl_l[q0 + li];
        drow[c] = drow_base + (unsigned)qrow * (unsigned)Tk;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) acc[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int cb = 0; cb < nkb; cb += 4) {
        const int nkj = min(4, nkb - cb);
        const bool full = (cb + 4) * 16 <= Tk;
        f32x4 ds[CNT][4];
#pragma unroll
        for (int kj = 0; kj < 4; ++kj) {
            if (kj >= nkj) {
#pragma unroll
                for (int c = 0; c < CNT; ++c) ds[c][kj] = f32x4{0.f, 0.f, 0.f, 0.f};
                continue;
            }
            const bf16x8 k0 = v2_row_frag(k_lds, (cb + kj) * 16, 0, lane), k1 = v2_row_frag(k_lds, (cb + kj) * 16, 1, lane);
            const bf16x8 v0 = v2_row_frag(v_lds, (cb + kj) * 16, 0, lane), v1 = v2_row_frag(v_lds, (cb + kj) * 16, 1, lane);
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k0, qf[c][0], a, 0, 0, 0);
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(k1, qf[c][1], a, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v0, df[c][0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(v1, df[c][1], dp, 0, 0, 0);
                bool keep[4] = {true, true, true, true};
                if constexpr (DROP) dropout_keep4_even(drop_key, drow[c] + (cb + kj) * 16 + 4 * g, drop_thr, keep);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pr = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - lse2[c]);       // (lse2 = +inf for rows past Tq: p = 0)
                    if (!full && (cb + kj) * 16 + 4 * g + r >= Tk) pr = 0.f;
                    float dpr = dp[r];
                    if constexpr (DROP) dpr = keep[r] ? dpr : 0.f;
                    ds[c][kj][r] = pr * fmaf(dpr, dscale, -dl[c]);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 t0 = v2_tr_frag(k_lds, cb * 16, dt * 16, lane);
#pragma unroll
            for (int c = 0; c < CNT; ++c) acc[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, pack_frag(ds[c][0], ds[c][1]), acc[c][dt], 0, 0, 0);
            if (nkj > 2) {
                const bf16x8 t1 = v2_tr_frag(k_lds, cb * 16 + 32, dt * 16, lane);
#pragma unroll
                for (int c = 0; c < CNT; ++c) acc[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, pack_frag(ds[c][2], ds[c][3]), acc[c][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int qrow = (w + 8 * c) * 16 + li;
        if (qrow < Tq) {
            const float f = (od_thr ? (dropout_keep(od_key, grow_base + (unsigned)qrow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
            bf16_t* op = dqb + (size_t)qrow * dq_rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 o = acc[c][dt] * f;
                *reinterpret_cast<u32x2*>(op + dt * 16 + 4 * g) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
    }
}

template <bool DROP, int CNT>
__device__ __forceinline__ void b2_phase_dkv(const unsigned char* q_lds, const unsigned char* k_lds, const unsigned char* v_lds,
                                             const unsigned char* do_lds, const float* lse_l, const float* dl_l, int w, int Tq, int Tk,
                                             unsigned drow_base, unsigned drop_key, unsigned drop_thr, float dscale, bf16_t* dkb, int dk_rs,
                                             bf16_t* dvb, int dv_rs, unsigned od_key, unsigned od_thr, float od_scale, unsigned grow_base,
                                             int lane) {
    const int g = lane >> 4, li = lane & 15;
    const int nqb = (Tq + 15) >> 4;
    bf16x8 kf[CNT][2], vf[CNT][2];
    f32x4 adk[CNT][4], adv[CNT][4];
    int key[CNT];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int k0 = ((7 - w) + 8 * c) * 16;                               // (dealt from the other end than phase A's query blocks)
        key[c] = k0 + li;
        kf[c][0] = v2_row_frag(k_lds, k0, 0, lane); kf[c][1] = v2_row_frag(k_lds, k0, 1, lane);
        vf[c][0] = v2_row_frag(v_lds, k0, 0, lane); vf[c][1] = v2_row_frag(v_lds, k0, 1, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) adk[c][dt] = adv[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    for (int qb = 0; qb < nqb; qb += 2) {                                    // 32 queries per step (rows past Tq: zero Q / dO rows, lse = +inf)
        f32x4 pp[CNT][2], ds[CNT][2];
#pragma unroll
        for (int qj = 0; qj < 2; ++qj) {
            const int q0 = (qb + qj) * 16;
            const bf16x8 q0f = v2_row_frag(q_lds, q0, 0, lane), q1f = v2_row_frag(q_lds, q0, 1, lane);
            const bf16x8 d0f = v2_row_frag(do_lds, q0, 0, lane), d1f = v2_row_frag(do_lds, q0, 1, lane);
            const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_l + q0 + 4 * g);
            const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_l + q0 + 4 * g);
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[c][0], a, 0, 0, 0);      // D[q][key]: rows q = 4 g + e, column key = li
                a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[c][1], a, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[c][0], dp, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[c][1], dp, 0, 0, 0);
                bool keep4[4] = {true, true, true, true};
                if constexpr (DROP) {      // a lane's 4 elements sit in 4 query rows: the quad of keys 4c..4c+3 shares one hash word per row (DPP)
                    const int j = li & 3;
                    const int qm = min(q0 + 4 * g + j, Tq - 1);
                    const int mine = (int)dropout_hash(drop_key, (drow_base + (unsigned)qm * (unsigned)Tk + (unsigned)key[c]) >> 2);
                    const unsigned sh = 8u * (unsigned)j;
                    keep4[0] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                    keep4[1] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                    keep4[2] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xaa, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                    keep4[3] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xff, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                }
                const bool kvalid = key[c] < Tk;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float pv = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - l4[r]);
                    pv = kvalid ? pv : 0.f;
                    float pd = pv, dpr = dp[r];
                    if constexpr (DROP) {
                        pd = keep4[r] ? pv : 0.f;
                        dpr = keep4[r] ? dpr : 0.f;
                    }
                    pp[c][qj][r] = pd;
                    ds[c][qj][r] = pv * fmaf(dpr, dscale, -d4[r]);
                }
            }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const bf16x8 dot = v2_tr_frag(do_lds, qb * 16, dt * 16, lane), qt = v2_tr_frag(q_lds, qb * 16, dt * 16, lane);
#pragma unroll
            for (int c = 0; c < CNT; ++c) {
                adv[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pack_frag(pp[c][0], pp[c][1]), adv[c][dt], 0, 0, 0);
                adk[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pack_frag(ds[c][0], ds[c][1]), adk[c][dt], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        if (key[c] < Tk) {
            const unsigned grow = grow_base + (unsigned)key[c];
            const float fk = (od_thr ? (dropout_keep(od_key + 1u, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
            const float fv = (od_thr ? (dropout_keep(od_key + 2u, grow, od_thr) ? od_scale : 0.f) : 1.f) * dscale;
            bf16_t* pk_ = dkb + (size_t)key[c] * dk_rs;
            bf16_t* pv_ = dvb + (size_t)key[c] * dv_rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 a = adk[c][dt] * fk, b = adv[c][dt] * fv;
                *reinterpret_cast<u32x2*>(pk_ + dt * 16 + 4 * g) = u32x2{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
                *reinterpret_cast<u32x2*>(pv_ + dt * 16 + 4 * g) = u32x2{pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
            }
        }
    }
}

__device__ __forceinline__ void b2_stage2(unsigned char* la, unsigned char* lb, const bf16_t* a, int a_rs, const bf16_t* b, int b_rs,
                                          int nrows, int rows_pad, int tid) {      // v2_stage2 for 512 threads
    const int total = rows_pad * 8;
    for (int c0 = tid; c0 < total; c0 += 2048) {
        u32x4 va[4], vb[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 512 * u, r = c >> 3, kc = c & 7;
            va[u] = vb[u] = u32x4{0u, 0u, 0u, 0u};
            if (c < total && r < nrows) {
                va[u] = *reinterpret_cast<const u32x4*>(a + (size_t)r * a_rs + kc * 8);
                vb[u] = *reinterpret_cast<const u32x4*>(b + (size_t)r * b_rs + kc * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int c = c0 + 512 * u, r = c >> 3, kc = c & 7;
            if (c < total) {
                const int off = r * V2_RB + ((kc ^ (r & 7)) << 4);
                *reinterpret_cast<u32x4*>(la + off) = va[u];
                *reinterpret_cast<u32x4*>(lb + off) = vb[u];
            }
        }
    }
}

template <bool DROP>
__global__ __launch_bounds__(512, 1) void attn_bwd2_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dQ, long dq_bs, int dq_rs, bf16_t* __restrict__ dK, long dk_bs,
                                                           int dk_rs, bf16_t* __restrict__ dV, long dv_bs, int dv_rs, int H, int Tq, int Tk,
                                                           unsigned drop_key, unsigned drop_thr, float drop_scale, unsigned od_key,
                                                           unsigned od_thr, float od_scale) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * V2_MAXROWS * V2_RB + 2 * V2_MAXROWS * 4];
    unsigned char* q_lds = smem;
    unsigned char* do_lds = smem + V2_MAXROWS * V2_RB;
    unsigned char* k_lds = smem + 2 * V2_MAXROWS * V2_RB;
    unsigned char* v_lds = smem + 3 * V2_MAXROWS * V2_RB;
    float* lse_l = reinterpret_cast<float*>(smem + 4 * V2_MAXROWS * V2_RB);
    float* dl_l = lse_l + V2_MAXROWS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const bf16_t* qb_ = Q.p + (size_t)b * Q.bs + h * 64;
    const bf16_t* dob = dO.p + (size_t)b * dO.bs + h * 64;
    const bf16_t* ob = O.p + (size_t)b * O.bs + h * 64;
    const int qpad = (Tq + 31) & ~31, kpad = (Tk + 31) & ~31;
    b2_stage2(q_lds, do_lds, qb_, Q.rs, dob, dO.rs, Tq, qpad, tid);
    b2_stage2(k_lds, v_lds, K.p + (size_t)b * K.bs + h * 64, K.rs, V.p + (size_t)b * V.bs + h * 64, V.rs, Tk, kpad, tid);
    // delta[q] = sum_d dO[q][d] O[q][d] (8 lanes per row: one 16-byte chunk each, both from global: dO is L2-hot) and lse in log2 units;
    // rows past Tq get lse = +inf (p = 0) and delta = 0
    const size_t stat_base = ((size_t)b * H + h) * Tq;
    for (int r0 = 0; r0 < qpad; r0 += 64) {
        const int r = r0 + (tid >> 3), kc = tid & 7;
        float s = 0.f;
        if (r < Tq) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(dob + (size_t)r * dO.rs + kc * 8), c = *reinterpret_cast<const u32x4*>(ob + (size_t)r * O.rs + kc * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf16lo(a[e]) * bf16lo(c[e]) + bf16hi(a[e]) * bf16hi(c[e]);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (kc == 0 && r < qpad) {
            dl_l[r] = r < Tq ? s : 0.f;
            lse_l[r] = r < Tq ? lse[stat_base + r] * LOG2E : INFINITY;
        }
    }
    __syncthreads();
    const float dscale = DROP ? drop_scale : 1.f;
    const unsigned drow_base = ((unsigned)b * H + h) * (unsigned)Tq * (unsigned)Tk, grow_q = (unsigned)b * Tq, grow_k = (unsigned)b * Tk;
    bf16_t* dqb = dQ + (size_t)b * dq_bs + h * 64;
    bf16_t* dkb = dK + (size_t)b * dk_bs + h * 64;
    bf16_t* dvb = dV + (size_t)b * dv_bs + h * 64;
    const int nqb = (Tq + 15) >> 4, nkb = (Tk + 15) >> 4;
    const int ca = (nqb - w + 7) >> 3;                                       // query blocks w, w + 8, ... < nqb
    const int cbk = (nkb - (7 - w) + 7) >> 3;                                // key blocks 7 - w, 15 - w, ... < nkb
#define B2_ARGS_A q_lds, k_lds, v_lds, do_lds, lse_l, dl_l, w, Tq, Tk, drow_base, drop_key, drop_thr, dscale, dqb, dq_rs, od_key, od_thr, od_scale, grow_q, lane
#define B2_ARGS_B q_lds, k_lds, v_lds, do_lds, lse_l, dl_l, w, Tq, Tk, drow_base, drop_key, drop_thr, dscale, dkb, dk_rs, dvb, dv_rs, od_key, od_thr, od_scale, grow_k, lane
    if (ca == 1) b2_phase_dq<DROP, 1>(B2_ARGS_A);
    else if (ca == 2) b2_phase_dq<DROP, 2>(B2_ARGS_A);
    else if (ca >= 3) b2_phase_dq<DROP, 3>(B2_ARGS_A);
    if (cbk == 1) b2_phase_dkv<DROP, 1>(B2_ARGS_B);
    else if (cbk == 2) b2_phase_dkv<DROP, 2>(B2_ARGS_B);
    else if (cbk >= 3) b2_phase_dkv<DROP, 3>(B2_ARGS_B);
#undef B2_ARGS_A
#undef B2_ARGS_B
}

// ================================================================================================== v3 backward: ONE softmax pass
// attn_bwd2 issues the exp / dropout-hash / dS arithmetic of every (query, key) pair TWICE (once per phase) and that VALU work is what
// bounds it (PMC: matrix pipe 18 % busy, both waves of a SIMD issuing 39 % of their cycles).  Here a pair is evaluated once, by the wave
// that owns its KEY block (S = Q K^T orientation: a lane owns a key column, so dV^T += dO^T P and dK^T += Q^T dS contract over the
// accumulator's own row index and stay in registers).  dQ contracts over KEYS, i.e. over the accumulator's column index and across
// waves: each 32-query chunk's dS goes to LDS once (bf16, [key][query], one ds_write_b64 per tile) and after ONE barrier per chunk
// the 8 waves each take one 16 x 16 tile of dQ^T[64 dims][32 queries] and contract it over all keys -- K^T and dS^T both as
// transposed LDS reads.  The dS buffer takes V's place in LDS (V is only needed as the owner's row fragments: loaded from global once);
// its 128-byte rows hold two chunks side by side (double buffer: chunk i + 1 is written while stragglers still read chunk i).
template <bool DROP, int CNT, int NWV>
__device__ __forceinline__ void b3_run(const unsigned char* q_lds, const unsigned char* k_lds, const unsigned char* do_lds, unsigned char* ds_lds,
                                       const float* lse_l, const float* dl_l, const bf16_t* vb, int v_rs, int w, int Tq, int Tk,
                                       unsigned drow_base, unsigned drop_key, unsigned drop_thr, float dscale, bf16_t* dqb, int dq_rs,
                                       bf16_t* dkb, int dk_rs, bf16_t* dvb, int dv_rs, unsigned od_key, unsigned od_thr, float od_scale,
                                       unsigned grow_q, unsigned grow_k, int lane) {
    constexpr int NC = CNT > 0 ? CNT : 1;
    const int g = lane >> 4, li = lane & 15;
    const int qpad = (Tq + 31) & ~31, kpad = (Tk + 31) & ~31;
    bf16x8 kf[NC][2], vf[NC][2];
    f32x4 adk[NC][4], adv[NC][4];
    int key[NC];
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        const int k0 = (w + NWV * c) * 16;
        key[c] = k0 + li;
        kf[c][0] = v2_row_frag(k_lds, k0, 0, lane); kf[c][1] = v2_row_frag(k_lds, k0, 1, lane);
        vf[c][0] = global_row_frag(vb, v_rs, k0, Tk, 0, lane); vf[c][1] = global_row_frag(vb, v_rs, k0, Tk, 1, lane);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) adk[c][dt] = adv[c][dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const int dt_w = w & 3, qt_w = (w >> 2) & 1;                              // this wave's tile of a chunk's dQ^T[4 x 16 dims][2 x 16 queries] (waves 0 .. 7)
    const int nks3 = ((kpad >> 5) + 2) / 3;                                   // 32-key steps in threes (rows up to 288 exist and are zero)
    // dQ^T[dims 16 dt_w ..][queries 16 (qb + qt_w) ..] = sum over keys K^T[dim][key] dS^T[key][query]: three independent accumulation
    // chains, the six transposed reads of a trip issued together
    auto dq_tile = [&](int qb) {
        const int col = ((qb >> 1) & 1) * 32 + qt_w * 16;
        f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0, a2 = a0;
        for (int k3 = 0; k3 < nks3; ++k3) {
            const int r = k3 * 96;
            const bf16x8 t0 = v2_tr_frag(k_lds, r, dt_w * 16, lane), t1 = v2_tr_frag(k_lds, r + 32, dt_w * 16, lane), t2 = v2_tr_frag(k_lds, r + 64, dt_w * 16, lane);
            const bf16x8 s0 = v2_tr_frag(ds_lds, r, col, lane), s1 = v2_tr_frag(ds_lds, r + 32, col, lane), s2 = v2_tr_frag(ds_lds, r + 64, col, lane);
            a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t0, s0, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t1, s1, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(t2, s2, a2, 0, 0, 0);
        }
        const int qrow = (qb + qt_w) * 16 + li;
        if (qrow < Tq) {
            const float f = (od_thr ? (dropout_keep(od_key, grow_q + (unsigned)qrow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
            const f32x4 o = (a0 + a1 + a2) * f;
            *reinterpret_cast<u32x2*>(dqb + (size_t)qrow * dq_rs + dt_w * 16 + 4 * g) = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        }
    };
    for (int qb = 0; qb < (qpad >> 4); qb += 2) {
        const int half = (qb >> 1) & 1;
        if constexpr (CNT > 0) {
            f32x4 pp[NC][2], ds[NC][2];
#pragma unroll
            for (int qj = 0; qj < 2; ++qj) {
                const int q0 = (qb + qj) * 16;
                if (q0 >= Tq) {          // wave-uniform: the padded half of the last chunk (T = 260: queries 272 .. 287) -- zeros, no arithmetic
#pragma unroll
                    for (int c = 0; c < CNT; ++c) {
                        pp[c][qj] = ds[c][qj] = f32x4{0.f, 0.f, 0.f, 0.f};
                        const int chunk = half * 4 + 2 * qj + (g >> 1);
                        *reinterpret_cast<u32x2*>(ds_lds + key[c] * V2_RB + ((chunk ^ (key[c] & 7)) << 4) + (g & 1) * 8) = u32x2{0u, 0u};
                    }
                    continue;
                }
                const bf16x8 q0f = v2_row_frag(q_lds, q0, 0, lane), q1f = v2_row_frag(q_lds, q0, 1, lane);
                const bf16x8 d0f = v2_row_frag(do_lds, q0, 0, lane), d1f = v2_row_frag(do_lds, q0, 1, lane);
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_l + q0 + 4 * g);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(dl_l + q0 + 4 * g);
#pragma unroll
                for (int c = 0; c < CNT; ++c) {
                    f32x4 a = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q0f, kf[c][0], a, 0, 0, 0);      // D[q][key]: rows q = 4 g + e, column key = li
                    a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q1f, kf[c][1], a, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d0f, vf[c][0], dp, 0, 0, 0);
                    dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(d1f, vf[c][1], dp, 0, 0, 0);
                    bool keep4[4] = {true, true, true, true};
                    if constexpr (DROP) {      // (as attn_bwd2's phase B: the quad of keys 4c..4c+3 shares one hash word per query row)
                        const int j = li & 3;
                        const int qm = min(q0 + 4 * g + j, Tq - 1);
                        const int mine = (int)dropout_hash(drop_key, (drow_base + (unsigned)qm * (unsigned)Tk + (unsigned)key[c]) >> 2);
                        const unsigned sh = 8u * (unsigned)j;
                        keep4[0] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x00, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[1] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0x55, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[2] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xaa, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                        keep4[3] = (((unsigned)__builtin_amdgcn_mov_dpp(mine, 0xff, 0xf, 0xf, true) >> sh) & 0xffu) >= drop_thr;
                    }
                    const bool kvalid = key[c] < Tk;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float pv = __builtin_amdgcn_exp2f(a[r] * (SCALE * LOG2E) - l4[r]);
                        pv = kvalid ? pv : 0.f;
                        float pd = pv, dpr = dp[r];
                        if constexpr (DROP) {
                            pd = keep4[r] ? pv : 0.f;
                            dpr = keep4[r] ? dpr : 0.f;
                        }
                        pp[c][qj][r] = pd;
                        ds[c][qj][r] = pv * fmaf(dpr, dscale, -d4[r]);
                    }
                    // dS -> LDS [key][query]: the lane's 4 consecutive queries of its key = 8 bytes of chunk (32 half + 16 qj + 4 g) / 8
                    const int chunk = half * 4 + 2 * qj + (g >> 1);
                    *reinterpret_cast<u32x2*>(ds_lds + key[c] * V2_RB + ((chunk ^ (key[c] & 7)) << 4) + (g & 1) * 8) =
                        u32x2{pack_bf16x2(ds[c][qj][0], ds[c][qj][1]), pack_bf16x2(ds[c][qj][2], ds[c][qj][3])};
                }
            }
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const bf16x8 dot = v2_tr_frag(do_lds, qb * 16, dt * 16, lane), qt = v2_tr_frag(q_lds, qb * 16, dt * 16, lane);
#pragma unroll
                for (int c = 0; c < CNT; ++c) {
                    adv[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(dot, pack_frag(pp[c][0], pp[c][1]), adv[c][dt], 0, 0, 0);
                    adk[c][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qt, pack_frag(ds[c][0], ds[c][1]), adk[c][dt], 0, 0, 0);
                }
            }
        }
        // The two waves of a SIMD (w and w + 4) take their dQ tiles at DIFFERENT times -- the lower four right after the barrier, the
        // upper four after their owner work on the NEXT chunk -- so that one wave's LDS-latency-bound contraction runs beside the other's
        // VALU-bound softmax (in lockstep both sat in the contraction together and the SIMD idled: 53 % of wave-cycles parked).  The
        // double buffer allows it: chunk i's half is next written after barrier i + 1.
        if (w >= 4 && w < 8 && qb > 0) dq_tile(qb - 2);
        __syncthreads();                                                     // the chunk's dS is complete (every key row, all owners)
        if (w < 4) dq_tile(qb);
    }
    if (w >= 4 && w < 8) dq_tile(((qpad >> 4) - 1) & ~1);
#pragma unroll
    for (int c = 0; c < CNT; ++c) {
        if (key[c] < Tk) {
            const unsigned grow = grow_k + (unsigned)key[c];
            const float fk = (od_thr ? (dropout_keep(od_key + 1u, grow, od_thr) ? od_scale : 0.f) : 1.f) * SCALE;
            const float fv = (od_thr ? (dropout_keep(od_key + 2u, grow, od_thr) ? od_scale : 0.f) : 1.f) * dscale;
            bf16_t* pk_ = dkb + (size_t)key[c] * dk_rs;
            bf16_t* pv_ = dvb + (size_t)key[c] * dv_rs;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const f32x4 a = adk[c][dt] * fk, b = adv[c][dt] * fv;
                *reinterpret_cast<u32x2*>(pk_ + dt * 16 + 4 * g) = u32x2{pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3])};
                *reinterpret_cast<u32x2*>(pv_ + dt * 16 + 4 * g) = u32x2{pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
            }
        }
    }
}

// NWV = 8 or 9 waves.  Key blocks are dealt w, w + NWV, ...: at T = 260 (17 blocks, the last one 4 keys wide) eight waves leave wave 0
// with THREE blocks and the other seven waiting for it at every chunk barrier; a ninth wave takes the odd block (it owns no dQ tile: the
// eight tiles of a chunk stay with waves 0 .. 7), every wave has at most two and the registers of the 3-block instantiation are never
// needed (9 waves = 3 on one SIMD: 168 registers).
template <bool DROP, int NWV>
__global__ __launch_bounds__(64 * NWV, 1) void attn_bwd3_kernel(AttnPtr Q, AttnPtr K, AttnPtr V, AttnPtr dO, AttnPtr O, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dQ, long dq_bs, int dq_rs, bf16_t* __restrict__ dK, long dk_bs,
                                                           int dk_rs, bf16_t* __restrict__ dV, long dv_bs, int dv_rs, int H, int Tq, int Tk,
                                                           unsigned drop_key, unsigned drop_thr, float drop_scale, unsigned od_key,
                                                           unsigned od_thr, float od_scale, int od_q_seq) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * V2_MAXROWS * V2_RB + 2 * V2_MAXROWS * 4];
    unsigned char* q_lds = smem;
    unsigned char* do_lds = smem + V2_MAXROWS * V2_RB;
    unsigned char* k_lds = smem + 2 * V2_MAXROWS * V2_RB;
    unsigned char* ds_lds = smem + 3 * V2_MAXROWS * V2_RB;
    float* lse_l = reinterpret_cast<float*>(smem + 4 * V2_MAXROWS * V2_RB);
    float* dl_l = lse_l + V2_MAXROWS;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int h = blockIdx.x % H, b = blockIdx.x / H;
    const bf16_t* qb_ = Q.p + (size_t)b * Q.bs + h * 64;
    const bf16_t* dob = dO.p + (size_t)b * dO.bs + h * 64;
    const bf16_t* ob = O.p + (size_t)b * O.bs + h * 64;
    const bf16_t* kb_ = K.p + (size_t)b * K.bs + h * 64;
    const int qpad = (Tq + 31) & ~31, kpad = (Tk + 31) & ~31;
    if (tid < 512) {                                                         // (the staging loops are written for 512 threads)
    b2_stage2(q_lds, do_lds, qb_, Q.rs, dob, dO.rs, Tq, qpad, tid);
    for (int c = tid; c < V2_MAXROWS * 8; c += 512) {                        // K alone (zero rows past Tk), and a dS buffer of zeros: its
        const int r = c >> 3, kc = c & 7;                                    // rows past the last owned key block are never written
        u32x4 v = {0u, 0u, 0u, 0u};
        if (r < Tk) v = *reinterpret_cast<const u32x4*>(kb_ + (size_t)r * K.rs + kc * 8);
        const int off = r * V2_RB + ((kc ^ (r & 7)) << 4);
        *reinterpret_cast<u32x4*>(k_lds + off) = v;
        *reinterpret_cast<u32x4*>(ds_lds + off) = u32x4{0u, 0u, 0u, 0u};
    }
    const size_t stat_base = ((size_t)b * H + h) * Tq;
    for (int r0 = 0; r0 < qpad; r0 += 64) {                                  // delta = rowsum(dO . O), lse in log2 units (attn_bwd2_kernel)
        const int r = r0 + (tid >> 3), kc = tid & 7;
        float s = 0.f;
        if (r < Tq) {
            const u32x4 a = *reinterpret_cast<const u32x4*>(dob + (size_t)r * dO.rs + kc * 8), c = *reinterpret_cast<const u32x4*>(ob + (size_t)r * O.rs + kc * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf16lo(a[e]) * bf16lo(c[e]) + bf16hi(a[e]) * bf16hi(c[e]);
        }
        s += __shfl_xor(s, 1, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 4, 64);
        if (kc == 0 && r < qpad) {
            dl_l[r] = r < Tq ? s : 0.f;
            lse_l[r] = r < Tq ? lse[stat_base + r] * LOG2E : INFINITY;
        }
    }
    }
    __syncthreads();
    const float dscale = DROP ? drop_scale : 1.f;
    const unsigned drow_base = ((unsigned)b * H + h) * (unsigned)Tq * (unsigned)Tk, grow_q = (unsigned)b * (od_q_seq ? od_q_seq : Tq),      // (od_q_seq: the queries are the first Tq rows of od_q_seq-row sequences)
                    grow_k = (unsigned)b * Tk;
    bf16_t* dqb = dQ + (size_t)b * dq_bs + h * 64;
    bf16_t* dkb = dK + (size_t)b * dk_bs + h * 64;
    bf16_t* dvb = dV + (size_t)b * dv_bs + h * 64;
    const bf16_t* vb = V.p + (size_t)b * V.bs + h * 64;
    const int nkb = (Tk + 15) >> 4;
    const int cnt = max(0, (nkb - w + NWV - 1) / NWV);                       // key blocks w, w + NWV, ... < nkb
#define B3_ARGS q_lds, k_lds, do_lds, ds_lds, lse_l, dl_l, vb, V.rs, w, Tq, Tk, drow_base, drop_key, drop_thr, dscale, dqb, dq_rs, dkb, dk_rs, dvb, dv_rs, \
                od_key, od_thr, od_scale, grow_q, grow_k, lane
    if (cnt == 0) b3_run<DROP, 0, NWV>(B3_ARGS);                             // (every wave meets every barrier: same trip count in all four)
    else if (cnt == 1) b3_run<DROP, 1, NWV>(B3_ARGS);
    else if (cnt == 2 || NWV == 9) b3_run<DROP, 2, NWV>(B3_ARGS);            // (9 waves x 2 blocks = V2_MAXROWS / 16: never a third)
    else b3_run<DROP, 3, NWV>(B3_ARGS);
#undef B3_ARGS
}

// the resident-operand kernels cover dense, non-causal calls whose operands fit (the encoders); I2T_ATTN_V2=0 keeps the tiled ones
bool v2_applies(int Tq, int Tk, int causal, const int* cu_q, const int* cu_k, unsigned drop_thr) {
    const char* e = getenv("I2T_ATTN_V2");                               // read per call: tests switch it inside one process
    const bool on = !(e && e[0] == '0');
    // (with dropout the 4 consecutive keys of a lane must be the 4 bytes of one hash: Tk % 4 == 0)
    return on && !causal && !cu_q && !cu_k && Tk <= V2_MAXROWS && Tq >= 64 && Tq <= V2_MAX_OTHER && (!drop_thr || (Tk & 3) == 0);
}

// variant of a kernel template <bool DROP, bool EVEN> for this call (EVEN = Tk % 4 == 0; only matters with dropout)
#define ATTN_DISPATCH(KERNEL, drop_thr, Tk, ...)                                   \
    do {                                                                           \
        if (!(drop_thr)) hipLaunchKernelGGL((KERNEL<false, true>), __VA_ARGS__);   \
        else if (((Tk) & 3) == 0) hipLaunchKernelGGL((KERNEL<true, true>), __VA_ARGS__); \
        else hipLaunchKernelGGL((KERNEL<true, false>), __VA_ARGS__);               \
    } while (0)

bool strides_ok(const void* p, long bs, int rs) { return p && ALIGNED16(p) && (bs % 8 == 0) && (rs % 8 == 0) && rs >= 64; }

}  // namespace

extern "C" int i2t_attention_fwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                                 const void* v, long v_bs, int v_rs, void* o, long o_bs, int o_rs, float* lse, int B,
                                 int H, int Tq, int Tk, int causal, unsigned drop_key, unsigned drop_thr, float drop_scale,
                                 const int* cu_q, const int* cu_k, int total_q) {
    I2T_REQUIRE(drop_thr == 0 || (double)B * H * Tq * Tk < 4294967296.0, "i2t_attention_fwd: dropout index overflows 32 bits");
    I2T_REQUIRE(!cu_q || total_q > 0, "i2t_attention_fwd: packed queries need total_q");
    I2T_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0, "i2t_attention_fwd: empty problem");
    I2T_REQUIRE(strides_ok(q, q_bs, q_rs) && strides_ok(k, k_bs, k_rs) && strides_ok(v, v_bs, v_rs) &&
                    strides_ok(o, o_bs, o_rs),
                "i2t_attention_fwd: operands must be 16-byte aligned with strides that are multiples of 8");
    I2T_REQUIRE(!causal || Tk >= Tq, "i2t_attention_fwd: causal needs Tk >= Tq");
    I2T_REQUIRE((double)((Tq + 63) / 64) * H * B < 2147483647.0, "i2t_attention_fwd: grid too large");
    AttnPtr Q{(const bf16_t*)q, q_bs, q_rs}, K{(const bf16_t*)k, k_bs, k_rs}, V{(const bf16_t*)v, v_bs, v_rs};
    if (v2_applies(Tq, Tk, causal, cu_q, cu_k, drop_thr)) {
#define V2_FWD(PER)                                                                                                                  \
    case PER:                                                                                                                        \
        if (!drop_thr) hipLaunchKernelGGL((attn_fwd2_kernel<false, PER>), dim3(H * B), dim3(256), 0, (hipStream_t)stream, Q, K, V,   \
                                          (bf16_t*)o, o_bs, o_rs, lse, H, Tq, Tk, drop_key, drop_thr, drop_scale);                   \
        else hipLaunchKernelGGL((attn_fwd2_kernel<true, PER>), dim3(H * B), dim3(256), 0, (hipStream_t)stream, Q, K, V,              \
                                (bf16_t*)o, o_bs, o_rs, lse, H, Tq, Tk, drop_key, drop_thr, drop_scale);                             \
        break;
        switch (((Tq + 15) >> 4) >> 2) {
            V2_FWD(1) V2_FWD(2) V2_FWD(3) V2_FWD(4)
        }
#undef V2_FWD
        I2T_CHECK_LAUNCH("i2t_attention_fwd(v2)");
        return I2T_OK;
    }
    dim3 grid(((Tq + 63) / 64) * H * B);
    ATTN_DISPATCH(attn_fwd_kernel, drop_thr, Tk, grid, dim3(256), 0, (hipStream_t)stream, Q, K, V, (bf16_t*)o, o_bs, o_rs, lse, H,
                  Tq, Tk, causal, drop_key, drop_thr, drop_scale, VarLen{cu_q, cu_k, total_q, B});
    I2T_CHECK_LAUNCH("i2t_attention_fwd");
    return I2T_OK;
}

extern "C" int i2t_attention_bwd(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                                 const void* v, long v_bs, int v_rs, const void* o, long o_bs, int o_rs,
                                 const void* d_o, long do_bs, int do_rs, const float* lse, float* delta_ws, void* dq,
                                 long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs, void* dv, long dv_bs, int dv_rs,
                                 int B, int H, int Tq, int Tk, int causal, unsigned drop_key, unsigned drop_thr, float drop_scale,
                                 const int* cu_q, const int* cu_k, int total_q, unsigned out_drop_key, unsigned out_drop_thr,
                                 float out_drop_scale) {
    return i2t_attention_bwd_ex(stream, q, q_bs, q_rs, k, k_bs, k_rs, v, v_bs, v_rs, o, o_bs, o_rs, d_o, do_bs, do_rs, lse, delta_ws, dq, dq_bs, dq_rs,
                                dk, dk_bs, dk_rs, dv, dv_bs, dv_rs, B, H, Tq, Tk, causal, drop_key, drop_thr, drop_scale, cu_q, cu_k, total_q,
                                out_drop_key, out_drop_thr, out_drop_scale, 0);
}

extern "C" int i2t_attention_bwd_ex(void* stream, const void* q, long q_bs, int q_rs, const void* k, long k_bs, int k_rs,
                                    const void* v, long v_bs, int v_rs, const void* o, long o_bs, int o_rs,
                                    const void* d_o, long do_bs, int do_rs, const float* lse, float* delta_ws, void* dq,
                                    long dq_bs, int dq_rs, void* dk, long dk_bs, int dk_rs, void* dv, long dv_bs, int dv_rs,
                                    int B, int H, int Tq, int Tk, int causal, unsigned drop_key, unsigned drop_thr, float drop_scale,
                                    const int* cu_q, const int* cu_k, int total_q, unsigned out_drop_key, unsigned out_drop_thr,
                                    float out_drop_scale, int out_drop_q_seq) {
    if (out_drop_q_seq == Tq || !out_drop_thr) out_drop_q_seq = 0;
    I2T_REQUIRE(B > 0 && H > 0 && Tq > 0 && Tk > 0 && lse && delta_ws, "i2t_attention_bwd: bad args");
    I2T_REQUIRE(!cu_q || total_q > 0, "i2t_attention_bwd: packed queries need total_q");
    I2T_REQUIRE(drop_thr == 0 || (double)B * H * Tq * Tk < 4294967296.0, "i2t_attention_bwd: dropout index overflows 32 bits");
    I2T_REQUIRE(strides_ok(q, q_bs, q_rs) && strides_ok(k, k_bs, k_rs) && strides_ok(v, v_bs, v_rs) &&
                    strides_ok(o, o_bs, o_rs) && strides_ok(d_o, do_bs, do_rs) && strides_ok(dq, dq_bs, dq_rs) &&
                    strides_ok(dk, dk_bs, dk_rs) && strides_ok(dv, dv_bs, dv_rs),
                "i2t_attention_bwd: operands must be 16-byte aligned with strides that are multiples of 8");
    I2T_REQUIRE(!causal || Tk >= Tq, "i2t_attention_bwd: causal needs Tk >= Tq");
    I2T_REQUIRE((double)((Tq + 63) / 64 + (Tk + 63) / 64) * H * B < 2147483647.0, "i2t_attention_bwd: grid too large");
    hipStream_t s = (hipStream_t)stream;
    AttnPtr Q{(const bf16_t*)q, q_bs, q_rs}, K{(const bf16_t*)k, k_bs, k_rs}, V{(const bf16_t*)v, v_bs, v_rs};
    AttnPtr DO{(const bf16_t*)d_o, do_bs, do_rs};
    const VarLen vl{cu_q, cu_k, total_q, B};
    const AttnPtr Ow{(const bf16_t*)o, o_bs, o_rs};
    // I2T_ATTN_BWD = 3 (default): the one-pass kernel | 2: the two-phase one (attn_bwd2) | 0: the tiled pair; I2T_ATTN_BWD2=0: as 0
    const char* be = getenv("I2T_ATTN_BWD");
    const int bmode = (getenv("I2T_ATTN_BWD2") && getenv("I2T_ATTN_BWD2")[0] == '0') ? 0 : (be ? atoi(be) : 3);
    I2T_REQUIRE(!out_drop_q_seq || (bmode == 3 && v2_applies(Tq, Tk, causal, cu_q, cu_k, drop_thr) && Tq <= V2_MAXROWS),
                "i2t_attention_bwd_ex: out_drop_q_seq=%d is offered by the one-pass resident kernel only (dense, Tk <= %d)", out_drop_q_seq, V2_MAXROWS);
    if (bmode == 3 && v2_applies(Tq, Tk, causal, cu_q, cu_k, drop_thr) && Tq <= V2_MAXROWS) {
        // 9 waves whenever 8 would leave a wave with a third key block (I2T_ATTN_BWD3_WAVES = 8 | 9 forces one: A/B runs)
        static const char* we = getenv("I2T_ATTN_BWD3_WAVES");
        const int nkb = (Tk + 15) >> 4;
        const bool nine = we && atoi(we) == 9;      // measured (B = 1024, T = 260, dropout): 1295 us against 1228 for 8 waves -- the third register-limited
        // wave of a SIMD and its spills cost more than wave 0's third block: off unless asked for
        (void)nkb;
#define B3_LAUNCH(D_, W_) hipLaunchKernelGGL((attn_bwd3_kernel<D_, W_>), dim3(H * B), dim3(64 * W_), 0, s, Q, K, V, DO, Ow, lse, (bf16_t*)dq, dq_bs, dq_rs, \
                                             (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, H, Tq, Tk, drop_key, drop_thr, drop_scale,          \
                                             out_drop_key, out_drop_thr, out_drop_scale, out_drop_q_seq)
        if (!drop_thr) { if (nine) B3_LAUNCH(false, 9); else B3_LAUNCH(false, 8); }
        else { if (nine) B3_LAUNCH(true, 9); else B3_LAUNCH(true, 8); }
#undef B3_LAUNCH
        I2T_CHECK_LAUNCH("i2t_attention_bwd(v3)");
        return I2T_OK;
    }
    if (bmode != 0 && v2_applies(Tq, Tk, causal, cu_q, cu_k, drop_thr) && Tq <= V2_MAXROWS) {
        if (!drop_thr) hipLaunchKernelGGL((attn_bwd2_kernel<false>), dim3(H * B), dim3(512), 0, s, Q, K, V, DO, Ow, lse, (bf16_t*)dq, dq_bs, dq_rs,
                                          (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, H, Tq, Tk, drop_key, drop_thr, drop_scale,
                                          out_drop_key, out_drop_thr, out_drop_scale);
        else hipLaunchKernelGGL((attn_bwd2_kernel<true>), dim3(H * B), dim3(512), 0, s, Q, K, V, DO, Ow, lse, (bf16_t*)dq, dq_bs, dq_rs,
                                (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, H, Tq, Tk, drop_key, drop_thr, drop_scale,
                                out_drop_key, out_drop_thr, out_drop_scale);
        I2T_CHECK_LAUNCH("i2t_attention_bwd(v2)");
        return I2T_OK;
    }
    {   // one tile of queries and keys (the decoder's attentions): the fused single-pass kernel (I2T_ATTN_BWD1=0: the tiled pair, for A/B runs)
        const char* e1 = getenv("I2T_ATTN_BWD1");
        if (Tq <= 64 && Tk <= 64 && !(e1 && e1[0] == '0')) {
            ATTN_DISPATCH(attn_bwd1_kernel, drop_thr, Tk, dim3(H * B), dim3(256), 0, s, Q, K, V, DO, Ow, lse, (bf16_t*)dq, dq_bs, dq_rs,
                          (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, H, Tq, Tk, causal, drop_key, drop_thr, drop_scale, vl,
                          out_drop_key, out_drop_thr, out_drop_scale);
            I2T_CHECK_LAUNCH("i2t_attention_bwd(1)");
            return I2T_OK;
        }
    }
    ATTN_DISPATCH(attn_bwd_dq_kernel, drop_thr, Tk, dim3(((Tq + 63) / 64) * H * B), dim3(256), 0, s, Q, K, V, DO, Ow, lse, delta_ws,
                  (bf16_t*)dq, dq_bs, dq_rs, H, Tq, Tk, causal, drop_key, drop_thr, drop_scale, vl, out_drop_key, out_drop_thr,
                  out_drop_scale);
    ATTN_DISPATCH(attn_bwd_dkv_kernel, drop_thr, Tk, dim3(((Tk + 63) / 64) * H * B), dim3(256), 0, s, Q, K, V, DO, lse, delta_ws,
                  (bf16_t*)dk, dk_bs, dk_rs, (bf16_t*)dv, dv_bs, dv_rs, H, Tq, Tk, causal, drop_key, drop_thr, drop_scale, vl,
                  out_drop_key, out_drop_thr, out_drop_scale);
    I2T_CHECK_LAUNCH("i2t_attention_bwd");
    return I2T_OK;
}
