// bf16 MFMA GEMM for gfx950 with fused epilogues (bias, GELU / GELU', dropout, residual, accumulate, bf16|f32 store).
//
//   C[M,N] = epilogue(alpha * op(A)[M,K] . op(B)[K,N])
//
// Three kernels behind one entry point (i2t_gemm_bf16 picks by shape; routing at the bottom of the file):
//   gemm256_kernel   persistent 256 x 256 x 64 tiles, 8 waves, LDS-DMA staging with counted waits -- every problem with
//                    >= 40 such tiles and every dW = dY^T.X problem (K slices + atomics): the workhorse, see its header;
//   gemm_bf16_kernel 128 x 128 x 64 tiles, 4 waves (2 x 2, 64 x 64 each), register-staged double buffer, one barrier per
//                    K-step, optional split-K with atomics -- small problems;
//   gemm_skinny_kernel  M <= 64 (decode steps at small batch): weights streamed straight to registers.
// LDS images shared by the tiled kernels:
//   R  image (reduction index contiguous in memory):  tile[r][64 k] bf16, 128-B rows, 16-B chunks XOR-swizzled by
//            (r & 7); fragments are one ds_read_b128 each (conflict-free: cdna_hip_programming.md T2).
//   Cf image (reduction index is the row index):       tile[64 k][128 r] bf16, 256-B rows whose 32-B windows are XOR-moved
//            by the k-row, so the 8 k-rows x 4 column quads a half-wave touches cover 64 distinct banks; fragments are
//            two ds_read_b64_tr_b16 each (T10) -- this is what lets dX = dY.W and dW = dY^T.X run without any
//            transposed copy of weights or activations in HBM.
// The MFMA is issued "swapped" (weight-side fragment as the A operand, activation-side as B) so that a lane ends
// up holding 4 CONSECUTIVE output columns of one output row: the epilogue reads bias/residual and stores C as one
// 8-/16-byte vector per lane (the split-K forms issue it un-swapped: 4 rows x 64 contiguous bytes per atomic instruction).
// Block -> tile maps are XCD-aware (8 XCDs, private L2s) and N-fastest inside groups of column tiles, so an activation row
// panel is fetched once per group and a weight group stays in one L2.
#include "common.h"
#include <stdlib.h>
#include <string.h>
#include <utility>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int R_ROW_BYTES = 128;                 // 64 bf16
constexpr int CF_ROW_BYTES = 288;                // 128 bf16 + 32 B pad
constexpr int OPERAND_BYTES = 128 * 128;         // 16 KiB: R image [128][64] or Cf image [64][128], both unpadded
constexpr int STAGE_BYTES = 2 * OPERAND_BYTES;   // A + B
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // double buffer: 73728

struct GemmParams {
    const bf16_t* A;
    const bf16_t* B;
    void* C;
    int M, N, K;
    int lda, ldb, ldc;
    float alpha;
    const float* alpha_sumsq;   // device scalar S (or null): alpha is multiplied by 1 / (sqrt(S) + 1e-6) -- a gradient normaliser folded into the GEMM
    const float* bias;
    int act;
    const bf16_t* aux_in;
    int ld_aux_in;
    bf16_t* aux_out;
    int ld_aux_out;
    const float* residual;
    int ldr;
    int c_is_f32;
    int accumulate;
    int tiles_m, tiles_n;
    int drop_mode;            // 0 none | 1 elementwise (idx = m*N + n) | 2 per (row, third of N) -- the q/k/v token multipliers
    unsigned drop_key, drop_thr;
    float drop_scale;
    float* ws;                // deterministic split-K (i2t_gemm_bf16_ws): slice `s` of the raw accumulators goes to ws[s][M][N]; null: atomics onto C
    int g2_splits, g2_nk;     // 256^2 kernel: K slices per output tile and K-tiles per slice (even)
    int g2_gn;                // 256^2 kernel: column tiles per group of the tile order
    int g2_dbg;               // experiment (I2T_G256_DBG): 1 = epilogue without its global stores, 2 = no epilogue at all
    const float* scale_a = nullptr;   // class 9 (fp8 operands, i2t_gemm_fp8): per-row scales of A [M] and of B [N] applied to the accumulators
    const float* scale_b = nullptr;
    int g2_stagger, g2_stagger_groups;   // experiment: start delay (units of s_sleep 127) x (workgroup index within its XCD mod groups)
    // fused cross-attention (epilogue class 8, see xattn_epilogue): queries, outputs and shapes
    const bf16_t* xq; long xq_bs; int xq_rs;      // Q [B, T, >= 64 H] (batch stride used when xcu is null) or packed [rows, >= 64 H]
    const int* xcu;                               // packed queries: rows of image b = [xcu[b], xcu[b+1])
    bf16_t* xo; long xo_bs; int xo_rs;            // attention output, same layout convention as Q
    float* xlse;                                  // [H][total_q] (packed) or [B][H][TqMax]
    int x_total_q, x_TqMax, x_H, x_d, x_B;
};

// Tile order inside an XCD's contiguous chunk of the grid.  PMC (round 1, B = 256): with M fastest the GEMM family moved
// 742 MB per launch against ~100 MB algorithmic (4.2 TB/s: bandwidth-bound) -- every 128-row activation tile was re-read
// for each of the N/128 column tiles, tens of MB apart.  Now N is the fast index inside groups of up to GN column tiles
// whose weight panels (GN x 128 x K bf16, <= ~3 MB) stay in the XCD's 4 MiB L2: an activation tile is fetched once per
// group and reused GN times from L2; the weight group is re-read (from L2) for every row tile.
constexpr int GN_MAX = 16;
__device__ __forceinline__ void tile_coords(const GemmParams& p, int swz, int& tile_m, int& tile_n, int gn_max = GN_MAX) {
    const int gn = p.tiles_n < gn_max ? p.tiles_n : gn_max;
    const int per_group = p.tiles_m * gn;
    int grp = swz / per_group;
    const int ngroups = (p.tiles_n + gn - 1) / gn;
    int rem = swz - grp * per_group, width = gn;
    if (grp >= ngroups - 1) {                      // the last group may be narrower
        grp = ngroups - 1;
        rem = swz - grp * per_group;
        width = p.tiles_n - grp * gn;
    }
    tile_m = rem / width;
    tile_n = grp * gn + rem % width;
}

// alpha as the epilogues apply it: the host's factor, times the gradient normaliser 1 / (||g|| + 1e-6) when the caller hands over
// sum(g^2) of the tensor the A operand was cut from (models/functions.py:19-24: the operand then stays un-normalised in memory)
__device__ __forceinline__ float eff_alpha(const GemmParams& p) {
    return p.alpha_sumsq ? p.alpha / (sqrtf(*p.alpha_sumsq) + 1e-6f) : p.alpha;
}

// One operand's staging registers: 4 x 16-byte chunks per thread per K-step.
struct Stage {
    u32x4 v[4];
};

// KMAJOR=false: memory is [rows][K] (K contiguous);  KMAJOR=true: memory is [K][rows] (rows contiguous)
template <bool KMAJOR>
__device__ __forceinline__ void stage_load(Stage& s, const bf16_t* __restrict__ base, int ld, int row0, int rows,
                                           int k0, int K, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int c = tid + 256 * u;
        u32x4 val = {0u, 0u, 0u, 0u};
        if (!KMAJOR) {
            int r = c >> 3, kc = c & 7;
            int gr = row0 + r, gk = k0 + kc * 8;
            if (gr < rows && gk < K) val = *reinterpret_cast<const u32x4*>(base + (size_t)gr * ld + gk);
        } else {
            int kr = c >> 4, rc = c & 15;
            int gk = k0 + kr, gr = row0 + rc * 8;
            if (gk < K && gr < rows) val = *reinterpret_cast<const u32x4*>(base + (size_t)gk * ld + gr);
        }
        s.v[u] = val;
    }
}

// Cf image of the v1 kernel: [k-row][16 chunks of 16 B], unpadded, chunk ^= swz(k) << 1 (see the DMA kernel's comment):
// conflict-free transposed reads for BOTH k orders (PMC round 1: the padded image cost 20 % LDS conflict cycles in dX).
template <bool PK>
__device__ __forceinline__ int cf_swz_v1(int kr) { return PK ? (kr & 7) : ((kr & 3) | (((kr >> 3) & 1) << 2)); }

template <bool KMAJOR, bool PK = false>
__device__ __forceinline__ void stage_store(const Stage& s, unsigned char* lds, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        int c = tid + 256 * u;
        int off;
        if (!KMAJOR) {
            int r = c >> 3, kc = c & 7;
            off = r * R_ROW_BYTES + ((kc ^ (r & 7)) << 4);
        } else {
            int kr = c >> 4, rc = c & 15;
            off = kr * 256 + ((rc ^ (cf_swz_v1<PK>(kr) << 1)) << 4);
        }
        *reinterpret_cast<u32x4*>(lds + off) = s.v[u];
    }
}

// Fragment of 16 rows starting at r0 for k-substep ks (32 wide): lane (g = lane>>4, i = lane&15) gets
// tile[r0 + i][32 ks + 8 g + 0..7].
// PERMUTE_K (both operands k-major): MFMA k-slot (g, j) is mapped to k = 32 ks + 16 (j>>2) + 4 g + (j&3) on BOTH
// operands (the reduction order is free), so that a half-wave's transposed reads touch 8 CONSECUTIVE k-rows: with the
// 288-byte row stride (8 banks per row) those are 8 disjoint 8-bank windows = conflict-free; the natural mapping
// (k = 8 g + j) reads rows {0-3, 8-11} per half-wave, which collide pairwise on this stride.
template <bool KMAJOR, bool PERMUTE_K = false>
__device__ __forceinline__ bf16x8 frag_read(const unsigned char* lds, int r0, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!KMAJOR) {
        int r = r0 + i;
        int kc = ks * 4 + g;
        u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * R_ROW_BYTES + ((kc ^ (r & 7)) << 4));
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int q = i >> 2, p = i & 3;
        const int kr0 = ks * 32 + (PERMUTE_K ? 4 : 8) * g + q, kr1 = kr0 + (PERMUTE_K ? 16 : 4);
        const int ch = (r0 >> 3) + (p >> 1), half = (p & 1) * 8;
        s16x4 lo = lds_read_tr16(lds + kr0 * 256 + ((ch ^ (cf_swz_v1<PERMUTE_K>(kr0) << 1)) << 4) + half);
        s16x4 hi = lds_read_tr16(lds + kr1 * 256 + ((ch ^ (cf_swz_v1<PERMUTE_K>(kr1) << 1)) << 4) + half);
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

// Epilogue shared by the GEMM kernels.  SPLITK: un-swapped accumulators, float atomics; else fused bias / GELU /
// GELU' / residual / accumulate with 8-/16-byte vector stores.
template <int MI, int NJ>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int lane);

template <bool SPLITK>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x4 (&acc)[4][4], int m0, int n0, int wm, int wn, int lane,
                                              int split = 0) {
    const int g = lane >> 4, li = lane & 15;
    if (SPLITK && p.ws) {
        // deterministic form: this K slice's raw accumulators to its own [M][N] plane; splitk_reduce_kernel adds the planes in a
        // fixed order and applies the fused epilogue (16 lanes = 64 contiguous bytes per store)
        float* W = p.ws + (size_t)split * p.M * p.N;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
                    if (m < p.M && n < p.N) W[(size_t)m * p.N + n] = acc[i][j][r];
                }
            }
        return;
    }
    if (SPLITK) {
        // un-swapped accumulators: lane holds C[m0 + wm*64 + 16 i + 4 g + r][n0 + wn*64 + 16 j + li], r = 0..3
        float* C = reinterpret_cast<float*>(p.C);
        const float alpha = eff_alpha(p);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int n = n0 + wn * 64 + j * 16 + li;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = m0 + wm * 64 + i * 16 + 4 * g + r;
                    if (m < p.M && n < p.N) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * alpha);
                }
            }
        return;
    }
    epilogue_tile<4, 4>(p, acc, m0 + wm * 64, n0 + wn * 64, lane);
}

// The fused epilogue for ONE lane's 4 consecutive columns C[m][n4 .. n4+3] (a4 = raw accumulators); m < M and n4 < N.
__device__ __forceinline__ void epilogue_quad(const GemmParams& p, const f32x4 a4, int m, int n4, bool vec_ok) {
    float v[4];
    const float alpha = eff_alpha(p);
#pragma unroll
    for (int r = 0; r < 4; ++r) v[r] = a4[r] * alpha;
    const int nv = (p.N - n4) < 4 ? (p.N - n4) : 4;
    if (p.bias) {
        if (nv == 4) {      // one 16-byte load (n4 % 4 == 0) instead of four scalar ones
            const f32x4 bv = *reinterpret_cast<const f32x4*>(p.bias + n4);
            v[0] += bv[0]; v[1] += bv[1]; v[2] += bv[2]; v[3] += bv[3];
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nv) v[r] += p.bias[n4 + r];
        }
    }
    if (p.act == I2T_ACT_GELU_DOUT) {      // activated output + the derivative (not the pre-activation) for the backward pass
        float dv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float x_ = v[r];
            gelu_tanh_both(x_, v[r], dv[r]);
        }
        if (p.aux_out) {
            bf16_t* ao = p.aux_out + (size_t)m * p.ld_aux_out + n4;
            for (int r = 0; r < nv; ++r) ao[r] = f32_to_bf16(dv[r]);
        }
    } else if (p.aux_out) {
        bf16_t* ao = p.aux_out + (size_t)m * p.ld_aux_out + n4;
        if (nv == 4 && (p.ld_aux_out & 3) == 0) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(ao) = pk;
        } else {
            for (int r = 0; r < nv; ++r) ao[r] = f32_to_bf16(v[r]);
        }
    }
    if (p.act == I2T_ACT_MUL_AUX) {
        const bf16_t* ai = p.aux_in + (size_t)m * p.ld_aux_in + n4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nv) v[r] *= bf16_to_f32(ai[r]);
    } else if (p.act == I2T_ACT_GELU) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
    } else if (p.act == I2T_ACT_GELU_ERF) {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = gelu_erf(v[r]);
    } else if (p.act == I2T_ACT_DGELU_ERF) {
        const bf16_t* ai = p.aux_in + (size_t)m * p.ld_aux_in + n4;
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if (r < nv) v[r] *= gelu_erf_grad(bf16_to_f32(ai[r]));
    } else if (p.act == I2T_ACT_DGELU) {
        const bf16_t* ai = p.aux_in + (size_t)m * p.ld_aux_in + n4;
        if (nv == 4 && (p.ld_aux_in & 3) == 0) {      // one 8-byte load of the 4 pre-activations
            const u32x2 pk = *reinterpret_cast<const u32x2*>(ai);
            v[0] *= gelu_tanh_grad(bf16lo(pk[0])); v[1] *= gelu_tanh_grad(bf16hi(pk[0]));
            v[2] *= gelu_tanh_grad(bf16lo(pk[1])); v[3] *= gelu_tanh_grad(bf16hi(pk[1]));
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (r < nv) v[r] *= gelu_tanh_grad(bf16_to_f32(ai[r]));
        }
    }
    if (p.drop_mode == 1) {
        bool keep[4];
        dropout_keep4(p.drop_key, (unsigned)m * (unsigned)p.N + (unsigned)n4, p.drop_thr, keep);
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = keep[r] ? v[r] * p.drop_scale : 0.f;
    } else if (p.drop_mode == 2) {
        const unsigned third = (unsigned)n4 / (unsigned)(p.N / 3);
        const float mult = dropout_keep(p.drop_key + third, (unsigned)m, p.drop_thr) ? p.drop_scale : 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] *= mult;
    }
    if (p.residual) {
        const float* rr = p.residual + (size_t)m * p.ldr + n4;
        if (nv == 4 && vec_ok) {
            f32x4 t = *reinterpret_cast<const f32x4*>(rr);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] += t[r];
        } else {
            for (int r = 0; r < nv; ++r) v[r] += rr[r];
        }
    }
    if (p.c_is_f32) {
        float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n4;
        if (nv == 4 && vec_ok) {
            f32x4 o = {v[0], v[1], v[2], v[3]};
            if (p.accumulate) {
                f32x4 t = *reinterpret_cast<const f32x4*>(c);
                o += t;
            }
            *reinterpret_cast<f32x4*>(c) = o;
        } else {
            for (int r = 0; r < nv; ++r) c[r] = p.accumulate ? c[r] + v[r] : v[r];
        }
    } else {
        bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n4;
        if (nv == 4 && vec_ok) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(c) = pk;
        } else {
            for (int r = 0; r < nv; ++r) c[r] = f32_to_bf16(v[r]);
        }
    }
}

// Fast form for 4 quads at once (all vector-aligned: N % 4 == 0 and every leading dimension % 4 == 0).  Per-quad code
// waits on each load before the next one is issued (a residual GEMM spent 30-60 us per 256^2 tile that way); here the
// loads of each KIND are issued back to back for the 4 quads -- from clamped, always-valid addresses, so that no
// predication splits them -- and only the stores are predicated.
__device__ __forceinline__ bool epilogue_fast_ok(const GemmParams& p) {
    return (p.N & 3) == 0 && (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) && (!p.aux_in || (p.ld_aux_in & 3) == 0) &&
           (!p.aux_out || (p.ld_aux_out & 3) == 0);
}
// EPI selects a compile-time epilogue class (the generic form keeps ~30 uniform values and every path alive: in the
// 256^2 kernel that cost SGPR and VGPR spills whose serialized scratch reloads were ~10 us per tile):
//   0 generic (all flags at run time)          1 bf16 C, optional bias, optional per-row/third dropout (mode 2)
//   2 bf16 C, GELU, optional bias / pre-act out 3 f32 C, optional bias / dropout (mode 1) / residual
//   4 bf16 C, GELU' of aux_in                   5 f32 C, optional accumulate
//   7 plain bf16 or f32 C with N % 4 != 0 (the lm_head's 50257 columns): the quad that straddles N is stored element-wise,
//     [N, ldc) untouched
// The loads of a batch (bias, GELU' input, residual / accumulate addend) are a separate step from the arithmetic and the
// stores, so that the 256^2 kernel can issue the loads of the NEXT 16-row group before the stores of the current one
// (epilogue_tile_tr): the compiler may not hoist them itself -- C may alias residual / aux_in -- and without that every
// group paid a full load latency (8 per tile).
#ifndef G2_EPI_DEPTH
#define G2_EPI_DEPTH 1
#endif
// Output tiles are written with non-temporal stores: a tile's 128-256 KB burst then does not have to find room in the
// XCD's 4 MB L2 (32 CUs x 128 KB arrive together), and the next output tile's first DMA fence -- which, vmcnt being one
// in-order counter, also waits for these stores -- comes 1-2 us sooner per tile (+1 % on the whole step).
#ifndef G2_PLAIN_STORES
#define G2_STORE(ptr, val) __builtin_nontemporal_store((val), (ptr))
#else
#define G2_STORE(ptr, val) (*(ptr) = (val))
#endif
struct EpiPre {
    f32x4 bv[4], add[4];
    u32x2 ax[4];
    f32x4 sb;            // class 9: the lane's 4 column scales (one vector per tile) and its 4 rows' scales
    float sa[4];
    float alpha;         // eff_alpha(p), evaluated once per tile
};
template <int EPI>
struct EpiFlags {
    static constexpr bool GEN = EPI == 0;
    bool f_bias, f_gelu, f_dgelu, f_auxout, f_drop1, f_drop2, f_res, f_acc, f_f32, f_gout, f_mul;
    __device__ __forceinline__ explicit EpiFlags(const GemmParams& p) {
        f_bias = (GEN || EPI == 1 || EPI == 2 || EPI == 3 || EPI == 7 || EPI == 9 || EPI == 10) ? (p.bias != nullptr) : false;
        f_gout = GEN ? (p.act == I2T_ACT_GELU_DOUT) : (EPI == 10);        // class 10: GELU + its derivative as the second output
        f_mul = GEN ? (p.act == I2T_ACT_MUL_AUX) : (EPI == 11);           // class 11: product with aux_in (the stored derivative)
        f_gelu = GEN ? (p.act == I2T_ACT_GELU || p.act == I2T_ACT_GELU_ERF) : (EPI == 2);        // (the erf flavours: generic class only)
        f_dgelu = GEN ? (p.act == I2T_ACT_DGELU || p.act == I2T_ACT_DGELU_ERF) : (EPI == 4);
        f_auxout = (GEN || EPI == 2 || EPI == 10) ? (p.aux_out != nullptr) : false;
        f_drop1 = (GEN || EPI == 3) ? (p.drop_mode == 1) : false;
        f_drop2 = (GEN || EPI == 1 || EPI == 7) ? (p.drop_mode == 2) : false;
        f_res = (GEN || EPI == 3 || EPI == 9) ? (p.residual != nullptr) : false;
        f_acc = (GEN || EPI == 5) ? (p.accumulate != 0) : false;
        f_f32 = (GEN || EPI == 7 || EPI == 9) ? (p.c_is_f32 != 0) : (EPI == 3 || EPI == 5);
    }
};
// EPI 7: the straddling quad loads bias from [N-1 .. N+2]: inside the 16-B padded vector
__device__ __forceinline__ int epi_clamp_n(const GemmParams& p, int n4) { return min(n4, ((p.N + 3) & ~3) - 4); }

// FULL: the caller guarantees that every quad lies inside C (no clamps, no store predicates)
template <int EPI, bool WITH_BIAS = true, bool FULL = false>
__device__ __forceinline__ void epilogue_loads4(const GemmParams& p, const int (&m)[4], const int (&n4)[4], EpiPre& L) {
    const EpiFlags<EPI> F(p);
    int mc[4], nc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        mc[q] = FULL ? m[q] : min(m[q], p.M - 1);
        nc[q] = FULL ? n4[q] : epi_clamp_n(p, n4[q]);
    }
    if (WITH_BIAS && F.f_bias) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.bv[q] = *reinterpret_cast<const f32x4*>(p.bias + nc[q]);
    }
#ifdef I2T_G2_DBG_NOLOADS      // timing experiment (tools/build_variant.sh): the per-element epilogue loads left out -- WRONG results
    if (F.f_dgelu) { for (int q = 0; q < 4; ++q) L.ax[q] = u32x2{0x3f803f80u, 0x3f803f80u}; }
    if (F.f_res || F.f_acc) { for (int q = 0; q < 4; ++q) L.add[q] = f32x4{1.f, 1.f, 1.f, 1.f}; }
    return;
#endif
#ifdef I2T_G2_DBG_SMALLLOADS   // timing experiment: the same load instructions, from the first 256 rows only (cache-resident) -- WRONG results
    if (F.f_dgelu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.ax[q] = *reinterpret_cast<const u32x2*>(p.aux_in + (size_t)(mc[q] & 255) * p.ld_aux_in + nc[q]);
    }
    if (F.f_res) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.add[q] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)(mc[q] & 255) * p.ldr + nc[q]);
    }
    if (F.f_acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.add[q] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + (size_t)(mc[q] & 255) * p.ldc + nc[q]);
    }
    return;
#endif
    if (F.f_dgelu || F.f_mul) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.ax[q] = *reinterpret_cast<const u32x2*>(p.aux_in + (size_t)mc[q] * p.ld_aux_in + nc[q]);
    }
    if constexpr (EPI == 9) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.sa[q] = p.scale_a[mc[q]];
    }
    // one set of addend registers serves residual and accumulate (both at once is rare: the second then waits on the first)
    if (F.f_res) {
#pragma unroll
        for (int q = 0; q < 4; ++q) L.add[q] = *reinterpret_cast<const f32x4*>(p.residual + (size_t)mc[q] * p.ldr + nc[q]);
    }
    if (F.f_acc) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 c = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + (size_t)mc[q] * p.ldc + nc[q]);
            L.add[q] = F.f_res ? L.add[q] + c : c;
        }
    }
}

template <int EPI, bool FULL = false>
__device__ __forceinline__ void epilogue_finish4(const GemmParams& p, const f32x4 (&a)[4], const int (&m)[4], const int (&n4)[4],
                                                 const EpiPre& L) {
    const EpiFlags<EPI> F(p);
    bool ok[4];
    int nc[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        ok[q] = FULL ? true : (m[q] < p.M && n4[q] < p.N && !(p.g2_dbg & 1));
        nc[q] = FULL ? n4[q] : epi_clamp_n(p, n4[q]);
    }
    f32x4 v[4];
    const bool has_add = F.f_res || F.f_acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v[q] = a[q] * L.alpha;
        if constexpr (EPI == 9) v[q] = v[q] * L.sa[q] * L.sb;          // fp8 operands: row scale of A x column scale of B
        if (F.f_bias) v[q] += L.bv[q];
        if constexpr (EPI == 9) {                                      // (forward-only users: a frozen backbone's MLP -- no pre-activation output)
            if (p.act == I2T_ACT_GELU_ERF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q][r] = gelu_erf(v[q][r]);
            } else if (p.act == I2T_ACT_GELU) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q][r] = gelu_tanh(v[q][r]);
            }
        }
    }
    if (F.f_gout) {                      // one sigmoid per element serves the activation and its derivative
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x4 dv;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float h_, d_;
                gelu_tanh_both(v[q][r], h_, d_);
                v[q][r] = h_; dv[r] = d_;
            }
            if (F.f_auxout && ok[q]) {
                const u32x2 pk = {pack_bf16x2(dv[0], dv[1]), pack_bf16x2(dv[2], dv[3])};
                G2_STORE(reinterpret_cast<u32x2*>(p.aux_out + (size_t)m[q] * p.ld_aux_out + n4[q]), pk);
            }
        }
    } else if (F.f_auxout) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ok[q]) {
                const u32x2 pk = {pack_bf16x2(v[q][0], v[q][1]), pack_bf16x2(v[q][2], v[q][3])};
                G2_STORE(reinterpret_cast<u32x2*>(p.aux_out + (size_t)m[q] * p.ld_aux_out + n4[q]), pk);
            }
    }
    if (F.f_mul) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q][0] *= bf16lo(L.ax[q][0]); v[q][1] *= bf16hi(L.ax[q][0]);
            v[q][2] *= bf16lo(L.ax[q][1]); v[q][3] *= bf16hi(L.ax[q][1]);
        }
    } else if (F.f_gout) {
    } else if (F.GEN && (p.act == I2T_ACT_GELU_ERF || p.act == I2T_ACT_DGELU_ERF)) {      // torchvision's ViT MLP: exact (erf) GELU
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (p.act == I2T_ACT_GELU_ERF) {
#pragma unroll
                for (int r = 0; r < 4; ++r) v[q][r] = gelu_erf(v[q][r]);
            } else {
                v[q][0] *= gelu_erf_grad(bf16lo(L.ax[q][0])); v[q][1] *= gelu_erf_grad(bf16hi(L.ax[q][0]));
                v[q][2] *= gelu_erf_grad(bf16lo(L.ax[q][1])); v[q][3] *= gelu_erf_grad(bf16hi(L.ax[q][1]));
            }
        }
    } else if (F.f_gelu) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) v[q][r] = gelu_tanh(v[q][r]);
    } else if (F.f_dgelu) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            v[q][0] *= gelu_tanh_grad(bf16lo(L.ax[q][0])); v[q][1] *= gelu_tanh_grad(bf16hi(L.ax[q][0]));
            v[q][2] *= gelu_tanh_grad(bf16lo(L.ax[q][1])); v[q][3] *= gelu_tanh_grad(bf16hi(L.ax[q][1]));
        }
    }
    if (F.f_drop1) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            bool keep[4];
            // N % 4 == 0 and n4 % 4 == 0 on this path (epilogue_fast_ok): the index is even -> two hashes, no alignment case
            dropout_keep4_even(p.drop_key, (unsigned)m[q] * (unsigned)p.N + (unsigned)n4[q], p.drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) v[q][r] = keep[r] ? v[q][r] * p.drop_scale : 0.f;
        }
    } else if (F.f_drop2) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned third = (unsigned)nc[q] / (unsigned)(p.N / 3);
            v[q] *= dropout_keep(p.drop_key + third, (unsigned)m[q], p.drop_thr) ? p.drop_scale : 0.f;
        }
    }
    if (has_add) {
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += L.add[q];
    }
    if (F.f_f32) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            if (!ok[q]) continue;
            float* c = reinterpret_cast<float*>(p.C) + (size_t)m[q] * p.ldc + n4[q];
            if (EPI == 7 && n4[q] + 4 > p.N) {              // the quad that straddles N
                for (int r = 0; r < p.N - n4[q]; ++r) c[r] = v[q][r];
            } else {
                G2_STORE(reinterpret_cast<f32x4*>(c), v[q]);
            }
        }
    } else {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (ok[q]) {
                bf16_t* c = reinterpret_cast<bf16_t*>(p.C) + (size_t)m[q] * p.ldc + n4[q];
                if (EPI == 7 && n4[q] + 4 > p.N) {          // the quad that straddles N
                    for (int r = 0; r < p.N - n4[q]; ++r) c[r] = f32_to_bf16(v[q][r]);
                } else {
                    const u32x2 pk = {pack_bf16x2(v[q][0], v[q][1]), pack_bf16x2(v[q][2], v[q][3])};
                    G2_STORE(reinterpret_cast<u32x2*>(c), pk);
                }
            }
    }
}

template <int EPI>
__device__ __forceinline__ void epilogue_batch4(const GemmParams& p, const f32x4 (&a)[4], const int (&m)[4], const int (&n4)[4]) {
    EpiPre L;
    L.alpha = eff_alpha(p);
    epilogue_loads4<EPI>(p, m, n4, L);
    epilogue_finish4<EPI>(p, a, m, n4, L);
}

// Fused epilogue of an (MI x 16) x (NJ x 16) wave tile held in swapped-issue accumulators: lane holds
// C[m][n4 .. n4+3], m = mbase + 16 i + (lane&15), n4 = nbase + 16 j + 4 (lane>>4).
template <int MI, int NJ>
__device__ __forceinline__ void epilogue_tile(const GemmParams& p, f32x4 (&acc)[MI][NJ], int mbase, int nbase, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0);
    // compile-time loops (lambdas over integral constants): with MI x NJ = 32 copies of this body hipcc left a
    // "#pragma unroll" loop rolled and moved the accumulators to scratch for the dynamic index
    if (NJ == 4 && epilogue_fast_ok(p)) {
        static_for<MI>([&](auto I_) {
            constexpr int i = decltype(I_)::value;
            const int mr = mbase + i * 16 + li;
            const int m[4] = {mr, mr, mr, mr};
            const int n4[4] = {nbase + 4 * g, nbase + 16 + 4 * g, nbase + 32 + 4 * g, nbase + 48 + 4 * g};
            const f32x4 a[4] = {acc[i][0], acc[i][1 % NJ], acc[i][2 % NJ], acc[i][3 % NJ]};
            epilogue_batch4<0>(p, a, m, n4);
        });
        return;
    }
    static_for<MI>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
        const int m = mbase + i * 16 + li;
        if (m >= p.M) return;
        static_for<NJ>([&](auto J_) {
            constexpr int j = decltype(J_)::value;
            const int n4 = nbase + j * 16 + 4 * g;
            if (n4 >= p.N) return;
            epilogue_quad(p, acc[i][j], m, n4, vec_ok);
        });
    });
}

// Same epilogue through a wave-private 4 KiB LDS transpose (16 rows x 64 fp32, 16-byte chunks XOR-swizzled by row & 7:
// conflict-free ds_write_b128 and ds_read_b128): afterwards a lane holds C[m][4c .. 4c+3] with m = 4k + (lane>>4),
// c = lane&15, so one store instruction covers 4 rows x 128 B (bf16) / 256 B (f32) of FULL cache lines instead of 16 rows
// x 32 / 64 B.  Measured need (round 1, K sweep): the direct form cost a fixed ~18 us per 256^2 tile -- 16 distinct lines
// per store instruction, row strides of 1-4 KiB camping on a few L2 channels.
template <int MI, int EPI, bool FULL = false>
__device__ __forceinline__ void epilogue_tile_tr(const GemmParams& p, f32x4 (&acc)[MI][4], int mbase, int nbase, int lane,
                                                 unsigned char* wave_lds) {
    const int g = lane >> 4, li = lane & 15;
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0);
    const bool fast = EPI != 0 || epilogue_fast_ok(p);        // the launcher picks a specialised class only when fast_ok holds
    // classes that LOAD per element (residual, accumulate, GELU' input) run the loads G2_EPI_DEPTH row groups ahead
    constexpr bool PIPE = EPI == 3 || EPI == 4 || EPI == 5 || EPI == 9 || EPI == 11;
    constexpr int DEPTH = PIPE ? G2_EPI_DEPTH : 0, NB = DEPTH + 1;
    EpiPre L[NB];
    {
        const float alpha = eff_alpha(p);
#pragma unroll
        for (int s = 0; s < NB; ++s) L[s].alpha = alpha;
    }
    const int n4c = nbase + 4 * li;                           // after the transpose a lane's 4 quads share their columns
    auto rows_of = [&](int i, int (&m)[4], int (&n4)[4]) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            m[k] = mbase + i * 16 + 4 * k + g;
            n4[k] = n4c;
        }
    };
    if constexpr (EPI != 0) {
        const EpiFlags<EPI> F(p);
        if (F.f_bias) {                                       // one bias vector per lane and tile
            const f32x4 b = *reinterpret_cast<const f32x4*>(p.bias + (FULL ? n4c : epi_clamp_n(p, n4c)));
#pragma unroll
            for (int s = 0; s < NB; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) L[s].bv[q] = b;
        }
        if constexpr (EPI == 9) {
            const f32x4 sb = *reinterpret_cast<const f32x4*>(p.scale_b + (FULL ? n4c : epi_clamp_n(p, n4c)));
#pragma unroll
            for (int s = 0; s < NB; ++s) L[s].sb = sb;
        }
        static_for<DEPTH>([&](auto D_) {
            constexpr int d = decltype(D_)::value;
            int m[4], n4[4];
            rows_of(d, m, n4);
            epilogue_loads4<EPI, false, FULL>(p, m, n4, L[d % NB]);
        });
    }
    static_for<MI>([&](auto I_) {
        constexpr int i = decltype(I_)::value;
        int m[4], n4[4];
        if constexpr (EPI != 0 && i + DEPTH < MI) {
            rows_of(i + DEPTH, m, n4);
            epilogue_loads4<EPI, false, FULL>(p, m, n4, L[(i + DEPTH) % NB]);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<f32x4*>(wave_lds + li * 256 + (((4 * j + g) ^ (li & 7)) << 4)) = acc[i][j];
        f32x4 a[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int row = 4 * k + g;
            a[k] = *reinterpret_cast<const f32x4*>(wave_lds + row * 256 + ((li ^ (row & 7)) << 4));
        }
        rows_of(i, m, n4);
        if constexpr (EPI != 0) {
            epilogue_finish4<EPI, FULL>(p, a, m, n4, L[i % NB]);
        } else if (fast) {
            epilogue_batch4<0>(p, a, m, n4);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (m[k] < p.M && n4[k] < p.N) epilogue_quad(p, a[k], m[k], n4[k], vec_ok);
        }
    });
}

// ----------------------------------------------------------------------------------------------------------------
// Fused cross-attention (epilogue class 8 of the 256^2 kernel): reference models/layers.py:537-542,600-605
// (nn.MultiheadAttention over the encoder output: K/V projection of the memory tokens -> softmax(Q K^T / sqrt(dh)) V).
//
// The K/V projection GEMM  [K | V](b) = mem(b) . [W_k ; W_v]^T + bias  is tiled so that ONE WAVE ends its K loop holding,
// for one image b (its 64 memory tokens) and one head h, K_h(b)^T and V_h(b) in its accumulators:
//     acc[i][j],   i < 4:  K^T[dim 16 i + 4 g + r][key 16 j + li]     (un-swapped issue: dims in registers, keys on lanes)
//     acc[4+i][j]       :  V  [key 16 j + 4 g + r][dim 16 i + li]     (swapped issue: keys in registers, dims on lanes)
// (workgroup tile = head pair x 4 images; "A" = the weight rows [K_h | V_h | K_h+1 | V_h+1], "B" = 256 memory rows).
// Those are exactly the operand forms of the two attention products when the scores are produced transposed
// (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand"):
//     S^T[key][q]  = sum_dim K^T[dim][key] Q^T[dim][q]   A = pack(acc[2s][j], acc[2s+1][j])        (sums over K^T's ROW index)
//     O^T[dim][q]  = sum_key V[key][dim]  P^T[key][q]    A = pack(acc[4+i][2s], acc[4+i][2s+1]),  B = pack(P^T tiles 2s, 2s+1)
// so the image's attention runs straight out of the GEMM's registers: K and V are written to HBM once (the backward pass
// reads them) and never read back in the forward -- the separate attention kernel re-read all of K and V (402 MB per layer
// at 2048 images; it was HBM-bound on exactly those bytes) -- and the scores / probabilities never leave the wave.
// Per wave: bias, K/V stores as full 128-byte lines through the wave's 4 KiB LDS pad, then for every 16-query block of the
// image's (packed) query rows 8 + 8 MFMAs around a softmax whose row reductions are 2 shuffles (a lane owns one query).
// Dropout on the probabilities uses the index space of i2t_attention_fwd, so the unfused backward kernels regenerate the mask.
__device__ __forceinline__ bf16x8 xa_pack(const f32x4& a, const f32x4& b) {
    const u32x4 v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}
__device__ __forceinline__ float xa_quad_max(float v) {
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float xa_quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

__device__ __forceinline__ void xattn_epilogue(const GemmParams& p, f32x4 (&acc)[8][4], int tile_m, int n0, int wr, int wc, int lane,
                                               unsigned char* pad) {
    const int g = lane >> 4, li = lane & 15;
    const int h = 2 * tile_m + wr, b = (n0 >> 6) + wc, d = p.x_d;
    if (b >= p.x_B || h >= p.x_H) return;                               // wave-uniform (no workgroup barrier in this epilogue)
    // ---- queries of this (image, head): issue the first block's loads before anything else
    int Tq = p.x_TqMax;
    size_t qoff = (size_t)b * p.xq_bs, ooff = (size_t)b * p.xo_bs, stat = ((size_t)b * p.x_H + h) * p.x_TqMax;
    if (p.xcu) {
        const int s0 = p.xcu[b];
        Tq = p.xcu[b + 1] - s0;
        qoff = (size_t)s0 * p.xq_rs; ooff = (size_t)s0 * p.xo_rs;
        stat = (size_t)h * p.x_total_q + s0;
    }
    const bf16_t* qb = p.xq + qoff + h * 64;
    auto load_q = [&](int qblk, u32x4 (&qf)[2]) {                       // Q^T B-operands: k-slot (g, e) <-> dim 32 s + 16 (e >> 2) + 4 g + (e & 3)
        const bf16_t* qp = qb + (size_t)min(16 * qblk + li, max(Tq - 1, 0)) * p.xq_rs + 4 * g;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            const u32x2 lo = *reinterpret_cast<const u32x2*>(qp + 32 * s2), hi = *reinterpret_cast<const u32x2*>(qp + 32 * s2 + 16);
            qf[s2] = u32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };
    // ALL query blocks of the pair (<= 4 x 16 rows) are requested here, ahead of the K / V store phase: the K loop's fragment
    // registers are dead by now, and the attention below then never waits on a global load (one block ahead was not enough: a
    // block's math is ~600 cycles, a load under this kernel's own store traffic 1-2k -- 3.7 us per tile, r02 profile)
    u32x4 qfa[4][2];
#pragma unroll
    for (int qb_ = 0; qb_ < 4; ++qb_) {
        qfa[qb_][0] = qfa[qb_][1] = u32x4{0u, 0u, 0u, 0u};
        if (16 * qb_ < Tq) load_q(qb_, qfa[qb_]);
    }
    // ---- bias (in_proj_bias rows d + 64 h .. for K, 2 d + 64 h .. for V; p.bias points at the K part)
    {
        const float* bk = p.bias + h * 64;
        const float* bv = p.bias + d + h * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x4 k4 = *reinterpret_cast<const f32x4*>(bk + 16 * i + 4 * g);
            const float v1 = bv[16 * i + li];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc[i][j] += k4;
                acc[4 + i][j] += f32x4{v1, v1, v1, v1};
            }
        }
    }
    // ---- K and V rows of the image -> C = kv[b][key][0:d | d:2d] as full 128-byte lines: per 16-key block the wave's pad holds
    // [16 keys][K dims 64 | V dims 64] bf16 (256-B rows, 16-B chunks XOR-swizzled by the row: conflict-free writes and reads)
    bf16_t* kvb = reinterpret_cast<bf16_t*>(p.C) + (size_t)b * 64 * p.ldc + h * 64;
    // V sits keys-in-registers / dims-on-lanes (the form O^T = V^T P^T wants): written from there a key's row is one 2-byte LDS store per
    // lane and value -- 256 ds_write_b16 per wave tile, most of this phase's time.  The matrix pipe transposes it instead: V^T as the A
    // operand (that IS the accumulator's layout, va_ below) times a 0 / 1 selection matrix gives V^T[dim 4 g + r][key li] -- the K half's
    // form, exact (one non-zero product per output) -- 16 MFMAs per wave tile, then the K half's 8-byte stores.
    bf16x8 va_[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) va_[i][s2] = xa_pack(acc[4 + i][2 * s2], acc[4 + i][2 * s2 + 1]);
    bf16x8 sel[2];                                                      // sel[t]: k-slot (g, e) of xa_pack <-> key 16 t + 4 g + (e & 3) of the pair; column li
    {
        const int e = li - 4 * g;
        s16x8 s0 = {0, 0, 0, 0, 0, 0, 0, 0}, s1 = s0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            s0[q] = (e == q) ? (short)0x3F80 : (short)0;
            s1[4 + q] = (e == q) ? (short)0x3F80 : (short)0;
        }
        sel[0] = __builtin_bit_cast(bf16x8, s0);
        sel[1] = __builtin_bit_cast(bf16x8, s1);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const u32x2 k2 = {pack_bf16x2(acc[i][j][0], acc[i][j][1]), pack_bf16x2(acc[i][j][2], acc[i][j][3])};
            *reinterpret_cast<u32x2*>(pad + li * 256 + (((2 * i + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = k2;       // row = key li
            const f32x4 vt = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_[i][j >> 1], sel[j & 1], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            const u32x2 v2 = {pack_bf16x2(vt[0], vt[1]), pack_bf16x2(vt[2], vt[3])};
            *reinterpret_cast<u32x2*>(pad + li * 256 + (((8 + 2 * i + (g >> 1)) ^ li) << 4) + (g & 1) * 8) = v2;   // row = key li, V part
        }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = 4 * it + g;
            const u32x4 v = *reinterpret_cast<const u32x4*>(pad + row * 256 + ((li ^ row) << 4));
            bf16_t* dst = kvb + (size_t)(16 * j + row) * p.ldc + (li < 8 ? li * 8 : d + (li - 8) * 8);
            G2_STORE(reinterpret_cast<u32x4*>(dst), v);
        }
    }
    if (Tq <= 0) return;
    // ---- operand fragments of the two attention products (bf16: the same rounding the stored K / V carry)
    bf16x8 ka[2][4];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
        for (int j = 0; j < 4; ++j) ka[s2][j] = xa_pack(acc[2 * s2][j], acc[2 * s2 + 1][j]);
    constexpr float kScale = 0.125f * 1.4426950408889634f;             // 1 / sqrt(64) x log2(e)
    const bool drop = p.drop_thr != 0;
    const int nqb = (Tq + 15) >> 4;
    for (int qg = 0; qg < nqb; qg += 4) {                              // groups of 4 query blocks (one group unless Tq > 64)
    if (qg > 0) {
#pragma unroll
        for (int qb_ = 0; qb_ < 4; ++qb_)
            if (16 * (qg + qb_) < Tq) load_q(qg + qb_, qfa[qb_]);
    }
#pragma unroll
    for (int qb4 = 0; qb4 < 4; ++qb4) {
        const int qblk = qg + qb4;
        if (qblk >= nqb) break;                                         // wave-uniform
        const bf16x8 q0 = __builtin_bit_cast(bf16x8, qfa[qb4][0]), q1 = __builtin_bit_cast(bf16x8, qfa[qb4][1]);
        const int qrow = 16 * qblk + li;
        f32x4 sc[4];
        float mx = -INFINITY;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[0][j], q0, a, 0, 0, 0);
            a = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[1][j], q1, a, 0, 0, 0);
            mx = fmaxf(fmaxf(mx, fmaxf(a[0], a[1])), fmaxf(a[2], a[3]));
            sc[j] = a;
        }
        mx = xa_quad_max(mx) * kScale;
        float rs = 0.f;
        const unsigned drow = (((unsigned)b * p.x_H + h) * (unsigned)p.x_TqMax + (unsigned)min(qrow, Tq - 1)) * 64u;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            bool keep[4] = {true, true, true, true};
            if (drop) dropout_keep4_even(p.drop_key, drow + 16 * j + 4 * g, p.drop_thr, keep);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float pr = __builtin_amdgcn_exp2f(sc[j][r] * kScale - mx);
                rs += pr;                                               // the softmax denominator is dropout-free
                sc[j][r] = keep[r] ? pr : 0.f;
            }
        }
        const float l = xa_quad_sum(rs);
        const bf16x8 p0 = xa_pack(sc[0], sc[1]), p1 = xa_pack(sc[2], sc[3]);
        const float inv = (drop ? p.drop_scale : 1.0f) / l;
        u32x2 ob[4];                                                    // (MFMAs stay outside the per-lane row predicate)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f32x4 o = {0.f, 0.f, 0.f, 0.f};
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_[i][0], p0, o, 0, 0, 0);
            o = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va_[i][1], p1, o, 0, 0, 0);
            ob[i] = u32x2{pack_bf16x2(o[0] * inv, o[1] * inv), pack_bf16x2(o[2] * inv, o[3] * inv)};
        }
        if (qrow < Tq) {
            bf16_t* op = p.xo + ooff + (size_t)qrow * p.xo_rs + h * 64 + 4 * g;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x2*>(op + 16 * i) = ob[i];
            if (g == 0 && p.xlse) p.xlse[stat + qrow] = (mx + log2f(l)) * 0.6931471805599453f;
        }
    }
    }
}

// SPLITK: gridDim.y slices of the reduction; every slice adds its partial tile into the fp32 C with float atomics
// (C must already hold the value to accumulate onto -- the gradient arena does).  The MFMA is issued un-swapped
// there so that one atomic wave-instruction covers 4 rows x 64 contiguous bytes instead of 16 rows x 4 scattered
// dwords (MI355X_MICROARCH.md, global float atomics: access shape).
template <bool A_KMAJOR, bool B_KMAJOR, bool SPLITK>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
    constexpr bool PK = A_KMAJOR && B_KMAJOR;
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware, bijective remap of the 1-D grid (cdna_hip_programming.md 5, "XCD swizzle must be bijective")
    const int nwg = gridDim.x;
    const int bid = blockIdx.x;
    int swz, split = 0;
    if (SPLITK) {
        // Workgroups go to the 8 XCDs round-robin in linear grid order.  split = linear % splits puts every tile of one
        // K-slice on the same XCD(s), so a slice of A and B is fetched from HBM once and shared through that XCD's L2
        // (PMC, round 1: with the slices spread over all XCDs the dW GEMMs read 2.3-7x their operand bytes).
        const int lin = bid + (int)blockIdx.y * nwg, splits = (int)gridDim.y;
        split = lin % splits;
        swz = lin / splits;
    } else {
        const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
        swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    }
    int tile_m, tile_n;
    tile_coords(p, swz, tile_m, tile_n);
    const int m0 = tile_m * BM, n0 = tile_n * BN;

    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // rows of an M/N-contiguous (k-major) operand may be read up to the next multiple of 8 (inside ld)
    const int a_rows = A_KMAJOR ? ((p.M + 7) & ~7) : p.M;
    const int b_rows = B_KMAJOR ? ((p.N + 7) & ~7) : p.N;
    const int nk_all = (p.K + BK - 1) / BK;
    const int nk_per = SPLITK ? (nk_all + (int)gridDim.y - 1) / (int)gridDim.y : nk_all;
    const int kt0 = SPLITK ? split * nk_per : 0;
    const int nk = min(nk_per, nk_all - kt0);
    if (nk <= 0 && !(SPLITK && p.ws)) return;                // block-uniform (a slice plane must be written even when its K range is empty)

    Stage sa, sb;
    if (nk > 0) {
        stage_load<A_KMAJOR>(sa, p.A, p.lda, m0, a_rows, kt0 * BK, p.K, tid);
        stage_load<B_KMAJOR>(sb, p.B, p.ldb, n0, b_rows, kt0 * BK, p.K, tid);
        stage_store<A_KMAJOR, PK>(sa, smem, tid);
        stage_store<B_KMAJOR, PK>(sb, smem + OPERAND_BYTES, tid);
    }
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const unsigned char* la = smem + (t & 1) * STAGE_BYTES;
        const unsigned char* lb = la + OPERAND_BYTES;
        const bool more = (t + 1) < nk;
        if (more) {   // issue the next tile's global loads before the MFMA phase (T14)
            stage_load<A_KMAJOR>(sa, p.A, p.lda, m0, a_rows, (kt0 + t + 1) * BK, p.K, tid);
            stage_load<B_KMAJOR>(sb, p.B, p.ldb, n0, b_rows, (kt0 + t + 1) * BK, p.K, tid);
        }
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) fa[i] = frag_read<A_KMAJOR, PK>(la, wm * 64 + i * 16, ks, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j) fb[j] = frag_read<B_KMAJOR, PK>(lb, wn * 64 + j * 16, ks, lane);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = SPLITK ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0)
                                       : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            unsigned char* na = smem + ((t + 1) & 1) * STAGE_BYTES;
            stage_store<A_KMAJOR, PK>(sa, na, tid);
            stage_store<B_KMAJOR, PK>(sb, na + OPERAND_BYTES, tid);
        }
        __syncthreads();
    }

    gemm_epilogue<SPLITK>(p, acc, m0, n0, wm, wn, lane, split);
}


// ----------------------------------------------------------------------------------------------------------------
// DMA-staged LDS images (used by the 256^2 kernel): operands go HBM -> LDS directly (buffer_load_dwordx4 ... lds, 1 KiB
// per wave-instruction, no VGPR staging and no ds_write pass).
// A DMA writes LDS linearly (wave-uniform base + lane * 16 B), so the conflict-avoiding permutation moves to the
// per-lane SOURCE address and the fragment reads apply the same involution (cdna_hip_programming.md rule 21):
//   R  image: [row][8 chunks of 16 B], chunk ^= row & 7                      (ds_read_b128, as before)
//   Cf image: [k-row][16 chunks] with NO padding; the 32-byte window a transposed read touches is XOR-moved by the
//             k-row: chunk ^= swz(k) << 1 with swz = k & 7 when both operands are k-major (a half-wave reads 8
//             consecutive k-rows) and swz = (k & 3) | ((k >> 3) & 1) << 2 for the natural k order (rows {0-3, 8-11}):
//             8 rows -> 8 disjoint 32-byte windows = all 64 banks.
// Out-of-range chunks (M/N/K tails) come back as zeros from the buffer descriptor's range check.


template <bool PK>
__device__ __forceinline__ int cf_swz(int kr) { return PK ? (kr & 7) : ((kr & 3) | (((kr >> 3) & 1) << 2)); }

template <bool KMAJOR, bool PK>
__device__ __forceinline__ bf16x8 dma_frag_read(const unsigned char* lds, int r0, int ks, int lane) {
    const int g = lane >> 4, i = lane & 15;
    if (!KMAJOR) {
        const int r = r0 + i, kc = ks * 4 + g;
        u32x4 v = *reinterpret_cast<const u32x4*>(lds + r * 128 + ((kc ^ (r & 7)) << 4));
        return __builtin_bit_cast(bf16x8, v);
    } else {
        const int q = i >> 2, p = i & 3;
        const int kr0 = ks * 32 + (PK ? 4 : 8) * g + q, kr1 = kr0 + (PK ? 16 : 4);
        const int ch = (r0 >> 3) + (p >> 1), half = (p & 1) * 8;
        s16x4 lo = lds_read_tr16(lds + kr0 * 256 + ((ch ^ (cf_swz<PK>(kr0) << 1)) << 4) + half);
        s16x4 hi = lds_read_tr16(lds + kr1 * 256 + ((ch ^ (cf_swz<PK>(kr1) << 1)) << 4) + half);
        s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        return __builtin_bit_cast(bf16x8, v);
    }
}

// ----------------------------------------------------------------------------------------------------------------
// 256 x 256 x 64 tile, 8 waves (2 x 4, 128 x 64 outputs each), one workgroup per CU: the large-tile kernel for the
// non-split problems (cdna_hip_programming.md 5: the 128^2 two-barrier structure tops out near 0.9 PF on zeros; what
// breaks that ceiling is ~1 block/CU, all-DMA staging kept in flight across raw barriers with counted vmcnt, and a
// per-phase interleave of LDS reads, DMA issue and MFMA).  Own schedule, built from those rules:
//
//   * a K-tile is staged as 4 UNITS of 16 KiB (128 rows x 64 k): e = 0 A-sub0, 1 B-first, 2 B-second, 3 A-sub1, where
//     sub s of A holds rows {wr*128 + s*64 + 0..63} (wr = wave row) and sub s of B holds columns {wc*64 + s*32 +
//     0..31}: every wave finds the 64 x 64-k (A) or 32 x 64-k (B) register subtile of a phase in ONE unit.  LDS = 2
//     tiles x 4 units = 128 KiB; unit q = 4 T + e (T = K-tile) lives in slot (T & 1) * 4 + e.
//   * a tile is 4 PHASES of 16 MFMA per wave (one 64 x 32 quadrant x K = 64) in snake order, so three of the four
//     register subtiles carry over:  even tile (A0,B0) (A0,B1) (A1,B1) (A1,B0);  odd tile (A0,B1) (A0,B0) (A1,B0)
//     (A1,B1)  ->  "B-first" is sub 0 on even tiles and sub 1 on odd ones.
//   * fragments are prefetched in registers so that no MFMA waits on an LDS read issued less than half a phase (8 MFMA)
//     earlier: B has two register sets (B0, B1: the next one is read a phase ahead); A has ONE (32 VGPR), refilled
//     by k-half: the k 0..31 half is re-read in the middle of the phase that last uses it, the k 32..63 half at the
//     start of the next phase.  Unit q is therefore read in phases q-2 .. q.
//   * phase P DMA-stages unit P + 6 into the slot whose previous tenant (unit P - 2) was last read in phase P - 2 (WAR:
//     a counted lgkmcnt before this phase's barrier retires those reads in every wave) and waits -- vmcnt(6): three
//     younger units stay in flight -- for unit P + 2, issued four phases earlier and first read after this barrier
//     (RAW: counted vmcnt, then a barrier the reader has passed).  vmcnt is never 0 inside the loop.
//   * operands go through buffer descriptors (buffer_load_dwordx4 ... lds): one loop-invariant 32-bit voffset per
//     thread and chunk, the K advance in the scalar offset, out-of-range rows / K-tail served as zeros by the range
//     check -- no 64-bit address math or predicates in the loop (the first version spilled 250+ VGPRs on them).
// s_setprio(1) around the MFMA clusters (cdna_hip_programming.md T5): measured -3 % end to end on this schedule (two
// builds, same box, 2 x 10 steps each), so it stays off.
#ifndef G2_SETPRIO
#define G2_SETPRIO 0
#endif
constexpr int G2_UNIT = 128 * 128;                     // 16 KiB
constexpr int G2_SMEM = 8 * G2_UNIT + 8 * 4096;        // 128 KiB of staging slots + a 4 KiB epilogue transpose pad per wave
typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct G2Tile {                       // wave-uniform description of one output tile's operand panels
    const bf16_t* a;
    const bf16_t* b;
    unsigned a_bytes, b_bytes;
    int m0, n0, t0;                   // t0: first K-tile of this work item's K slice
};

template <bool A_KMAJOR, bool B_KMAJOR, bool UNSWAP = false, bool XA = false>
struct G2 {
    static constexpr bool PK = A_KMAJOR && B_KMAJOR;
    static constexpr int FA = A_KMAJOR ? 2 : 1, FB = B_KMAJOR ? 2 : 1;      // LGKM ops per fragment read
    static constexpr int cap(int n) { return n > 15 ? 15 : n; }
    static constexpr int LG0 = cap(4 * FA + 4 * FB), LG1 = cap(4 * FA), LG2 = cap(4 * FA), LG3 = cap(4 * FA + 4 * FB);

    // Physical placement of slot PAR*4 + e: the A units in 16-KiB slots 0-3, the B units in 4-7, so that every A read is
    // base + immediate (< 64 KiB, the ds offset field) and every B read is (base + 64 KiB) + immediate: 2 + 2 address
    // VGPRs instead of 4 + 4 (three spilled address registers cost a vmcnt(0) drain per K-loop iteration).
    static constexpr int phys(int slot) { return ((slot & 3) == 0 || (slot & 3) == 3) ? (slot >> 2) * 2 + ((slot & 3) == 3) : 4 + (slot >> 2) * 2 + ((slot & 3) == 2); }
    unsigned char* smem;
    int lane, wr, wc, wave_off, nk, tb, ta;
    int va[2][2], vb[2][2];          // voffset[sub][chunk]
    unsigned step_a, step_b;         // bytes per K-tile
    unsigned lds0;                   // LDS byte address of smem
    G2Tile cur, nxt;                 // the tile being computed and the one whose first units are already being staged

    __device__ __forceinline__ static G2Tile tile_desc(const GemmParams& p, int idx, int ntiles) {
        G2Tile d;
        if (idx >= ntiles) {           // past the end: zero-sized panels, every DMA returns zeros
            d.a = p.A; d.b = p.B; d.a_bytes = 0; d.b_bytes = 0; d.m0 = 0; d.n0 = 0; d.t0 = 0;
            return d;
        }
        // work item = (K slice, output tile), slice-major: an XCD's contiguous chunk of items shares its slice of A and B
        const int tiles = p.tiles_m * p.tiles_n, slice = idx / tiles;
        d.t0 = slice * p.g2_nk;
        int tile_m, tile_n;
        tile_coords(p, idx - slice * tiles, tile_m, tile_n, p.g2_gn);
        d.m0 = tile_m * 256; d.n0 = tile_n * 256;
        if (XA) {
            // fused cross-attention: the tile's 256 "A rows" are [K_h | V_h | K_h+1 | V_h+1] for the head pair h = 2 tile_m: p.A points
            // at W_k (in_proj rows d ..), W_v lies x_d rows further on; init_lane bakes the sub-panel offsets into the voffsets
            d.a = p.A + (size_t)tile_m * 128 * p.lda;
            d.a_bytes = (unsigned)((p.x_d + 128) * p.lda * 2);
        } else if (!A_KMAJOR) {
            d.a = p.A + (size_t)d.m0 * p.lda;
            d.a_bytes = (unsigned)(min(p.M - d.m0, 256) * p.lda * 2);
        } else {
            d.a = p.A + d.m0;
            d.a_bytes = (unsigned)(((size_t)(p.K - 1) * p.lda + min(((p.M + 7) & ~7) - d.m0, 256)) * 2);
        }
        if (!B_KMAJOR) {
            d.b = p.B + (size_t)d.n0 * p.ldb;
            d.b_bytes = (unsigned)(min(p.N - d.n0, 256) * p.ldb * 2);
        } else {
            d.b = p.B + d.n0;
            d.b_bytes = (unsigned)(((size_t)(p.K - 1) * p.ldb + min(((p.N + 7) & ~7) - d.n0, 256)) * 2);
        }
        return d;
    }

    __device__ __forceinline__ void init(const GemmParams& p, unsigned char* smem_, int tid) {
        smem = smem_;
        lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem_;
        const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wr = wave >> 2; wc = wave & 3; wave_off = wave * 1024;
        nk = p.g2_nk;
        step_a = A_KMAJOR ? 128u * (unsigned)p.lda : 128u;
        step_b = B_KMAJOR ? 128u * (unsigned)p.ldb : 128u;
        init_lane(p, tid);
    }
    // Per-lane K-loop state (DMA voffsets; the fragment read addresses derive from `lane`).  Recomputed after every
    // epilogue from a LAUNDERED thread id, so that it is not live while the epilogue needs the registers (and the
    // epilogue's own per-lane constants are not live across the K loop): kept live, the two sets spilled to scratch
    // and every tile paid ~10 us of serialized scratch reloads.
    __device__ __forceinline__ void init_lane(const GemmParams& p, int tid) {
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        {   // bases of the k-major transposed reads: k-row (8 or, permuted order, 4) g + q, swizzled chunk of the wave's columns
            const int g4 = lane >> 4, q = (lane >> 2) & 3, pp = lane & 3;
            const int kr = (PK ? 4 : 8) * g4 + q;
            tb = kr * 256 + ((((wc * 4) + (pp >> 1)) ^ (cf_swz<PK>(kr) << 1)) << 4) + (pp & 1) * 8;
            ta = kr * 256 + ((((wr * 8) + (pp >> 1)) ^ (cf_swz<PK>(kr) << 1)) << 4) + (pp & 1) * 8;
        }
#pragma unroll
        for (int sub = 0; sub < 2; ++sub)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const int c = u * 512 + tid;
                if (XA) {            // wave row r >> 6 = head of the pair; sub 0 = its 64 W_k rows, sub 1 = its 64 W_v rows
                    const int r = c >> 3, kc = (c & 7) ^ (r & 7);
                    va[sub][u] = ((r >> 6) * 64 + (r & 63) + sub * p.x_d) * p.lda * 2 + kc * 16;
                } else if (!A_KMAJOR) {
                    const int r = c >> 3, kc = (c & 7) ^ (r & 7);
                    va[sub][u] = ((r >> 6) * 128 + sub * 64 + (r & 63)) * p.lda * 2 + kc * 16;
                } else {
                    const int kr = c >> 4, ul = ((c & 15) ^ (cf_swz<PK>(kr) << 1)) * 8;
                    va[sub][u] = (kr * p.lda + (ul >> 6) * 128 + sub * 64 + (ul & 63)) * 2;
                }
                if (!B_KMAJOR) {
                    const int r = c >> 3, kc = (c & 7) ^ (r & 7);
                    vb[sub][u] = ((r >> 5) * 64 + sub * 32 + (r & 31)) * p.ldb * 2 + kc * 16;
                } else {
                    const int kr = c >> 4, ul = ((c & 15) ^ (cf_swz<PK>(kr) << 1)) * 8;
                    vb[sub][u] = (kr * p.ldb + (ul >> 5) * 64 + sub * 32 + (ul & 31)) * 2;
                }
            }
    }

    // unit e (0 A-sub0 | 1 B-first | 2 B-second | 3 A-sub1) of K-tile T, whose parity is PAR -> slot PAR*4 + e.
    // T >= nk continues into the NEXT output tile (nk is even, so slot parities carry over): no pipeline drain between tiles.
    // The DMA is issued from inline asm ON PURPOSE: hipcc orders a builtin LDS-DMA against later LDS reads it cannot
    // disambiguate -- the transposed-read intrinsics -- with s_waitcnt vmcnt(0), which drained the pipeline 3-4 times per
    // 8 phases in every k-major variant.  All ordering of these loads is done by hand (fence(), the prologue, the final
    // vmcnt(0)); the compiler's own vmcnt waits for its epilogue loads stay correct, older or younger DMAs only make them
    // wait longer.  M0 = LDS byte address of the wave's 1 KiB piece (wave-uniform base + lane * 16).
    __device__ __forceinline__ void dma2(const bf16_t* base, unsigned bytes, unsigned lds_addr, int v0, int v1, unsigned so) {
        const unsigned long long b = (unsigned long long)base;
        const u32x4 rs = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, bytes, 0x00020000u};
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %2, %4, %5 offen lds\n\t"
                     "s_mov_b32 m0, %1\n\tbuffer_load_dwordx4 %3, %4, %5 offen lds"
                     ::"s"(lds_addr), "s"(lds_addr + 8192u), "v"(v0), "v"(v1), "s"(rs), "s"(so) : "memory");
    }
    template <int PAR, int E>
    __device__ __forceinline__ void stage(int T) {
        const unsigned slot = lds0 + phys(PAR * 4 + E) * G2_UNIT + wave_off;
        constexpr int sub = (E == 0) ? 0 : (E == 3) ? 1 : (E == 1) ? PAR : (PAR ^ 1);
        const bool nx = T >= nk;
        const int Te = nx ? T - nk : T;
        if (E == 0 || E == 3)
            dma2(nx ? nxt.a : cur.a, nx ? nxt.a_bytes : cur.a_bytes, slot, va[sub][0], va[sub][1], (unsigned)(Te + (nx ? nxt.t0 : cur.t0)) * step_a);
        else
            dma2(nx ? nxt.b : cur.b, nx ? nxt.b_bytes : cur.b_bytes, slot, vb[sub][0], vb[sub][1], (unsigned)(Te + (nx ? nxt.t0 : cur.t0)) * step_b);
    }
    template <int SLOT, int KS>
    __device__ __forceinline__ void read_a(bf16x8 (&ra)[8]) {
        if constexpr (A_KMAJOR) {        // as read_b: fragment i flips chunk bits 1-2, the second half sits 4 (16 if permuted) k-rows on
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned char* a = smem + phys(SLOT) * G2_UNIT + KS * 8192 + (ta ^ (i << 5));
                const s16x4 lo = lds_read_tr16(a), hi = lds_read_tr16(a + (PK ? 4096 : 1024));
                const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                ra[KS * 4 + i] = __builtin_bit_cast(bf16x8, v);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[KS * 4 + i] = dma_frag_read<A_KMAJOR, PK>(smem + phys(SLOT) * G2_UNIT, wr * 64 + i * 16, KS, lane);
        }
    }
    template <int SLOT>
    __device__ __forceinline__ void read_b(bf16x8 (&rb)[4]) {
        if constexpr (B_KMAJOR) {
            // k-major B, natural k order: the XOR swizzle of a lane's chunk does not depend on ks or on the lo/hi half, and
            // fragment j only flips chunk bit 1, so every address is (tb ^ (j << 5)) + immediate -- written that way the
            // 8 transposed reads share one VGPR (as "(ch ^ swz) << 4" per fragment hipcc kept 16 and spilled in the K loop)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const unsigned char* a = smem + phys(SLOT) * G2_UNIT + ks * 8192 + (tb ^ (j << 5));
                    const s16x4 lo = lds_read_tr16(a), hi = lds_read_tr16(a + (PK ? 4096 : 1024));
                    const s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    rb[ks * 2 + j] = __builtin_bit_cast(bf16x8, v);
                }
        } else {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int j = 0; j < 2; ++j) rb[ks * 2 + j] = dma_frag_read<B_KMAJOR, PK>(smem + phys(SLOT) * G2_UNIT, wc * 32 + j * 16, ks, lane);
        }
    }
    template <int SUBA, int SUBB, int KS>
    __device__ __forceinline__ void mma(f32x4 (&acc)[8][4], const bf16x8 (&ra)[8], const bf16x8 (&rb)[4]) {
        if (G2_SETPRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                acc[SUBA * 4 + i][SUBB * 2 + j] =     // UNSWAP (atomic epilogue): lane holds C[4 (lane>>4) + r][lane&15] of each 16 x 16 block
                    // XA: the K half of the wave tile (SUBA 0) is issued un-swapped -> K^T[dim][key] (dims in registers, keys on lanes);
                    // the V half swapped -> V[key][dim]: exactly the forms the attention epilogue chains into its MFMAs
                    (UNSWAP || (XA && SUBA == 0)) ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(ra[KS * 4 + i], rb[KS * 2 + j], acc[SUBA * 4 + i][SUBB * 2 + j], 0, 0, 0)
                           : __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[KS * 2 + j], ra[KS * 4 + i], acc[SUBA * 4 + i][SUBB * 2 + j], 0, 0, 0);
        if (G2_SETPRIO) __builtin_amdgcn_s_setprio(0);
    }
    template <int LG>
    __device__ __forceinline__ void fence() {     // my part of unit P+2 landed, my reads of phase P-2 retired; then everyone's
        asm volatile("s_waitcnt vmcnt(6) lgkmcnt(%0)" ::"n"(LG) : "memory");
        __builtin_amdgcn_s_barrier();
    }

    // 8 phases = K-tiles t (even) and t + 1 (odd).  Slot = parity * 4 + e.  LAST (the output tile's final pair) leaves the
    // two reads that belong to the next output tile's K-tile 0 to the caller (next_tile_reads, after the epilogue), so
    // that the fragment registers are dead while the epilogue needs its own.
    template <bool LAST>
    __device__ __forceinline__ void two_tiles(int t, f32x4 (&acc)[8][4], bf16x8 (&ra)[8], bf16x8 (&rb0)[4], bf16x8 (&rb1)[4]) {
        // ---- even tile: B-first = sub 0 (rb0), B-second = sub 1 (rb1)
        fence<LG3>();  read_a<0, 1>(ra);  read_b<2>(rb1);  stage<1, 2>(t + 1);
        mma<0, 0, 0>(acc, ra, rb0);  mma<0, 0, 1>(acc, ra, rb0);
        fence<LG0>();  stage<1, 3>(t + 1);
        mma<0, 1, 0>(acc, ra, rb1);  read_a<3, 0>(ra);  mma<0, 1, 1>(acc, ra, rb1);
        fence<LG1>();  read_a<3, 1>(ra);  stage<0, 0>(t + 2);
        mma<1, 1, 0>(acc, ra, rb1);  mma<1, 1, 1>(acc, ra, rb1);
        fence<LG2>();  read_b<5>(rb1);  stage<0, 1>(t + 2);
        mma<1, 0, 0>(acc, ra, rb0);  read_a<4, 0>(ra);  mma<1, 0, 1>(acc, ra, rb0);
        // ---- odd tile: B-first = sub 1 (rb1), B-second = sub 0 (rb0)
        fence<LG3>();  read_a<4, 1>(ra);  read_b<6>(rb0);  stage<0, 2>(t + 2);
        mma<0, 1, 0>(acc, ra, rb1);  mma<0, 1, 1>(acc, ra, rb1);
        fence<LG0>();  stage<0, 3>(t + 2);
        mma<0, 0, 0>(acc, ra, rb0);  read_a<7, 0>(ra);  mma<0, 0, 1>(acc, ra, rb0);
        fence<LG1>();  read_a<7, 1>(ra);  stage<1, 0>(t + 3);
        mma<1, 0, 0>(acc, ra, rb0);  mma<1, 0, 1>(acc, ra, rb0);
        fence<LG2>();  if (!LAST) read_b<1>(rb0);  stage<1, 1>(t + 3);
        mma<1, 1, 0>(acc, ra, rb1);  if (!LAST) read_a<0, 0>(ra);  mma<1, 1, 1>(acc, ra, rb1);
    }
    __device__ __forceinline__ void next_tile_reads(bf16x8 (&ra)[8], bf16x8 (&rb0)[4]) {
        read_b<1>(rb0);
        read_a<0, 0>(ra);
    }

    // ---- fp8 (e4m3) operands: the SAME bytes, units, slots and DMA schedule -- a 128-byte LDS row is 128 k values instead of 64 --
    // on v_mfma_scale_f32_16x16x128_f8f6f4 (unit block scales), which consumes 32 bytes per lane and operand: the two 16-byte chunks
    // the bf16 form reads for k-steps 0 and 1 of a K-tile (chunks g and 4 + g: the scaled MFMA pairs byte j of lane l in A with byte
    // j of lane l in B, so any k assignment works as long as both operands use the same one).  8 MFMAs of 32 cycles per phase instead of
    // 16 of 16: the K loop's time per byte is unchanged, its flops double.  One MFMA needs BOTH halves of a fragment, so the A set
    // cannot be refilled by k-half behind the k-step-0 MFMAs as above: it is refilled by FRAGMENT -- after the two MFMAs that last use
    // fragment i (in the phases after which the A subtile changes: 2, 4, 6, 8), ~6 MFMAs ahead of its next use.
    typedef __attribute__((ext_vector_type(8))) int i32x8;
    template <int SLOT, int I>
    __device__ __forceinline__ void f8_read_a(i32x8 (&fa)[4]) {
        const u32x4 lo = __builtin_bit_cast(u32x4, dma_frag_read<false, false>(smem + phys(SLOT) * G2_UNIT, wr * 64 + I * 16, 0, lane));
        const u32x4 hi = __builtin_bit_cast(u32x4, dma_frag_read<false, false>(smem + phys(SLOT) * G2_UNIT, wr * 64 + I * 16, 1, lane));
        fa[I] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
    }
    template <int SLOT>
    __device__ __forceinline__ void f8_read_a_all(i32x8 (&fa)[4]) {
        f8_read_a<SLOT, 0>(fa); f8_read_a<SLOT, 1>(fa); f8_read_a<SLOT, 2>(fa); f8_read_a<SLOT, 3>(fa);
    }
    template <int SLOT>
    __device__ __forceinline__ void f8_read_b(i32x8 (&fb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const u32x4 lo = __builtin_bit_cast(u32x4, dma_frag_read<false, false>(smem + phys(SLOT) * G2_UNIT, wc * 32 + j * 16, 0, lane));
            const u32x4 hi = __builtin_bit_cast(u32x4, dma_frag_read<false, false>(smem + phys(SLOT) * G2_UNIT, wc * 32 + j * 16, 1, lane));
            fb[j] = i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        }
    }
    template <int SUBA, int SUBB, int I>
    __device__ __forceinline__ void f8_mma_i(f32x4 (&acc)[8][4], const i32x8 (&fa)[4], const i32x8 (&fb)[2]) {
#pragma unroll
        for (int j = 0; j < 2; ++j)          // swapped issue, as the bf16 form: lane holds C[16 i + (lane & 15)][16 j + 4 (lane >> 4) .. + 3]
            // From inline asm, accumulator tied in place: through the builtin hipcc allocates this instruction's results away from its
            // srcC (copies of accumulators, 98-120 spilled registers in this loop -- the same schedule on bf16 MFMAs: none).  What the
            // compiler then no longer knows: the result latency (two_tiles_fp8's caller pads it before the epilogue reads acc; inside
            // the loop an accumulator is next touched a K-tile later); s_waitcnt for the LDS reads feeding it is still its job.
            asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %3 op_sel_hi:[0,0,0]"
                         : "+v"(acc[SUBA * 4 + I][SUBB * 2 + j]) : "v"(fb[j]), "v"(fa[I]), "v"(0x7f7f7f7f));
    }
    template <int SUBA, int SUBB>
    __device__ __forceinline__ void f8_mma(f32x4 (&acc)[8][4], const i32x8 (&fa)[4], const i32x8 (&fb)[2]) {
        f8_mma_i<SUBA, SUBB, 0>(acc, fa, fb); f8_mma_i<SUBA, SUBB, 1>(acc, fa, fb); f8_mma_i<SUBA, SUBB, 2>(acc, fa, fb); f8_mma_i<SUBA, SUBB, 3>(acc, fa, fb);
    }
    // the phase whose A subtile changes afterwards: fragment i of the NEXT subtile (slot NSLOT) is read behind the MFMAs that last use i
    template <int SUBA, int SUBB, int NSLOT, bool READ>
    __device__ __forceinline__ void f8_mma_refill(f32x4 (&acc)[8][4], i32x8 (&fa)[4], const i32x8 (&fb)[2]) {
        f8_mma_i<SUBA, SUBB, 0>(acc, fa, fb); if (READ) f8_read_a<NSLOT, 0>(fa);
        f8_mma_i<SUBA, SUBB, 1>(acc, fa, fb); if (READ) f8_read_a<NSLOT, 1>(fa);
        f8_mma_i<SUBA, SUBB, 2>(acc, fa, fb); if (READ) f8_read_a<NSLOT, 2>(fa);
        f8_mma_i<SUBA, SUBB, 3>(acc, fa, fb); if (READ) f8_read_a<NSLOT, 3>(fa);
    }
    // LDS reads issued per phase: 4 (B) | 8 (A refill) | 0 | 12 -> the fence of the NEXT phase lets exactly those stay in flight (the reads
    // of the phase before them -- the tenants of the slot this phase's DMA overwrites -- have then retired)
    template <bool LAST>
    __device__ __forceinline__ void two_tiles_fp8(int t, f32x4 (&acc)[8][4], i32x8 (&fa)[4], i32x8 (&fb0)[2], i32x8 (&fb1)[2]) {
        // ---- even tile
        fence<12>();  f8_read_b<2>(fb1);  stage<1, 2>(t + 1);
        f8_mma<0, 0>(acc, fa, fb0);
        fence<4>();   stage<1, 3>(t + 1);
        f8_mma_refill<0, 1, 3, true>(acc, fa, fb1);
        fence<8>();   stage<0, 0>(t + 2);
        f8_mma<1, 1>(acc, fa, fb1);
        fence<0>();   f8_read_b<5>(fb1);  stage<0, 1>(t + 2);
        f8_mma_refill<1, 0, 4, true>(acc, fa, fb0);
        // ---- odd tile
        fence<12>();  f8_read_b<6>(fb0);  stage<0, 2>(t + 2);
        f8_mma<0, 1>(acc, fa, fb1);
        fence<4>();   stage<0, 3>(t + 2);
        f8_mma_refill<0, 0, 7, true>(acc, fa, fb0);
        fence<8>();   stage<1, 0>(t + 3);
        f8_mma<1, 0>(acc, fa, fb0);
        fence<0>();   if (!LAST) f8_read_b<1>(fb0);  stage<1, 1>(t + 3);
        f8_mma_refill<1, 1, 0, !LAST>(acc, fa, fb1);
    }
    __device__ __forceinline__ void next_tile_reads_fp8(i32x8 (&fa)[4], i32x8 (&fb0)[2]) {
        f8_read_b<1>(fb0);
        f8_read_a_all<0>(fa);
    }
};

// Epilogue class 12 (i2t_gemm_bf16_top2: the lm_head of a greedy decode step).  The output tile is NOT stored: every 64-column segment
// of a row leaves its two largest values and their columns -- C[row][segment] = {v1, column1, v2, column2}, value descending, the lower
// column first on ties (torch.argmax's rule) -- 16 bytes instead of 256.  The 823 MB of fp32 logits a 4096-caption step wrote and
// the n-gram-ban / argmax kernel read back become 51 MB; i2t_top2_ngram_argmax (decode.hip) merges the segments.
__device__ __forceinline__ void top2_epilogue(const GemmParams& p, f32x4 (&acc)[8][4], int mbase, int nbase, int lane) {
    const int g = lane >> 4, li = lane & 15;
    const int seg = nbase >> 6;
    if (seg >= p.ldc) return;                          // wave-uniform: a segment wholly past N
    f32x4* out = reinterpret_cast<f32x4*>(p.C);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        float v1 = -INFINITY, v2 = -INFINITY;
        int i1 = 0x7fffffff, i2 = 0x7fffffff;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {              // a lane's columns in ascending order: strict > keeps the first of equals
                const int col = nbase + 16 * j + 4 * g + r;
                const float v = col < p.N ? acc[i][j][r] : -INFINITY;
                if (v > v1) {
                    v2 = v1; i2 = i1; v1 = v; i1 = col;
                } else if (v > v2) {
                    v2 = v; i2 = col;
                }
            }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {           // the four lanes that hold a row's other columns
            const float ov1 = __shfl_xor(v1, o, 64), ov2 = __shfl_xor(v2, o, 64);
            const int oi1 = __shfl_xor(i1, o, 64), oi2 = __shfl_xor(i2, o, 64);
            const bool of = ov1 > v1 || (ov1 == v1 && oi1 < i1);                   // the other pair holds the best
            const float l1 = of ? v1 : ov1, w2 = of ? ov2 : v2;                    // loser's best against winner's second
            const int li1 = of ? i1 : oi1, wi2 = of ? oi2 : i2;
            const bool ls = l1 > w2 || (l1 == w2 && li1 < wi2);
            v1 = of ? ov1 : v1; i1 = of ? oi1 : i1;
            v2 = ls ? l1 : w2; i2 = ls ? li1 : wi2;
        }
        const int m = mbase + 16 * i + li;
        if (g == 0 && m < p.M) out[(size_t)m * p.ldc + seg] = f32x4{v1, __int_as_float(i1), v2, __int_as_float(i2)};
    }
}

// The kernel's own argument block (GemmParams is the only argument: offset 0 of the kernarg segment), through a pointer the
// compiler cannot connect to the argument `p`: loads through it are scalar loads issued where they are used.
__device__ __forceinline__ const GemmParams* epilogue_params() {
    auto k = __builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(k));
    return (const GemmParams*)k;
}

// Persistent: gridDim.x = min(#CUs, tiles) workgroups, each walks tiles idx, idx + grid, ... ; the DMA stream runs 6
// units ahead of the MFMAs and simply continues into the next tile, so the next tile's first K-tiles land while this
// tile's epilogue runs.  XCD x (workgroups = x mod 8) takes a contiguous chunk of every round of tiles.
template <bool A_KMAJOR, bool B_KMAJOR, int EPI>
__global__ __launch_bounds__(512) void gemm256_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[G2_SMEM];
    const int tid = threadIdx.x;
    const int G = gridDim.x, bid = blockIdx.x, ntiles = p.tiles_m * p.tiles_n * p.g2_splits;      // work items
    const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
    const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);

    if (p.g2_stagger > 0) {             // de-phase the workgroups of an XCD: their epilogues (HBM bursts) then interleave
        const int k = ((bid >> 3) % p.g2_stagger_groups) * p.g2_stagger;
        for (int i = 0; i < k; ++i) __builtin_amdgcn_s_sleep(127);
    }
    G2<A_KMAJOR, B_KMAJOR, EPI == 6, EPI == 8> g;
    g.init(p, smem, tid);
    g.cur = g.tile_desc(p, first, ntiles);
    g.nxt = g.tile_desc(p, first + G, ntiles);
    f32x4 acc[8][4];
    if constexpr (EPI == 9) {            // fp8 operands (i2t_gemm_fp8): same pipeline, the fragment registers hold 32-byte operands
        typename decltype(g)::i32x8 fa[4], fb0[2], fb1[2];
        g.template stage<0, 0>(0); g.template stage<0, 1>(0); g.template stage<0, 2>(0); g.template stage<0, 3>(0);
        g.template stage<1, 0>(1); g.template stage<1, 1>(1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        g.next_tile_reads_fp8(fa, fb0);
        for (int idx = first; idx < ntiles; idx += G) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
            for (int t = 0; t < g.nk - 2; t += 2) g.template two_tiles_fp8<false>(t, acc, fa, fb0, fb1);
            g.template two_tiles_fp8<true>(g.nk - 2, acc, fa, fb0, fb1);
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (8 passes each) before the epilogue reads acc
            const int m0 = g.cur.m0, n0 = g.cur.n0;
            g.cur = g.nxt;
            g.nxt = g.tile_desc(p, idx + 2 * G, ntiles);
            int lane_e = tid & 63;
            asm volatile("" : "+v"(lane_e));
            const GemmParams& pe = *epilogue_params();       // (see the bf16 classes below: epilogue scalars are not kept live across the K loop)
            const int inside = __builtin_amdgcn_readfirstlane((m0 + g.wr * 128 + 128 <= pe.M && n0 + g.wc * 64 + 64 <= pe.N) ? 1 : 0);
            if (inside) epilogue_tile_tr<8, 9, true>(pe, acc, m0 + g.wr * 128, n0 + g.wc * 64, lane_e, smem + 8 * G2_UNIT + (g.wave_off << 2));
            else epilogue_tile_tr<8, 9, false>(pe, acc, m0 + g.wr * 128, n0 + g.wc * 64, lane_e, smem + 8 * G2_UNIT + (g.wave_off << 2));
            g.init_lane(p, tid);
            g.next_tile_reads_fp8(fa, fb0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        return;
    }
    bf16x8 ra[8], rb0[4], rb1[4];

    // prologue: units 0..5 (K-tile 0 and the first half of K-tile 1), then the reads that precede phase 0
    g.template stage<0, 0>(0); g.template stage<0, 1>(0); g.template stage<0, 2>(0); g.template stage<0, 3>(0);
    g.template stage<1, 0>(1); g.template stage<1, 1>(1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    g.template read_b<1>(rb0);
    g.template read_a<0, 0>(ra);

    for (int idx = first; idx < ntiles; idx += G) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int t = 0; t < g.nk - 2; t += 2) g.template two_tiles<false>(t, acc, ra, rb0, rb1);
        g.template two_tiles<true>(g.nk - 2, acc, ra, rb0, rb1);
        const int m0 = g.cur.m0, n0 = g.cur.n0;
        g.cur = g.nxt;
        g.nxt = g.tile_desc(p, idx + 2 * G, ntiles);
        if constexpr (EPI == 8) {        // fused cross-attention: the wave's (image, head) attends out of the accumulators
            int lane_e = tid & 63;
            asm volatile("" : "+v"(lane_e));
            xattn_epilogue(*epilogue_params(), acc, m0 >> 8, n0, g.wr, g.wc, lane_e, smem + 8 * G2_UNIT + (g.wave_off << 2));
        } else if constexpr (EPI == 12) {       // the two largest of every 64-column row segment instead of the tile (greedy decode's lm_head)
            int lane_e = tid & 63;
            asm volatile("" : "+v"(lane_e));
            top2_epilogue(*epilogue_params(), acc, m0 + g.wr * 128, n0 + g.wc * 64, lane_e);
        } else if constexpr (EPI == 6) {        // split-K partial: fp32 atomics, one wave-instruction = 4 rows x 64 contiguous bytes
            int lane_e = tid & 63;
            asm volatile("" : "+v"(lane_e));
            float* C = reinterpret_cast<float*>(p.C);
            const float alpha = eff_alpha(p);
            const int mb = m0 + g.wr * 128 + 4 * (lane_e >> 4), n_ = n0 + g.wc * 64 + (lane_e & 15);
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int m = mb + i * 16 + r, n = n_ + j * 16;
                        if (m < p.M && n < p.N && !(p.g2_dbg & 2)) atomicAdd(C + (size_t)m * p.ldc + n, acc[i][j][r] * alpha);
                    }
        } else {
            int lane_e = tid & 63;
            asm volatile("" : "+v"(lane_e));
            // The epilogue reads its scalars (C, leading dimensions, bias / aux / residual pointers, dropout constants ...) from a
            // LAUNDERED pointer to the kernel-argument segment: taken from `p` they are loaded at kernel entry and stay live across the
            // K loop -- 18-30 SGPRs more than the file holds, and with all 256 VGPRs taken the spills went to SCRATCH, whose reloads
            // (scratch_load + s_waitcnt vmcnt(0), one per 16-row group) drained every store of the tile and the next tile's DMA stream.
            const GemmParams& pe = *epilogue_params();
            // most tiles lie inside C: that copy of the epilogue carries no clamps and no store predicates (wave-uniform choice)
            const int inside = (EPI != 0 && pe.g2_dbg == 0) ?
                __builtin_amdgcn_readfirstlane((m0 + g.wr * 128 + 128 <= pe.M && n0 + g.wc * 64 + 64 <= pe.N) ? 1 : 0) : 0;
            if (pe.g2_dbg & 2) {
                if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(pe.C)[0] = acc[7][3][3];      // keep the accumulators alive
            } else if (inside) {
                epilogue_tile_tr<8, EPI, true>(pe, acc, m0 + g.wr * 128, n0 + g.wc * 64, lane_e, smem + 8 * G2_UNIT + (g.wave_off << 2));
            } else {
                epilogue_tile_tr<8, EPI, false>(pe, acc, m0 + g.wr * 128, n0 + g.wc * 64, lane_e, smem + 8 * G2_UNIT + (g.wave_off << 2));
            }
        }
        g.init_lane(p, tid);
        g.next_tile_reads(ra, rb0);      // units 0, 1 of the next tile landed before the last fence; same LGKM count as in-loop
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ----------------------------------------------------------------------------------------------------------------
// gemm3: 256 x 128 x 64 tiles with TWO accumulator sets per wave -- the epilogue of output tile i runs inside the K loop of
// tile i + 1.  In the 256^2 kernel above a tile's epilogue (bias / GELU / conversions, LDS transposes, 128-256 KB of stores:
// 6.5-13 us) runs with the matrix pipe idle, which is what holds the K = 512 .. 768 GEMMs of the model at 0.7-1.0 PF while the
// K loop alone sustains 1.4 PF.  Here a wave owns a 64 x 64 tile (8 waves as 4 x 2): 64 accumulator registers per set, two sets
// = the 128 of the 256^2 kernel; while set A accumulates tile i + 1, set B (tile i) is drained one 16-row group per K-tile:
// scaled / biased / activated in registers, transposed through the wave's 2 KiB LDS pad as bf16, stored as full 128-byte
// lines.  Price: 1.5x the L2->LDS bytes and 1.33x the LDS reads per MFMA of the 256^2 tile (both have room), a 3-stage
// 48 KiB ring (one barrier per K-tile, DMA two K-tiles ahead) instead of the 8-phase schedule.
// Every VMEM operation inside the loop is issued from inline asm and waited for by hand (counted vmcnt): one compiler-inserted
// s_waitcnt vmcnt(0) for an epilogue load would drain the DMA ring once per K-tile.
constexpr int G3_STAGE = 48 * 1024, G3_B_OFF = 32 * 1024, G3_PAD = 2048;
constexpr int G3_SMEM = 3 * G3_STAGE + 8 * G3_PAD;          // 160 KiB

__device__ __forceinline__ void g3_store16_nt(void* ptr, const u32x4& v) {
    // s_nop: a store of more than 8 bytes reads its data registers after issue; the compiler pads that hazard for its own stores
    // but not for inline asm (seen: the next instruction's zero-initialisation landing in the stored tile)
    asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(ptr), "v"(v) : "memory");
}
// In-place ("+v") on purpose: the destination is an existing variable that the load overwrites where it lives.  With a fresh
// "=v" result assigned to a loop-carried variable the compiler copies the result registers at the join / back-edge -- BEFORE the
// hand-placed wait, i.e. it copies registers the load has not filled yet (seen: garbage bias in every tile but the first).
__device__ __forceinline__ void g3_load16(f32x4& v, const void* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(v) : "v"(ptr) : "memory");
}

struct G3Tile {
    const bf16_t* a;
    const bf16_t* b;
    unsigned a_bytes, b_bytes;
    int m0, n0;
};

template <int EPI>
struct G3 {
    unsigned lds0;
    unsigned char* smem;
    int lane, wave, wr, wc;
    int va0, vb0;
    unsigned sa64, sb64;
    int nk;

    __device__ __forceinline__ static G3Tile tile_desc(const GemmParams& p, int idx, int ntiles) {
        G3Tile d;
        if (idx >= ntiles) { d.a = p.A; d.b = p.B; d.a_bytes = 0; d.b_bytes = 0; d.m0 = 0; d.n0 = 0; return d; }
        int tile_m, tile_n;
        tile_coords(p, idx, tile_m, tile_n, p.g2_gn);
        d.m0 = tile_m * 256; d.n0 = tile_n * 128;
        d.a = p.A + (size_t)d.m0 * p.lda;
        d.a_bytes = (unsigned)(min(p.M - d.m0, 256) * p.lda * 2);
        d.b = p.B + (size_t)d.n0 * p.ldb;
        d.b_bytes = (unsigned)(min(p.N - d.n0, 128) * p.ldb * 2);
        return d;
    }
    __device__ __forceinline__ void init(const GemmParams& p, unsigned char* smem_, int tid) {
        smem = smem_;
        lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)smem_;
        lane = tid & 63;
        wave = __builtin_amdgcn_readfirstlane(tid >> 6);
        wr = wave >> 1; wc = wave & 1;
        nk = (p.K + 63) >> 6;
        const int rl = lane >> 3, c = (lane & 7) ^ rl;               // R image: chunk ^= row & 7, applied on the SOURCE side
        // piece u of a wave covers rows 8 (8 u + wave) + rl: the 64-row step between a wave's pieces goes into the scalar offset
        va0 = (8 * wave + rl) * p.lda * 2 + c * 16;
        vb0 = (8 * wave + rl) * p.ldb * 2 + c * 16;
        sa64 = 64u * (unsigned)p.lda * 2u; sb64 = 64u * (unsigned)p.ldb * 2u;
        init_frag_bases();
    }
    // K-tile kt of tile d -> ring slot `slot`: 4 + 2 one-KiB pieces per wave
    __device__ __forceinline__ void stage(const G3Tile& d, int kt, int slot) {
        const unsigned base = lds0 + slot * G3_STAGE + wave * 1024;
        const unsigned so = (unsigned)kt * 128u;
        const unsigned long long a = (unsigned long long)d.a, b = (unsigned long long)d.b;
        const u32x4 ra_ = {(unsigned)a, (unsigned)(a >> 32) & 0xffffu, d.a_bytes, 0x00020000u};
        const u32x4 rb_ = {(unsigned)b, (unsigned)(b >> 32) & 0xffffu, d.b_bytes, 0x00020000u};
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %4, %5, %6 offen lds\n\t"
                     "s_mov_b32 m0, %1\n\tbuffer_load_dwordx4 %4, %5, %7 offen lds\n\t"
                     "s_mov_b32 m0, %2\n\tbuffer_load_dwordx4 %4, %5, %8 offen lds\n\t"
                     "s_mov_b32 m0, %3\n\tbuffer_load_dwordx4 %4, %5, %9 offen lds"
                     ::"s"(base), "s"(base + 8192u), "s"(base + 16384u), "s"(base + 24576u), "v"(va0), "s"(ra_),
                       "s"(so), "s"(so + sa64), "s"(so + 2u * sa64), "s"(so + 3u * sa64) : "memory");
        asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\t"
                     "s_mov_b32 m0, %1\n\tbuffer_load_dwordx4 %2, %3, %5 offen lds"
                     ::"s"(base + G3_B_OFF), "s"(base + G3_B_OFF + 8192u), "v"(vb0), "s"(rb_), "s"(so), "s"(so + sb64) : "memory");
    }
    // Fragment i of k-step ks: tile[r0 + 16 i + li][32 ks + 8 g .. + 7] at byte (r0 + 16 i + li) * 128 + (((4 ks + g) ^ (li & 7)) << 4): the
    // swizzle does not depend on i (16 i is a multiple of 8) and ks only flips byte-offset bit 6, so every read is one per-lane
    // base (fa / fb, set up once) + the slot base (one add per K-tile) + an immediate: 2 address registers per operand.
    int fa, fb;
    __device__ __forceinline__ void init_frag_bases() {
        const int g = lane >> 4, li = lane & 15;
        const int x = ((g ^ (li & 7)) & 7) << 4;                         // ks = 0 (g < 4: bit 2 of the chunk index comes from ks alone)
        fa = (wr * 64 + li) * 128 + x;
        fb = G3_B_OFF + (wc * 64 + li) * 128 + x;
    }
    template <int KS>
    __device__ __forceinline__ void read_frags(int slot, bf16x8 (&ra)[4], bf16x8 (&rb)[4]) {
        const unsigned char* sa = smem + slot * G3_STAGE + (fa ^ (KS * 64));
        const unsigned char* sb = smem + slot * G3_STAGE + (fb ^ (KS * 64));
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sa + i * 2048));
#pragma unroll
        for (int j = 0; j < 4; ++j) rb[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(sb + j * 2048));
    }
    __device__ __forceinline__ static void mma(f32x4 (&acc)[4][4], const bf16x8 (&ra)[4], const bf16x8 (&rb)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)      // swapped issue: lane holds C[row 16 i + (lane & 15)][cols 16 j + 4 (lane >> 4) .. + 3]
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(rb[j], ra[i], acc[i][j], 0, 0, 0);
    }
};

// One 16-row group (i) of a finished 64 x 64 wave tile, class 1 (bf16 C, optional bias, optional per-(row, third) multipliers):
// registers -> bf16 -> the wave's pad (16 rows x 128 B, 16-B chunks XOR-swizzled by row & 7) -> 2 stores of 8 rows x 128 B.
struct G3Pend {
    int m0, n0;          // tile origin of the pending set (its wave offset is added by the step)
    int live;            // the pending set holds a finished tile
};
// Epilogue steps of a finished 64 x 64 wave tile: class 1 (bf16 C [+ per-(row, third) multipliers]) = 4 steps, one 16-row group
// each; class 2 (GELU with the bf16 pre-activation as a second output) = 7 steps: pre-activation of group s >> 1, then its GELU
// (s = 0 .. 5), and both halves of group 3 in step 6 -- one K-tile of every tile must stay free of a step (the bias quads are
// fetched there, when the registers of `old` are dead), and K = 512 has 8 K-tiles.
template <int EPI>
constexpr int g3_nsteps() { return EPI == 2 ? 7 : 4; }

template <int EPI, int i, bool ACT, bool PRE>
__device__ __forceinline__ void g3_epi_part(const GemmParams& p, const f32x4 (&acc)[4][4], const G3Pend& pd, int wr, int wc, int lane,
                                            unsigned char* pad) {
    if (PRE && !p.aux_out) return;                              // (uniform) nothing to emit
    const int g = lane >> 4, li = lane & 15;
    const int m = pd.m0 + wr * 64 + 16 * i + li, nb = pd.n0 + wc * 64;
    // per-(row, third) multiplier: a wave's 64 columns lie inside ONE third whenever N / 3 is a multiple of 64 (q / k / v of width
    // 512, 768, ...), so the division and the hash are done once per row here instead of once per quad (4 x ~55 VALU instructions
    // in a step that has to fit behind one MFMA block)
    const unsigned sec = (unsigned)(p.N / 3);
    const bool one_third = EPI == 1 && p.drop_mode == 2 && (sec & 63u) == 0;
    float mult = 1.f;
    if (one_third) mult = dropout_keep(p.drop_key + (unsigned)nb / sec, (unsigned)min(m, p.M - 1), p.drop_thr) ? p.drop_scale : 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        f32x4 v = acc[i][j];                                    // alpha = 1 (host), bias already inside
        if (ACT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = gelu_tanh(v[r]);
        }
        if (EPI == 1 && p.drop_mode == 2) {
            if (one_third) {
                v *= mult;
            } else {
                const unsigned third = (unsigned)(nb + 16 * j + 4 * g) / sec;
                v *= dropout_keep(p.drop_key + third, (unsigned)min(m, p.M - 1), p.drop_thr) ? p.drop_scale : 0.f;
            }
        }
        const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        const int slot8 = 4 * j + g;                            // 8-byte slot within the 128-B row
        *reinterpret_cast<u32x2*>(pad + li * 128 + ((((slot8 >> 1) ^ (li & 7))) << 4) + (slot8 & 1) * 8) = pk;
        if (ACT) __builtin_amdgcn_sched_barrier(0);             // one quad's GELU at a time: interleaving all four spilled registers
    }
    bf16_t* dst = PRE ? p.aux_out : reinterpret_cast<bf16_t*>(p.C);
    const int ld = PRE ? p.ld_aux_out : p.ldc;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
        const int row = 8 * it + (lane >> 3), c = lane & 7;
        const u32x4 v = *reinterpret_cast<const u32x4*>(pad + row * 128 + ((c ^ (row & 7)) << 4));
        const int mm = pd.m0 + wr * 64 + 16 * i + row, nn = nb + 8 * c;
        if (mm < p.M && nn < p.N) g3_store16_nt(dst + (size_t)mm * ld + nn, v);
    }
}

template <int EPI, int S>
__device__ __forceinline__ void g3_epi_step(const GemmParams& p, const f32x4 (&acc)[4][4], const G3Pend& pd, int wr, int wc, int lane,
                                            unsigned char* pad) {
    if constexpr (EPI == 2) {
        if constexpr (S == 6) {
            g3_epi_part<EPI, 3, false, true>(p, acc, pd, wr, wc, lane, pad);
            g3_epi_part<EPI, 3, true, false>(p, acc, pd, wr, wc, lane, pad);
        } else {
            g3_epi_part<EPI, (S >> 1), (S & 1) != 0, (S & 1) == 0>(p, acc, pd, wr, wc, lane, pad);
        }
    } else {
        g3_epi_part<EPI, S, false, false>(p, acc, pd, wr, wc, lane, pad);
    }
}

template <int EPI>
__device__ __forceinline__ void g3_epi_all(const GemmParams& p, const f32x4 (&acc)[4][4], const G3Pend& pd, int wr, int wc, int lane,
                                           unsigned char* pad) {
    static_for<g3_nsteps<EPI>()>([&](auto S_) { g3_epi_step<EPI, decltype(S_)::value>(p, acc, pd, wr, wc, lane, pad); });
}

template <int EPI, bool OVERLAP>
__global__ __launch_bounds__(512) void gemm3_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[G3_SMEM];
    const int tid = threadIdx.x;
    const int G = gridDim.x, bid = blockIdx.x, ntiles = p.tiles_m * p.tiles_n;
    const int xcd = bid & 7, q8 = G >> 3, r8 = G & 7;
    const int first = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    G3<EPI> g;
    g.init(p, smem, tid);
    unsigned char* pad = smem + 3 * G3_STAGE + g.wave * G3_PAD;
    const int nk = g.nk;
    // producer state: the K-tile the DMA stream stages next (runs two K-tiles ahead of the MFMAs, across output tiles)
    G3Tile pt = g.tile_desc(p, first, ntiles);
    int p_idx = first, p_kt = 0, p_slot = 0;
    auto produce = [&]() {
        g.stage(pt, p_kt, p_slot);
        p_slot = p_slot == 2 ? 0 : p_slot + 1;
        if (++p_kt == nk) { p_kt = 0; p_idx += G; pt = g.tile_desc(p, p_idx, ntiles); }
    };
    produce();
    produce();
    asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    bf16x8 ra0[4], rb0[4], ra1[4], rb1[4];
    int c_slot = 0;
    g.template read_frags<0>(0, ra0, rb0);
    // cur accumulates the tile in flight; old holds the finished tile whose epilogue is being drained (copied from cur at the
    // tile boundary: 64 register moves per tile, instead of two copies of the loop with the roles swapped)
    f32x4 cur[4][4], old[4][4];
    G3Pend pend{0, 0, 0};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) old[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

#pragma unroll 1
    for (int idx = first; idx < ntiles; idx += G) {
        int tile_m, tile_n;
        tile_coords(p, idx, tile_m, tile_n, p.g2_gn);
        const int m0 = tile_m * 256, n0 = tile_n * 128;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) cur[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const bool drain = OVERLAP && pend.live && !(p.g2_dbg & 2);
        // One K-tile.  LAST: the tile's final K-tile also fetches the bias quads of the tile (issued before this iteration's DMAs,
        // so its wait covers them); they are added when the finished accumulators are moved to `old` right after the loop -- the
        // accumulators themselves only ever see MFMAs (same summation order as the 256^2 kernel: bit-equal results).
        // STEP >= 0: the iteration also drains 16-row group STEP of the previous tile (`old`); peeled, so STEP is a constant.
        f32x4 bias4[4];
        auto k_iter = [&](auto STEP_, auto LAST_) {
            constexpr int STEP = decltype(STEP_)::value;
            constexpr bool LAST = decltype(LAST_)::value;
            const int n_slot = c_slot == 2 ? 0 : c_slot + 1;
            if constexpr (LAST) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bias4[j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    if (p.bias) g3_load16(bias4[j], p.bias + min(n0 + g.wc * 64 + 16 * j + 4 * (g.lane >> 4), p.N - 4));
                }
            }
            produce();                                            // K-tile q + 2 -> the slot whose last readers passed the previous barrier
            g.template read_frags<1>(c_slot, ra1, rb1);
            G3<EPI>::mma(cur, ra0, rb0);
            // K-tile q + 1 landed <=> at most the 6 DMAs of q + 2 are still in flight.  (Loads retire in order, so this holds however
            // the epilogue stores of the previous iteration retire; a count of 8 "for the 2 stores" would be wrong if they retire early.)
            // (Staging q + 3 right after the barrier instead -- two full iterations ahead -- measured 4 % SLOWER: the loop is bound by
            // LDS bandwidth, 176 KB per K-tile = 0.65 us against 0.49 us of MFMA, not by DMA latency.)
            asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            g.template read_frags<0>(n_slot, ra0, rb0);
            // The two waves of a SIMD (w, w + 4) take their epilogue step on opposite sides of the second MFMA block: while one
            // converts / transposes / stores, the other has the matrix pipe to itself.
            if constexpr (STEP >= 0) {
                if (g.wave < 4) g3_epi_step<EPI, STEP>(p, old, pend, g.wr, g.wc, g.lane, pad);
            }
            G3<EPI>::mma(cur, ra1, rb1);
            if constexpr (STEP >= 0) {
                if (g.wave >= 4) g3_epi_step<EPI, STEP>(p, old, pend, g.wr, g.wc, g.lane, pad);
            }
            c_slot = n_slot;
        };
        using T_ = std::true_type; using F_ = std::false_type;
        constexpr int NS = g3_nsteps<EPI>();
        int kt = 0;
        if (drain) {                       // nk > NS (host): the draining K-tiles first
            static_for<NS>([&](auto S_) { k_iter(S_, F_{}); });
            kt = NS;
        }
#pragma unroll 1
        for (; kt < nk - 1; ++kt) k_iter(std::integral_constant<int, -1>{}, F_{});
        k_iter(std::integral_constant<int, -1>{}, T_{});
        if (!OVERLAP) {
            const G3Pend now{m0, n0, 1};
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) cur[i][j] += bias4[j];
            if (!(p.g2_dbg & 2)) {
                g3_epi_all<EPI>(p, cur, now, g.wr, g.wc, g.lane, pad);
            } else if (cur[0][0][0] == 12345.678f) {
                reinterpret_cast<float*>(p.C)[0] = cur[3][3][3];
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) old[i][j] = cur[i][j] + bias4[j];
            pend = G3Pend{m0, n0, 1};
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (OVERLAP && pend.live) {
        if (!(p.g2_dbg & 2)) {
            g3_epi_all<EPI>(p, old, pend, g.wr, g.wc, g.lane, pad);
        } else if (old[0][0][0] == 12345.678f) {
            reinterpret_cast<float*>(p.C)[0] = old[3][3][3];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
}

int g256_epilogue_class(const GemmParams& p) {
    const bool fast4 = (p.ldc & 3) == 0 && (!p.residual || (p.ldr & 3) == 0) && (!p.aux_in || (p.ld_aux_in & 3) == 0) &&
                       (!p.aux_out || (p.ld_aux_out & 3) == 0);
    const bool none = p.act == I2T_ACT_NONE && !p.aux_out;
    // bias-free only: a straddling quad would read bias[N .. N+2]
    if (fast4 && (p.N & 3) != 0 && none && !p.bias && !p.residual && !p.accumulate && !p.drop_mode) return 7;
    if (!fast4 || (p.N & 3) != 0) return 0;
    if (!p.c_is_f32 && none && !p.residual && !p.accumulate && p.drop_mode != 1) return 1;
    if (!p.c_is_f32 && p.act == I2T_ACT_GELU && !p.drop_mode && !p.residual && !p.accumulate) return 2;
    if (!p.c_is_f32 && p.act == I2T_ACT_GELU_DOUT && !p.drop_mode && !p.residual && !p.accumulate) return 10;
    if (!p.c_is_f32 && p.act == I2T_ACT_MUL_AUX && !p.bias && !p.aux_out && !p.drop_mode && !p.residual && !p.accumulate) return 11;
    if (p.c_is_f32 && none && !p.accumulate && p.drop_mode != 2 && (p.residual || p.bias || p.drop_mode)) return 3;
    if (!p.c_is_f32 && p.act == I2T_ACT_DGELU && !p.bias && !p.aux_out && !p.drop_mode && !p.residual && !p.accumulate) return 4;
    if (p.c_is_f32 && none && !p.bias && !p.residual && !p.drop_mode) return 5;
    return 0;
}

// CUs the persistent kernels may occupy: all of them, unless i2t_gemm_set_cu_limit reserved some (see include/i2t.h)
int g_cu_reserve = 0;
int g256_cus() {
    static int n_cu = 0;
    if (!n_cu) {
        int dev = 0;
        hipDeviceProp_t prop;
        n_cu = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    const int r = __atomic_load_n(&g_cu_reserve, __ATOMIC_RELAXED);
    return (r > 0 && r < n_cu - 8) ? n_cu - r : n_cu;
}

void launch_g3(hipStream_t s, GemmParams p, bool overlap) {
    const int n_cu = g256_cus();
    { const char* e = getenv("I2T_G3_DBG"); p.g2_stagger = e ? atoi(e) : 0; }
    p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 127) / 128;
    const int tiles = p.tiles_m * p.tiles_n;
    const dim3 grid(tiles < n_cu ? tiles : n_cu), block(512);
    const int cls = g256_epilogue_class(p);
    if (cls == 2) {
        if (overlap) hipLaunchKernelGGL((gemm3_kernel<2, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm3_kernel<2, false>), grid, block, 0, s, p);
    } else {
        if (overlap) hipLaunchKernelGGL((gemm3_kernel<1, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm3_kernel<1, false>), grid, block, 0, s, p);
    }
}

template <bool B_KMAJOR>
void launch_g256(hipStream_t s, GemmParams p) {
    const int n_cu = g256_cus();
    p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
    p.g2_splits = 1; p.g2_nk = (((p.K + 63) >> 6) + 1) & ~1;
    const int tiles = p.tiles_m * p.tiles_n;
    const dim3 grid(tiles < n_cu ? tiles : n_cu), block(512);
    // forward GEMMs (B^T form) meet classes 1-3, the dX GEMMs (B form) classes 1, 4, 5; anything else runs the generic one
    static const bool log_cls = getenv("I2T_GEMM_LOG") != nullptr;
    if (log_cls)
        fprintf(stderr, "[g256] class %d bk=%d M=%d N=%d K=%d f32=%d bias=%d act=%d auxo=%d auxi=%d res=%d acc=%d drop=%d ldc=%d\n",
                g256_epilogue_class(p), (int)B_KMAJOR, p.M, p.N, p.K, p.c_is_f32, p.bias != nullptr, p.act, p.aux_out != nullptr,
                p.aux_in != nullptr, p.residual != nullptr, p.accumulate, p.drop_mode, p.ldc);
    // forward GEMMs (B^T form) meet classes 1, 2, 3, 5, 7, the dX GEMMs (B form) classes 1, 4, 5; a class that is not built for
    // the layout runs the generic kernel (NO fall-through between cases: a wrong class dereferences a null epilogue operand)
    int cls = g256_epilogue_class(p);
    if (B_KMAJOR ? (cls == 2 || cls == 3 || cls == 7 || cls == 10) : (cls == 4 || cls == 11)) cls = 0;
    if (cls == 1) hipLaunchKernelGGL((gemm256_kernel<false, B_KMAJOR, 1>), grid, block, 0, s, p);
    else if (cls == 5) hipLaunchKernelGGL((gemm256_kernel<false, B_KMAJOR, 5>), grid, block, 0, s, p);
    else if (cls == 2) hipLaunchKernelGGL((gemm256_kernel<false, false, 2>), grid, block, 0, s, p);
    else if (cls == 3) hipLaunchKernelGGL((gemm256_kernel<false, false, 3>), grid, block, 0, s, p);
    else if (cls == 7) hipLaunchKernelGGL((gemm256_kernel<false, false, 7>), grid, block, 0, s, p);
    else if (cls == 4) hipLaunchKernelGGL((gemm256_kernel<false, true, 4>), grid, block, 0, s, p);
    else if (cls == 10) hipLaunchKernelGGL((gemm256_kernel<false, false, 10>), grid, block, 0, s, p);
    else if (cls == 11) hipLaunchKernelGGL((gemm256_kernel<false, true, 11>), grid, block, 0, s, p);
    else hipLaunchKernelGGL((gemm256_kernel<false, B_KMAJOR, 0>), grid, block, 0, s, p);
}

}  // namespace
// fp8 operands on the persistent kernel (called by i2t_gemm_fp8, csrc/fp8.hip): A8 [M][lda] and B8 [N][ldb] e4m3 bytes, C = (A8 . B8^T)
// sa[m] sb[n] (+ bias) (+ residual).  The kernel sees the byte matrices as bf16 matrices of half the width (same bytes per row and per
// K-tile).  False = not eligible (the caller keeps its own 128 x 128 kernel): K % 256, alignment, fewer tiles than the hand-over point.
bool i2t_g256_fp8_try(hipStream_t s, const void* A8, int lda, const float* sa, const void* B8, int ldb, const float* sb, void* C, int ldc,
                      int c_is_f32, int M, int N, int K, const float* bias, int act, const float* residual, int ldr) {
    const char* off = getenv("I2T_FP8_G256");
    if (off && off[0] == '0') return false;
    if (K % 256 != 0 || (lda & 15) || (ldb & 15) || (N & 3) || (ldc & 3) || (residual && (ldr & 3)) || !ALIGNED16(A8) || !ALIGNED16(B8) || !ALIGNED16(C) ||
        (bias && !ALIGNED16(bias)) || !ALIGNED16(sb) || (size_t)M * lda >= (1ull << 32) || (size_t)N * ldb >= (1ull << 32))
        return false;
    if ((long)((M + 255) / 256) * ((N + 255) / 256) < 40) return false;
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)A8; p.B = (const bf16_t*)B8; p.C = C;
    p.M = M; p.N = N; p.K = K / 2; p.lda = lda / 2; p.ldb = ldb / 2; p.ldc = ldc;
    p.alpha = 1.0f; p.bias = bias; p.act = act; p.residual = residual; p.ldr = ldr; p.c_is_f32 = c_is_f32;
    p.scale_a = sa; p.scale_b = sb;
    p.g2_gn = 8;
    const int n_cu = g256_cus();
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    p.g2_splits = 1; p.g2_nk = p.K >> 6;                               // K % 256 == 0: an even number of 128-byte K-tiles
    const int tiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm256_kernel<false, false, 9>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, s, p);
    return true;
}
namespace {

// dW = A^T . B accumulated into an fp32 C (both operands k-major): K slices spread over the CUs when the output has too
// few 256^2 tiles, partial tiles combined with float atomics (C already holds the value to accumulate onto); with enough
// tiles (the tied lm_head / embedding gradient) one slice and a plain read-add-write epilogue.
bool launch_g256_dw(hipStream_t s, GemmParams p) {
    const int n_cu = g256_cus();
    p.tiles_m = (p.M + 255) / 256; p.tiles_n = (p.N + 255) / 256;
    const int tiles = p.tiles_m * p.tiles_n, nk_all = (p.K + 63) >> 6;
    if (tiles >= n_cu || n_cu / tiles < 2 || i2t_det()) {      // (deterministic mode: one K slice, no atomics)
        // (more than half a round of tiles but less than one -- a Qwen2-1.5B down_proj dW, 1536 x 8960 = 210 tiles -- cannot be split:
        // one tile per workgroup on the persistent kernel still beats the 128^2 fallback it used to take, 711 TF)
        if (g256_epilogue_class(p) != 5) return false;
        p.g2_splits = 1; p.g2_nk = (nk_all + 1) & ~1;
        hipLaunchKernelGGL((gemm256_kernel<true, true, 5>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, s, p);
        return true;
    }
    int splits = n_cu / tiles;
    if (const char* e = getenv("I2T_DW_SPLITS")) splits = atoi(e);      // experiments
    int per = ((nk_all + splits - 1) / splits + 1) & ~1;          // even number of K-tiles per slice
    if (per < 8) per = 8;
    splits = (nk_all + per - 1) / per;
    if (splits < 2) return false;
    p.g2_splits = splits; p.g2_nk = per;
    const int items = tiles * splits;
    hipLaunchKernelGGL((gemm256_kernel<true, true, 6>), dim3(items < n_cu ? items : n_cu), dim3(512), 0, s, p);
    return true;
}

// out[n] (+)= sum_m X[m][n]: 16-byte loads (8 columns per lane, 512 columns per wave-row), the 4 waves of a
// workgroup stride the rows of its slice, LDS combine, one float atomic per column per workgroup.
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ X, int ld, int M, int N,
                                                     float* __restrict__ out, const float* __restrict__ alpha_sumsq) {
    __shared__ float part[4][512];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c0 = blockIdx.x * 512 + lane * 8;
    const int rows_per = (M + gridDim.y - 1) / gridDim.y;
    const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 + 8 <= N) {
        int m = r0 + w;
        for (; m + 12 < r1; m += 16) {          // 4 independent 16-byte loads in flight per lane (1 was latency-bound: ~1.4 TB/s)
            u32x4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const u32x4*>(X + (size_t)(m + 4 * u) * ld + c0);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc[2 * e] += bf16lo(v[u][e]);
                    acc[2 * e + 1] += bf16hi(v[u][e]);
                }
        }
        for (; m < r1; m += 4) {
            const u32x4 v = *reinterpret_cast<const u32x4*>(X + (size_t)m * ld + c0);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += bf16lo(v[e]);
                acc[2 * e + 1] += bf16hi(v[e]);
            }
        }
    } else if (c0 < N) {
        for (int m = r0 + w; m < r1; m += 4)
            for (int e = 0; e < 8 && c0 + e < N; ++e) acc[e] += bf16_to_f32(X[(size_t)m * ld + c0 + e]);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[w][lane * 8 + e] = acc[e];
    __syncthreads();
    for (int c = threadIdx.x; c < 512; c += 256) {
        const int n = blockIdx.x * 512 + c;
        if (n < N) atomicAdd(out + n, (part[0][c] + part[1][c] + part[2][c] + part[3][c]) * (alpha_sumsq ? 1.0f / (sqrtf(*alpha_sumsq) + 1e-6f) : 1.0f));
    }
}

// ----------------------------------------------------------------------------------------------------------------
// Skinny GEMM (M <= 64: one decode step for up to 64 captions).  Weight-streaming bound: every weight byte is read
// exactly once, straight from HBM into VGPRs (no LDS round trip -- cdna_hip_programming.md, "GEMV / M <= 16 decode
// weights" row), 16 weight rows x 64 k per wave-step; the few activation rows are re-read from L1/L2.
//   grid.x = N / 16 column tiles, the 4 waves of a workgroup interleave over 64-wide K-steps and combine through LDS;
//   grid.y > 1 (only for the in-place residual form C += x W^T + b with fp32 C): K is also split across workgroups and
//   the partial tiles are added with float atomics, which lifts the N = 768 projections from 48 to 384 workgroups.
// MFMA is issued with the weight fragment as the A operand (rows = n), so a lane owns 4 consecutive n of one m.
template <int MT>
__global__ __launch_bounds__(256) void gemm_skinny_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) float red[4][MT][64][4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int n0 = blockIdx.x * 16;
    const int nsteps_all = (p.K + 63) / 64;
    const int per = (nsteps_all + (int)gridDim.y - 1) / (int)gridDim.y;
    const int s0 = (int)blockIdx.y * per, s1 = min(nsteps_all, s0 + per);
    f32x4 acc[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nrow = n0 + li;
    const bf16_t* wrow = p.B + (size_t)min(nrow, p.N - 1) * p.ldb;
    const bool nvalid = nrow < p.N;
    for (int st = s0 + w; st < s1; st += 4) {
        const int k0 = st * 64 + 8 * g;
        u32x4 w0 = {0u, 0u, 0u, 0u}, w1 = {0u, 0u, 0u, 0u};
        if (nvalid && k0 < p.K) w0 = *reinterpret_cast<const u32x4*>(wrow + k0);
        if (nvalid && k0 + 32 < p.K) w1 = *reinterpret_cast<const u32x4*>(wrow + k0 + 32);
#pragma unroll
        for (int t = 0; t < MT; ++t) {
            const int m = t * 16 + li;
            u32x4 x0 = {0u, 0u, 0u, 0u}, x1 = {0u, 0u, 0u, 0u};
            if (m < p.M && k0 < p.K) x0 = *reinterpret_cast<const u32x4*>(p.A + (size_t)m * p.lda + k0);
            if (m < p.M && k0 + 32 < p.K) x1 = *reinterpret_cast<const u32x4*>(p.A + (size_t)m * p.lda + k0 + 32);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w0), __builtin_bit_cast(bf16x8, x0), acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w1), __builtin_bit_cast(bf16x8, x1), acc[t], 0, 0, 0);
        }
    }
#pragma unroll
    for (int t = 0; t < MT; ++t) *reinterpret_cast<f32x4*>(red[w][t][lane]) = acc[t];
    __syncthreads();
    // wave w finishes m-subtile w, w+4, ... : lane holds C[m = 16 t + li][n0 + 4 g .. +3]
    for (int t = w; t < MT; t += 4) {
        f32x4 v = *reinterpret_cast<const f32x4*>(red[0][t][lane]);
#pragma unroll
        for (int ww = 1; ww < 4; ++ww) v += *reinterpret_cast<const f32x4*>(red[ww][t][lane]);
        const int m = t * 16 + li, n4 = n0 + 4 * g;
        if (m >= p.M) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = n4 + r;
            if (n >= p.N) continue;
            float val = v[r] * eff_alpha(p);
            if (gridDim.y > 1) {            // split-K: C already holds the residual; slice 0 contributes the bias
                if (p.bias && blockIdx.y == 0) val += p.bias[n];
                atomicAdd(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n, val);
                continue;
            }
            if (p.bias) val += p.bias[n];
            if (p.act == I2T_ACT_GELU) val = gelu_tanh(val);
            if (p.residual) val += p.residual[(size_t)m * p.ldr + n];
            if (p.c_is_f32) {
                float* c = reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n;
                *c = p.accumulate ? *c + val : val;
            } else {
                reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + n] = f32_to_bf16(val);
            }
        }
    }
}

// Second stage of the deterministic split-K form: C = epilogue(sum_s ws[s]) with the planes added in slice order (bit-reproducible,
// unlike atomics); a thread owns 4 consecutive columns of one row and runs the ordinary fused epilogue on the sum.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmParams p, int splits) {
    const int nq = (p.N + 3) >> 2;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)p.M * nq) return;
    const int m = (int)(i / nq), n4 = (int)(i - (long)m * nq) * 4;
    const size_t plane = (size_t)p.M * p.N;
    const float* w = p.ws + (size_t)m * p.N + n4;
    f32x4 a = {0.f, 0.f, 0.f, 0.f};
    const bool full = n4 + 4 <= p.N && (p.N & 3) == 0;
    for (int s_ = 0; s_ < splits; ++s_) {
        if (full) {
            a += *reinterpret_cast<const f32x4*>(w + s_ * plane);
        } else {
            for (int r = 0; r < 4; ++r)
                if (n4 + r < p.N) a[r] += w[s_ * plane + r];
        }
    }
    const bool vec_ok = ((p.ldc & 3) == 0) && (!p.residual || (p.ldr & 3) == 0);
    epilogue_quad(p, a, m, n4, vec_ok);
}

template <int MT>
void launch_skinny(hipStream_t s, const GemmParams& p, int ksplit) {
    dim3 grid((p.N + 15) / 16, ksplit);
    hipLaunchKernelGGL(gemm_skinny_kernel<MT>, grid, dim3(256), 0, s, p);
}

}  // namespace

extern "C" int i2t_gemm_bf16(void* stream, const void* A, int lda, int a_kmajor, const void* B, int ldb,
                             int b_kmajor, void* C, int ldc, int c_is_f32, int M, int N, int K, float alpha,
                             const float* bias, int act, const void* aux_in, int ld_aux_in, void* aux_out,
                             int ld_aux_out, const float* residual, int ldr, int accumulate, int drop_mode,
                             unsigned drop_key, unsigned drop_thr, float drop_scale) {
    return i2t_gemm_bf16_ex(stream, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, c_is_f32, M, N, K, alpha, bias, act, aux_in, ld_aux_in, aux_out,
                            ld_aux_out, residual, ldr, accumulate, drop_mode, drop_key, drop_thr, drop_scale, nullptr);
}

extern "C" int i2t_gemm_bf16_ex(void* stream, const void* A, int lda, int a_kmajor, const void* B, int ldb,
                                int b_kmajor, void* C, int ldc, int c_is_f32, int M, int N, int K, float alpha,
                                const float* bias, int act, const void* aux_in, int ld_aux_in, void* aux_out,
                                int ld_aux_out, const float* residual, int ldr, int accumulate, int drop_mode,
                                unsigned drop_key, unsigned drop_thr, float drop_scale, const float* alpha_sumsq) {
    I2T_REQUIRE(A && B && C, "i2t_gemm_bf16: null operand");
    I2T_REQUIRE(M > 0 && N > 0 && K > 0, "i2t_gemm_bf16: empty problem M=%d N=%d K=%d", M, N, K);
    // K need not be a multiple of 8, but an operand whose reduction index is contiguous is then read up to the
    // next multiple of 8 inside its leading dimension: the caller keeps those pad elements ZERO.
    I2T_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0, "i2t_gemm_bf16: lda=%d ldb=%d must be multiples of 8", lda, ldb);
    I2T_REQUIRE(ALIGNED16(A) && ALIGNED16(B), "i2t_gemm_bf16: A/B must be 16-byte aligned");
    I2T_REQUIRE(lda >= (a_kmajor ? ((M + 7) & ~7) : ((K + 7) & ~7)), "i2t_gemm_bf16: lda=%d too small", lda);
    I2T_REQUIRE(ldb >= (b_kmajor ? ((N + 7) & ~7) : ((K + 7) & ~7)), "i2t_gemm_bf16: ldb=%d too small", ldb);
    I2T_REQUIRE(ldc >= N, "i2t_gemm_bf16: ldc=%d < N=%d", ldc, N);
    I2T_REQUIRE(!accumulate || c_is_f32, "i2t_gemm_bf16: accumulate needs an f32 C");
    I2T_REQUIRE((act != I2T_ACT_DGELU && act != I2T_ACT_DGELU_ERF && act != I2T_ACT_MUL_AUX) || aux_in, "i2t_gemm_bf16: DGELU / MUL_AUX need aux_in");
    I2T_REQUIRE(act >= I2T_ACT_NONE && act <= I2T_ACT_MUL_AUX, "i2t_gemm_bf16: unknown act %d", act);
    I2T_REQUIRE(((uintptr_t)C & (c_is_f32 ? 15 : 7)) == 0, "i2t_gemm_bf16: C misaligned");
    GemmParams p;
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = alpha; p.alpha_sumsq = alpha_sumsq; p.bias = bias; p.act = act;
    p.aux_in = (const bf16_t*)aux_in; p.ld_aux_in = ld_aux_in;
    p.aux_out = (bf16_t*)aux_out; p.ld_aux_out = ld_aux_out;
    p.residual = residual; p.ldr = ldr; p.c_is_f32 = c_is_f32; p.accumulate = accumulate;
    p.drop_mode = drop_mode; p.drop_key = drop_key; p.drop_thr = drop_thr; p.drop_scale = drop_scale;
    p.g2_splits = 1; p.g2_nk = 0; p.ws = nullptr;
    { static const char* e = getenv("I2T_G256_GN"); static const int gn = e ? atoi(e) : 8; p.g2_gn = gn > 0 ? gn : 8; }
    {
        static const char* e = getenv("I2T_G256_STAGGER");                   // "units[,groups]"
        static const int su = e ? atoi(e) : 0;
        static const int sg = (e && strchr(e, ',')) ? atoi(strchr(e, ',') + 1) : 2;
        p.g2_stagger = su; p.g2_stagger_groups = sg > 1 ? sg : 2;
        static const char* d = getenv("I2T_G256_DBG");
        static const int dbg = d ? atoi(d) : 0;
        p.g2_dbg = dbg;
    }
    I2T_REQUIRE(drop_mode == 0 || (drop_mode == 1 && (long)M * N < (1L << 32)) || (drop_mode == 2 && N % 12 == 0),
                "i2t_gemm_bf16: dropout mode %d unsupported for M=%d N=%d", drop_mode, M, N);
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    hipStream_t s = (hipStream_t)stream;
    if (M <= 64 && !a_kmajor && !b_kmajor && !aux_out && act <= I2T_ACT_GELU && !accumulate && !drop_mode) {
        // decode-step shape: weight-streaming kernel.  In-place residual form (C is fp32 and IS the residual) may also
        // split K across workgroups when there are too few column tiles to pull HBM bandwidth from every CU.
        int ksplit = 1;
        const int ntiles = (N + 15) / 16;
        // NOTE: the cross-workgroup split is OFF by default: float atomics make the sum order, hence the last bits of
        // the logits, vary from run to run, and greedy decoding must be token-exact reproducible.  The N = 768
        // projections then run on 48 workgroups; they are launch-latency-sized anyway (1.2 - 4.7 MB of weights).
        if (getenv("I2T_SKINNY_KSPLIT") && c_is_f32 && residual == (const float*)C && ldr == ldc && act == I2T_ACT_NONE && !accumulate)
            while (ntiles * ksplit < 256 && (K / 64) / (ksplit * 2) >= 2 && ksplit < 16) ksplit *= 2;
        const int mt = (M + 15) / 16;
        if (mt == 1) launch_skinny<1>(s, p, ksplit);
        else if (mt == 2) launch_skinny<2>(s, p, ksplit);
        else if (mt == 3) launch_skinny<3>(s, p, ksplit);
        else launch_skinny<4>(s, p, ksplit);
        I2T_CHECK_LAUNCH("i2t_gemm_bf16(skinny)");
        return I2T_OK;
    }
    static const char* sel = getenv("I2T_GEMM");
    static const bool no_g256 = sel && !strcmp(sel, "v1");
    if (!no_g256 && a_kmajor && b_kmajor && accumulate && c_is_f32 && !bias && act == I2T_ACT_NONE && !aux_out && !residual &&
        !drop_mode && M >= 256 && N >= 256 && (ldc & 3) == 0 && (N & 3) == 0) {
        // The k-major panels are addressed through 32-bit buffer offsets: (K + 512) rows x ld x 2 bytes must stay below 4 GiB.
        // A longer reduction (B = 2048: the tied lm_head's dW reads 74 k rows of 50 264 logits, the projector's 401 k rows of
        // 8192) runs as consecutive K chunks that accumulate into the same C -- it used to fall back to the 128^2 kernel
        // (731 / 789 TF instead of ~1.1 / 1.3 PF).
        const size_t ld_max = (size_t)(lda > ldb ? lda : ldb);
        const long k_fit = (long)((1ull << 32) / (2 * ld_max)) - 512;
        if (k_fit >= 1024) {
            const long kc = (K <= k_fit) ? K : (k_fit / 128) * 128;
            bool ok = true;
            for (long k0 = 0; k0 < K && ok; k0 += kc) {
                GemmParams q = p;
                q.A = p.A + (size_t)k0 * lda;
                q.B = p.B + (size_t)k0 * ldb;
                q.K = (int)((K - k0 < kc) ? (K - k0) : kc);
                ok = launch_g256_dw(s, q);
                if (!ok && k0 > 0) { i2t_set_error("i2t_gemm_bf16: K chunk %ld of a chunked dW GEMM has no large-tile form", k0); return I2T_EINVAL; }
            }
            if (ok) {
                I2T_CHECK_LAUNCH("i2t_gemm_bf16(256 dW)");
                return I2T_OK;
            }
        }
    }
    dim3 grid(p.tiles_m * p.tiles_n), block(256);
    // split-K for accumulate-into-fp32 problems whose tile grid cannot fill the 256 CUs (the dW = dY^T.X GEMMs: small
    // M x N, very long K): enough slices to reach ~2 workgroups per CU, each slice at least 4 K-steps long
    int splits = 1;
    const int plain_epilogue = !bias && act == I2T_ACT_NONE && !aux_out && !residual && !drop_mode;
    if (accumulate && c_is_f32 && plain_epilogue) {
        const int tiles = p.tiles_m * p.tiles_n, nk_all = (K + BK - 1) / BK;
        while (tiles * splits < 384 && nk_all / (splits * 2) >= 4 && splits < 64) splits *= 2;
        if (i2t_det()) splits = 1;            // deterministic mode: one K slice per tile, plain read-add-write epilogue
    }
    // large-tile kernel for the non-split problems with enough 256^2 tiles to occupy the chip (I2T_GEMM=v1 keeps the 128^2 one)
    // K % 128 == 0: K-tiles run in pairs and the DMA stream chains output tiles; k-major panels must fit a 32-bit byte offset
    // K % 128 == 0 (K-tiles run in pairs), or any K when B is k-major: the range check then returns B rows >= K as zeros, so
    // whatever finite values a row-major A delivers past K (its zero pads, then the head of the next row) contribute nothing
    const bool g256_ok = (K % 128 == 0 || b_kmajor) && (!a_kmajor || (size_t)(K + 512) * lda * 2 < (1ull << 32)) &&
                         (!b_kmajor || (size_t)(K + 512) * ldb * 2 < (1ull << 32));
    // I2T_G256_MIN_TILES (read per call so that a test can flip it): tile count from which the large-tile kernel takes over
    const char* mt_env = getenv("I2T_G256_MIN_TILES");
    const long min_tiles = mt_env ? atol(mt_env) : 40;
    {   // gemm3 (256 x 128 tiles, the previous tile's epilogue inside the K loop): bit-equal to the 256^2 kernel; measured on MI355X
        // (tools/bench_gemm3.py) +8.5 % at K = 512 (class 1: 865 vs 797 TF), a tie at K = 768 (993 vs 979, 963 vs 953, 1026 vs 1058)
        // and -14 % at K = 2048 (its K loop moves 1.33x the LDS bytes per MFMA and is LDS-bound); the GELU class LOSES (458 vs 704 TF:
        // a 64-value GELU step per wave outlasts the MFMA block it is meant to hide behind).  Default: class 1 with K <= 512 only ...
        // I2T_GEMM3 = 0 never | 1 always, epilogue after the K loop | 2 always, overlapped | unset: the default rule.
        const char* e3 = getenv("I2T_GEMM3");          // (read per call: a test flips it)
        // ... and only without the per-row dropout multipliers and up to ~4e5 rows: at the benchmark's M = 798 720 (B = 3072) the
        // 256^2 kernel is the faster one (1363 vs 1654 us with dropout, 1417 vs 1528 without), and with dropout gemm3 does not win
        // at M = 266 240 either (515 vs 513 us)
        // Round 3 re-measurement (tools/ab_gemm_classes.py, M = 99 840, K = 512, class 1): since the 256^2 kernel's stores became
        // non-temporal it is the faster one here too -- 843 / 850 TF against gemm3's 775 / 798 (N = 2048 / 1536) -- so the default
        // rule is OFF; gemm3 stays reachable through I2T_GEMM3 for the record of the overlap experiment.
        const int g3 = e3 ? atoi(e3) : 0;
        const int cls3 = g256_epilogue_class(p);
        if (g3 && splits == 1 && !a_kmajor && !b_kmajor && K % 64 == 0 && alpha == 1.0f && !alpha_sumsq && (N & 7) == 0 && (ldc & 7) == 0 && ALIGNED16(C) &&
            ((cls3 == 1 && K >= 5 * 64) || (cls3 == 2 && K >= 8 * 64 && (!aux_out || ((ld_aux_out & 7) == 0 && ALIGNED16(aux_out))))) &&
            (long)((M + 255) / 256) * ((N + 127) / 128) >= 2 * min_tiles) {
            launch_g3(s, p, g3 == 2);
            I2T_CHECK_LAUNCH("i2t_gemm_bf16(g3)");
            return I2T_OK;
        }
    }
    // N <= 128 (the LoRA adapters' rank-padded products u = x A^T, du = dY (s B)): a 256-column tile is half empty and M / 256 row tiles
    // leave most CUs idle -- the 128^2 kernel runs twice the workgroups on full tiles (I2T_G256_NARROW=1: the old routing, for A/B)
    static const bool narrow_256 = getenv("I2T_G256_NARROW") && getenv("I2T_G256_NARROW")[0] == '1';
    if (splits == 1 && !no_g256 && g256_ok && !a_kmajor && (N > 128 || narrow_256) && (long)((M + 255) / 256) * ((N + 255) / 256) >= min_tiles) {
        if (b_kmajor) launch_g256<true>(s, p);
        else launch_g256<false>(s, p);
        I2T_CHECK_LAUNCH("i2t_gemm_bf16(256)");
        return I2T_OK;
    }
    if (splits > 1) {
        grid.y = splits;
        if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true>), grid, block, 0, s, p);
        else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, true>), grid, block, 0, s, p);
        else if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, true>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, true>), grid, block, 0, s, p);
    } else {
        if (!a_kmajor && !b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, false, false>), grid, block, 0, s, p);
        else if (!a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<false, true, false>), grid, block, 0, s, p);
        else if (a_kmajor && b_kmajor) hipLaunchKernelGGL((gemm_bf16_kernel<true, true, false>), grid, block, 0, s, p);
        else hipLaunchKernelGGL((gemm_bf16_kernel<true, false, false>), grid, block, 0, s, p);
    }
    I2T_CHECK_LAUNCH("i2t_gemm_bf16");
    return I2T_OK;
}

extern "C" int i2t_gemm_bf16_top2(void* stream, const void* A, int lda, const void* B, int ldb, int M, int N, int K, float* top2, int nseg) {
    I2T_REQUIRE(A && B && top2 && M > 0 && N > 0 && K > 0 && nseg == (N + 63) / 64, "i2t_gemm_bf16_top2: bad args (nseg must be ceil(N / 64))");
    I2T_REQUIRE(K % 128 == 0 && (lda & 7) == 0 && (ldb & 7) == 0 && lda >= K && ldb >= K && ALIGNED16(A) && ALIGNED16(B) && ALIGNED16(top2),
                "i2t_gemm_bf16_top2: K=%d must be a multiple of 128, operands 16-byte aligned with leading dimensions %% 8 == 0", K);
    I2T_REQUIRE((size_t)256 * lda * 2 < (1ull << 32) && (size_t)256 * ldb * 2 < (1ull << 32), "i2t_gemm_bf16_top2: rows too long");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = top2;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = nseg;
    p.alpha = 1.0f; p.c_is_f32 = 1;
    { static const char* e = getenv("I2T_G256_GN"); static const int gn = e ? atoi(e) : 8; p.g2_gn = gn > 0 ? gn : 8; }
    const int n_cu = g256_cus();
    p.tiles_m = (M + 255) / 256; p.tiles_n = (N + 255) / 256;
    p.g2_splits = 1; p.g2_nk = (((K + 63) >> 6) + 1) & ~1;
    const int tiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm256_kernel<false, false, 12>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, (hipStream_t)stream, p);
    I2T_CHECK_LAUNCH("i2t_gemm_bf16_top2");
    return I2T_OK;
}

extern "C" int i2t_gemm_bf16_ws(void* stream, const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_is_f32, int M,
                                int N, int K, const float* bias, int act, const float* residual, int ldr, float* workspace,
                                long ws_floats) {
    // Row-major GEMMs with FEW output tiles and a LONG K (a decode step at a mid-sized caption batch: 64 < M <= ~2048 rows against
    // N = d .. 3 d, K = d .. 4 d) leave most of the 256 CUs idle in a tile-per-workgroup launch.  Here K is cut into slices, every
    // (tile, slice) is a workgroup of the 128^2 kernel writing its raw accumulators to its own plane of `workspace`, and a second
    // launch adds the planes in slice order and applies the fused epilogue: deterministic (greedy decoding stays token-reproducible),
    // no atomics.  Falls through to i2t_gemm_bf16 when the split does not pay or the workspace is too small.
    const int tiles = ((M + BM - 1) / BM) * ((N + BN - 1) / BN), nk_all = (K + BK - 1) / BK;
    int splits = 1;
    while (tiles * splits < 256 && nk_all / (splits * 2) >= 4 && splits < 32) splits *= 2;
    while (splits > 1 && (long)splits * M * N > ws_floats) splits >>= 1;
    // (M > 2048: the planes' write + read -- splits x M x N fp32 each way -- is HBM traffic the decode step cannot spare once several
    // 4096-caption batches run concurrently: the benchmark's decode leg lost 7 % with the split form and keeps the persistent kernel)
    if (M <= 64 || M > 2048 || splits == 1 || !workspace || (act != I2T_ACT_NONE && act != I2T_ACT_GELU))
        return i2t_gemm_bf16(stream, A, lda, 0, B, ldb, 0, C, ldc, c_is_f32, M, N, K, 1.0f, bias, act, nullptr, 0, nullptr, 0, residual,
                             ldr, 0, 0, 0u, 0u, 1.0f);
    I2T_REQUIRE(A && B && C, "i2t_gemm_bf16_ws: null operand");
    I2T_REQUIRE((lda & 7) == 0 && (ldb & 7) == 0 && ALIGNED16(A) && ALIGNED16(B) && lda >= ((K + 7) & ~7) && ldb >= ((K + 7) & ~7) && ldc >= N,
                "i2t_gemm_bf16_ws: operand layout (lda=%d ldb=%d ldc=%d)", lda, ldb, ldc);
    I2T_REQUIRE(((uintptr_t)C & (c_is_f32 ? 15 : 7)) == 0 && ALIGNED16(workspace), "i2t_gemm_bf16_ws: C / workspace misaligned");
    GemmParams p = {};
    p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = C;
    p.M = M; p.N = N; p.K = K; p.lda = lda; p.ldb = ldb; p.ldc = ldc;
    p.alpha = 1.0f; p.bias = bias; p.act = act; p.residual = residual; p.ldr = ldr; p.c_is_f32 = c_is_f32;
    p.tiles_m = (M + BM - 1) / BM; p.tiles_n = (N + BN - 1) / BN;
    p.g2_splits = 1; p.g2_gn = 8; p.ws = workspace;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL((gemm_bf16_kernel<false, false, true>), dim3(tiles, splits), dim3(256), 0, s, p);
    const long quads = (long)M * ((N + 3) / 4);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, p, splits);
    I2T_CHECK_LAUNCH("i2t_gemm_bf16_ws");
    return I2T_OK;
}

extern "C" int i2t_xattn_kv_fused(void* stream, const void* mem, int ld_mem, const void* w_kv, int ld_w, const float* bias_kv,
                                  const void* q, long q_bs, int q_rs, const int* cu_q, int total_q, void* kv, int ld_kv, void* o,
                                  long o_bs, int o_rs, float* lse, int B, int S, int H, int Tq, unsigned drop_key,
                                  unsigned drop_thr, float drop_scale) {
    const int d = 64 * H;
    I2T_REQUIRE(mem && w_kv && bias_kv && q && kv && o && B > 0 && H > 0 && Tq > 0, "i2t_xattn_kv_fused: bad args");
    I2T_REQUIRE(S == 64 && (H & 1) == 0, "i2t_xattn_kv_fused: needs 64 memory tokens per image and an even head count (S=%d H=%d)", S, H);
    I2T_REQUIRE((ld_mem & 7) == 0 && (ld_w & 7) == 0 && ld_mem >= d && ld_w >= d && ALIGNED16(mem) && ALIGNED16(w_kv),
                "i2t_xattn_kv_fused: mem / w_kv must be 16-byte aligned with leading dimensions >= d, %% 8 == 0");
    I2T_REQUIRE((ld_kv & 7) == 0 && ld_kv >= 2 * d && ALIGNED16(kv), "i2t_xattn_kv_fused: kv rows must hold [K | V] (2 d), 16-byte aligned");
    I2T_REQUIRE(((uintptr_t)q & 7) == 0 && ((uintptr_t)o & 7) == 0 && (q_rs & 3) == 0 && (o_rs & 3) == 0 && (q_bs & 3) == 0 && (o_bs & 3) == 0 &&
                    q_rs >= d && o_rs >= d, "i2t_xattn_kv_fused: q / o must be 8-byte aligned with strides %% 4 == 0");
    I2T_REQUIRE(!cu_q || total_q > 0, "i2t_xattn_kv_fused: packed queries need total_q");
    I2T_REQUIRE(drop_thr == 0 || (double)B * H * Tq * 64 < 4294967296.0, "i2t_xattn_kv_fused: dropout index overflows 32 bits");
    I2T_REQUIRE((double)(d + 128) * ld_w * 2 < 4294967296.0 && (double)B * S * ld_mem * 2 < 1.8e19, "i2t_xattn_kv_fused: operand too large");
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = (const bf16_t*)w_kv; p.lda = ld_w;                  // "A" = the weight rows: [K_h | V_h] per wave tile
    p.B = (const bf16_t*)mem; p.ldb = ld_mem;                 // "B" = 256 memory rows = 4 images per workgroup tile
    p.C = kv; p.ldc = ld_kv;
    p.M = 2 * d; p.N = B * S; p.K = d;
    p.alpha = 1.f; p.bias = bias_kv;
    p.drop_key = drop_key; p.drop_thr = drop_thr; p.drop_scale = drop_scale;
    p.xq = (const bf16_t*)q; p.xq_bs = q_bs; p.xq_rs = q_rs; p.xcu = cu_q;
    p.xo = (bf16_t*)o; p.xo_bs = o_bs; p.xo_rs = o_rs; p.xlse = lse;
    p.x_total_q = total_q; p.x_TqMax = Tq; p.x_H = H; p.x_d = d; p.x_B = B;
    p.g2_gn = 8;
    p.tiles_m = H / 2; p.tiles_n = (p.N + 255) / 256;
    p.g2_splits = 1; p.g2_nk = (((p.K + 63) >> 6) + 1) & ~1;
    const int n_cu = g256_cus(), tiles = p.tiles_m * p.tiles_n;
    hipLaunchKernelGGL((gemm256_kernel<false, false, 8>), dim3(tiles < n_cu ? tiles : n_cu), dim3(512), 0, (hipStream_t)stream, p);
    I2T_CHECK_LAUNCH("i2t_xattn_kv_fused");
    return I2T_OK;
}

extern "C" int i2t_gemm_reserve_cus(int n_reserved) {
    I2T_REQUIRE(n_reserved >= 0, "i2t_gemm_reserve_cus: negative count");
    __atomic_store_n(&g_cu_reserve, n_reserved, __ATOMIC_RELAXED);
    return I2T_OK;
}

extern "C" int i2t_gemm_reserved_cus(void) { return __atomic_load_n(&g_cu_reserve, __ATOMIC_RELAXED); }

extern "C" int i2t_colsum_bf16(void* stream, const void* X, int ld, int M, int N, float* out, int accumulate) {
    return i2t_colsum_bf16_ex(stream, X, ld, M, N, out, accumulate, nullptr);
}

extern "C" int i2t_colsum_bf16_ex(void* stream, const void* X, int ld, int M, int N, float* out, int accumulate, const float* alpha_sumsq) {
    I2T_REQUIRE(X && out && M > 0 && N > 0, "i2t_colsum_bf16: bad args");
    I2T_REQUIRE((ld & 7) == 0 && ALIGNED16(X), "i2t_colsum_bf16: X must be 16-byte aligned with ld %% 8 == 0");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(out, 0, sizeof(float) * (size_t)N, s);
        if (e != hipSuccess) { i2t_set_error("i2t_colsum_bf16: memset: %s", hipGetErrorString(e)); return I2T_EHIP; }
    }
    const int col_blocks = (N + 511) / 512;
    int splits = (M + 63) / 64;                       // >= 64 rows per workgroup
    const int want = 512 / col_blocks;                // ~2 workgroups per CU overall
    if (splits > want) splits = want;
    if (splits < 1 || i2t_det()) splits = 1;           // (deterministic mode: one workgroup per column block = one add per column)
    hipLaunchKernelGGL(colsum_kernel, dim3(col_blocks, splits), dim3(256), 0, s, (const bf16_t*)X, ld, M, N, out, alpha_sumsq);
    I2T_CHECK_LAUNCH("i2t_colsum_bf16");
    return I2T_OK;
}
