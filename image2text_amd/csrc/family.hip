// Kernels of the nano-mini block family that are not GEMMs or attention (reference models/layers.py):
//   * sparse token subsets (TransformerBlock :545-577, :609-614): row gather / scatter by an index list;
//   * MoELinear (:301-346) routing: gate MLP tail + softmax + top-k + expert GELU in one pass over a row, and its backward;
//   * the per-token q/k/v dropout multipliers of MultiQueryAttention (:412-420) on separate q and k|v projections.
// All of it is HBM-bound row work: one 64-lane wave per row, 16-byte accesses where the layout allows.
//
// MoELinear as two GEMMs (engine_family.py):  U = x [W1_0; ..; W1_{E-1}; Wg]^T + b  (fp32, one GEMM for every expert's l1 and
// the gate's first layer),  A = moe_gate(U)  (this file),  y = A W2aug^T  with  A[m] = [w_0 gelu(U_0) | .. | w_{E-1} gelu(U_{E-1}) |
// w_0 .. w_{E-1} | 0] and W2aug = [l2_0.weight | .. | l2_{E-1}.weight | l2_0.bias .. l2_{E-1}.bias | 0]: the unselected experts
// have w_e = 0, so the dense product equals the reference's gather / scatter over the top-k experts, and the expert biases ride in
// the K panel.  Every expert is evaluated for every token -- E P = 64 columns against an output width of 1024..4096 -- which is
// cheaper than routing: the products are GEMM-shaped and nothing is permuted.
#include "common.h"

namespace {

constexpr int MAX_E = 16;       // experts
constexpr int MAX_G = 64;       // gate hidden width

// ---------------------------------------------------------------------------------------------------- row gather / scatter
template <bool F32_OUT, bool BF_OUT>
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                          float* __restrict__ out_f, bf16_t* __restrict__ out_b, long n4, int d4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long r = i / d4;
    const int c = (int)(i - r * d4);
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + ((size_t)idx[r] * d4 + c) * 4);
    if (F32_OUT) *reinterpret_cast<f32x4*>(out_f + i * 4) = v;
    if (BF_OUT) {
        u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
        *reinterpret_cast<u32x2*>(out_b + i * 4) = pk;
    }
}

__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ src, const int* __restrict__ idx,
                                                           float* __restrict__ dst, long n4, int d4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long r = i / d4;
    const int c = (int)(i - r * d4);
    *reinterpret_cast<f32x4*>(dst + ((size_t)idx[r] * d4 + c) * 4) = *reinterpret_cast<const f32x4*>(src + i * 4);
}

// x[m][n] *= keep(key0 + n / sec, m) ? scale : 0   (bf16, in place)
__global__ __launch_bounds__(256) void row_sections_kernel(bf16_t* __restrict__ x, int ld, long n4, int cols4, int sec, unsigned key0,
                                                           unsigned thr, float scale) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const long m = i / cols4;
    const int n = (int)(i - m * cols4) * 4;
    const float f = dropout_keep(key0 + (unsigned)(n / sec), (unsigned)m, thr) ? scale : 0.f;
    u32x2* p = reinterpret_cast<u32x2*>(x + (size_t)m * ld + n);
    const u32x2 v = *p;
    *p = u32x2{pack_bf16x2(bf16lo(v[0]) * f, bf16hi(v[0]) * f), pack_bf16x2(bf16lo(v[1]) * f, bf16hi(v[1]) * f)};
}

// ---------------------------------------------------------------------------------------------------- MoE gate
struct MoeShape {
    int M, E, P, G;           // rows, experts, expert rank, gate hidden width (0: the gate is one Linear, its logits are U's last E columns)
    int top_k, ldu, Kp;       // U row stride (floats); A / dA row stride = padded K of the second GEMM (bf16 elements)
    float inv_sqrt_in;
};

// logits (pre-softmax, already / sqrt(in)) of one row; every lane computes all E of them (E G <= 1024 FMAs)
__device__ __forceinline__ void gate_logits(const float* __restrict__ urow, const MoeShape& s, const float* __restrict__ wg2,
                                            const float* __restrict__ bg2, const float* hid /* LDS, G */, float (&z)[MAX_E]) {
    const int EP = s.E * s.P;
#pragma unroll 1
    for (int e = 0; e < s.E; ++e) {
        float a;
        if (s.G) {
            a = bg2 ? bg2[e] : 0.f;
            for (int k = 0; k < s.G; ++k) a = fmaf(hid[k], wg2[e * s.G + k], a);
        } else {
            a = urow[EP + e];
        }
        z[e] = a * s.inv_sqrt_in;
    }
}

// softmax over E and the top-k choice: w[e] = gate value when expert e is among the k largest, else 0.  Ties go to the lower
// expert index (torch.topk's CPU order); a NaN gate propagates.
__device__ __forceinline__ void gate_route(const MoeShape& s, const float (&z)[MAX_E], float (&g)[MAX_E], float (&w)[MAX_E]) {
    float mx = -INFINITY, sum = 0.f;
    for (int e = 0; e < s.E; ++e) mx = fmaxf(mx, z[e]);
    for (int e = 0; e < s.E; ++e) {
        g[e] = __expf(z[e] - mx);
        sum += g[e];
    }
    const float inv = 1.0f / sum;
    for (int e = 0; e < s.E; ++e) {
        g[e] *= inv;
        w[e] = 0.f;
    }
    unsigned taken = 0;
    for (int k = 0; k < s.top_k; ++k) {
        int best = -1;
        float bv = 0.f;
        for (int e = 0; e < s.E; ++e)
            if (!((taken >> e) & 1u) && (best < 0 || g[e] > bv)) {
                best = e;
                bv = g[e];
            }
        taken |= 1u << best;
        w[best] = g[best];
    }
}

// One wave per row.  U f32 [M][ldu]: columns [0, E P) the experts' l1 pre-activations, then G gate-hidden pre-activations (or E
// logits when G == 0).  Writes A bf16 [M][Kp], the gate values g f32 [M][E] and the routing weights w f32 [M][E] (0 = not chosen).
__global__ __launch_bounds__(256) void moe_gate_fwd_kernel(const float* __restrict__ U, const float* __restrict__ wg2,
                                                           const float* __restrict__ bg2, bf16_t* __restrict__ A, float* __restrict__ gates,
                                                           float* __restrict__ wsel, MoeShape s) {
    __shared__ float hid_s[4][MAX_G];
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wv;
    if (m >= s.M) return;                                   // wave-uniform; no block barrier below
    const float* urow = U + (size_t)m * s.ldu;
    const int EP = s.E * s.P;
    if (s.G) {
        for (int k = lane; k < s.G; k += 64) hid_s[wv][k] = gelu_tanh(urow[EP + k]);
        __builtin_amdgcn_wave_barrier();                    // one wave, in-order LDS pipeline: the reads below see these writes
    }
    float z[MAX_E], g[MAX_E], w[MAX_E];
    gate_logits(urow, s, wg2, bg2, hid_s[wv], z);
    gate_route(s, z, g, w);
    bf16_t* arow = A + (size_t)m * s.Kp;
    for (int c = lane; c < s.Kp; c += 64) {
        float v = 0.f;
        if (c < EP) {
            const int e = c / s.P;
            v = w[e] != 0.f ? w[e] * gelu_tanh(urow[c]) : 0.f;
        } else if (c < EP + s.E) {
            v = w[c - EP];
        }
        arow[c] = f32_to_bf16(v);
    }
    if (lane < s.E) {
        gates[(size_t)m * s.E + lane] = g[lane];
        wsel[(size_t)m * s.E + lane] = w[lane];
    }
}

// Backward of the routing, one wave per row: dA bf16 [M][Kp] (= dy W2aug) -> D1 bf16 [M][ldd] = gradient w.r.t. U (expert
// pre-activations and the gate's first-layer output), and per-workgroup partial sums of the gate's second layer gradient
// (part f32 [gridDim.x][E G + E]; moe_gate_reduce_kernel adds them up in a fixed order).
__global__ __launch_bounds__(256) void moe_gate_bwd_kernel(const bf16_t* __restrict__ dA, const float* __restrict__ U,
                                                           const float* __restrict__ gates, const float* __restrict__ wsel,
                                                           const float* __restrict__ wg2, bf16_t* __restrict__ D1, int ldd,
                                                           float* __restrict__ part, MoeShape s, int rows_per_block) {
    __shared__ float acc_s[4][MAX_E * MAX_G + MAX_E];       // per wave: partial d(wg2), d(bg2) -- a lane owns its entries, no atomics
    const int wv = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int EP = s.E * s.P, NG = s.E * s.G + s.E;
    for (int i = threadIdx.x; i < 4 * (MAX_E * MAX_G + MAX_E); i += 256) (&acc_s[0][0])[i] = 0.f;
    __syncthreads();
    const int m0 = blockIdx.x * rows_per_block;
    for (int r = wv; r < rows_per_block; r += 4) {          // every wave runs the same trip count (barriers inside)
        const int m = m0 + r;
        const bool on = m < s.M;
        const float* urow = U + (size_t)(on ? m : 0) * s.ldu;
        const bf16_t* drow = dA + (size_t)(on ? m : 0) * s.Kp;
        bf16_t* orow = D1 + (size_t)(on ? m : 0) * ldd;
        float g[MAX_E], w[MAX_E], dwe[MAX_E];
        for (int e = 0; e < s.E; ++e) {
            g[e] = gates[(size_t)(on ? m : 0) * s.E + e];
            w[e] = wsel[(size_t)(on ? m : 0) * s.E + e];
            dwe[e] = 0.f;
        }
        // expert part: D1[c] = w_e dA[c] gelu'(U[c]);  d(w_e) = sum_j dA[e P + j] gelu(U[e P + j]) + dA[E P + e]
        for (int c0 = 0; c0 < EP; c0 += 64) {
            const int c = c0 + lane;
            float contrib = 0.f;
            int e = 0;
            if (c < EP) {
                e = c / s.P;
                const float u = urow[c], da = bf16_to_f32(drow[c]);
                if (w[e] != 0.f) {
                    contrib = da * gelu_tanh(u);
                    if (on) orow[c] = f32_to_bf16(w[e] * da * gelu_tanh_grad(u));
                } else if (on) {
                    orow[c] = 0;
                }
            }
            // segmented sum over the P lanes of one expert (P is a multiple of 8; lanes of an expert are contiguous)
            for (int ee = c0 / s.P; ee < s.E && ee * s.P < c0 + 64; ++ee) {
                const float t = wave_sum((c < EP && e == ee) ? contrib : 0.f);
                dwe[ee] += t;
            }
        }
        float dot = 0.f;
        for (int e = 0; e < s.E; ++e) {
            dwe[e] = (w[e] != 0.f) ? dwe[e] + bf16_to_f32(drow[EP + e]) : 0.f;       // only chosen experts' weights reach the output
            dot += g[e] * dwe[e];
        }
        float dlog[MAX_E];
        for (int e = 0; e < s.E; ++e) dlog[e] = on ? g[e] * (dwe[e] - dot) * s.inv_sqrt_in : 0.f;
        if (s.G) {
            for (int k = lane; k < s.G; k += 64) {
                const float u = urow[EP + k];
                const float hk = gelu_tanh(u);
                float dh = 0.f;
                for (int e = 0; e < s.E; ++e) {
                    dh = fmaf(dlog[e], wg2[e * s.G + k], dh);
                    acc_s[wv][e * s.G + k] += dlog[e] * hk;
                }
                if (on) orow[EP + k] = f32_to_bf16(dh * gelu_tanh_grad(u));
            }
            if (lane < s.E) acc_s[wv][s.E * s.G + lane] += dlog[lane];
        } else if (on && lane < s.E) {
            orow[EP + lane] = f32_to_bf16(dlog[lane]);
        }
        // zero the pad columns of D1 (they multiply zero weight rows, but must not be NaN)
        if (on)
            for (int c = EP + (s.G ? s.G : s.E) + lane; c < ldd; c += 64) orow[c] = 0;
    }
    __syncthreads();
    if (s.G)
        for (int i = threadIdx.x; i < NG; i += 256)
            part[(size_t)blockIdx.x * NG + i] = (acc_s[0][i] + acc_s[1][i]) + (acc_s[2][i] + acc_s[3][i]);
}

// dwg2 [E][G] += sum_b part[b][e G + k]; dbg2 [E] += sum_b part[b][E G + e]: one workgroup per output, a fixed summation tree
// (thread t adds rows t, t + 256, ..; then the block reduction) -- run-to-run identical
__global__ __launch_bounds__(256) void moe_gate_reduce_kernel(const float* __restrict__ part, int nb, int NG, int EG, float* __restrict__ dwg2,
                                                              float* __restrict__ dbg2) {
    __shared__ float red[16];
    const int i = blockIdx.x;
    float a = 0.f;
    for (int b = threadIdx.x; b < nb; b += 256) a += part[(size_t)b * NG + i];
    a = block_sum(a, red);
    if (threadIdx.x == 0) {
        if (i < EG) dwg2[i] += a;
        else if (dbg2) dbg2[i - EG] += a;
    }
}

// W2aug bf16 [out][Kp] from the stacked expert parameters: l2w bf16 [E][out][P], l2b f32 [E][out]
__global__ __launch_bounds__(256) void moe_pack_w2_kernel(const bf16_t* __restrict__ l2w, const float* __restrict__ l2b,
                                                          bf16_t* __restrict__ W, int out, int E, int P, int Kp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)out * Kp) return;
    const int o = (int)(i / Kp), c = (int)(i - (long)o * Kp);
    const int EP = E * P;
    bf16_t v = 0;
    if (c < EP) v = l2w[((size_t)(c / P) * out + o) * P + (c % P)];
    else if (c < EP + E) v = f32_to_bf16(l2b[(size_t)(c - EP) * out + o]);
    W[i] = v;
}

// gradients back onto the stacked parameters: gw f32 [E][out][P] += dW[o][e P + j], gb f32 [E][out] += dW[o][E P + e]
__global__ __launch_bounds__(256) void moe_unpack_dw2_kernel(const float* __restrict__ dW, float* __restrict__ gw, float* __restrict__ gb,
                                                             int out, int E, int P, int Kp) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const int EP = E * P;
    if (i >= (long)out * (EP + E)) return;
    const int o = (int)(i / (EP + E)), c = (int)(i - (long)o * (EP + E));
    const float v = dW[(size_t)o * Kp + c];
    if (c < EP) gw[((size_t)(c / P) * out + o) * P + (c % P)] += v;
    else gb[(size_t)(c - EP) * out + o] += v;
}


// ---------------------------------------------------------------------------------------------------- decode step (KV cache)
// One wave per (sequence, query head): the new query against the cached keys 0 .. *pos_ptr of its key/value head, head width HD.
// Same access pattern as decode.hip::decode_attention_kernel (HD / 8 lanes per key, 16-byte loads, xor-shuffle reductions), with
// the query heads of a group reading one shared K/V head (multi-query: the cache is H times smaller than the multi-head one, and
// the step is HBM-bound on exactly that cache).  k_new / v_new: this token's key / value rows (appended to the cache at slot
// *pos_ptr by the group's first head, and attended to from LDS by all of them); null for a fixed memory of n_keys_fixed keys.
constexpr int GDEC_MAX_KEYS = 1024;
template <int HD>
__global__ __launch_bounds__(64) void gq_decode_attention_kernel(const bf16_t* __restrict__ q, int q_rs, const bf16_t* __restrict__ k_new,
                                                                 const bf16_t* __restrict__ v_new, int kv_rs, bf16_t* __restrict__ kc,
                                                                 bf16_t* __restrict__ vc, long cache_bs, int cache_rs,
                                                                 bf16_t* __restrict__ o, int o_rs, const int* __restrict__ pos_ptr,
                                                                 int n_keys_fixed, int G, float scale) {
    constexpr int LPK = HD / 8, KPP = 64 / LPK;             // lanes per key, keys per pass
    __shared__ float qs[HD], kn[HD], vn[HD];
    __shared__ float ps[GDEC_MAX_KEYS];
    const int h = blockIdx.x, b = blockIdx.y, lane = threadIdx.x, hk = h / G;
    const bool append = k_new != nullptr;
    const int n = pos_ptr ? (*pos_ptr + 1) : n_keys_fixed;
    bf16_t* kb = kc + (size_t)b * cache_bs + hk * HD;
    bf16_t* vb = vc + (size_t)b * cache_bs + hk * HD;
    for (int i = lane; i < HD; i += 64) {
        qs[i] = bf16_to_f32(q[(size_t)b * q_rs + h * HD + i]);
        if (append) {
            const bf16_t kv = k_new[(size_t)b * kv_rs + hk * HD + i], vv = v_new[(size_t)b * kv_rs + hk * HD + i];
            kn[i] = bf16_to_f32(kv);
            vn[i] = bf16_to_f32(vv);
            if (h % G == 0) {
                kb[(size_t)(n - 1) * cache_rs + i] = kv;
                vb[(size_t)(n - 1) * cache_rs + i] = vv;
            }
        }
    }
    __syncthreads();
    const int n_cached = append ? n - 1 : n;
    const int kg = lane / LPK, c = lane % LPK;
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = qs[c * 8 + e];
    float mx = -INFINITY;
    if (append && lane == 0) {
        float s = 0.f;
        for (int e = 0; e < HD; ++e) s += kn[e] * qs[e];
        s *= scale;
        ps[n - 1] = s;
        mx = s;
    }
    for (int k0 = 0; k0 < n_cached; k0 += KPP) {
        const int key = k0 + kg;
        float s = 0.f;
        if (key < n_cached) {
            const u32x4 kk = *reinterpret_cast<const u32x4*>(kb + (size_t)key * cache_rs + c * 8);
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf16lo(kk[e]) * qv[2 * e] + bf16hi(kk[e]) * qv[2 * e + 1];
        }
#pragma unroll
        for (int o_ = 1; o_ < LPK; o_ <<= 1) s += __shfl_xor(s, o_, 64);
        s *= scale;
        if (key < n_cached) {
            if (c == 0) ps[key] = s;
            mx = fmaxf(mx, s);
        }
    }
    mx = wave_max(mx);
    __syncthreads();
    float sum = 0.f;
    for (int key = lane; key < n; key += 64) {
        const float p = __expf(ps[key] - mx);
        ps[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < n_cached; k0 += KPP) {
        const int key = k0 + kg;
        if (key < n_cached) {
            const u32x4 vv = *reinterpret_cast<const u32x4*>(vb + (size_t)key * cache_rs + c * 8);
            const float p = ps[key];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += p * bf16lo(vv[e]);
                acc[2 * e + 1] += p * bf16hi(vv[e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int o_ = LPK; o_ < 64; o_ <<= 1) acc[e] += __shfl_xor(acc[e], o_, 64);      // sum the key stripes (lanes with equal c)
    if (kg == 0) {
        const float inv = 1.0f / sum;
        if (append) {
            const float pn = ps[n - 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pn * vn[c * 8 + e];
        }
        const u32x4 pk = {pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv),
                          pack_bf16x2(acc[4] * inv, acc[5] * inv), pack_bf16x2(acc[6] * inv, acc[7] * inv)};
        *reinterpret_cast<u32x4*>(o + (size_t)b * o_rs + h * HD + c * 8) = pk;
    }
}

// per-layer cache slot and membership of the token at *pos_ptr: lpos[l] = rank[l][pos], lmem[l] = member[l][pos]
__global__ void sparse_step_setup_kernel(const int* __restrict__ pos_ptr, const int* __restrict__ rank, const int* __restrict__ member,
                                         int* __restrict__ lpos, int* __restrict__ lmem, int L, int tmax) {
    const int l = blockIdx.x * 64 + threadIdx.x;
    if (l >= L) return;
    const int pos = min(*pos_ptr, tmax - 1);
    lpos[l] = rank[(size_t)l * tmax + pos];
    lmem[l] = member[(size_t)l * tmax + pos];
}

// out = *flag ? a : b   (fp32, n multiple of 4)
__global__ __launch_bounds__(256) void select_rows_kernel(const int* __restrict__ flag, const float* __restrict__ a,
                                                          const float* __restrict__ b, float* __restrict__ out, long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float* src = *flag ? a : b;
    *reinterpret_cast<f32x4*>(out + i * 4) = *reinterpret_cast<const f32x4*>(src + i * 4);
}

long blocks_for(long n) { return (n + 255) / 256; }

int moe_check(const char* who, int M, int E, int P, int G, int top_k, int ldu, int Kp) {
    I2T_REQUIRE(M > 0 && E >= 1 && E <= MAX_E && P >= 8 && P % 8 == 0 && G >= 0 && G <= MAX_G && top_k >= 1 && top_k <= E,
                "%s: unsupported shape M=%d E=%d P=%d gate hidden=%d top_k=%d", who, M, E, P, G, top_k);
    I2T_REQUIRE(ldu >= E * P + (G ? G : E) && Kp >= E * P + E && Kp % 8 == 0, "%s: ldu=%d / Kp=%d too small for E=%d P=%d", who, ldu, Kp, E, P);
    return I2T_OK;
}

}  // namespace

extern "C" int i2t_gather_rows(void* stream, const float* src, const int* idx, float* out_f32, void* out_bf16, long n, int d) {
    I2T_REQUIRE(src && idx && (out_f32 || out_bf16) && n > 0 && d > 0 && d % 4 == 0 && ALIGNED16(src), "i2t_gather_rows: bad args");
    const long n4 = n * (d / 4);
    dim3 grid((unsigned)blocks_for(n4));
    hipStream_t s = (hipStream_t)stream;
    if (out_f32 && out_bf16) hipLaunchKernelGGL((gather_rows_kernel<true, true>), grid, dim3(256), 0, s, src, idx, out_f32, (bf16_t*)out_bf16, n4, d / 4);
    else if (out_f32) hipLaunchKernelGGL((gather_rows_kernel<true, false>), grid, dim3(256), 0, s, src, idx, out_f32, (bf16_t*)nullptr, n4, d / 4);
    else hipLaunchKernelGGL((gather_rows_kernel<false, true>), grid, dim3(256), 0, s, src, idx, (float*)nullptr, (bf16_t*)out_bf16, n4, d / 4);
    I2T_CHECK_LAUNCH("i2t_gather_rows");
    return I2T_OK;
}

extern "C" int i2t_scatter_rows(void* stream, const float* src, const int* idx, float* dst, long n, int d) {
    I2T_REQUIRE(src && idx && dst && n > 0 && d > 0 && d % 4 == 0 && ALIGNED16(src) && ALIGNED16(dst), "i2t_scatter_rows: bad args");
    const long n4 = n * (d / 4);
    hipLaunchKernelGGL(scatter_rows_kernel, dim3((unsigned)blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, src, idx, dst, n4, d / 4);
    I2T_CHECK_LAUNCH("i2t_scatter_rows");
    return I2T_OK;
}

extern "C" int i2t_row_sections_dropout(void* stream, void* x, int ld, long rows, int cols, int section, unsigned key0, unsigned thr,
                                        float scale) {
    I2T_REQUIRE(x && rows > 0 && cols > 0 && cols % 4 == 0 && ld % 4 == 0 && section > 0 && section % 4 == 0 && rows < (1L << 32),
                "i2t_row_sections_dropout: bad args");
    if (!thr) return I2T_OK;
    const long n4 = rows * (cols / 4);
    hipLaunchKernelGGL(row_sections_kernel, dim3((unsigned)blocks_for(n4)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, ld, n4, cols / 4,
                       section, key0, thr, scale);
    I2T_CHECK_LAUNCH("i2t_row_sections_dropout");
    return I2T_OK;
}

extern "C" int i2t_moe_gate_fwd(void* stream, const float* U, int ldu, const float* wg2, const float* bg2, void* A, int Kp, float* gates,
                                float* wsel, int M, int E, int P, int G, int top_k, float inv_sqrt_in) {
    if (int rc = moe_check("i2t_moe_gate_fwd", M, E, P, G, top_k, ldu, Kp)) return rc;
    I2T_REQUIRE(U && A && gates && wsel && (G == 0 || wg2), "i2t_moe_gate_fwd: null operand");
    const MoeShape s{M, E, P, G, top_k, ldu, Kp, inv_sqrt_in};
    hipLaunchKernelGGL(moe_gate_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, U, wg2, bg2, (bf16_t*)A, gates, wsel, s);
    I2T_CHECK_LAUNCH("i2t_moe_gate_fwd");
    return I2T_OK;
}

// enough waves to hide the per-row latency chain (loads -> shuffles): 8 rows per workgroup (2 per wave) up to 4096 workgroups
extern "C" int i2t_moe_gate_bwd_blocks(int M) { return M >= 4096 * 8 ? 4096 : (M + 7) / 8; }

extern "C" int i2t_moe_gate_bwd(void* stream, const void* dA, int Kp, const float* U, int ldu, const float* gates, const float* wsel,
                                const float* wg2, void* D1, int ldd, float* dwg2, float* dbg2, float* part_ws, int M, int E, int P, int G,
                                int top_k, float inv_sqrt_in) {
    if (int rc = moe_check("i2t_moe_gate_bwd", M, E, P, G, top_k, ldu, Kp)) return rc;
    I2T_REQUIRE(dA && U && gates && wsel && D1 && ldd >= E * P + (G ? G : E), "i2t_moe_gate_bwd: bad args");
    I2T_REQUIRE(G == 0 || (wg2 && dwg2 && part_ws), "i2t_moe_gate_bwd: a gate hidden layer needs wg2, dwg2 and the partial-sum workspace");
    const int nb = i2t_moe_gate_bwd_blocks(M);
    const int rpb = (M + nb - 1) / nb;
    const MoeShape s{M, E, P, G, top_k, ldu, Kp, inv_sqrt_in};
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(moe_gate_bwd_kernel, dim3(nb), dim3(256), 0, st, (const bf16_t*)dA, U, gates, wsel, wg2, (bf16_t*)D1, ldd, part_ws, s, rpb);
    if (G) {
        const int NG = E * G + E;
        hipLaunchKernelGGL(moe_gate_reduce_kernel, dim3(NG), dim3(256), 0, st, part_ws, nb, NG, E * G, dwg2, dbg2);
    }
    I2T_CHECK_LAUNCH("i2t_moe_gate_bwd");
    return I2T_OK;
}

extern "C" int i2t_moe_pack_w2(void* stream, const void* l2w, const float* l2b, void* W, int out, int E, int P, int Kp) {
    I2T_REQUIRE(l2w && l2b && W && out > 0 && E >= 1 && E <= MAX_E && P > 0 && Kp >= E * P + E, "i2t_moe_pack_w2: bad args");
    hipLaunchKernelGGL(moe_pack_w2_kernel, dim3((unsigned)blocks_for((long)out * Kp)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)l2w, l2b,
                       (bf16_t*)W, out, E, P, Kp);
    I2T_CHECK_LAUNCH("i2t_moe_pack_w2");
    return I2T_OK;
}

extern "C" int i2t_moe_unpack_dw2(void* stream, const float* dW, float* gw, float* gb, int out, int E, int P, int Kp) {
    I2T_REQUIRE(dW && gw && gb && out > 0 && E >= 1 && E <= MAX_E && P > 0 && Kp >= E * P + E, "i2t_moe_unpack_dw2: bad args");
    hipLaunchKernelGGL(moe_unpack_dw2_kernel, dim3((unsigned)blocks_for((long)out * (E * P + E))), dim3(256), 0, (hipStream_t)stream, dW, gw, gb,
                       out, E, P, Kp);
    I2T_CHECK_LAUNCH("i2t_moe_unpack_dw2");
    return I2T_OK;
}

extern "C" int i2t_gq_decode_attention(void* stream, const void* q, int q_rs, const void* k_new, const void* v_new, int kv_rs, void* kcache,
                                       void* vcache, long cache_bs, int cache_rs, void* out, int out_rs, const int* pos_ptr,
                                       int n_keys_fixed, int max_keys, int B, int H, int Hkv, int hd) {
    I2T_REQUIRE(q && kcache && vcache && out && B > 0 && H > 0 && Hkv > 0 && H % Hkv == 0, "i2t_gq_decode_attention: bad args");
    I2T_REQUIRE(hd == 16 || hd == 32 || hd == 64 || hd == 128, "i2t_gq_decode_attention: head_dim %d (16, 32, 64 or 128)", hd);
    I2T_REQUIRE((k_new != nullptr) == (v_new != nullptr) && (pos_ptr || n_keys_fixed > 0), "i2t_gq_decode_attention: no key count");
    I2T_REQUIRE(max_keys > 0 && max_keys <= GDEC_MAX_KEYS && n_keys_fixed <= max_keys, "i2t_gq_decode_attention: at most %d keys", GDEC_MAX_KEYS);
    I2T_REQUIRE(cache_rs % 8 == 0 && cache_bs % 8 == 0 && out_rs % 8 == 0 && ALIGNED16(kcache) && ALIGNED16(vcache) && ALIGNED16(out),
                "i2t_gq_decode_attention: cache / output rows must be 16-byte aligned");
    const float scale = 1.0f / sqrtf((float)hd);
    dim3 grid(H, B);
    hipStream_t s = (hipStream_t)stream;
#define GDEC_LAUNCH(HD)                                                                                                              \
    hipLaunchKernelGGL(gq_decode_attention_kernel<HD>, grid, dim3(64), 0, s, (const bf16_t*)q, q_rs, (const bf16_t*)k_new,             \
                       (const bf16_t*)v_new, kv_rs, (bf16_t*)kcache, (bf16_t*)vcache, cache_bs, cache_rs, (bf16_t*)out, out_rs, pos_ptr, \
                       n_keys_fixed, H / Hkv, scale)
    if (hd == 16) GDEC_LAUNCH(16);
    else if (hd == 32) GDEC_LAUNCH(32);
    else if (hd == 64) GDEC_LAUNCH(64);
    else GDEC_LAUNCH(128);
#undef GDEC_LAUNCH
    I2T_CHECK_LAUNCH("i2t_gq_decode_attention");
    return I2T_OK;
}

extern "C" int i2t_sparse_step_setup(void* stream, const int* pos_ptr, const int* rank, const int* member, int* lpos, int* lmem, int L,
                                     int tmax) {
    I2T_REQUIRE(pos_ptr && rank && member && lpos && lmem && L > 0 && tmax > 0, "i2t_sparse_step_setup: bad args");
    hipLaunchKernelGGL(sparse_step_setup_kernel, dim3((L + 63) / 64), dim3(64), 0, (hipStream_t)stream, pos_ptr, rank, member, lpos, lmem, L, tmax);
    I2T_CHECK_LAUNCH("i2t_sparse_step_setup");
    return I2T_OK;
}

extern "C" int i2t_select_rows(void* stream, const int* flag, const float* a, const float* b, float* out, long n) {
    I2T_REQUIRE(flag && a && b && out && n > 0 && n % 4 == 0 && ALIGNED16(a) && ALIGNED16(b) && ALIGNED16(out), "i2t_select_rows: bad args");
    hipLaunchKernelGGL(select_rows_kernel, dim3((unsigned)blocks_for(n / 4)), dim3(256), 0, (hipStream_t)stream, flag, a, b, out, n / 4);
    I2T_CHECK_LAUNCH("i2t_select_rows");
    return I2T_OK;
}
