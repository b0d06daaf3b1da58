// Greedy-decode step kernels: everything position-dependent reads its position from DEVICE memory, so one captured
// hipGraph replays unchanged for every generated token (static KV cache, no host round trip per token).
#include "common.h"

namespace {

// x[b][:] = wte[ids[b][len-1]] + wpe[len-1+pos_offset]
__global__ __launch_bounds__(256) void embed_step_kernel(const int64_t* __restrict__ ids, int ids_ld,
                                                         const int* __restrict__ len_ptr, const float* __restrict__ wte,
                                                         const float* __restrict__ wpe, float* __restrict__ x, int d,
                                                         int pos_offset, int vocab) {
    const int b = blockIdx.x;
    const int t = *len_ptr - 1;
    long id = ids[(size_t)b * ids_ld + t];
    id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
    const f32x4* e = reinterpret_cast<const f32x4*>(wte + (size_t)id * d);
    f32x4* o = reinterpret_cast<f32x4*>(x + (size_t)b * d);
    if (wpe) {
        const f32x4* p = reinterpret_cast<const f32x4*>(wpe + (size_t)(t + pos_offset) * d);
        for (int c = threadIdx.x; c < (d >> 2); c += 256) o[c] = e[c] + p[c];
    } else {
        for (int c = threadIdx.x; c < (d >> 2); c += 256) o[c] = e[c];
    }
}

// kcache[b][pos][:] = qkv[b][d:2d], vcache[b][pos][:] = qkv[b][2d:3d]
__global__ __launch_bounds__(256) void kv_append_kernel(const bf16_t* __restrict__ qkv, int qkv_rs, bf16_t* __restrict__ kc,
                                                        bf16_t* __restrict__ vc, long cache_bs, int cache_rs,
                                                        const int* __restrict__ pos_ptr, int d) {
    const int b = blockIdx.x, pos = *pos_ptr;
    const u32x4* src = reinterpret_cast<const u32x4*>(qkv + (size_t)b * qkv_rs + d);
    u32x4* kd = reinterpret_cast<u32x4*>(kc + (size_t)b * cache_bs + (size_t)pos * cache_rs);
    u32x4* vd = reinterpret_cast<u32x4*>(vc + (size_t)b * cache_bs + (size_t)pos * cache_rs);
    const int d8 = d >> 3;
    for (int c = threadIdx.x; c < d8; c += 256) {
        kd[c] = src[c];
        vd[c] = src[d8 + c];
    }
}

// One wave per (b, h).  Both passes over the cache use 16-byte loads with 8 lanes per key (lane = 8 * key_in_group + c,
// c = 16-byte chunk of the 64-wide head): one wave-instruction fetches 8 keys x 128 B = 8 full cache lines (the
// lane-per-key form issued 8 loads that each touched 64 lines).  Scores: per-lane partial dot over its 8 dims, 3 xor
// shuffles inside the 8-lane group.  Output: per-lane partial sum over its key stripe for its 8 dims, 3 xor shuffles
// across the 8 stripes.  The step is HBM-bound on the K/V cache (B x H x t x 256 B per call).
constexpr int DEC_MAX_KEYS = 1024;
// append_dm > 0: q points at a packed [q | k | v] row of width 3*append_dm; the new token's k/v (this head's 64
// columns) are written into the cache at *pos by this workgroup and attended to from LDS (fused kv_append).
template <int WPB>         // waves (= heads) per workgroup: 49 152 one-wave workgroups per launch were dispatch-rate bound
__global__ __launch_bounds__(64 * WPB) void decode_attention_kernel(const bf16_t* __restrict__ q, int q_rs,
                                                                    bf16_t* __restrict__ kc, bf16_t* __restrict__ vc,
                                                                    long cache_bs, int cache_rs, long cache_hs, bf16_t* __restrict__ o,
                                                                    int o_rs, const int* __restrict__ pos_ptr, int n_keys_fixed,
                                                                    int append_dm) {
    __shared__ float qs_[WPB][64], kn_[WPB][64], vn_[WPB][64];
    __shared__ float ps_[WPB][DEC_MAX_KEYS];
    const int wv = threadIdx.x >> 6;
    float* qs = qs_[wv];
    float* kn = kn_[wv];
    float* vn = vn_[wv];
    float* ps = ps_[wv];
    const int h = blockIdx.x * WPB + wv, b = blockIdx.y, lane = threadIdx.x & 63;
    const int n = pos_ptr ? (*pos_ptr + 1) : n_keys_fixed;
    const bf16_t* qrow = q + (size_t)b * q_rs + h * 64 + lane;
    qs[lane] = bf16_to_f32(qrow[0]);
    bf16_t* kb = kc + (size_t)b * cache_bs + (size_t)h * cache_hs;      // cache_hs = 64: token-major rows [t][H][64];
    bf16_t* vb = vc + (size_t)b * cache_bs + (size_t)h * cache_hs;      // cache_hs = tmax * 64 (cache_rs = 64): head-major [H][t][64]
    const int n_cached = append_dm > 0 ? n - 1 : n;          // keys read back from the cache
    if (append_dm > 0) {
        const bf16_t kv = qrow[append_dm], vv = qrow[2 * append_dm];
        kn[lane] = bf16_to_f32(kv);
        vn[lane] = bf16_to_f32(vv);
        kb[(size_t)(n - 1) * cache_rs + lane] = kv;
        vb[(size_t)(n - 1) * cache_rs + lane] = vv;
    }
    __syncthreads();
    const int kg = lane >> 3, c = lane & 7;                  // key within a group of 8, 16-byte chunk of the head
    float qv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) qv[e] = qs[c * 8 + e];
    float mx = -INFINITY;
    if (append_dm > 0 && lane == 0) {
        float s = 0.f;
        for (int e = 0; e < 64; ++e) s += kn[e] * qs[e];
        s *= 0.125f;
        ps[n - 1] = s;
        mx = s;
    }
    // 4 groups of 8 keys per trip: the four 16-byte loads are issued back to back (a one-group loop exposed the full HBM latency per
    // 8 keys: a caption's whole cache is 4-8 such groups, so the kernel ran latency-bound at 0.6 of the HBM roof)
    for (int k0 = 0; k0 < n_cached; k0 += 32) {
        u32x4 kk[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + kg;
            kk[u] = u32x4{0u, 0u, 0u, 0u};
            if (key < n_cached) kk[u] = *reinterpret_cast<const u32x4*>(kb + (size_t)key * cache_rs + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + kg;
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) s += bf16lo(kk[u][e]) * qv[2 * e] + bf16hi(kk[u][e]) * qv[2 * e + 1];
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            s += __shfl_xor(s, 4, 64);
            s *= 0.125f;
            if (key < n_cached) {
                if (c == 0) ps[key] = s;
                mx = fmaxf(mx, s);
            }
        }
    }
    mx = wave_max(mx);
    __syncthreads();
    float sum = 0.f;
    for (int key = lane; key < n; key += 64) {
        float p = __expf(ps[key] - mx);
        ps[key] = p;
        sum += p;
    }
    sum = wave_sum(sum);
    __syncthreads();
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < n_cached; k0 += 32) {
        u32x4 vv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + kg;
            vv[u] = u32x4{0u, 0u, 0u, 0u};
            if (key < n_cached) vv[u] = *reinterpret_cast<const u32x4*>(vb + (size_t)key * cache_rs + c * 8);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int key = k0 + 8 * u + kg;
            const float p = key < n_cached ? ps[key] : 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc[2 * e] += p * bf16lo(vv[u][e]);
                acc[2 * e + 1] += p * bf16hi(vv[u][e]);
            }
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {                            // sum the 8 key stripes (lanes with equal c)
        acc[e] += __shfl_xor(acc[e], 8, 64);
        acc[e] += __shfl_xor(acc[e], 16, 64);
        acc[e] += __shfl_xor(acc[e], 32, 64);
    }
    if (kg == 0) {
        const float inv = 1.0f / sum;
        if (append_dm > 0) {
            const float pn = ps[n - 1];
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += pn * vn[c * 8 + e];
        }
        const u32x4 pk = {pack_bf16x2(acc[0] * inv, acc[1] * inv), pack_bf16x2(acc[2] * inv, acc[3] * inv),
                          pack_bf16x2(acc[4] * inv, acc[5] * inv), pack_bf16x2(acc[6] * inv, acc[7] * inv)};
        *reinterpret_cast<u32x4*>(o + (size_t)b * o_rs + h * 64 + c * 8) = pk;
    }
}

// HF NoRepeatNGramLogitsProcessor + argmax for one caption per workgroup
constexpr int BAN_THREADS = 1024;
constexpr int MAX_BANNED = 1024;
template <bool F32>
__global__ __launch_bounds__(BAN_THREADS) void ngram_ban_argmax_kernel(const void* __restrict__ logits, int ld,
                                                                       int64_t* __restrict__ ids, int ids_ld,
                                                                       const int* __restrict__ len_ptr,
                                                                       const int* __restrict__ ngram_sizes, int n_sizes,
                                                                       int V, float* __restrict__ margin_out) {
    __shared__ int banned[MAX_BANNED];
    __shared__ int n_banned;
    __shared__ float rv[BAN_THREADS / 64], rv2[BAN_THREADS / 64];
    __shared__ int ri[BAN_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = *len_ptr;
    int64_t* row = ids + (size_t)b * ids_ld;
    if (tid == 0) n_banned = 0;
    __syncthreads();
    for (int si = 0; si < n_sizes; ++si) {
        const int n = ngram_sizes[si];
        if (n < 1 || len + 1 < n) continue;              // block-uniform
        // candidate n-gram starts i in [0, len - n]; it repeats the current (n-1)-token tail iff ids[i+j] == ids[len-n+1+j]
        for (int i = tid; i <= len - n; i += BAN_THREADS) {
            bool same = true;
            for (int j = 0; j < n - 1; ++j) same = same && (row[i + j] == row[len - n + 1 + j]);
            if (same) {
                int slot = atomicAdd(&n_banned, 1);
                if (slot < MAX_BANNED) banned[slot] = (int)row[i + n - 1];
            }
        }
    }
    __syncthreads();
    const int nb = min(n_banned, MAX_BANNED);
    float best = -INFINITY, second = -INFINITY;
    int bi = 0x7fffffff;
    // fp32 rows with ld % 4 == 0 (the decode path pads to 8): 16-byte loads, 4 consecutive columns per lane per step
    // (the scalar scan ran at ~2.3 TB/s).  Per lane the columns are still visited in ascending order: first index wins ties.
    const bool vec4 = F32 && (ld & 3) == 0 && ((uintptr_t)logits & 15) == 0;
    const int vend = vec4 ? (V & ~3) : 0;
    for (int c4 = tid * 4; c4 < vend; c4 += BAN_THREADS * 4) {
        const f32x4 q = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(logits) + (size_t)b * ld + c4);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = q[e];
            const int c = c4 + e;
            for (int k = 0; k < nb; ++k)
                if (banned[k] == c) v = -INFINITY;
            if (v > best) {
                second = best;
                best = v;
                bi = c;
            } else if (v > second) {
                second = v;
            }
        }
    }
    for (int c = vend + tid; c < V; c += BAN_THREADS) {
        float v = F32 ? reinterpret_cast<const float*>(logits)[(size_t)b * ld + c]
                      : bf16_to_f32(reinterpret_cast<const bf16_t*>(logits)[(size_t)b * ld + c]);
        for (int k = 0; k < nb; ++k)
            if (banned[k] == c) v = -INFINITY;
        if (v > best) {                                   // strided ascending scan: first index wins ties
            second = best;
            best = v;
            bi = c;
        } else if (v > second) {
            second = v;
        }
    }
    // wave reduce (value desc, index asc), tracking the runner-up for the margin
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        float ov = __shfl_xor(best, o, 64), os = __shfl_xor(second, o, 64);
        int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) {
            second = fmaxf(best, os);
            best = ov;
            bi = oi;
        } else {
            second = fmaxf(second, ov);
        }
    }
    const int w = tid >> 6;
    if ((tid & 63) == 0) {
        rv[w] = best;
        rv2[w] = second;
        ri[w] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < BAN_THREADS / 64; ++k) {
            if (rv[k] > best || (rv[k] == best && ri[k] < bi)) {
                second = fmaxf(best, rv2[k]);
                best = rv[k];
                bi = ri[k];
            } else {
                second = fmaxf(second, rv[k]);
            }
        }
        row[len] = bi;
        if (margin_out) margin_out[b] = best - second;
    }
}

// The same token choice from the lm_head's SEGMENT maxima (i2t_gemm_bf16_top2: the two largest logits of every 64-column segment of a
// row, value descending / column ascending) instead of the logits themselves: a segment whose best column is not banned contributes
// it, one whose best is banned contributes its second, and one whose two best are BOTH banned (rare: two continuations of repeated
// n-grams among 64 neighbouring token ids, both ahead of everything else there) is re-evaluated exactly here -- 64 dot products of
// the hidden row with the head's rows, banned columns left out.  One caption per workgroup.
constexpr int T2_THREADS = 256, T2_MAX_REDO = 64;
__global__ __launch_bounds__(T2_THREADS) void top2_ngram_argmax_kernel(const f32x4* __restrict__ top2, int nseg, const bf16_t* __restrict__ hid,
                                                                        int ld_h, const bf16_t* __restrict__ W, int ldw, int d,
                                                                        int64_t* __restrict__ ids, int ids_ld, const int* __restrict__ len_ptr,
                                                                        const int* __restrict__ ngram_sizes, int n_sizes, int V) {
    __shared__ int banned[MAX_BANNED];
    __shared__ int n_banned, n_redo;
    __shared__ int redo[T2_MAX_REDO];
    __shared__ float rv[T2_THREADS / 64];
    __shared__ int ri[T2_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int len = *len_ptr;
    int64_t* row = ids + (size_t)b * ids_ld;
    if (tid == 0) n_banned = n_redo = 0;
    __syncthreads();
    for (int si = 0; si < n_sizes; ++si) {                  // (as ngram_ban_argmax_kernel)
        const int n = ngram_sizes[si];
        if (n < 1 || len + 1 < n) continue;
        for (int i = tid; i <= len - n; i += T2_THREADS) {
            bool same = true;
            for (int j = 0; j < n - 1; ++j) same = same && (row[i + j] == row[len - n + 1 + j]);
            if (same) {
                int slot = atomicAdd(&n_banned, 1);
                if (slot < MAX_BANNED) banned[slot] = (int)row[i + n - 1];
            }
        }
    }
    __syncthreads();
    const int nb = min(n_banned, MAX_BANNED);
    auto is_banned = [&](int c) {
        bool hit = false;
        for (int k = 0; k < nb; ++k) hit = hit || (banned[k] == c);
        return hit;
    };
    float best = -INFINITY;
    int bi = 0x7fffffff;
    auto offer = [&](float v, int c) {
        if (v > best || (v == best && c < bi)) {
            best = v;
            bi = c;
        }
    };
    for (int sgm = tid; sgm < nseg; sgm += T2_THREADS) {
        const f32x4 t = top2[(size_t)b * nseg + sgm];
        const int i1 = __float_as_int(t[1]), i2 = __float_as_int(t[3]);
        if (!(t[0] > -INFINITY)) continue;                  // nothing valid in the segment
        if (!is_banned(i1)) {
            offer(t[0], i1);
        } else if (t[2] > -INFINITY && !is_banned(i2)) {
            offer(t[2], i2);
        } else if (t[2] > -INFINITY) {                      // both leaders banned: what is left of the segment is unknown
            const int slot = atomicAdd(&n_redo, 1);
            if (slot < T2_MAX_REDO) redo[slot] = sgm;
        }
    }
    __syncthreads();
    const int nr = min(n_redo, T2_MAX_REDO);
    for (int r = 0; r < nr; ++r) {                          // exact re-evaluation of a segment, one column per thread of wave 0
        if (tid < 64) {
            const int c = redo[r] * 64 + tid;
            if (c < V && !is_banned(c)) {
                const bf16_t* wr = W + (size_t)c * ldw;
                const bf16_t* hr = hid + (size_t)b * ld_h;
                float acc = 0.f;
                for (int k = 0; k < d; k += 8) {
                    const u32x4 wv = *reinterpret_cast<const u32x4*>(wr + k), hv = *reinterpret_cast<const u32x4*>(hr + k);
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc += bf16lo(wv[e]) * bf16lo(hv[e]) + bf16hi(wv[e]) * bf16hi(hv[e]);
                }
                offer(acc, c);
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        offer(ov, oi);
    }
    if ((tid & 63) == 0) {
        rv[tid >> 6] = best;
        ri[tid >> 6] = bi;
    }
    __syncthreads();
    if (tid == 0) {
        for (int k = 1; k < T2_THREADS / 64; ++k) offer(rv[k], ri[k]);
        row[len] = bi;
    }
}

__global__ void advance_kernel(int* counters, int n, int delta) {
    if ((int)threadIdx.x < n) counters[threadIdx.x] += delta;
}

}  // namespace

extern "C" int i2t_embed_step(void* stream, const int64_t* ids, int ids_ld, const int* len_ptr, const float* wte,
                              const float* wpe, float* x, int B, int d, int pos_offset, int vocab) {
    I2T_REQUIRE(ids && len_ptr && wte && x && B > 0 && d % 4 == 0, "i2t_embed_step: bad args");
    hipLaunchKernelGGL(embed_step_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, ids, ids_ld, len_ptr, wte, wpe, x, d,
                       pos_offset, vocab);
    I2T_CHECK_LAUNCH("i2t_embed_step");
    return I2T_OK;
}

extern "C" int i2t_kv_append(void* stream, const void* qkv, int qkv_rs, void* kcache, void* vcache, long cache_bs,
                             int cache_rs, const int* pos_ptr, int B, int d) {
    I2T_REQUIRE(qkv && kcache && vcache && pos_ptr && B > 0 && d % 8 == 0 && qkv_rs % 8 == 0 && cache_rs % 8 == 0 &&
                    cache_bs % 8 == 0,
                "i2t_kv_append: bad args");
    hipLaunchKernelGGL(kv_append_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, qkv_rs,
                       (bf16_t*)kcache, (bf16_t*)vcache, cache_bs, cache_rs, pos_ptr, d);
    I2T_CHECK_LAUNCH("i2t_kv_append");
    return I2T_OK;
}

extern "C" int i2t_decode_attention(void* stream, const void* q, int q_rs, void* kcache, void* vcache,
                                    long cache_bs, int cache_rs, long cache_hs, void* o, int o_rs, const int* pos_ptr, int n_keys_fixed,
                                    int append_dm, int B, int H) {
    I2T_REQUIRE(cache_hs >= 64 && cache_hs % 8 == 0, "i2t_decode_attention: head stride %ld", cache_hs);
    I2T_REQUIRE(append_dm == 0 || (pos_ptr && append_dm == 64 * H), "i2t_decode_attention: append needs pos_ptr and a packed qkv row");
    I2T_REQUIRE(q && kcache && vcache && o && B > 0 && H > 0, "i2t_decode_attention: bad args");
    I2T_REQUIRE(pos_ptr || (n_keys_fixed > 0 && n_keys_fixed <= DEC_MAX_KEYS), "i2t_decode_attention: key count out of range");
    I2T_REQUIRE(cache_rs % 8 == 0 && cache_bs % 8 == 0 && ALIGNED16(kcache) && ALIGNED16(vcache), "i2t_decode_attention: cache misaligned");
    I2T_REQUIRE(o_rs % 8 == 0 && ALIGNED16(o), "i2t_decode_attention: output rows must be 16-byte aligned");
    if (H % 4 == 0)
        hipLaunchKernelGGL(decode_attention_kernel<4>, dim3(H / 4, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)q, q_rs,
                           (bf16_t*)kcache, (bf16_t*)vcache, cache_bs, cache_rs, cache_hs, (bf16_t*)o, o_rs, pos_ptr, n_keys_fixed, append_dm);
    else
        hipLaunchKernelGGL(decode_attention_kernel<1>, dim3(H, B), dim3(64), 0, (hipStream_t)stream, (const bf16_t*)q, q_rs,
                           (bf16_t*)kcache, (bf16_t*)vcache, cache_bs, cache_rs, cache_hs, (bf16_t*)o, o_rs, pos_ptr, n_keys_fixed, append_dm);
    I2T_CHECK_LAUNCH("i2t_decode_attention");
    return I2T_OK;
}

extern "C" int i2t_ngram_ban_argmax(void* stream, const void* logits, int ld, int logits_is_f32, int64_t* ids, int ids_ld,
                                    int* len_ptr, const int* ngram_sizes, int n_sizes, int B, int V, float* margin_out) {
    I2T_REQUIRE(logits && ids && len_ptr && B > 0 && V > 0 && (n_sizes == 0 || ngram_sizes), "i2t_ngram_ban_argmax: bad args");
    if (logits_is_f32)
        hipLaunchKernelGGL(ngram_ban_argmax_kernel<true>, dim3(B), dim3(BAN_THREADS), 0, (hipStream_t)stream, logits, ld, ids,
                           ids_ld, len_ptr, ngram_sizes, n_sizes, V, margin_out);
    else
        hipLaunchKernelGGL(ngram_ban_argmax_kernel<false>, dim3(B), dim3(BAN_THREADS), 0, (hipStream_t)stream, logits, ld, ids,
                           ids_ld, len_ptr, ngram_sizes, n_sizes, V, margin_out);
    I2T_CHECK_LAUNCH("i2t_ngram_ban_argmax");
    return I2T_OK;
}

extern "C" int i2t_top2_ngram_argmax(void* stream, const float* top2, int nseg, const void* hidden, int ld_hidden, const void* w_head,
                                     int ld_w, int d, int64_t* ids, int ids_ld, int* len_ptr, const int* ngram_sizes, int n_sizes, int B, int V) {
    I2T_REQUIRE(top2 && hidden && w_head && ids && len_ptr && B > 0 && V > 0 && nseg == (V + 63) / 64 && (n_sizes == 0 || ngram_sizes),
                "i2t_top2_ngram_argmax: bad args (nseg must be ceil(V / 64))");
    I2T_REQUIRE(d % 8 == 0 && (ld_hidden & 7) == 0 && (ld_w & 7) == 0 && ALIGNED16(top2) && ALIGNED16(hidden) && ALIGNED16(w_head),
                "i2t_top2_ngram_argmax: hidden / head rows must be 16-byte aligned, d %% 8 == 0");
    hipLaunchKernelGGL(top2_ngram_argmax_kernel, dim3(B), dim3(T2_THREADS), 0, (hipStream_t)stream, (const f32x4*)top2, nseg,
                       (const bf16_t*)hidden, ld_hidden, (const bf16_t*)w_head, ld_w, d, ids, ids_ld, len_ptr, ngram_sizes, n_sizes, V);
    I2T_CHECK_LAUNCH("i2t_top2_ngram_argmax");
    return I2T_OK;
}

extern "C" int i2t_advance(void* stream, int* counters, int n, int delta) {
    I2T_REQUIRE(counters && n > 0 && n <= 64, "i2t_advance: bad args");
    hipLaunchKernelGGL(advance_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, counters, n, delta);
    I2T_CHECK_LAUNCH("i2t_advance");
    return I2T_OK;
}
