// On-device sampling step of generate() (reference models/vision_encoder_decoder.py:150-180): temperature, no-repeat-n-gram
// ban, top-k crop, softmax, nucleus (top-p) cut, renormalisation and the draw -- one workgroup per caption, the caption's
// whole logits row resident in registers (512 threads x EPT values), no sort:
//   * the k-th largest logit and the nucleus boundary are found by BISECTION over the bits of an order-preserving integer
//     image of the floats: 32 (31) block-wide count / mass reductions each, instead of a 50 k-element sort per row;
//   * the draw is an inverse-CDF walk over the kept distribution in VOCABULARY order with one counter-based uniform per
//     (seed, step, row) -- reproducible and independent of batch composition (torch.multinomial's Philox stream, which the
//     reference uses, cannot be matched; the DISTRIBUTION drawn from is the reference's, pinned by tests/golden/tiny_sampling.npz);
//   * every position-dependent input (current length, seed) is read from device memory, so the step is captured once into the
//     decode hipGraph and replayed per token like the greedy step.
// Kept set, exactly as the reference computes it: top-k keeps every logit >= the k-th largest (ties at the threshold stay);
// the nucleus keeps the prefix of the descending-sorted probabilities whose running sum is <= max(nucleus_p, largest
// probability) -- complete groups of equal values, then c members of the boundary group, as many as still fit (torch.sort
// leaves the order inside such a group unspecified; c is 0 unless probabilities tie exactly at the cut).
#include "common.h"

namespace {

// 512 threads: up to 256 registers per lane, so a 50 k-entry row (EPT = 100 values per lane) stays in registers without
// spilling (at 1024 threads the 128-register cap spilled the row to scratch inside the bisection loops)
constexpr int ST = 512, SW = ST / 64;

__device__ __forceinline__ unsigned okey(float f) {            // monotone float -> unsigned (no NaNs on this path)
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ unsigned mix32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ int block_sum_i(int v, int* red) {
    v = wave_sum_i(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    int t = 0;
#pragma unroll
    for (int i = 0; i < SW; ++i) t += red[i];
    return t;
}

template <int EPT>
__global__ __launch_bounds__(ST) void sample_kernel(const float* __restrict__ logits, int ld, int64_t* __restrict__ ids, int ids_ld,
                                                    const int* __restrict__ len_ptr, const int* __restrict__ ngram_sizes,
                                                    int n_sizes, int V, float temperature, int top_k, float nucleus_p,
                                                    const unsigned* __restrict__ seed, float* __restrict__ dist_out, int dist_ld) {
    __shared__ unsigned banbits[EPT * ST / 32];
    __shared__ float redf[SW];
    __shared__ int redi[SW];
    __shared__ float wsum[SW][EPT];
    __shared__ float slot_tot[EPT];
    __shared__ float sh_target;
    __shared__ int sh_slot, sh_tok;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int len = *len_ptr;
    int64_t* row = ids + (size_t)b * ids_ld;

    // ---- no-repeat-n-gram ban (transformers NoRepeatNGramLogitsProcessor): token t is banned when (last n-1 ids) + (t) already occurs
    for (int w = tid; w < EPT * ST / 32; w += ST) banbits[w] = 0u;
    __syncthreads();
    for (int si = 0; si < n_sizes; ++si) {
        const int n = ngram_sizes[si];
        if (n < 1 || len + 1 < n) continue;                  // block-uniform
        for (int i = tid; i <= len - n; i += ST) {
            bool same = true;
            for (int j = 0; j < n - 1; ++j) same = same && (row[i + j] == row[len - n + 1 + j]);
            if (same) {
                const int t = (int)row[i + n - 1];
                if (t >= 0 && t < V) atomicOr(&banbits[t >> 5], 1u << (t & 31));
            }
        }
    }
    __syncthreads();

    // ---- the row: element i = k * ST + tid (coalesced), scaled by 1 / temperature, banned / out-of-range -> -inf
    float x[EPT];
    const float* src = logits + (size_t)b * ld;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = k * ST + tid;
        float v = -INFINITY;
        if (i < V) {
            v = src[i] / temperature;
            if ((banbits[i >> 5] >> (i & 31)) & 1u) v = -INFINITY;
        }
        x[k] = v;
    }

    // ---- top-k: k-th largest value by bisection on the key bits; everything below it is cropped.  The row is turned into its
    // integer keys IN PLACE for the search (a second register array of keys spilled to scratch at EPT = 50)
    if (top_k > 0 && top_k < V) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) x[k] = __uint_as_float(okey(x[k]));
        unsigned kth = 0u;
#pragma unroll 1
        for (int bit = 31; bit >= 0; --bit) {
            const unsigned cand = kth | (1u << bit);
            int c = 0;
#pragma unroll
            for (int k = 0; k < EPT; ++k) c += __float_as_uint(x[k]) >= cand ? 1 : 0;
            if (block_sum_i(c, redi) >= top_k) kth = cand;
        }
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const unsigned key = __float_as_uint(x[k]);
            x[k] = key < kth ? -INFINITY : __uint_as_float((key & 0x80000000u) ? (key ^ 0x80000000u) : ~key);
        }
    }

    // ---- softmax (x becomes the probability)
    float m = -INFINITY;
#pragma unroll
    for (int k = 0; k < EPT; ++k) m = fmaxf(m, x[k]);
    m = block_max(m, redf);
    float z = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        x[k] = x[k] == -INFINITY ? 0.f : __expf(x[k] - m);
        z += x[k];
    }
    z = block_sum(z, redf);
    const float inv = 1.0f / z;
#pragma unroll
    for (int k = 0; k < EPT; ++k) x[k] *= inv;

    // ---- nucleus: boundary value u = the largest probability whose "mass of everything >= it" still exceeds the threshold
    if (nucleus_p >= 0.f) {
        const float thr = fmaxf(nucleus_p, inv);             // inv = the largest probability (exp(0) / z)
        float tot = 0.f;
#pragma unroll
        for (int k = 0; k < EPT; ++k) tot += x[k];
        tot = block_sum(tot, redf);
        if (tot > thr) {
            unsigned u = 0u;
#pragma unroll 1
            for (int bit = 30; bit >= 0; --bit) {            // probabilities are in [0, 1]: their float bits order as integers
                const unsigned cand = u | (1u << bit);
                float ms = 0.f;
#pragma unroll
                for (int k = 0; k < EPT; ++k) ms += __float_as_uint(x[k]) >= cand ? x[k] : 0.f;
                if (block_sum(ms, redf) > thr) u = cand;
            }
            float mgt = 0.f;
            int cu = 0;
#pragma unroll
            for (int k = 0; k < EPT; ++k) {
                const unsigned key = __float_as_uint(x[k]);
                mgt += key > u ? x[k] : 0.f;
                cu += key == u ? 1 : 0;
            }
            mgt = block_sum(mgt, redf);
            cu = block_sum_i(cu, redi);
            const float pu = __uint_as_float(u);
            int c = 0;                                       // members of the boundary group whose running sum still fits
            while (c < cu && mgt + (float)(c + 1) * pu <= thr) ++c;
            if (c == 0) {
#pragma unroll
                for (int k = 0; k < EPT; ++k)
                    if (__float_as_uint(x[k]) <= u) x[k] = 0.f;
            } else {       // rare (equal probabilities at the cut): keep c members of the boundary group -- torch.sort leaves the order
                           // inside such a group unspecified, here it is (thread, slot) order: one block scan of per-thread tie counts
                unsigned long long tm[2] = {0ull, 0ull};      // EPT <= 128
#pragma unroll
                for (int k = 0; k < EPT; ++k) tm[k >> 6] |= (__float_as_uint(x[k]) == u) ? (1ull << (k & 63)) : 0ull;
                const int mine = __popcll(tm[0]) + __popcll(tm[1]);
                int incl_t = mine;
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) {
                    const int t = __shfl_up(incl_t, o, 64);
                    if (lane >= o) incl_t += t;
                }
                __syncthreads();
                if (lane == 63) redi[wave] = incl_t;
                __syncthreads();
                int before = incl_t - mine;
#pragma unroll
                for (int w = 0; w < SW; ++w) before += w < wave ? redi[w] : 0;
#pragma unroll
                for (int k = 0; k < EPT; ++k) {
                    const unsigned key = __float_as_uint(x[k]);
                    const int r = before + (k >= 64 ? __popcll(tm[0]) + __popcll(tm[1] & ((1ull << (k & 63)) - 1ull))
                                                    : __popcll(tm[0] & ((1ull << (k & 63)) - 1ull)));
                    if (key < u || (key == u && r >= c)) x[k] = 0.f;
                }
            }
        }
    }

    // ---- draw: inverse CDF over the kept distribution in vocabulary order (slot-major: index = k * ST + tid)
    float zk = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) zk += x[k];
    zk = block_sum(zk, redf);
    if (dist_out) {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = k * ST + tid;
            if (i < V) dist_out[(size_t)b * dist_ld + i] = x[k] / zk;
        }
    }
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const float s = wave_sum(x[k]);
        if (lane == 0) wsum[wave][k] = s;
    }
    __syncthreads();
    if (tid < EPT) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < SW; ++w) s += wsum[w][tid];
        slot_tot[tid] = s;
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned h = mix32(mix32(seed[0] ^ ((unsigned)b * 0x9E3779B9u)) + seed[1] + (unsigned)len * 0x85EBCA6Bu);
        float target = (float)(h >> 8) * (1.0f / 16777216.0f) * zk;
        int ks = EPT - 1;
        float run = 0.f;
        for (int k = 0; k < EPT; ++k) {
            if (run + slot_tot[k] > target) { ks = k; break; }
            run += slot_tot[k];
        }
        // target beyond the last slot's end (rounding): stay in the last slot that holds any mass
        if (!(run + slot_tot[ks] > target)) {
            ks = 0; run = 0.f;
            float r2 = 0.f;
            for (int k = 0; k < EPT; ++k) { if (slot_tot[k] > 0.f) { ks = k; run = r2; } r2 += slot_tot[k]; }
        }
        sh_slot = ks;
        sh_target = target - run;
        sh_tok = 0x7fffffff;
    }
    __syncthreads();
    const int ks = sh_slot;
    const float target = sh_target;
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) v = k == ks ? x[k] : v;
    float incl = v;                                           // inclusive scan over the slot's ST values: wave scan + wave offsets
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const float t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) redf[wave] = incl;
    __syncthreads();
    float woff = 0.f;
#pragma unroll
    for (int w = 0; w < SW; ++w) woff += w < wave ? redf[w] : 0.f;
    incl += woff;
    const bool hit = v > 0.f && incl > target;
    const unsigned long long bal = __ballot(hit);
    if (bal != 0ull && lane == 0) atomicMin(&sh_tok, ks * ST + wave * 64 + (int)__builtin_ctzll(bal));
    __syncthreads();
    int tok = sh_tok;                                         // the same LDS word in every thread: block-uniform branch below
    if (tok == 0x7fffffff) {                                  // rounding left the target past the slot's last kept element: take that one
        __syncthreads();
        if (tid == 0) sh_tok = -1;
        __syncthreads();
        if (v > 0.f) atomicMax(&sh_tok, ks * ST + tid);
        __syncthreads();
        tok = sh_tok;
    }
    if (tid == 0) row[len] = (int64_t)(tok < 0 ? 0 : tok);
}

}  // namespace

#define LAUNCH_SAMPLE(E)                                                                                                        \
    hipLaunchKernelGGL(sample_kernel<E>, dim3(B), dim3(ST), 0, (hipStream_t)stream, logits, ld, ids, ids_ld, len_ptr, ngram_sizes, \
                       n_sizes, V, temperature, top_k, nucleus_p, seed, dist_out, dist_ld)

extern "C" int i2t_sample_token(void* stream, const float* logits, int ld, int64_t* ids, int ids_ld, const int* len_ptr,
                                const int* ngram_sizes, int n_sizes, int B, int V, float temperature, int top_k, float nucleus_p,
                                const unsigned* seed, float* dist_out, int dist_ld) {
    I2T_REQUIRE(logits && ids && len_ptr && seed && B > 0 && V > 0 && (n_sizes == 0 || ngram_sizes), "i2t_sample_token: bad args");
    I2T_REQUIRE(temperature > 0.f, "i2t_sample_token: temperature must be positive");
    I2T_REQUIRE(V <= 128 * ST, "i2t_sample_token: vocabulary %d exceeds the register-resident row (%d)", V, 128 * ST);
    I2T_REQUIRE(!dist_out || dist_ld >= V, "i2t_sample_token: dist_ld < V");
    const int ept = (V + ST - 1) / ST;
    if (ept <= 2) LAUNCH_SAMPLE(2);
    else if (ept <= 16) LAUNCH_SAMPLE(16);
    else if (ept <= 64) LAUNCH_SAMPLE(64);
    else if (ept <= 100) LAUNCH_SAMPLE(100);
    else LAUNCH_SAMPLE(128);
    I2T_CHECK_LAUNCH("i2t_sample_token");
    return I2T_OK;
}
