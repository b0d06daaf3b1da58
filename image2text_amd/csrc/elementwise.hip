// HBM-bound pieces of the path: embeddings, weighted cross-entropy over bf16 logits, the gradient normaliser,
// fused AdamW over a flat arena, casts and small strided copies.  All use 16-byte accesses per lane and
// 64-lane shuffle reductions; cross-workgroup sums use one float atomic per workgroup (cdna guide, Guideline 12).
#include "common.h"

namespace {

// ---------------------------------------------------------------------------------------------- embeddings
__global__ __launch_bounds__(256) void embed_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ wte,
                                                        const float* __restrict__ wpe, float* __restrict__ x, int T,
                                                        int d, int pos_offset, int vocab, int rows,
                                                        const int* __restrict__ pos) {
    const int d4 = d >> 2;
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const int t = pos ? pos[row] : row % T;
        long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const f32x4* e = reinterpret_cast<const f32x4*>(wte + (size_t)id * d);
        f32x4* o = reinterpret_cast<f32x4*>(x + (size_t)row * d);
        if (wpe) {
            const f32x4* p = reinterpret_cast<const f32x4*>(wpe + (size_t)(t + pos_offset) * d);
            for (int c = threadIdx.x; c < d4; c += 256) o[c] = e[c] + p[c];
        } else {                                    // token embedding only (the position term is a per-position MLP, grouped.hip)
            for (int c = threadIdx.x; c < d4; c += 256) o[c] = e[c];
        }
    }
}

__global__ __launch_bounds__(256) void embed_bwd_wte_kernel(const int64_t* __restrict__ ids, const float* __restrict__ dx,
                                                            float* __restrict__ dwte, int d, int vocab, int rows) {
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        long id = ids[row];
        id = id < 0 ? 0 : (id >= vocab ? vocab - 1 : id);
        const float* g = dx + (size_t)row * d;
        float* o = dwte + (size_t)id * d;
        if (gridDim.x == 1) {      // one workgroup walks every row in order (deterministic mode): a column has ONE owner thread -> plain adds
            for (int c = threadIdx.x; c < d; c += 256) o[c] += g[c];
        } else {
            for (int c = threadIdx.x; c < d; c += 256) atomicAdd(o + c, g[c]);   // 256 contiguous bytes per wave-instr
        }
    }
}

// dst[r][:] (+)= sum_b x[b][r][:]   (one thread per float4 column chunk of a row; loops over its slice of the batch --
// blockIdx.y -- with 4 loads in flight; more than one slice: float atomics onto a dst that already holds its base value)
__global__ __launch_bounds__(256) void sum_over_batch_kernel(const float* __restrict__ x, long x_bs,
                                                             float* __restrict__ dst, int B, int rows, int d,
                                                             int accumulate) {
    const int d4 = d >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)rows * d4) return;
    const int r = (int)(idx / d4), c = (int)(idx % d4);
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
    const f32x4* xp = reinterpret_cast<const f32x4*>(x + (size_t)r * d) + c;
    const size_t bs4 = (size_t)x_bs >> 2;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    int b = b0;
    for (; b + 3 < b1; b += 4) {
        const f32x4 v0 = xp[(size_t)b * bs4], v1 = xp[(size_t)(b + 1) * bs4], v2 = xp[(size_t)(b + 2) * bs4], v3 = xp[(size_t)(b + 3) * bs4];
        s += (v0 + v1) + (v2 + v3);
    }
    for (; b < b1; ++b) s += xp[(size_t)b * bs4];
    float* o = dst + (size_t)r * d + 4 * c;
    if (gridDim.y == 1) {
        f32x4* o4 = reinterpret_cast<f32x4*>(o);
        if (accumulate) s += *o4;
        *o4 = s;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) atomicAdd(o + e, s[e]);
    }
}

// launch helper: batch slices when the column grid alone cannot fill the chip (>= 32 batch entries per slice)
static void launch_sum_over_batch(hipStream_t s, const float* x, long x_bs, float* dst, int B, int rows, int d, int accumulate) {
    const long items = (long)rows * (d >> 2);
    const int col_blocks = (int)((items + 255) / 256);
    int slices = 1;
    while (slices < 32 && col_blocks * slices < 1024 && B / (slices * 2) >= 32) slices *= 2;
    if (i2t_det()) slices = 1;                  // one slice: plain adds in batch order
    if (slices > 1 && !accumulate) (void)hipMemsetAsync(dst, 0, (size_t)rows * d * sizeof(float), s);
    hipLaunchKernelGGL(sum_over_batch_kernel, dim3(col_blocks, slices), dim3(256), 0, s, x, x_bs, dst, B, rows, d, accumulate);
}

__global__ __launch_bounds__(256) void bcast_rows_kernel(const float* __restrict__ src, float* __restrict__ y, long y_bs,
                                                         int rows, int d, unsigned drop_key, unsigned drop_thr, float drop_scale) {
    const int d4 = d >> 2;
    const int b = blockIdx.y;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)rows * d4; idx += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(src)[idx];
        if (drop_thr) {      // elementwise dropout of the tensor y heads a slab of: index = element offset in it
            bool keep[4];
            dropout_keep4(drop_key, (unsigned)((size_t)b * y_bs + 4 * (size_t)idx), drop_thr, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * drop_scale : 0.f;
        }
        reinterpret_cast<f32x4*>(y + (size_t)b * y_bs)[idx] = v;
    }
}

template <bool Y_BF16>
__global__ __launch_bounds__(256) void copy_rows_kernel(const float* __restrict__ x, long x_bs, void* __restrict__ y,
                                                        long y_bs, int rows, int d) {
    const int d4 = d >> 2;
    const int b = blockIdx.y;
    for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < (long)rows * d4; idx += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)b * x_bs)[idx];
        if (Y_BF16) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + (size_t)b * y_bs)[idx] = pk;
        } else {
            reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (size_t)b * y_bs)[idx] = v;
        }
    }
}

__global__ __launch_bounds__(256) void add_kernel(float* __restrict__ dst, const float* __restrict__ src, long n4, long n) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256)
        reinterpret_cast<f32x4*>(dst)[i] += reinterpret_cast<const f32x4*>(src)[i];
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[n4 * 4 + threadIdx.x] += src[n4 * 4 + threadIdx.x];
}

__global__ __launch_bounds__(256) void cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n) {
    const long n8 = n >> 3;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        f32x4 a = reinterpret_cast<const f32x4*>(src)[2 * i], b = reinterpret_cast<const f32x4*>(src)[2 * i + 1];
        u32x4 pk = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
        reinterpret_cast<u32x4*>(dst)[i] = pk;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) dst[n8 * 8 + threadIdx.x] = f32_to_bf16(src[n8 * 8 + threadIdx.x]);
}

// I2T_PRECISE=1 (the opt-in inference parity mode, ops.py): an fp32 buffer as the sum of two bf16 terms, hi = bf16(f(x)) and
// lo = bf16(f(x) - hi), f = identity / GELU(tanh) / exact GELU.  A GEMM fed hi and lo of both operands (three products through the
// accumulate class) sees them to 2^-17 instead of 2^-9.
__global__ __launch_bounds__(256) void split_kernel(const float* __restrict__ src, bf16_t* __restrict__ hi, bf16_t* __restrict__ lo, long n,
                                                    int act) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = src[i];
        if (act == I2T_ACT_GELU) v = gelu_tanh(v);
        else if (act == I2T_ACT_GELU_ERF) v = gelu_erf(v);
        const bf16_t h = f32_to_bf16(v);
        hi[i] = h;
        if (lo) lo[i] = f32_to_bf16(v - bf16_to_f32(h));
    }
}

// ---------------------------------------------------------------------------------------------- cross entropy
// one workgroup per row; V bf16 logits are read once with an online (max, sum) per thread, combined across the block
constexpr int CE_THREADS = 512;

__device__ __forceinline__ void online_add(float& m, float& s, float v) {
    if (v > m) {
        s = s * __expf(m - v) + 1.f;
        m = v;
    } else {
        s += __expf(v - m);
    }
}

__device__ __forceinline__ float row_lse(const bf16_t* row, int V, float inv_temp, float* red) {
    float m = -INFINITY, s = 0.f;
    const int v8 = V >> 3;
    for (int c = threadIdx.x; c < v8; c += CE_THREADS) {
        u32x4 pk = reinterpret_cast<const u32x4*>(row)[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            online_add(m, s, bf16lo(pk[e]) * inv_temp);
            online_add(m, s, bf16hi(pk[e]) * inv_temp);
        }
    }
    for (int c = v8 * 8 + threadIdx.x; c < V; c += CE_THREADS) online_add(m, s, bf16_to_f32(row[c]) * inv_temp);
    const float bm = block_max(m, red);
    const float part = (m == -INFINITY) ? 0.f : s * __expf(m - bm);
    const float bs = block_sum(part, red);
    return bm + __logf(bs);
}

__global__ __launch_bounds__(CE_THREADS) void ce_fwd_kernel(const bf16_t* __restrict__ logits, int ld,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ w,
                                                            float inv_temp, int64_t ignore_index, float* __restrict__ lse,
                                                            float* __restrict__ loss, int V) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int64_t lab = labels[row];
    if (lab == ignore_index || lab < 0 || lab >= V) {   // block-uniform
        if (threadIdx.x == 0) lse[row] = 0.f;
        return;
    }
    const bf16_t* r = logits + (size_t)row * ld;
    const float l = row_lse(r, V, inv_temp, red);
    if (threadIdx.x == 0) {
        lse[row] = l;
        atomicAdd(loss, w[row] * (l - bf16_to_f32(r[lab]) * inv_temp));
    }
}

__global__ __launch_bounds__(CE_THREADS) void ce_bwd_kernel(bf16_t* __restrict__ logits, int ld,
                                                            const int64_t* __restrict__ labels, const float* __restrict__ w,
                                                            float inv_temp, int64_t ignore_index, const float* __restrict__ lse,
                                                            const float* __restrict__ gscale_ptr, int V) {
    const int row = blockIdx.x;
    const int64_t lab = labels[row];
    bf16_t* r = logits + (size_t)row * ld;
    const bool live = !(lab == ignore_index || lab < 0 || lab >= V);
    const float coef = live ? (*gscale_ptr) * w[row] * inv_temp : 0.f;
    const float l = live ? lse[row] : 0.f;
    const int v8 = V >> 3;
    for (int c = threadIdx.x; c < v8; c += CE_THREADS) {
        u32x4 pk = reinterpret_cast<u32x4*>(r)[c];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int col = c * 8 + 2 * e;
            float a = coef * (__expf(bf16lo(pk[e]) * inv_temp - l) - (col == lab ? 1.f : 0.f));
            float b = coef * (__expf(bf16hi(pk[e]) * inv_temp - l) - (col + 1 == lab ? 1.f : 0.f));
            o[e] = live ? pack_bf16x2(a, b) : 0u;
        }
        reinterpret_cast<u32x4*>(r)[c] = o;
    }
    for (int c = v8 * 8 + threadIdx.x; c < V; c += CE_THREADS) {
        float a = coef * (__expf(bf16_to_f32(r[c]) * inv_temp - l) - (c == lab ? 1.f : 0.f));
        r[c] = live ? f32_to_bf16(a) : (bf16_t)0;
    }
}

// Single-pass form for training steps: the row (V <= 65536 bf16 = 16 x 16-byte chunks per thread of a 512-thread workgroup) is read
// ONCE into registers, reduced to its logsumexp, and overwritten with the UN-scaled gradient w/T (softmax(z/T) - onehot) -- the
// loss term and lse leave as in ce_fwd.  ce_fwd + ce_bwd read the logits twice and write them once (11 GB each way at the
// benchmark's 110 k x 50 257 logits: 2.1 + 4.3 ms); this reads once and writes once.  The upstream gradient (a device scalar, 1.0
// for a plain loss.backward()) is applied afterwards by scale_bf16_kernel, which returns at once when it is 1.
// NCH = 16-byte chunks per thread (compile-time: the row lives in 4 NCH registers); WPS = waves per SIMD the register budget must
// leave room for -- 3 workgroups per CU at NCH = 13 (the GPT-2 vocabulary), so that one workgroup's load phase runs beside
// another's exp / store phase (two per CU ran 4.2 TB/s against the 5.1 of the streaming ce_bwd)
constexpr int CE_FUSED_CHUNKS = 16;
template <int NCH, int WPS>
__global__ __launch_bounds__(CE_THREADS, WPS) void ce_fused_kernel(bf16_t* __restrict__ logits, int ld, const int64_t* __restrict__ labels,
                                                              const float* __restrict__ w, float inv_temp, int64_t ignore_index,
                                                              float* __restrict__ lse, float* __restrict__ loss, int V) {
    __shared__ float red[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int64_t lab = labels[row];
    bf16_t* r = logits + (size_t)row * ld;
    const int v8 = V >> 3;
    if (lab == ignore_index || lab < 0 || lab >= V) {   // block-uniform: a dead row's gradient is zero
        for (int c = tid; c < v8; c += CE_THREADS) reinterpret_cast<u32x4*>(r)[c] = u32x4{0u, 0u, 0u, 0u};
        for (int c = v8 * 8 + tid; c < V; c += CE_THREADS) r[c] = (bf16_t)0;
        if (tid == 0) lse[row] = 0.f;
        return;
    }
    const float zlab = bf16_to_f32(r[lab]) * inv_temp;                 // (every thread: read before anyone overwrites the row)
    u32x4 pk[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = tid + k * CE_THREADS;
        pk[k] = c < v8 ? __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(r) + c) : u32x4{0u, 0u, 0u, 0u};
    }
    const int ct = v8 * 8 + tid;                                       // the V % 8 tail: one element for threads 0 .. (V & 7) - 1
    const float tail = (tid < (V & 7)) ? bf16_to_f32(r[ct]) * inv_temp : -INFINITY;
    // max first, then ONE exp per element (the online form spends an exp per element per running-max update)
    float m = tail;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (tid + k * CE_THREADS < v8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(bf16lo(pk[k][e]), bf16hi(pk[k][e])) * inv_temp);
        }
    }
    const float bm = block_max(m, red);
    float s = (tid < (V & 7)) ? __expf(tail - bm) : 0.f;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        if (tid + k * CE_THREADS < v8) {
#pragma unroll
            for (int e = 0; e < 4; ++e) s += __expf(bf16lo(pk[k][e]) * inv_temp - bm) + __expf(bf16hi(pk[k][e]) * inv_temp - bm);
        }
    }
    const float bs = block_sum(s, red);
    const float l = bm + __logf(bs);
    const float wr = w[row];
    if (tid == 0) {
        lse[row] = l;
        atomicAdd(loss, wr * (l - zlab));
    }
    const float coef = wr * inv_temp;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
        const int c = tid + k * CE_THREADS;
        if (c < v8) {
            u32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int col = c * 8 + 2 * e;
                const float a = coef * (__expf(bf16lo(pk[k][e]) * inv_temp - l) - (col == lab ? 1.f : 0.f));
                const float b = coef * (__expf(bf16hi(pk[k][e]) * inv_temp - l) - (col + 1 == lab ? 1.f : 0.f));
                o[e] = pack_bf16x2(a, b);
            }
            reinterpret_cast<u32x4*>(r)[c] = o;
        }
    }
    if (tid < (V & 7)) r[ct] = f32_to_bf16(coef * (__expf(tail - l) - (ct == lab ? 1.f : 0.f)));
}

// x[0 .. n) *= *scale (bf16, in place); nothing is touched when the scale is exactly 1 (the common upstream gradient)
__global__ __launch_bounds__(256) void scale_bf16_kernel(bf16_t* __restrict__ x, long n8, long n, const float* __restrict__ scale) {
    const float sc = *scale;
    if (sc == 1.0f) return;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        u32x4 v = reinterpret_cast<u32x4*>(x)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(bf16lo(v[e]) * sc, bf16hi(v[e]) * sc);
        reinterpret_cast<u32x4*>(x)[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) x[n8 * 8 + threadIdx.x] = f32_to_bf16(bf16_to_f32(x[n8 * 8 + threadIdx.x]) * sc);
}

// Momentum-distillation form of the loss (reference training/wrapper.py:134-144): the target of a labelled row is
// alpha * softmax(teacher / T) + (1 - alpha) * onehot(label), so
//   loss_row = lse(z/T) - (1 - alpha) z[label]/T - alpha/T * sum_v softmax(teacher/T)[v] z[v]
//   dz[v]    = g w/T * (softmax(z/T)[v] - (1 - alpha) [v == label] - alpha softmax(teacher/T)[v])
// `teacher` = the momentum model's logits for the same rows (bf16, never differentiated); its row lse is kept for backward.
__device__ __forceinline__ float row_dot_softmax(const bf16_t* z, const bf16_t* t, int V, float inv_temp, float lse_t, float* red) {
    float s = 0.f;
    const int v8 = V >> 3;
    for (int c = threadIdx.x; c < v8; c += CE_THREADS) {
        const u32x4 a = reinterpret_cast<const u32x4*>(z)[c], b = reinterpret_cast<const u32x4*>(t)[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s += __expf(bf16lo(b[e]) * inv_temp - lse_t) * bf16lo(a[e]);
            s += __expf(bf16hi(b[e]) * inv_temp - lse_t) * bf16hi(a[e]);
        }
    }
    for (int c = v8 * 8 + threadIdx.x; c < V; c += CE_THREADS) s += __expf(bf16_to_f32(t[c]) * inv_temp - lse_t) * bf16_to_f32(z[c]);
    return block_sum(s, red);
}

__global__ __launch_bounds__(CE_THREADS) void ce_distill_fwd_kernel(const bf16_t* __restrict__ logits, int ld, const bf16_t* __restrict__ teacher,
                                                                    int ld_t, float alpha, const int64_t* __restrict__ labels,
                                                                    const float* __restrict__ w, float inv_temp, int64_t ignore_index,
                                                                    float* __restrict__ lse, float* __restrict__ lse_t,
                                                                    float* __restrict__ loss, int V) {
    __shared__ float red[16];
    const int row = blockIdx.x;
    const int64_t lab = labels[row];
    if (lab == ignore_index || lab < 0 || lab >= V) {   // block-uniform (an ignored row's weight is zero)
        if (threadIdx.x == 0) lse[row] = lse_t[row] = 0.f;
        return;
    }
    const bf16_t* r = logits + (size_t)row * ld;
    const bf16_t* t = teacher + (size_t)row * ld_t;
    const float l = row_lse(r, V, inv_temp, red);
    const float lt = row_lse(t, V, inv_temp, red);
    const float dot = row_dot_softmax(r, t, V, inv_temp, lt, red);
    if (threadIdx.x == 0) {
        lse[row] = l;
        lse_t[row] = lt;
        atomicAdd(loss, w[row] * (l - (1.f - alpha) * bf16_to_f32(r[lab]) * inv_temp - alpha * inv_temp * dot));
    }
}

__global__ __launch_bounds__(CE_THREADS) void ce_distill_bwd_kernel(bf16_t* __restrict__ logits, int ld, const bf16_t* __restrict__ teacher,
                                                                    int ld_t, float alpha, const int64_t* __restrict__ labels,
                                                                    const float* __restrict__ w, float inv_temp, int64_t ignore_index,
                                                                    const float* __restrict__ lse, const float* __restrict__ lse_t,
                                                                    const float* __restrict__ gscale_ptr, int V) {
    const int row = blockIdx.x;
    const int64_t lab = labels[row];
    bf16_t* r = logits + (size_t)row * ld;
    const bf16_t* t = teacher + (size_t)row * ld_t;
    const bool live = !(lab == ignore_index || lab < 0 || lab >= V);
    const float coef = live ? (*gscale_ptr) * w[row] * inv_temp : 0.f;
    const float l = live ? lse[row] : 0.f, lt = live ? lse_t[row] : 0.f;
    const float hard = 1.f - alpha;
    for (int c = threadIdx.x; c < V; c += CE_THREADS) {
        const float a = coef * (__expf(bf16_to_f32(r[c]) * inv_temp - l) - (c == lab ? hard : 0.f) - alpha * __expf(bf16_to_f32(t[c]) * inv_temp - lt));
        r[c] = live ? f32_to_bf16(a) : (bf16_t)0;
    }
}

// Momentum (EMA) update of the distillation twin over the flat arenas (reference wrapper.py:52-59): pm <- pm m + p (1 - m), plus
// the twin's bf16 shadow in the same pass.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ pm, const float* __restrict__ p, bf16_t* __restrict__ pmb, long n4,
                                                  float momentum) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 a = reinterpret_cast<const f32x4*>(pm)[i], b = reinterpret_cast<const f32x4*>(p)[i];
        const f32x4 v = a * momentum + b * (1.f - momentum);
        reinterpret_cast<f32x4*>(pm)[i] = v;
        if (pmb) {
            const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            reinterpret_cast<u32x2*>(pmb)[i] = pk;
        }
    }
}

// Decoder inputs of a training step from its labels, with the optional MLM corruption (reference wrapper.py:154-196): position 0
// is BOS, position t the (corrupted) label t - 1; an ignored label reads EOS.  Corruption of a labelled token: with probability
// mask_fraction it becomes the MASK id -- or, with probability random_mask_fraction of those, a uniformly random id.  The three
// draws per token come from two counter hashes of (seed, element index): u_mask = top 24 bits of h1, u_rand = top 24 bits of h2,
// id = h2's low bits folded into [0, vocab) by a 64-bit multiply (image2text_amd/rng.py::mlm_draws is the host replica).
__device__ __forceinline__ unsigned mix32_(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__global__ __launch_bounds__(256) void lm_inputs_kernel(const int64_t* __restrict__ labels, int64_t* __restrict__ ids, long n, int L,
                                                        int64_t bos, int64_t eos, int64_t mask_id, int vocab, int64_t ignore_index,
                                                        float mask_fraction, float random_fraction, unsigned seed_lo, unsigned seed_hi) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const int t = (int)(i % L);
        int64_t v = bos;
        if (t > 0) {
            const long src = i - 1;
            const int64_t lab = labels[src];
            v = lab == ignore_index ? eos : lab;
            if (mask_fraction > 0.f && lab != ignore_index) {
                const unsigned h1 = mix32_(mix32_(seed_lo ^ (unsigned)src) + seed_hi + (unsigned)(src >> 32) * 0x9E3779B9u);
                const unsigned h2 = mix32_(h1 ^ 0x85EBCA6Bu);
                const float u_mask = (float)(h1 >> 8) * (1.0f / 16777216.0f), u_rand = (float)(h2 >> 8) * (1.0f / 16777216.0f);
                if (u_mask <= mask_fraction)
                    v = u_rand <= random_fraction ? (int64_t)(((unsigned long long)mix32_(h2 + 0x27D4EB2Fu) * (unsigned long long)vocab) >> 32) : mask_id;
            }
        }
        ids[i] = v;
    }
}

// ---------------------------------------------------------------------------------------------- gradient normaliser
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, long n4, long n, float* __restrict__ ws) {
    __shared__ float red[16];
    float s = 0.f;
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {            // 4 independent 16-byte loads in flight per lane
        f32x4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = reinterpret_cast<const f32x4*>(g)[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) s += v[u][0] * v[u][0] + v[u][1] * v[u][1] + v[u][2] * v[u][2] + v[u][3] * v[u][3];
    }
    for (; i < n4; i += stride) {
        f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        float t = g[n4 * 4 + threadIdx.x];
        s += t * t;
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) atomicAdd(ws, s);
}

__global__ __launch_bounds__(256) void scale_by_norm_kernel(float* __restrict__ g, long n4, long n, const float* __restrict__ ws,
                                                            bf16_t* __restrict__ gb, unsigned drop_key, unsigned drop_thr,
                                                            float drop_scale, int keep_f32) {
    // keep_f32: g itself is left un-normalised (its first consumer applies the factor, i2t_layernorm_bwd dx_pre_sumsq);
    // only the bf16 copy is produced -- 6 bytes per element instead of 10
    const float inv = 1.0f / (sqrtf(*ws) + 1e-6f);
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<f32x4*>(g)[i] * inv;
        if (!keep_f32) reinterpret_cast<f32x4*>(g)[i] = v;
        if (gb) {
            if (drop_thr) {        // the bf16 copy feeds a dropped-out branch: same elementwise mask as its forward
                bool keep[4];
                dropout_keep4(drop_key, (unsigned)(i * 4), drop_thr, keep);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = keep[e] ? v[e] * drop_scale : 0.f;
            }
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            reinterpret_cast<u32x2*>(gb)[i] = pk;
        }
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long e = n4 * 4 + threadIdx.x;
        float v = g[e] * inv;
        if (!keep_f32) g[e] = v;
        if (gb) gb[e] = f32_to_bf16((drop_thr && !dropout_keep(drop_key, (unsigned)e, drop_thr)) ? 0.f : (drop_thr ? v * drop_scale : v));
    }
}

// ---------------------------------------------------------------------------------------------- AdamW
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    bf16_t* __restrict__ pb, long n, const long* __restrict__ seg_end,
                                                    const float* __restrict__ seg_lr, const float* __restrict__ seg_wd,
                                                    int nseg, float beta1, float beta2, float eps, float bc1, float bc2s,
                                                    float grad_scale) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e0 = i * 4;
        int lo = 0, hi = nseg - 1;            // first segment whose end > e0 (segments are 4-element aligned)
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (seg_end[mid] > e0) hi = mid; else lo = mid + 1;
        }
        const float lr = seg_lr[lo], wd = seg_wd[lo];
        if (lr < 0.f) continue;      // a frozen parameter / one outside every group (lr = -1 in the table): no state, no update, no traffic
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gv[e] * grad_scale;
            pv[e] *= (1.f - lr * wd);
            mv[e] = beta1 * mv[e] + (1.f - beta1) * gr;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * gr * gr;
            const float denom = sqrtf(vv[e]) / bc2s + eps;
            pv[e] -= (lr / bc1) * (mv[e] / denom);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (pb) {
            u32x2 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
            reinterpret_cast<u32x2*>(pb)[i] = pk;
        }
    }
}

// SNRAdam (reference models/optimizer.py:56-113): Adam whose second moment tracks the VARIANCE of the gradient around the
// bias-corrected running mean of the previous step instead of its energy: d = g - m_prev / (1 - beta1^(t-1)) (t = 1: d = g),
// v = beta2 v + (1 - beta2) d^2, step = lr * (m / (1 - beta1^t)) / (sqrt(v / (1 - beta2^t)) + eps); decoupled weight decay
// applied first.  Same arena / segment-table / bf16-shadow contract as adamw_kernel.
__global__ __launch_bounds__(256) void snradam_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                      float* __restrict__ m, float* __restrict__ v,
                                                      bf16_t* __restrict__ pb, long n, const long* __restrict__ seg_end,
                                                      const float* __restrict__ seg_lr, const float* __restrict__ seg_wd,
                                                      int nseg, float beta1, float beta2, float eps, float inv_bc1_prev,
                                                      float inv_bc1, float inv_bc2, float grad_scale) {
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e0 = i * 4;
        int lo = 0, hi = nseg - 1;
        while (lo < hi) {
            int mid = (lo + hi) >> 1;
            if (seg_end[mid] > e0) hi = mid; else lo = mid + 1;
        }
        const float lr = seg_lr[lo], wd = seg_wd[lo];
        if (lr < 0.f) continue;      // frozen / outside every group (lr = -1): no state, no update.  lr == 0 (a warm-up from 0) still tracks the
                                     // moments, as the reference's SNRAdam does (models/optimizer.py:98-108: exp_avg / exp_avg_sq / iter_ advance at any lr)
        f32x4 pv = reinterpret_cast<f32x4*>(p)[i], gv = reinterpret_cast<const f32x4*>(g)[i];
        f32x4 mv = reinterpret_cast<f32x4*>(m)[i], vv = reinterpret_cast<f32x4*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gv[e] * grad_scale;
            pv[e] *= (1.f - lr * wd);
            const float dev = gr - mv[e] * inv_bc1_prev;
            mv[e] = beta1 * mv[e] + (1.f - beta1) * gr;
            vv[e] = beta2 * vv[e] + (1.f - beta2) * dev * dev;
            pv[e] -= lr * (mv[e] * inv_bc1) / (sqrtf(vv[e] * inv_bc2) + eps);
        }
        reinterpret_cast<f32x4*>(p)[i] = pv;
        reinterpret_cast<f32x4*>(m)[i] = mv;
        reinterpret_cast<f32x4*>(v)[i] = vv;
        if (pb) {
            u32x2 pk = {pack_bf16x2(pv[0], pv[1]), pack_bf16x2(pv[2], pv[3])};
            reinterpret_cast<u32x2*>(pb)[i] = pk;
        }
    }
}

// in-place dropout of a [rows][cols] tensor: mode 1 elementwise (idx = r*cols + c), mode 2 per (row, third of cols)
template <bool F32>
__global__ __launch_bounds__(256) void dropout_apply_kernel(void* __restrict__ x, long n4, int cols, int mode, unsigned key,
                                                            unsigned thr, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long e0 = i * 4;
        const unsigned r = (unsigned)(e0 / cols), c = (unsigned)(e0 % cols);
        float m[4];
        if (mode == 2) {
            const float f = dropout_keep(key + c / (unsigned)(cols / 3), r, thr) ? scale : 0.f;
            m[0] = m[1] = m[2] = m[3] = f;
        } else {
            bool keep[4];
            dropout_keep4(key, (unsigned)e0, thr, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = keep[e] ? scale : 0.f;
        }
        if (F32) {
            f32x4 v = reinterpret_cast<f32x4*>(x)[i];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] *= m[e];
            reinterpret_cast<f32x4*>(x)[i] = v;
        } else {
            u32x2 v = reinterpret_cast<u32x2*>(x)[i];
            v[0] = pack_bf16x2(bf16lo(v[0]) * m[0], bf16hi(v[0]) * m[1]);
            v[1] = pack_bf16x2(bf16lo(v[1]) * m[2], bf16hi(v[1]) * m[3]);
            reinterpret_cast<u32x2*>(x)[i] = v;
        }
    }
}

int grid_for(long work_items, int cap = 2048) {
    long b = (work_items + 255) / 256;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

}  // namespace

// dwpe[pos[row] + off][:] += dx[row][:] for packed rows (positions repeat across sequences -> atomics)
__global__ __launch_bounds__(256) void embed_bwd_wpe_packed_kernel(const int* __restrict__ pos, const float* __restrict__ dx,
                                                                   float* __restrict__ dwpe, int d, int pos_offset, int rows) {
    for (int row = blockIdx.x; row < rows; row += gridDim.x) {
        const float* g = dx + (size_t)row * d;
        float* o = dwpe + (size_t)(pos[row] + pos_offset) * d;
        if (gridDim.x == 1) {      // (as embed_bwd_wte_kernel)
            for (int c = threadIdx.x; c < d; c += 256) o[c] += g[c];
        } else {
            for (int c = threadIdx.x; c < d; c += 256) atomicAdd(o + c, g[c]);
        }
    }
}

extern "C" int i2t_embed_fwd(void* stream, const int64_t* ids, const float* wte, const float* wpe, float* x, int B, int T,
                             int d, int pos_offset, int vocab, const int* pos) {
    I2T_REQUIRE(ids && wte && x && B > 0 && T > 0 && d % 4 == 0, "i2t_embed_fwd: bad args");
    const int rows = B * T;
    hipLaunchKernelGGL(embed_fwd_kernel, dim3(rows < 4096 ? rows : 4096), dim3(256), 0, (hipStream_t)stream, ids, wte, wpe, x,
                       T, d, pos_offset, vocab, rows, pos);
    I2T_CHECK_LAUNCH("i2t_embed_fwd");
    return I2T_OK;
}

extern "C" int i2t_embed_bwd(void* stream, const int64_t* ids, const float* dx, float* dwte, float* dwpe, int B, int T,
                             int d, int pos_offset, int vocab, const int* pos) {
    I2T_REQUIRE((ids || !dwte) && dx && B > 0 && T > 0 && d % 4 == 0, "i2t_embed_bwd: bad args");
    const int rows = B * T;
    hipStream_t s = (hipStream_t)stream;
    if (dwte)
        hipLaunchKernelGGL(embed_bwd_wte_kernel, dim3(i2t_det() ? 1 : (rows < 4096 ? rows : 4096)), dim3(256), 0, s, ids, dx, dwte, d, vocab, rows);      // (deterministic: ONE workgroup walks the rows in order)
    if (dwpe && pos) {
        hipLaunchKernelGGL(embed_bwd_wpe_packed_kernel, dim3(i2t_det() ? 1 : (rows < 4096 ? rows : 4096)), dim3(256), 0, s, pos, dx, dwpe, d, pos_offset, rows);
    } else if (dwpe) {
        launch_sum_over_batch(s, dx, (long)T * d, dwpe + (size_t)pos_offset * d, B, T, d, 1);
    }
    I2T_CHECK_LAUNCH("i2t_embed_bwd");
    return I2T_OK;
}

extern "C" int i2t_sum_over_batch(void* stream, const float* x, long x_batch_stride, float* dst, int B, int rows, int d,
                                  int accumulate) {
    I2T_REQUIRE(x && dst && B > 0 && rows > 0 && d % 4 == 0 && x_batch_stride % 4 == 0, "i2t_sum_over_batch: bad args");
    launch_sum_over_batch((hipStream_t)stream, x, x_batch_stride, dst, B, rows, d, accumulate);
    I2T_CHECK_LAUNCH("i2t_sum_over_batch");
    return I2T_OK;
}

extern "C" int i2t_bcast_rows(void* stream, const float* src, float* y, long y_batch_stride, int B, int rows, int d) {
    return i2t_bcast_rows_drop(stream, src, y, y_batch_stride, B, rows, d, 0u, 0u, 1.0f);
}

extern "C" int i2t_bcast_rows_drop(void* stream, const float* src, float* y, long y_batch_stride, int B, int rows, int d, unsigned drop_key,
                                   unsigned drop_thr, float drop_scale) {
    I2T_REQUIRE(src && y && B > 0 && rows > 0 && d % 4 == 0 && y_batch_stride % 4 == 0, "i2t_bcast_rows: bad args");
    I2T_REQUIRE(!drop_thr || (double)B * y_batch_stride < 4294967296.0, "i2t_bcast_rows_drop: dropout needs B * y_batch_stride < 2^32");
    hipLaunchKernelGGL(bcast_rows_kernel, dim3(grid_for((long)rows * (d >> 2), 64), B), dim3(256), 0, (hipStream_t)stream, src,
                       y, y_batch_stride, rows, d, drop_key, drop_thr, drop_scale);
    I2T_CHECK_LAUNCH("i2t_bcast_rows");
    return I2T_OK;
}

extern "C" int i2t_copy_rows(void* stream, const float* x, long x_bs, void* y, long y_bs, int y_is_bf16, int B, int rows,
                             int d) {
    I2T_REQUIRE(x && y && B > 0 && rows > 0 && d % 4 == 0 && x_bs % 4 == 0 && y_bs % 4 == 0, "i2t_copy_rows: bad args");
    dim3 grid(grid_for((long)rows * (d >> 2), 64), B);
    if (y_is_bf16) hipLaunchKernelGGL(copy_rows_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, rows, d);
    else hipLaunchKernelGGL(copy_rows_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, x, x_bs, y, y_bs, rows, d);
    I2T_CHECK_LAUNCH("i2t_copy_rows");
    return I2T_OK;
}

extern "C" int i2t_add_f32(void* stream, float* dst, const float* src, long n) {
    I2T_REQUIRE(dst && src && n > 0 && ALIGNED16(dst) && ALIGNED16(src), "i2t_add_f32: bad args");
    hipLaunchKernelGGL(add_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, (hipStream_t)stream, dst, src, n >> 2, n);
    I2T_CHECK_LAUNCH("i2t_add_f32");
    return I2T_OK;
}

extern "C" int i2t_dropout_apply(void* stream, void* x, int is_f32, long rows, int cols, int mode, unsigned key, unsigned thr,
                                 float scale) {
    I2T_REQUIRE(x && rows > 0 && cols > 0 && cols % 4 == 0 && ALIGNED16(x), "i2t_dropout_apply: bad args");
    I2T_REQUIRE((mode == 1 && rows * cols < (1L << 32)) || (mode == 2 && cols % 12 == 0), "i2t_dropout_apply: mode %d unsupported here", mode);
    const long n4 = rows * cols / 4;
    if (is_f32)
        hipLaunchKernelGGL(dropout_apply_kernel<true>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, n4, cols, mode, key, thr, scale);
    else
        hipLaunchKernelGGL(dropout_apply_kernel<false>, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, x, n4, cols, mode, key, thr, scale);
    I2T_CHECK_LAUNCH("i2t_dropout_apply");
    return I2T_OK;
}

extern "C" int i2t_cast_f32_bf16(void* stream, const float* src, void* dst, long n) {
    I2T_REQUIRE(src && dst && n > 0 && ALIGNED16(src) && ALIGNED16(dst), "i2t_cast_f32_bf16: bad args");
    hipLaunchKernelGGL(cast_kernel, dim3(grid_for(n >> 3)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst, n);
    I2T_CHECK_LAUNCH("i2t_cast_f32_bf16");
    return I2T_OK;
}

extern "C" int i2t_split_f32_bf16(void* stream, const float* src, void* hi, void* lo, long n, int act) {
    I2T_REQUIRE(src && hi && n > 0, "i2t_split_f32_bf16: bad args");
    I2T_REQUIRE(act == I2T_ACT_NONE || act == I2T_ACT_GELU || act == I2T_ACT_GELU_ERF, "i2t_split_f32_bf16: act=%d (none, GELU(tanh) or exact GELU)", act);
    hipLaunchKernelGGL(split_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)hi, (bf16_t*)lo, n, act);
    I2T_CHECK_LAUNCH("i2t_split_f32_bf16");
    return I2T_OK;
}

extern "C" int i2t_ce_fwd(void* stream, const void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
                          int64_t ignore_index, float* lse, float* loss, int M, int V) {
    I2T_REQUIRE(logits && labels && w && lse && loss && M > 0 && V > 0, "i2t_ce_fwd: bad args");
    I2T_REQUIRE(ld % 8 == 0 && ld >= V && ALIGNED16(logits), "i2t_ce_fwd: ld=%d must be a multiple of 8 and >= V", ld);
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(M), dim3(CE_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, labels, w,
                       inv_temp, ignore_index, lse, loss, V);
    I2T_CHECK_LAUNCH("i2t_ce_fwd");
    return I2T_OK;
}

extern "C" int i2t_ce_bwd(void* stream, void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
                          int64_t ignore_index, const float* lse, const float* gscale_ptr, int M, int V) {
    I2T_REQUIRE(logits && labels && w && lse && gscale_ptr && M > 0 && V > 0, "i2t_ce_bwd: bad args");
    I2T_REQUIRE(ld % 8 == 0 && ld >= V && ALIGNED16(logits), "i2t_ce_bwd: ld=%d must be a multiple of 8 and >= V", ld);
    hipLaunchKernelGGL(ce_bwd_kernel, dim3(M), dim3(CE_THREADS), 0, (hipStream_t)stream, (bf16_t*)logits, ld, labels, w,
                       inv_temp, ignore_index, lse, gscale_ptr, V);
    I2T_CHECK_LAUNCH("i2t_ce_bwd");
    return I2T_OK;
}

extern "C" int i2t_ce_fwd_bwd(void* stream, void* logits, int ld, const int64_t* labels, const float* w, float inv_temp,
                              int64_t ignore_index, float* lse, float* loss, int M, int V) {
    I2T_REQUIRE(logits && labels && w && lse && loss && M > 0 && V > 0, "i2t_ce_fwd_bwd: bad args");
    I2T_REQUIRE(ld % 8 == 0 && ld >= V && ALIGNED16(logits), "i2t_ce_fwd_bwd: ld=%d must be a multiple of 8 and >= V", ld);
    I2T_REQUIRE((V >> 3) <= CE_FUSED_CHUNKS * CE_THREADS, "i2t_ce_fwd_bwd: V=%d exceeds the one-pass form's %d columns (use i2t_ce_fwd + i2t_ce_bwd)", V,
                8 * CE_FUSED_CHUNKS * CE_THREADS);
    const int nch = ((V >> 3) + CE_THREADS - 1) / CE_THREADS;
#define CE_FUSED_LAUNCH(N_, W_) hipLaunchKernelGGL((ce_fused_kernel<N_, W_>), dim3(M), dim3(CE_THREADS), 0, (hipStream_t)stream, (bf16_t*)logits, ld, \
                                                   labels, w, inv_temp, ignore_index, lse, loss, V)
    if (nch <= 4) CE_FUSED_LAUNCH(4, 8);
    else if (nch <= 8) CE_FUSED_LAUNCH(8, 8);
    else if (nch <= 13) CE_FUSED_LAUNCH(13, 6);
    else CE_FUSED_LAUNCH(16, 4);
#undef CE_FUSED_LAUNCH
    I2T_CHECK_LAUNCH("i2t_ce_fwd_bwd");
    return I2T_OK;
}

extern "C" int i2t_scale_bf16(void* stream, void* x, long n, const float* scale_ptr) {
    I2T_REQUIRE(x && scale_ptr && n > 0 && ALIGNED16(x), "i2t_scale_bf16: bad args");
    hipLaunchKernelGGL(scale_bf16_kernel, dim3(grid_for(n >> 3, 2048)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, n >> 3, n, scale_ptr);
    I2T_CHECK_LAUNCH("i2t_scale_bf16");
    return I2T_OK;
}

extern "C" int i2t_grad_normalize(void* stream, float* g, long n, float* ws, void* g_bf16, unsigned drop_key, unsigned drop_thr,
                                  float drop_scale, int presummed, float* clear_after) {
    I2T_REQUIRE(g && ws && n > 0 && ALIGNED16(g), "i2t_grad_normalize: bad args");
    I2T_REQUIRE(drop_thr == 0 || (g_bf16 && n < (1L << 32)), "i2t_grad_normalize: dropout needs the bf16 copy and n < 2^32");
    hipStream_t s = (hipStream_t)stream;
    I2T_REQUIRE(!(presummed & 2) || g_bf16, "i2t_grad_normalize: flag 2 (fp32 left as is) needs the bf16 copy");
    if (!(presummed & 1)) {    // presummed (bit 0): the producer of g (i2t_layernorm_bwd's sumsq_out) already accumulated sum(g^2) into ws
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float), s);
        if (e != hipSuccess) { i2t_set_error("i2t_grad_normalize: memset: %s", hipGetErrorString(e)); return I2T_EHIP; }
        const int grid = i2t_det() ? 1 : grid_for(n >> 2, 1024);          // (deterministic: one workgroup, fixed-order block sum)
        hipLaunchKernelGGL(sumsq_kernel, dim3(grid), dim3(256), 0, s, g, n >> 2, n, ws);
    }
    hipLaunchKernelGGL(scale_by_norm_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, s, g, n >> 2, n, ws, (bf16_t*)g_bf16, drop_key,
                       drop_thr, drop_scale, (presummed & 2) ? 1 : 0);
    if (clear_after) {         // zero the accumulator the NEXT producer will add into (a different float than ws)
        hipError_t e = hipMemsetAsync(clear_after, 0, sizeof(float), s);
        if (e != hipSuccess) { i2t_set_error("i2t_grad_normalize: memset: %s", hipGetErrorString(e)); return I2T_EHIP; }
    }
    I2T_CHECK_LAUNCH("i2t_grad_normalize");
    return I2T_OK;
}

extern "C" int i2t_sumsq(void* stream, const float* g, long n, float* ws, int accumulate) {
    I2T_REQUIRE(g && ws && n > 0 && ALIGNED16(g), "i2t_sumsq: bad args");
    hipStream_t s = (hipStream_t)stream;
    if (!accumulate) {
        hipError_t e = hipMemsetAsync(ws, 0, sizeof(float), s);
        if (e != hipSuccess) { i2t_set_error("i2t_sumsq: memset: %s", hipGetErrorString(e)); return I2T_EHIP; }
    }
    hipLaunchKernelGGL(sumsq_kernel, dim3(i2t_det() ? 1 : grid_for(n >> 2, 1024)), dim3(256), 0, s, g, n >> 2, n, ws);
    I2T_CHECK_LAUNCH("i2t_sumsq");
    return I2T_OK;
}

extern "C" int i2t_adamw_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, long n,
                              const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg, float beta1,
                              float beta2, float eps, int step, float grad_scale) {
    I2T_REQUIRE(p && g && m && v && seg_end && seg_lr && seg_wd && nseg > 0 && n > 0 && step > 0, "i2t_adamw_step: bad args");
    I2T_REQUIRE(n % 4 == 0 && ALIGNED16(p) && ALIGNED16(g) && ALIGNED16(m) && ALIGNED16(v), "i2t_adamw_step: arena must be 16-byte aligned, n %% 4 == 0");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2s = sqrtf(1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(n >> 2, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                       n, seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, bc1, bc2s, grad_scale);
    I2T_CHECK_LAUNCH("i2t_adamw_step");
    return I2T_OK;
}

extern "C" int i2t_snradam_step(void* stream, float* p, const float* g, float* m, float* v, void* p_bf16, long n,
                                const long* seg_end, const float* seg_lr, const float* seg_wd, int nseg, float beta1,
                                float beta2, float eps, int step, float grad_scale) {
    I2T_REQUIRE(p && g && m && v && seg_end && seg_lr && seg_wd && nseg > 0 && n > 0 && step > 0, "i2t_snradam_step: bad args");
    I2T_REQUIRE(n % 4 == 0 && ALIGNED16(p) && ALIGNED16(g) && ALIGNED16(m) && ALIGNED16(v), "i2t_snradam_step: arena must be 16-byte aligned, n %% 4 == 0");
    const float inv_bc1_prev = step == 1 ? 1.f : 1.f / (1.f - powf(beta1, (float)(step - 1)));
    const float inv_bc1 = 1.f / (1.f - powf(beta1, (float)step));
    const float inv_bc2 = 1.f / (1.f - powf(beta2, (float)step));
    hipLaunchKernelGGL(snradam_kernel, dim3(grid_for(n >> 2, 4096)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (bf16_t*)p_bf16,
                       n, seg_end, seg_lr, seg_wd, nseg, beta1, beta2, eps, inv_bc1_prev, inv_bc1, inv_bc2, grad_scale);
    I2T_CHECK_LAUNCH("i2t_snradam_step");
    return I2T_OK;
}

extern "C" int i2t_ce_distill_fwd(void* stream, const void* logits, int ld, const void* teacher, int ld_t, float alpha, const int64_t* labels,
                                  const float* w, float inv_temp, int64_t ignore_index, float* lse, float* lse_t, float* loss, int M, int V) {
    I2T_REQUIRE(logits && teacher && labels && w && lse && lse_t && loss && M > 0 && V > 0, "i2t_ce_distill_fwd: bad args");
    I2T_REQUIRE(ld % 8 == 0 && ld >= V && ld_t % 8 == 0 && ld_t >= V && ALIGNED16(logits) && ALIGNED16(teacher), "i2t_ce_distill_fwd: leading dimensions must be multiples of 8 and >= V");
    I2T_REQUIRE(alpha >= 0.f && alpha <= 1.f, "i2t_ce_distill_fwd: alpha out of [0, 1]");
    hipLaunchKernelGGL(ce_distill_fwd_kernel, dim3(M), dim3(CE_THREADS), 0, (hipStream_t)stream, (const bf16_t*)logits, ld, (const bf16_t*)teacher,
                       ld_t, alpha, labels, w, inv_temp, ignore_index, lse, lse_t, loss, V);
    I2T_CHECK_LAUNCH("i2t_ce_distill_fwd");
    return I2T_OK;
}

extern "C" int i2t_ce_distill_bwd(void* stream, void* logits, int ld, const void* teacher, int ld_t, float alpha, const int64_t* labels,
                                  const float* w, float inv_temp, int64_t ignore_index, const float* lse, const float* lse_t,
                                  const float* gscale_ptr, int M, int V) {
    I2T_REQUIRE(logits && teacher && labels && w && lse && lse_t && gscale_ptr && M > 0 && V > 0, "i2t_ce_distill_bwd: bad args");
    I2T_REQUIRE(ld >= V && ld_t >= V, "i2t_ce_distill_bwd: leading dimensions must be >= V");
    hipLaunchKernelGGL(ce_distill_bwd_kernel, dim3(M), dim3(CE_THREADS), 0, (hipStream_t)stream, (bf16_t*)logits, ld, (const bf16_t*)teacher,
                       ld_t, alpha, labels, w, inv_temp, ignore_index, lse, lse_t, gscale_ptr, V);
    I2T_CHECK_LAUNCH("i2t_ce_distill_bwd");
    return I2T_OK;
}

extern "C" int i2t_ema_update(void* stream, float* pm, const float* p, void* pm_bf16, long n, float momentum) {
    I2T_REQUIRE(pm && p && n > 0 && n % 4 == 0 && ALIGNED16(pm) && ALIGNED16(p), "i2t_ema_update: arenas must be 16-byte aligned, n %% 4 == 0");
    I2T_REQUIRE(momentum >= 0.f && momentum <= 1.f, "i2t_ema_update: momentum out of [0, 1]");
    hipLaunchKernelGGL(ema_kernel, dim3(grid_for(n >> 2, 4096)), dim3(256), 0, (hipStream_t)stream, pm, p, (bf16_t*)pm_bf16, n >> 2, momentum);
    I2T_CHECK_LAUNCH("i2t_ema_update");
    return I2T_OK;
}

extern "C" int i2t_lm_inputs(void* stream, const int64_t* labels, int64_t* ids, int B, int L, int64_t bos, int64_t eos, int64_t mask_id,
                             int vocab, int64_t ignore_index, float mask_fraction, float random_fraction, unsigned seed_lo,
                             unsigned seed_hi) {
    I2T_REQUIRE(labels && ids && B > 0 && L > 0 && vocab > 0, "i2t_lm_inputs: bad args");
    I2T_REQUIRE(mask_fraction >= 0.f && mask_fraction <= 1.f && random_fraction >= 0.f && random_fraction <= 1.f, "i2t_lm_inputs: fractions out of [0, 1]");
    const long n = (long)B * L;
    hipLaunchKernelGGL(lm_inputs_kernel, dim3(grid_for(n, 1024)), dim3(256), 0, (hipStream_t)stream, labels, ids, n, L, bos, eos, mask_id, vocab,
                       ignore_index, mask_fraction, random_fraction, seed_lo, seed_hi);
    I2T_CHECK_LAUNCH("i2t_lm_inputs");
    return I2T_OK;
}
