// LayerNorm (row) and LayerNormND (per-image slab) forward/backward for gfx950.  HBM-bound elementwise +
// reduction work: 16-byte loads per lane, wave-shuffle (64-lane) reductions, one wave per row for the row form.
#include "common.h"

namespace {

constexpr float LN_EPS = 1e-5f;
constexpr int MAXC = 4;   // float4 chunks per lane -> d <= 1024

// ------------------------------------------------------------------------------------------------ row LN fwd
template <bool Y_F32>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, void* __restrict__ y,
                                                     float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                     int M, int d, float eps) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nc = d >> 2;
    for (int row = blockIdx.x * 4 + w; row < M; row += gridDim.x * 4) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
        f32x4 v[MAXC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            int c = lane + 64 * i;
            v[i] = (c < nc) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mean = wave_sum(s) / d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            int c = lane + 64 * i;
            if (c < nc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float t = v[i][e] - mean;
                    q += t * t;
                }
            }
        }
        const float rstd = rsqrtf(wave_sum(q) / d + eps);
        if (lane == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            int c = lane + 64 * i;
            if (c < nc) {
                f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
                f32x4 bt = beta ? reinterpret_cast<const f32x4*>(beta)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
                if (Y_F32) {
                    reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (size_t)row * d)[c] = o;
                } else {
                    u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + (size_t)row * d)[c] = pk;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ row LN bwd
// Each block owns ROWS_PER_BLOCK consecutive rows; each wave walks its share, keeping the dgamma/dbeta partial of
// the columns its lanes own in registers; one LDS reduction + one atomic per column per block at the end.
constexpr int LN_BWD_ROWS = 32;

template <bool DY_F32>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, float* __restrict__ dx,
                                                     int dx_accumulate, bf16_t* __restrict__ dx_bf16,
                                                     float* __restrict__ dgamma, float* __restrict__ dbeta, int M,
                                                     int d, unsigned drop_key, unsigned drop_thr, float drop_scale,
                                                     float* __restrict__ sumsq_out, const float* __restrict__ dx_pre_sumsq, int blk0,
                                                     unsigned dxm_key, unsigned dxm_thr, float dxm_scale, int acc_period, int acc_rows) {
    __shared__ float red[2][4][MAXC * 256];   // [gamma|beta][wave][column]  (32 KiB)
    // dx_pre_sumsq: the dx this launch accumulates onto is still UN-normalised; its normaliser 1 / (||dx|| + 1e-6) -- the
    // gradient normaliser of the block boundary above, whose fp32 rescale pass this replaces -- is applied while adding
    const float pre = dx_pre_sumsq ? 1.0f / (sqrtf(*dx_pre_sumsq) + 1e-6f) : 1.0f;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int nc = d >> 2;
    float ssq = 0.f;                          // sum of squares of the f32 dx this block writes (for the gradient normaliser)
    f32x4 pg[MAXC], pb[MAXC];
#pragma unroll
    for (int i = 0; i < MAXC; ++i) pg[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bid = blockIdx.x + blk0;        // (blk0: deterministic mode launches the workgroups one at a time, in order)
    const int row_end = min(M, (bid + 1) * LN_BWD_ROWS);
    for (int row = bid * LN_BWD_ROWS + w; row < row_end; row += 4) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[MAXC], g[MAXC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            int c = lane + 64 * i;
            xh[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nc) {
                f32x4 xv = reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c];
                f32x4 dyv;
                if (DY_F32) {
                    dyv = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + (size_t)row * d)[c];
                } else {
                    u32x2 pk = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + (size_t)row * d)[c];
                    dyv = f32x4{bf16lo(pk[0]), bf16hi(pk[0]), bf16lo(pk[1]), bf16hi(pk[1])};
                }
                f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = dyv[e] * gm[e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    pg[i][e] += dyv[e] * xh[i][e];
                    pb[i][e] += dyv[e];
                }
            }
        }
        const float c1 = wave_sum(s1) / d, c2 = wave_sum(s2) / d;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            int c = lane + 64 * i;
            if (c < nc) {
                f32x4* dxp = reinterpret_cast<f32x4*>(dx + (size_t)row * d) + c;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
                if (dx_accumulate && (acc_period == 0 || row % acc_period < acc_rows)) o += *dxp * pre;      // (acc_period: only the first acc_rows rows of every period hold a value)
                ssq += o[0] * o[0] + o[1] * o[1] + o[2] * o[2] + o[3] * o[3];
                if (dxm_thr) {     // the tower's lowest block: the embedding dropout's mask on the f32 gradient it hands to the embedding
                    bool keepm[4];           // backward (idx = row * d + column, as i2t_dropout_apply mode 1); sum of squares and bf16 copy: unmasked
                    dropout_keep4(dxm_key, (unsigned)row * (unsigned)d + 4u * (unsigned)c, dxm_thr, keepm);
                    f32x4 om;
#pragma unroll
                    for (int e = 0; e < 4; ++e) om[e] = keepm[e] ? o[e] * dxm_scale : 0.f;
                    *dxp = om;
                } else {
                    *dxp = o;
                }
                if (dx_bf16) {
                    if (drop_thr) {    // the bf16 copy feeds a dropped-out branch: its forward mask, idx = row * d + column
                        bool keep[4];
                        dropout_keep4_even(drop_key, (unsigned)row * (unsigned)d + 4u * (unsigned)c, drop_thr, keep);      // d % 4 == 0: even index
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = keep[e] ? o[e] * drop_scale : 0.f;
                    }
                    u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    reinterpret_cast<u32x2*>(dx_bf16 + (size_t)row * d)[c] = pk;
                }
            }
        }
    }
    if (sumsq_out) {                          // one atomic per workgroup (block-uniform branch)
        ssq = wave_sum(ssq);
        if (lane == 0) red[0][0][w] = ssq;
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(sumsq_out, red[0][0][0] + red[0][0][1] + red[0][0][2] + red[0][0][3]);
        __syncthreads();
    }
    if (!dgamma && !dbeta) return;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        int c = lane + 64 * i;
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                red[0][w][c * 4 + e] = pg[i][e];
                red[1][w][c * 4 + e] = pb[i][e];
            }
        }
    }
    __syncthreads();
    for (int col = threadIdx.x; col < d; col += 256) {
        float sg = red[0][0][col] + red[0][1][col] + red[0][2][col] + red[0][3][col];
        float sb = red[1][0][col] + red[1][1][col] + red[1][2][col] + red[1][3][col];
        if (dgamma) atomicAdd(dgamma + col, sg);
        if (dbeta) atomicAdd(dbeta + col, sb);
    }
}

// ------------------------------------------------------------------------------------------------ wide rows (1024 < d <= 8192)
// One WORKGROUP per row (a thread owns chunks tid + 256 i): the rows of the big Hugging Face decoders (Falcon-7B: d = 4544).  Same
// arithmetic as the one-wave-per-row kernels above, the reductions go through LDS (block_sum).
constexpr int WIDEC = 8;
constexpr int LN_BWD_WIDE_ROWS = 8;

template <bool Y_F32>
__global__ __launch_bounds__(256) void ln_fwd_wide_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, void* __restrict__ y,
                                                          float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int d, float eps) {
    __shared__ float red[16];
    const int nc = d >> 2;
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
        const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
        f32x4 v[WIDEC];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < WIDEC; ++i) {
            const int c = threadIdx.x + 256 * i;
            v[i] = (c < nc) ? xr[c] : f32x4{0.f, 0.f, 0.f, 0.f};
            s += v[i][0] + v[i][1] + v[i][2] + v[i][3];
        }
        const float mean = block_sum(s, red) / d;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < WIDEC; ++i)
            if (threadIdx.x + 256 * i < nc) {
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = v[i][e] - mean;
                    q += t * t;
                }
            }
        const float rstd = rsqrtf(block_sum(q, red) / d + eps);
        if (threadIdx.x == 0) {
            if (mean_out) mean_out[row] = mean;
            if (rstd_out) rstd_out[row] = rstd;
        }
#pragma unroll
        for (int i = 0; i < WIDEC; ++i) {
            const int c = threadIdx.x + 256 * i;
            if (c < nc) {
                const f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
                const f32x4 bt = beta ? reinterpret_cast<const f32x4*>(beta)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = (v[i][e] - mean) * rstd * gm[e] + bt[e];
                if (Y_F32) {
                    reinterpret_cast<f32x4*>(reinterpret_cast<float*>(y) + (size_t)row * d)[c] = o;
                } else {
                    const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y) + (size_t)row * d)[c] = pk;
                }
            }
        }
    }
}

// a workgroup walks LN_BWD_WIDE_ROWS consecutive rows, its threads keep the dgamma / dbeta partials of their own columns in
// registers: one atomic per column and workgroup at the end (no dropout / gradient-normaliser extras on this form)
template <bool DY_F32>
__global__ __launch_bounds__(256) void ln_bwd_wide_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, float* __restrict__ dx, int dx_accumulate,
                                                          bf16_t* __restrict__ dx_bf16, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, int M, int d, int blk0) {
    __shared__ float red[16];
    const int nc = d >> 2;
    f32x4 pg[WIDEC], pb[WIDEC];
#pragma unroll
    for (int i = 0; i < WIDEC; ++i) pg[i] = pb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bid = blockIdx.x + blk0;
    const int row_end = min(M, (bid + 1) * LN_BWD_WIDE_ROWS);
    for (int row = bid * LN_BWD_WIDE_ROWS; row < row_end; ++row) {
        const float mu = mean[row], rs = rstd[row];
        f32x4 xh[WIDEC], g[WIDEC];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int i = 0; i < WIDEC; ++i) {
            const int c = threadIdx.x + 256 * i;
            xh[i] = g[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            if (c < nc) {
                const f32x4 xv = reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c];
                f32x4 dyv;
                if (DY_F32) {
                    dyv = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + (size_t)row * d)[c];
                } else {
                    const u32x2 pk = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + (size_t)row * d)[c];
                    dyv = f32x4{bf16lo(pk[0]), bf16hi(pk[0]), bf16lo(pk[1]), bf16hi(pk[1])};
                }
                const f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    xh[i][e] = (xv[e] - mu) * rs;
                    g[i][e] = dyv[e] * gm[e];
                    s1 += g[i][e];
                    s2 += g[i][e] * xh[i][e];
                    pg[i][e] += dyv[e] * xh[i][e];
                    pb[i][e] += dyv[e];
                }
            }
        }
        const float c1 = block_sum(s1, red) / d, c2 = block_sum(s2, red) / d;
#pragma unroll
        for (int i = 0; i < WIDEC; ++i) {
            const int c = threadIdx.x + 256 * i;
            if (c < nc) {
                f32x4* dxp = reinterpret_cast<f32x4*>(dx + (size_t)row * d) + c;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) o[e] = rs * (g[i][e] - c1 - xh[i][e] * c2);
                if (dx_accumulate) o += *dxp;
                *dxp = o;
                if (dx_bf16) {
                    const u32x2 pk = {pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
                    reinterpret_cast<u32x2*>(dx_bf16 + (size_t)row * d)[c] = pk;
                }
            }
        }
    }
    if (!dgamma && !dbeta) return;
#pragma unroll
    for (int i = 0; i < WIDEC; ++i) {
        const int c = threadIdx.x + 256 * i;
        if (c < nc) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (dgamma) atomicAdd(dgamma + c * 4 + e, pg[i][e]);
                if (dbeta) atomicAdd(dbeta + c * 4 + e, pb[i][e]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ LayerNormND
// stats layout per image: [0]=mean [1]=rstd [2 + 2s], [3 + 2s] = partial (mean_s, M2_s) of split s (fwd)
//                         or partial (sum g, sum g*xhat) (bwd);  NSPLIT splits per image.
constexpr int NSPLIT = 16;
constexpr int STATS_STRIDE = 2 + 2 * NSPLIT;

__global__ __launch_bounds__(256) void lnnd_partial_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                           float* __restrict__ stats, int n) {
    __shared__ float red[16];
    const int b = blockIdx.x, s = blockIdx.y;
    const int n4 = n >> 2;
    const int per = (n4 + NSPLIT - 1) / NSPLIT;
    const int c0 = s * per, c1 = min(n4, c0 + per);
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (size_t)b * n);
    const f32x4* ab = reinterpret_cast<const f32x4*>(add);
    float sum = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) {
        f32x4 v = xb[c];
        if (add) v += ab[c];
        sum += v[0] + v[1] + v[2] + v[3];
    }
    const int cnt = max(0, c1 - c0) * 4;
    const float mean = cnt > 0 ? block_sum(sum, red) / cnt : 0.f;
    float m2 = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) {
        f32x4 v = xb[c];
        if (add) v += ab[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float t = v[e] - mean;
            m2 += t * t;
        }
    }
    m2 = block_sum(m2, red);
    if (threadIdx.x == 0) {
        stats[(size_t)b * STATS_STRIDE + 2 + 2 * s] = mean;
        stats[(size_t)b * STATS_STRIDE + 3 + 2 * s] = m2;
    }
}

// combine NSPLIT (mean, M2) partials of equal-ish counts (Chan et al.)
__device__ __forceinline__ void lnnd_combine(const float* st, int n, float& mean, float& rstd) {
    const int n4 = n >> 2;
    const int per = (n4 + NSPLIT - 1) / NSPLIT;
    float cnt = 0.f, mu = 0.f, m2 = 0.f;
    for (int s = 0; s < NSPLIT; ++s) {
        int c0 = s * per, c1 = min(n4, c0 + per);
        float cs = (float)(max(0, c1 - c0) * 4);
        if (cs == 0.f) continue;
        float ms = st[2 + 2 * s], qs = st[3 + 2 * s];
        float tot = cnt + cs;
        float dlt = ms - mu;
        mu += dlt * (cs / tot);
        m2 += qs + dlt * dlt * (cnt * cs / tot);
        cnt = tot;
    }
    mean = mu;
    rstd = rsqrtf(m2 / cnt + LN_EPS);
}

__global__ __launch_bounds__(256) void lnnd_apply_kernel(const float* __restrict__ x, const float* __restrict__ add,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         float* __restrict__ y, long y_bs, float* __restrict__ stats,
                                                         int n, unsigned drop_key, unsigned drop_thr, float drop_scale, long drop_base) {
    const int b = blockIdx.x;
    float mean, rstd;
    lnnd_combine(stats + (size_t)b * STATS_STRIDE, n, mean, rstd);
    const int n4 = n >> 2;
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (size_t)b * n);
    f32x4* yb = reinterpret_cast<f32x4*>(y + (size_t)b * y_bs);
    for (int c = blockIdx.y * 256 + threadIdx.x; c < n4; c += gridDim.y * 256) {
        f32x4 v = xb[c];
        if (add) v += reinterpret_cast<const f32x4*>(add)[c];
        f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
        f32x4 bt = beta ? reinterpret_cast<const f32x4*>(beta)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = (v[e] - mean) * rstd * gm[e] + bt[e];
        if (drop_thr) {      // elementwise dropout of the tensor y is a slab of (the embedding dropout, encoder.py:170): index = element offset in it
            bool keep[4];
            dropout_keep4(drop_key, (unsigned)((size_t)b * y_bs + drop_base + 4 * (size_t)c), drop_thr, keep);
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = keep[e] ? o[e] * drop_scale : 0.f;
        }
        yb[c] = o;
    }
    __syncthreads();   // every thread of block (b,0) has read the partials before they are overwritten
    if (blockIdx.y == 0 && threadIdx.x == 0) {
        stats[(size_t)b * STATS_STRIDE + 0] = mean;
        stats[(size_t)b * STATS_STRIDE + 1] = rstd;
    }
}

// bwd pass 1: per (image, split) partial sums of g = dy*gamma and g*xhat
__global__ __launch_bounds__(256) void lnnd_bwd_partial_kernel(const float* __restrict__ dy, long dy_bs,
                                                               const float* __restrict__ x,
                                                               const float* __restrict__ add,
                                                               const float* __restrict__ gamma,
                                                               float* __restrict__ stats, int n) {
    __shared__ float red[16];
    const int b = blockIdx.x, s = blockIdx.y;
    const float mean = stats[(size_t)b * STATS_STRIDE], rstd = stats[(size_t)b * STATS_STRIDE + 1];
    const int n4 = n >> 2;
    const int per = (n4 + NSPLIT - 1) / NSPLIT;
    const int c0 = s * per, c1 = min(n4, c0 + per);
    const f32x4* xb = reinterpret_cast<const f32x4*>(x + (size_t)b * n);
    const f32x4* db = reinterpret_cast<const f32x4*>(dy + (size_t)b * dy_bs);
    float s1 = 0.f, s2 = 0.f;
    for (int c = c0 + threadIdx.x; c < c1; c += 256) {
        f32x4 v = xb[c];
        if (add) v += reinterpret_cast<const f32x4*>(add)[c];
        f32x4 g = db[c] * reinterpret_cast<const f32x4*>(gamma)[c];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            s1 += g[e];
            s2 += g[e] * (v[e] - mean) * rstd;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        stats[(size_t)b * STATS_STRIDE + 2 + 2 * s] = s1;
        stats[(size_t)b * STATS_STRIDE + 3 + 2 * s] = s2;
    }
}

// bwd pass 2: one thread per float4 of the slab, looping over ITS SLICE of the batch (blockIdx.y): dx written,
// dgamma/dbeta/dadd summed in registers and added once per slice (float atomics when there is more than one slice).
// One slice for the whole batch meant 98 workgroups for a 196 x 512 slab and two loads in flight per thread: 2.4 TB/s.
__global__ __launch_bounds__(256) void lnnd_bwd_apply_kernel(const float* __restrict__ dy, long dy_bs,
                                                             const float* __restrict__ x, const float* __restrict__ add,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ stats, float* __restrict__ dx,
                                                             float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                             float* __restrict__ dadd, int B, int n) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int n4 = n >> 2;
    if (c >= n4) return;
    const f32x4 gm = reinterpret_cast<const f32x4*>(gamma)[c];
    const f32x4 ad = add ? reinterpret_cast<const f32x4*>(add)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 ag = {0.f, 0.f, 0.f, 0.f}, ab = ag, aa = ag;
    const int per = (B + gridDim.y - 1) / gridDim.y;
    const int b0 = blockIdx.y * per, b1 = min(B, b0 + per);
#pragma unroll 2
    for (int b = b0; b < b1; ++b) {
        const float* st = stats + (size_t)b * STATS_STRIDE;
        const float mean = st[0], rstd = st[1];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int s = 0; s < NSPLIT; ++s) {
            s1 += st[2 + 2 * s];
            s2 += st[3 + 2 * s];
        }
        const float c1 = s1 / n, c2 = s2 / n;
        f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)b * n)[c] + ad;
        f32x4 d = reinterpret_cast<const f32x4*>(dy + (size_t)b * dy_bs)[c];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float xh = (v[e] - mean) * rstd;
            o[e] = rstd * (d[e] * gm[e] - c1 - xh * c2);
            ag[e] += d[e] * xh;
            ab[e] += d[e];
            aa[e] += o[e];
        }
        reinterpret_cast<f32x4*>(dx + (size_t)b * n)[c] = o;
    }
    if (gridDim.y == 1) {
        if (dgamma) reinterpret_cast<f32x4*>(dgamma)[c] += ag;
        if (dbeta) reinterpret_cast<f32x4*>(dbeta)[c] += ab;
        if (dadd) reinterpret_cast<f32x4*>(dadd)[c] += aa;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (dgamma) atomicAdd(dgamma + 4 * (size_t)c + e, ag[e]);
            if (dbeta) atomicAdd(dbeta + 4 * (size_t)c + e, ab[e]);
            if (dadd) atomicAdd(dadd + 4 * (size_t)c + e, aa[e]);
        }
    }
}

}  // namespace

extern "C" int i2t_layernorm_fwd(void* stream, const float* x, const float* gamma, const float* beta, void* y,
                                 int y_is_f32, float* mean, float* rstd, int M, int d) {
    return i2t_layernorm_fwd_eps(stream, x, gamma, beta, y, y_is_f32, mean, rstd, M, d, LN_EPS);
}

extern "C" int i2t_layernorm_fwd_eps(void* stream, const float* x, const float* gamma, const float* beta, void* y,
                                     int y_is_f32, float* mean, float* rstd, int M, int d, float eps) {
    I2T_REQUIRE(x && gamma && y && M > 0 && eps > 0.f, "i2t_layernorm_fwd: bad args");
    I2T_REQUIRE(d % 4 == 0 && d <= WIDEC * 1024, "i2t_layernorm_fwd: d=%d must be a multiple of 4 and <= %d", d, WIDEC * 1024);
    hipStream_t s = (hipStream_t)stream;
    if (d > MAXC * 256) {                 // wide rows: one workgroup per row
        const int wgrid = M < 16384 ? M : 16384;
        if (y_is_f32) hipLaunchKernelGGL(ln_fwd_wide_kernel<true>, dim3(wgrid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, M, d, eps);
        else hipLaunchKernelGGL(ln_fwd_wide_kernel<false>, dim3(wgrid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, M, d, eps);
        I2T_CHECK_LAUNCH("i2t_layernorm_fwd");
        return I2T_OK;
    }
    int grid = (M + 3) / 4;
    if (grid > 8192) grid = 8192;
    if (y_is_f32) hipLaunchKernelGGL(ln_fwd_kernel<true>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, M, d, eps);
    else hipLaunchKernelGGL(ln_fwd_kernel<false>, dim3(grid), dim3(256), 0, s, x, gamma, beta, y, mean, rstd, M, d, eps);
    I2T_CHECK_LAUNCH("i2t_layernorm_fwd");
    return I2T_OK;
}

extern "C" int i2t_layernorm_bwd(void* stream, const void* dy, int dy_is_f32, const float* x, const float* gamma,
                                 const float* mean, const float* rstd, float* dx, int dx_accumulate, void* dx_bf16,
                                 float* dgamma, float* dbeta, int M, int d, unsigned drop_key, unsigned drop_thr,
                                 float drop_scale, float* sumsq_out, const float* dx_pre_sumsq) {
    return i2t_layernorm_bwd_ex(stream, dy, dy_is_f32, x, gamma, mean, rstd, dx, dx_accumulate, dx_bf16, dgamma, dbeta, M, d, drop_key, drop_thr,
                                drop_scale, sumsq_out, dx_pre_sumsq, 0u, 0u, 1.0f, 0, 0);
}

extern "C" int i2t_layernorm_bwd_ex(void* stream, const void* dy, int dy_is_f32, const float* x, const float* gamma,
                                    const float* mean, const float* rstd, float* dx, int dx_accumulate, void* dx_bf16,
                                    float* dgamma, float* dbeta, int M, int d, unsigned drop_key, unsigned drop_thr,
                                    float drop_scale, float* sumsq_out, const float* dx_pre_sumsq, unsigned dx_mask_key,
                                    unsigned dx_mask_thr, float dx_mask_scale, int acc_period, int acc_rows) {
    I2T_REQUIRE(acc_period == 0 || (dx_accumulate && acc_rows >= 0 && acc_rows <= acc_period && d <= MAXC * 256),
                "i2t_layernorm_bwd_ex: acc_period=%d acc_rows=%d (needs dx_accumulate, 0 <= acc_rows <= acc_period)", acc_period, acc_rows);
    I2T_REQUIRE(!dx_mask_thr || ((long)M * d < (1L << 32) && d <= MAXC * 256), "i2t_layernorm_bwd_ex: the f32 mask needs M*d < 2^32 and d <= %d", MAXC * 256);
    I2T_REQUIRE(dy && x && gamma && mean && rstd && dx && M > 0, "i2t_layernorm_bwd: bad args");
    I2T_REQUIRE(!dx_pre_sumsq || dx_accumulate, "i2t_layernorm_bwd: dx_pre_sumsq only applies when accumulating onto dx");
    I2T_REQUIRE(drop_thr == 0 || (dx_bf16 && (long)M * d < (1L << 32)), "i2t_layernorm_bwd: dropout needs dx_bf16 and M*d < 2^32");
    I2T_REQUIRE(d % 4 == 0 && d <= WIDEC * 1024, "i2t_layernorm_bwd: d=%d must be a multiple of 4 and <= %d", d, WIDEC * 1024);
    hipStream_t s = (hipStream_t)stream;
    if (d > MAXC * 256) {                 // wide rows: one workgroup per row group; the extras of the narrow form are not offered here
        I2T_REQUIRE(drop_thr == 0 && !sumsq_out && !dx_pre_sumsq, "i2t_layernorm_bwd: d=%d > %d supports neither dropout nor the gradient-normaliser extras", d, MAXC * 256);
        const int wgrid = (M + LN_BWD_WIDE_ROWS - 1) / LN_BWD_WIDE_ROWS, wper = i2t_det() ? 1 : wgrid;
        for (int b0 = 0; b0 < wgrid; b0 += wper) {
            if (dy_is_f32)
                hipLaunchKernelGGL(ln_bwd_wide_kernel<true>, dim3(wper), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx, dx_accumulate,
                                   (bf16_t*)dx_bf16, dgamma, dbeta, M, d, b0);
            else
                hipLaunchKernelGGL(ln_bwd_wide_kernel<false>, dim3(wper), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx, dx_accumulate,
                                   (bf16_t*)dx_bf16, dgamma, dbeta, M, d, b0);
        }
        I2T_CHECK_LAUNCH("i2t_layernorm_bwd");
        return I2T_OK;
    }
    int grid = (M + LN_BWD_ROWS - 1) / LN_BWD_ROWS;
    // deterministic mode: the workgroups' atomics onto dgamma / dbeta / sumsq_out land in workgroup order (one launch each)
    const int per = i2t_det() ? 1 : grid;
    for (int b0 = 0; b0 < grid; b0 += per) {
        if (dy_is_f32)
            hipLaunchKernelGGL(ln_bwd_kernel<true>, dim3(per), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx, dx_accumulate,
                               (bf16_t*)dx_bf16, dgamma, dbeta, M, d, drop_key, drop_thr, drop_scale, sumsq_out, dx_pre_sumsq, b0, dx_mask_key, dx_mask_thr,
                               dx_mask_scale, acc_period, acc_rows);
        else
            hipLaunchKernelGGL(ln_bwd_kernel<false>, dim3(per), dim3(256), 0, s, dy, x, gamma, mean, rstd, dx, dx_accumulate,
                               (bf16_t*)dx_bf16, dgamma, dbeta, M, d, drop_key, drop_thr, drop_scale, sumsq_out, dx_pre_sumsq, b0, dx_mask_key, dx_mask_thr,
                               dx_mask_scale, acc_period, acc_rows);
    }
    I2T_CHECK_LAUNCH("i2t_layernorm_bwd");
    return I2T_OK;
}

extern "C" int i2t_layernorm_nd_fwd(void* stream, const float* x, const float* add, const float* gamma,
                                    const float* beta, float* y, long y_batch_stride, float* stats, int B, int rows,
                                    int d) {
    return i2t_layernorm_nd_fwd_drop(stream, x, add, gamma, beta, y, y_batch_stride, stats, B, rows, d, 0u, 0u, 1.0f, 0);
}

extern "C" int i2t_layernorm_nd_fwd_drop(void* stream, const float* x, const float* add, const float* gamma,
                                         const float* beta, float* y, long y_batch_stride, float* stats, int B, int rows,
                                         int d, unsigned drop_key, unsigned drop_thr, float drop_scale, long drop_base) {
    I2T_REQUIRE(x && gamma && y && stats && B > 0, "i2t_layernorm_nd_fwd: bad args");
    const long n = (long)rows * d;
    I2T_REQUIRE(n % 4 == 0 && n < (1L << 30) && y_batch_stride % 4 == 0, "i2t_layernorm_nd_fwd: slab size %ld unsupported", n);
    I2T_REQUIRE(!drop_thr || (drop_base >= 0 && drop_base % 4 == 0 && (double)B * y_batch_stride < 4294967296.0),
                "i2t_layernorm_nd_fwd_drop: dropout needs drop_base %% 4 == 0 and B * y_batch_stride < 2^32");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lnnd_partial_kernel, dim3(B, NSPLIT), dim3(256), 0, s, x, add, stats, (int)n);
    hipLaunchKernelGGL(lnnd_apply_kernel, dim3(B, NSPLIT), dim3(256), 0, s, x, add, gamma, beta, y, y_batch_stride, stats, (int)n,
                       drop_key, drop_thr, drop_scale, drop_base);
    I2T_CHECK_LAUNCH("i2t_layernorm_nd_fwd");
    return I2T_OK;
}

extern "C" int i2t_layernorm_nd_bwd(void* stream, const float* dy, long dy_batch_stride, const float* x,
                                    const float* add, const float* gamma, const float* stats_c, float* dx,
                                    float* dgamma, float* dbeta, float* dadd, int B, int rows, int d) {
    I2T_REQUIRE(dy && x && gamma && stats_c && dx && B > 0, "i2t_layernorm_nd_bwd: bad args");
    const long n = (long)rows * d;
    I2T_REQUIRE(n % 4 == 0 && n < (1L << 30) && dy_batch_stride % 4 == 0, "i2t_layernorm_nd_bwd: slab size %ld unsupported", n);
    float* stats = const_cast<float*>(stats_c);   // partial slots [2..] are scratch; mean/rstd stay intact
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lnnd_bwd_partial_kernel, dim3(B, NSPLIT), dim3(256), 0, s, dy, dy_batch_stride, x, add, gamma, stats, (int)n);
    const int n4 = (int)(n >> 2);
    const int col_blocks = (n4 + 255) / 256;
    int slices = 1;                                            // >= 32 images per slice, ~1024 workgroups at most
    while (slices < 16 && col_blocks * slices < 1024 && B / (slices * 2) >= 32) slices *= 2;
    if (i2t_det()) slices = 1;                                 // one slice: plain adds, no atomics
    hipLaunchKernelGGL(lnnd_bwd_apply_kernel, dim3(col_blocks, slices), dim3(256), 0, s, dy, dy_batch_stride, x, add, gamma,
                       stats, dx, dgamma, dbeta, dadd, B, (int)n);
    I2T_CHECK_LAUNCH("i2t_layernorm_nd_bwd");
    return I2T_OK;
}
