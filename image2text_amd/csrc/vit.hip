// PretrainedViT (reference models/encoder.py:56-127): the pieces around the GEMM / attention / LayerNorm kernels.
//   * patchify: torchvision's conv_proj (p x p, stride p) is a GEMM over im2col rows -- fp32 NCHW image -> bf16 [B P^2][3 p p]
//   * token assembly: [class_token | patch embeddings] + pos_embedding
//   * L2 normalisation rows (F.normalize) around the per-slot MLP head
//   * PEER product-key lookup (models/layers.py:37-109): top-k of the two half-key score rows, top-k of their k x k sums, softmax,
//     gathered expert rows -- one workgroup per (image, slot) row, heads in a loop, no atomics in the forward
//   * LSH cosine embeddings (models/layers.py:112-143): fp32 projections (bucket decisions are discontinuous: no bf16 here),
//     bucketize against the module's own grid, EmbeddingBag(mean)
// All HBM/latency-bound: rows = images x n_cls slots, a few thousand at most.
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------------------------ patchify
// out[(b, py, px)][(c, ky, kx)] = img[b][c][py p + ky][px p + kx]; one thread = 8 consecutive kx (32 B in, 16 B out)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, bf16_t* __restrict__ out, long n8, int C, int H, int W, int p) {
    const int p8 = p / 8, gw = W / p, gh = H / p;
    const int row_chunks = C * p * p8;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long r = i / row_chunks;
        int q = (int)(i - r * row_chunks);
        const int kx8 = q % p8; q /= p8;
        const int ky = q % p, c = q / p;
        const int px = (int)(r % gw);
        const long t = r / gw;
        const int py = (int)(t % gh);
        const long b = t / gh;
        const float* src = img + (((size_t)b * C + c) * H + (size_t)py * p + ky) * W + (size_t)px * p + kx8 * 8;
        const f32x4 a = *reinterpret_cast<const f32x4*>(src), bb = *reinterpret_cast<const f32x4*>(src + 4);
        const u32x4 o = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(bb[0], bb[1]), pack_bf16x2(bb[2], bb[3])};
        *reinterpret_cast<u32x4*>(out + i * 8) = o;
    }
}

// x[b][0] = cls + pos[0]; x[b][1 + j] = proj[b P2 + j] + pos[1 + j]
__global__ __launch_bounds__(256) void vit_tokens_kernel(const float* __restrict__ proj, const float* __restrict__ cls, const float* __restrict__ pos,
                                                         float* __restrict__ x, long n4, int T, int d4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const long r = i / d4;
        const int c = (int)(i - r * d4);
        const int t = (int)(r % T);
        const long b = r / T;
        const f32x4 a = t == 0 ? reinterpret_cast<const f32x4*>(cls)[c]
                               : reinterpret_cast<const f32x4*>(proj)[((size_t)b * (T - 1) + (t - 1)) * d4 + c];
        reinterpret_cast<f32x4*>(x)[i] = a + reinterpret_cast<const f32x4*>(pos)[(size_t)t * d4 + c];
    }
}

// ------------------------------------------------------------------------------------------------------------ L2 rows
// y = x / max(||x||, 1e-12) (F.normalize, p = 2); one wave per row; inv[row] saved for the backward
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, bf16_t* __restrict__ yb,
                                                         float* __restrict__ inv, int M, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * d;
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c);
        s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    s = wave_sum(s);
    const float r = 1.0f / fmaxf(sqrtf(s), 1e-12f);
    if (lane == 0 && inv) inv[row] = r;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c) * r;
        if (y) *reinterpret_cast<f32x4*>(y + (size_t)row * d + c) = v;
        if (yb) *reinterpret_cast<u32x2*>(yb + (size_t)row * d + c) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}

// dx = inv (dy - y <y, dy>), y = x inv  (rows whose norm hit the 1e-12 floor: dx = inv dy, the clamp is constant there)
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ inv,
                                                         float* __restrict__ dx, int accumulate, int M, int d) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    const float* xr = x + (size_t)row * d;
    const float* gr = dy + (size_t)row * d;
    const float r = inv[row];
    float s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), g = *reinterpret_cast<const f32x4*>(gr + c);
        s += v[0] * g[0] + v[1] * g[1] + v[2] * g[2] + v[3] * g[3];
    }
    s = wave_sum(s) * r * r;                                     // <y, dy> / ||x||
    if (r >= 1e12f) s = 0.f;
    for (int c = lane * 4; c < d; c += 256) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(xr + c), g = *reinterpret_cast<const f32x4*>(gr + c);
        f32x4 o = (g - v * s) * r;
        float* dst = dx + (size_t)row * d + c;
        if (accumulate) o += *reinterpret_cast<const f32x4*>(dst);
        *reinterpret_cast<f32x4*>(dst) = o;
    }
}

// dst[b][c][r] = src[b][r][c] (fp32 in; fp32 and / or bf16 out); tiny matrices (R = n_cls <= 64 on one side)
__global__ __launch_bounds__(256) void transpose_last2_kernel(const float* __restrict__ src, float* __restrict__ dst, bf16_t* __restrict__ dstb,
                                                              long n, int R, int C) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {      // i indexes dst: (b, c, r)
        const int r = (int)(i % R);
        const long t = i / R;
        const int c = (int)(t % C);
        const long b = t / C;
        const float v = src[((size_t)b * R + r) * C + c];
        if (dst) dst[i] = v;
        if (dstb) dstb[i] = f32_to_bf16(v);
    }
}

// ------------------------------------------------------------------------------------------------------------ PEER
struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax wave_argmax(ArgMax a) {      // larger value wins, ties -> smaller index
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(a.v, o, 64);
        const int oi = __shfl_xor(a.i, o, 64);
        if (ov > a.v || (ov == a.v && oi < a.i)) { a.v = ov; a.i = oi; }
    }
    return a;
}

constexpr int PEER_MAXK = 16, PEER_MAXQ = 1024;

// k rounds of wave-wide argmax over vals[0..n) held in LDS (chosen entries are overwritten with -inf); one wave
__device__ __forceinline__ void wave_topk(float* vals, int n, int k, float* out_v, int* out_i, int lane) {
    for (int r = 0; r < k; ++r) {
        ArgMax a = {-INFINITY, 0x7fffffff};
        for (int j = lane; j < n; j += 64) {
            const float v = vals[j];
            if (v > a.v) { a.v = v; a.i = j; }
        }
        a = wave_argmax(a);
        if (lane == 0) {
            out_v[r] = a.v;
            out_i[r] = a.i;
            vals[a.i] = -INFINITY;
        }
        __builtin_amdgcn_wave_barrier();
        __threadfence_block();
    }
}

// One workgroup (4 waves) per row r = (image, slot).  S fp32 [M nh][2 nq] = [left scores | right scores] of every (row, head);
// ip bf16 [M][nh din] = key_linear(inp); res fp32 [M][dout] = residual(inp); e_in bf16 [units][din]; e_out bf16 [units][dout].
// Saves per (row, head, j): unit, left / right query-unit index, softmax score, pre-GELU dot.
__global__ __launch_bounds__(256) void peer_lookup_fwd_kernel(const float* __restrict__ S, const bf16_t* __restrict__ ip, const float* __restrict__ res,
                                                              const bf16_t* __restrict__ e_in, const bf16_t* __restrict__ e_out, float* __restrict__ out,
                                                              int* __restrict__ sv_unit, int* __restrict__ sv_lr, float* __restrict__ sv_score,
                                                              float* __restrict__ sv_dot, int nh, int nq, int k, int din, int dout) {
    __shared__ float sl[PEER_MAXQ], sr[PEER_MAXQ], cross[PEER_MAXK * PEER_MAXK];
    __shared__ float lv[PEER_MAXK], rv[PEER_MAXK], cv[PEER_MAXK], fw[PEER_MAXK];
    __shared__ int li[PEER_MAXK], ri[PEER_MAXK], ci[PEER_MAXK], unit[PEER_MAXK];
    const int row = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    constexpr int MAXC = 8;                                     // out columns per thread: dout <= 256 * 4 * MAXC / ... (float4 chunks)
    f32x4 acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int d4 = dout / 4;
    for (int h = 0; h < nh; ++h) {
        const float* s = S + ((size_t)row * nh + h) * 2 * nq;
        for (int j = tid; j < nq; j += 256) { sl[j] = s[j]; sr[j] = s[nq + j]; }
        __syncthreads();
        if (wave == 0) wave_topk(sl, nq, k, lv, li, lane);
        if (wave == 1) wave_topk(sr, nq, k, rv, ri, lane);
        __syncthreads();
        for (int j = tid; j < k * k; j += 256) cross[j] = lv[j / k] + rv[j % k];
        __syncthreads();
        if (wave == 0) {
            wave_topk(cross, k * k, k, cv, ci, lane);
            if (lane == 0) {
                float mx = cv[0], den = 0.f;
                for (int j = 0; j < k; ++j) den += expf(cv[j] - mx);
                for (int j = 0; j < k; ++j) {
                    const int l = li[ci[j] / k], r = ri[ci[j] % k];
                    unit[j] = l * k + r;                        // reference layers.py:93-96: stride topk (kept as is)
                    const size_t o = ((size_t)row * nh + h) * k + j;
                    sv_unit[o] = unit[j];
                    sv_lr[2 * o] = l;
                    sv_lr[2 * o + 1] = r;
                    const float sc = expf(cv[j] - mx) / den;
                    sv_score[o] = sc;
                    cv[j] = sc;
                }
            }
        }
        __syncthreads();
        const bf16_t* ipr = ip + ((size_t)row * nh + h) * din;
        for (int j = wave; j < k; j += 4) {                     // one wave per candidate: <e_in[unit], ip>
            const bf16_t* e = e_in + (size_t)unit[j] * din;
            float t = 0.f;
            for (int c = lane * 8; c < din; c += 512) {
                const u32x4 a = *reinterpret_cast<const u32x4*>(e + c), b = *reinterpret_cast<const u32x4*>(ipr + c);
#pragma unroll
                for (int q = 0; q < 4; ++q) t += bf16lo(a[q]) * bf16lo(b[q]) + bf16hi(a[q]) * bf16hi(b[q]);
            }
            t = wave_sum(t);
            if (lane == 0) {
                sv_dot[((size_t)row * nh + h) * k + j] = t;
                fw[j] = cv[j] * gelu_tanh(t);
            }
        }
        __syncthreads();
        for (int j = 0; j < k; ++j) {
            const bf16_t* e = e_out + (size_t)unit[j] * dout;
            const float w = fw[j];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int col = tid + 256 * c;
                if (col < d4) {
                    const u32x2 v = *reinterpret_cast<const u32x2*>(e + 4 * col);
                    acc[c] += f32x4{bf16lo(v[0]), bf16hi(v[0]), bf16lo(v[1]), bf16hi(v[1])} * w;
                }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
        const int col = tid + 256 * c;
        if (col < d4) reinterpret_cast<f32x4*>(out + (size_t)row * dout)[col] = acc[c] + reinterpret_cast<const f32x4*>(res + (size_t)row * dout)[col];
    }
}

// Backward of the lookup for one row: dout fp32 [M][dout] -> dS (fp32, pre-zeroed by the caller) at the chosen left / right entries,
// dip bf16 [M][nh din], and the expert tables' gradients by fp32 atomics (several rows may hit the same unit).
__global__ __launch_bounds__(256) void peer_lookup_bwd_kernel(const float* __restrict__ dout, const bf16_t* __restrict__ ip,
                                                              const bf16_t* __restrict__ e_in, const bf16_t* __restrict__ e_out,
                                                              const int* __restrict__ sv_unit, const int* __restrict__ sv_lr,
                                                              const float* __restrict__ sv_score, const float* __restrict__ sv_dot,
                                                              float* __restrict__ dS, bf16_t* __restrict__ dip, float* __restrict__ g_in,
                                                              float* __restrict__ g_out, int nh, int nq, int k, int din, int dout_w, int M) {
    __shared__ float dfw[PEER_MAXK], dt[PEER_MAXK];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    // (deterministic mode launches ONE workgroup that walks the rows in order: the table gradients are then summed in a fixed order)
    for (int row = blockIdx.x; row < M; row += gridDim.x) {
    const float* g = dout + (size_t)row * dout_w;
    for (int h = 0; h < nh; ++h) {
        const size_t base = ((size_t)row * nh + h) * k;
        for (int j = wave; j < k; j += 4) {                     // d fw_j = <dout, e_out[unit_j]>;  g_out[unit_j] += fw_j dout
            const int u = sv_unit[base + j];
            const float t = sv_dot[base + j], sc = sv_score[base + j];
            const float fwj = sc * gelu_tanh(t);
            const bf16_t* e = e_out + (size_t)u * dout_w;
            float acc = 0.f;
            for (int c = lane * 4; c < dout_w; c += 256) {
                const u32x2 v = *reinterpret_cast<const u32x2*>(e + c);
                const f32x4 gg = *reinterpret_cast<const f32x4*>(g + c);
                acc += bf16lo(v[0]) * gg[0] + bf16hi(v[0]) * gg[1] + bf16lo(v[1]) * gg[2] + bf16hi(v[1]) * gg[3];
                if (g_out) {
                    float* go = g_out + (size_t)u * dout_w + c;
                    atomicAdd(go, fwj * gg[0]); atomicAdd(go + 1, fwj * gg[1]); atomicAdd(go + 2, fwj * gg[2]); atomicAdd(go + 3, fwj * gg[3]);
                }
            }
            acc = wave_sum(acc);
            if (lane == 0) dfw[j] = acc;
        }
        __syncthreads();
        if (tid == 0) {                                         // softmax / GELU backward over the k candidates, scatter into dS
            float dot_sd = 0.f, ds[PEER_MAXK];
            for (int j = 0; j < k; ++j) {
                const float t = sv_dot[base + j], sc = sv_score[base + j];
                ds[j] = dfw[j] * gelu_tanh(t);
                dt[j] = dfw[j] * sc * gelu_tanh_grad(t);
                dot_sd += sc * ds[j];
            }
            float* dsr = dS + ((size_t)row * nh + h) * 2 * nq;
            for (int j = 0; j < k; ++j) {
                const float dd = sv_score[base + j] * (ds[j] - dot_sd);
                dsr[sv_lr[2 * (base + j)]] += dd;
                dsr[nq + sv_lr[2 * (base + j) + 1]] += dd;
            }
        }
        __syncthreads();
        const bf16_t* ipr = ip + ((size_t)row * nh + h) * din;
        for (int c = tid * 2; c < din; c += 512) {              // dip = sum_j dt_j e_in[unit_j];  g_in[unit_j] += dt_j ip
            const unsigned pv = *reinterpret_cast<const unsigned*>(ipr + c);
            float a0 = 0.f, a1 = 0.f;
            for (int j = 0; j < k; ++j) {
                const int u = sv_unit[base + j];
                const unsigned ev = *reinterpret_cast<const unsigned*>(e_in + (size_t)u * din + c);
                a0 += dt[j] * bf16lo(ev);
                a1 += dt[j] * bf16hi(ev);
                if (g_in) {
                    atomicAdd(g_in + (size_t)u * din + c, dt[j] * bf16lo(pv));
                    atomicAdd(g_in + (size_t)u * din + c + 1, dt[j] * bf16hi(pv));
                }
            }
            *reinterpret_cast<unsigned*>(dip + ((size_t)row * nh + h) * din + c) = pack_bf16x2(a0, a1);
        }
        __syncthreads();
    }
    }
}

// ------------------------------------------------------------------------------------------------------------ LSH
// z[M][N] = x[M][K] . P[K][N] in fp32 (64 x 64 tiles, 4 x 4 per thread)
__global__ __launch_bounds__(256) void f32_gemm_kernel(const float* __restrict__ x, const float* __restrict__ P, float* __restrict__ z, int M, int N, int K) {
    __shared__ float xs[16][64 + 1], ps[16][64];
    const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    float acc[4][4] = {};
    for (int k0 = 0; k0 < K; k0 += 16) {
        for (int i = threadIdx.x; i < 64 * 16; i += 256) {
            const int r = i >> 4, kk = i & 15;
            xs[kk][r] = (m0 + r < M && k0 + kk < K) ? x[(size_t)(m0 + r) * K + k0 + kk] : 0.f;
            const int kp = i >> 6, c = i & 63;
            ps[kp][c] = (k0 + kp < K && n0 + c < N) ? P[(size_t)(k0 + kp) * N + n0 + c] : 0.f;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) {
            float a[4], b[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] = xs[kk][ty * 4 + i]; b[i] = ps[kk][tx * 4 + i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(a[i], b[j], acc[i][j]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (m0 + ty * 4 + i < M && n0 + tx * 4 + j < N) z[(size_t)(m0 + ty * 4 + i) * N + n0 + tx * 4 + j] = acc[i][j];
}

// One workgroup per (image b, slot s).  z fp32 [B][n_cls nK n_proj]; table k of slot s = tab + s slot_stride + tab_off[k], fp32
// [(nb_k + 1) n_proj][dout]; grid values of table k = grids + grid_off[k] (nb_k floats).  torch.bucketize(z, grid) = number of grid
// points strictly below z.  rows int32 [B][n_cls][nK][n_proj] saved for the backward.
__global__ __launch_bounds__(256) void lsh_embed_fwd_kernel(const float* __restrict__ z, const float* __restrict__ tab, long slot_stride,
                                                            const long* __restrict__ tab_off, const int* __restrict__ nbins,
                                                            const float* __restrict__ grids, const int* __restrict__ grid_off,
                                                            float* __restrict__ out, int* __restrict__ rows, int n_cls, int nK, int n_proj, int dout) {
    __shared__ int rid[1024];
    const int b = blockIdx.x / n_cls, s = blockIdx.x % n_cls, tid = threadIdx.x;
    const float* zr = z + ((size_t)b * n_cls + s) * nK * n_proj;
    for (int i = tid; i < nK * n_proj; i += 256) {
        const int kk = i / n_proj, j = i % n_proj, nb = nbins[kk];
        const float v = zr[i];
        const float* gr = grids + grid_off[kk];
        int cnt = 0;
        for (int q = 0; q < nb; ++q) cnt += gr[q] < v ? 1 : 0;
        rid[i] = cnt + (nb + 1) * j;
        rows[(size_t)blockIdx.x * nK * n_proj + i] = rid[i];
    }
    __syncthreads();
    const float inv = 1.0f / (float)n_proj;
    for (int c = tid * 4; c < dout; c += 1024) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
        for (int kk = 0; kk < nK; ++kk) {
            const float* t = tab + (size_t)s * slot_stride + tab_off[kk];
            f32x4 a = {0.f, 0.f, 0.f, 0.f};
            for (int j = 0; j < n_proj; ++j) a += *reinterpret_cast<const f32x4*>(t + (size_t)rid[kk * n_proj + j] * dout + c);
            acc += a * inv;
        }
        *reinterpret_cast<f32x4*>(out + (size_t)blockIdx.x * dout + c) = acc;
    }
}

__global__ __launch_bounds__(256) void lsh_embed_bwd_kernel(const float* __restrict__ dy, const int* __restrict__ rows, float* __restrict__ gtab,
                                                            long slot_stride, const long* __restrict__ tab_off, int n_cls, int nK, int n_proj,
                                                            int dout, int nblk) {
    const int tid = threadIdx.x;
    const float inv = 1.0f / (float)n_proj;
    // (deterministic mode launches ONE workgroup that walks the (image, slot) rows in order; a column belongs to one thread throughout)
    for (int blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
        const int s = blk % n_cls;
        const int* rr = rows + (size_t)blk * nK * n_proj;
        for (int c = tid; c < dout; c += 256) {
            const float g = dy[(size_t)blk * dout + c] * inv;
            for (int kk = 0; kk < nK; ++kk) {
                float* t = gtab + (size_t)s * slot_stride + tab_off[kk];
                for (int j = 0; j < n_proj; ++j) atomicAdd(t + (size_t)rr[kk * n_proj + j] * dout + c, g);
            }
        }
    }
}

inline unsigned grid_for(long n) {
    const long blocks = (n + 255) / 256;
    return (unsigned)(blocks < 65536 ? blocks : 65536);
}

}  // namespace

extern "C" int i2t_patchify(void* stream, const float* images, void* out, int B, int C, int H, int W, int p) {
    I2T_REQUIRE(images && out && B > 0 && C > 0 && p >= 8 && p % 8 == 0 && H % p == 0 && W % p == 0 && ALIGNED16(images) && ALIGNED16(out),
                "i2t_patchify: bad args (patch %d must be a multiple of 8 and divide %dx%d)", p, H, W);
    const long n8 = (long)B * C * H * W / 8;
    hipLaunchKernelGGL(patchify_kernel, dim3(grid_for(n8)), dim3(256), 0, (hipStream_t)stream, images, (bf16_t*)out, n8, C, H, W, p);
    I2T_CHECK_LAUNCH("i2t_patchify");
    return I2T_OK;
}

extern "C" int i2t_vit_tokens(void* stream, const float* proj, const float* cls, const float* pos, float* x, int B, int T, int d) {
    I2T_REQUIRE(proj && cls && pos && x && B > 0 && T > 1 && d > 0 && d % 4 == 0 && ALIGNED16(proj) && ALIGNED16(cls) && ALIGNED16(pos) && ALIGNED16(x),
                "i2t_vit_tokens: bad args (d=%d must be a multiple of 4)", d);
    const long n4 = (long)B * T * (d / 4);
    hipLaunchKernelGGL(vit_tokens_kernel, dim3(grid_for(n4)), dim3(256), 0, (hipStream_t)stream, proj, cls, pos, x, n4, T, d / 4);
    I2T_CHECK_LAUNCH("i2t_vit_tokens");
    return I2T_OK;
}

extern "C" int i2t_l2norm_fwd(void* stream, const float* x, float* y, void* y_bf16, float* inv_norm, int M, int d) {
    I2T_REQUIRE(x && (y || y_bf16) && M > 0 && d > 0 && d % 4 == 0 && ALIGNED16(x) && (!y || ALIGNED16(y)) && (!y_bf16 || ALIGNED16(y_bf16)),
                "i2t_l2norm_fwd: bad args (d=%d must be a multiple of 4)", d);
    hipLaunchKernelGGL(l2norm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, y, (bf16_t*)y_bf16, inv_norm, M, d);
    I2T_CHECK_LAUNCH("i2t_l2norm_fwd");
    return I2T_OK;
}

extern "C" int i2t_l2norm_bwd(void* stream, const float* dy, const float* x, const float* inv_norm, float* dx, int accumulate, int M, int d) {
    I2T_REQUIRE(dy && x && inv_norm && dx && M > 0 && d > 0 && d % 4 == 0 && ALIGNED16(dy) && ALIGNED16(x) && ALIGNED16(dx),
                "i2t_l2norm_bwd: bad args (d=%d must be a multiple of 4)", d);
    hipLaunchKernelGGL(l2norm_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, dy, x, inv_norm, dx, accumulate, M, d);
    I2T_CHECK_LAUNCH("i2t_l2norm_bwd");
    return I2T_OK;
}

extern "C" int i2t_transpose_last2(void* stream, const float* src, float* dst, void* dst_bf16, long B, int R, int C) {
    I2T_REQUIRE(src && (dst || dst_bf16) && B > 0 && R > 0 && C > 0, "i2t_transpose_last2: bad args");
    const long n = B * R * C;
    hipLaunchKernelGGL(transpose_last2_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, src, dst, (bf16_t*)dst_bf16, n, R, C);
    I2T_CHECK_LAUNCH("i2t_transpose_last2");
    return I2T_OK;
}

extern "C" int i2t_peer_lookup_fwd(void* stream, const float* scores, const void* inp_proj, const float* residual, const void* emb_in,
                                   const void* emb_out, float* out, int* sv_unit, int* sv_lr, float* sv_score, float* sv_dot, int M, int nhead,
                                   int nq, int topk, int din, int dout) {
    I2T_REQUIRE(scores && inp_proj && residual && emb_in && emb_out && out && sv_unit && sv_lr && sv_score && sv_dot && M > 0 && nhead > 0 &&
                    nq >= topk && nq <= PEER_MAXQ && topk >= 1 && topk <= PEER_MAXK && din % 8 == 0 && dout % 4 == 0 && dout <= 8192 &&
                    ALIGNED16(inp_proj) && ALIGNED16(emb_in) && ALIGNED16(emb_out) && ALIGNED16(out) && ALIGNED16(residual),
                "i2t_peer_lookup_fwd: bad args (nq=%d <= %d, topk=%d <= %d, din=%d %% 8, dout=%d %% 4 and <= 8192)", nq, PEER_MAXQ, topk,
                PEER_MAXK, din, dout);
    hipLaunchKernelGGL(peer_lookup_fwd_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, scores, (const bf16_t*)inp_proj, residual,
                       (const bf16_t*)emb_in, (const bf16_t*)emb_out, out, sv_unit, sv_lr, sv_score, sv_dot, nhead, nq, topk, din, dout);
    I2T_CHECK_LAUNCH("i2t_peer_lookup_fwd");
    return I2T_OK;
}

extern "C" int i2t_peer_lookup_bwd(void* stream, const float* dout, const void* inp_proj, const void* emb_in, const void* emb_out,
                                   const int* sv_unit, const int* sv_lr, const float* sv_score, const float* sv_dot, float* dscores,
                                   void* dinp_proj, float* g_emb_in, float* g_emb_out, int M, int nhead, int nq, int topk, int din, int dout_w) {
    I2T_REQUIRE(dout && inp_proj && emb_in && emb_out && sv_unit && sv_lr && sv_score && sv_dot && dscores && dinp_proj && M > 0 && nhead > 0 &&
                    nq <= PEER_MAXQ && topk >= 1 && topk <= PEER_MAXK && din % 8 == 0 && dout_w % 4 == 0 && ALIGNED16(dout) && ALIGNED16(emb_out),
                "i2t_peer_lookup_bwd: bad args");
    hipLaunchKernelGGL(peer_lookup_bwd_kernel, dim3(i2t_det() ? 1 : M), dim3(256), 0, (hipStream_t)stream, dout, (const bf16_t*)inp_proj,
                       (const bf16_t*)emb_in, (const bf16_t*)emb_out, sv_unit, sv_lr, sv_score, sv_dot, dscores, (bf16_t*)dinp_proj, g_emb_in, g_emb_out,
                       nhead, nq, topk, din, dout_w, M);
    I2T_CHECK_LAUNCH("i2t_peer_lookup_bwd");
    return I2T_OK;
}

extern "C" int i2t_gemm_f32(void* stream, const float* x, const float* P, float* z, int M, int N, int K) {
    I2T_REQUIRE(x && P && z && M > 0 && N > 0 && K > 0, "i2t_gemm_f32: bad args");
    hipLaunchKernelGGL(f32_gemm_kernel, dim3((N + 63) / 64, (M + 63) / 64), dim3(256), 0, (hipStream_t)stream, x, P, z, M, N, K);
    I2T_CHECK_LAUNCH("i2t_gemm_f32");
    return I2T_OK;
}

extern "C" int i2t_lsh_embed_fwd(void* stream, const float* z, const float* tables, long slot_stride, const long* tab_off, const int* nbins,
                                 const float* grids, const int* grid_off, float* out, int* rows, int B, int n_cls, int nK, int n_proj, int dout) {
    I2T_REQUIRE(z && tables && tab_off && nbins && grids && grid_off && out && rows && B > 0 && n_cls > 0 && nK > 0 && n_proj > 0 &&
                    nK * n_proj <= 1024 && dout % 4 == 0 && ALIGNED16(tables) && ALIGNED16(out) && slot_stride % 4 == 0,
                "i2t_lsh_embed_fwd: bad args (nK * n_proj = %d <= 1024, dout=%d %% 4)", nK * n_proj, dout);
    hipLaunchKernelGGL(lsh_embed_fwd_kernel, dim3(B * n_cls), dim3(256), 0, (hipStream_t)stream, z, tables, slot_stride, tab_off, nbins, grids,
                       grid_off, out, rows, n_cls, nK, n_proj, dout);
    I2T_CHECK_LAUNCH("i2t_lsh_embed_fwd");
    return I2T_OK;
}

extern "C" int i2t_lsh_embed_bwd(void* stream, const float* dy, const int* rows, float* g_tables, long slot_stride, const long* tab_off, int B,
                                 int n_cls, int nK, int n_proj, int dout) {
    I2T_REQUIRE(dy && rows && g_tables && tab_off && B > 0 && n_cls > 0 && nK > 0 && n_proj > 0, "i2t_lsh_embed_bwd: bad args");
    hipLaunchKernelGGL(lsh_embed_bwd_kernel, dim3(i2t_det() ? 1 : B * n_cls), dim3(256), 0, (hipStream_t)stream, dy, rows, g_tables, slot_stride,
                       tab_off, n_cls, nK, n_proj, dout, B * n_cls);
    I2T_CHECK_LAUNCH("i2t_lsh_embed_bwd");
    return I2T_OK;
}
