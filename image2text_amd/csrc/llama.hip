// Row kernels of the Llama-2 / Qwen2 decoder blocks (reference models/decoder.py:404-440: Llama2HuggingfaceDecoder /
// Qwen2HuggingfaceDecoder wrap transformers' LlamaForCausalLM / Qwen2ForCausalLM; the arithmetic restated here is
// transformers 5.x models/llama/modeling_llama.py -- LlamaRMSNorm, apply_rotary_pos_emb (rotate_half convention), LlamaMLP):
//   * RMSNorm forward / backward:        y = w * x * rsqrt(mean(x^2) + eps)
//   * rotary position embedding:         [x1 | x2] -> [x1 cos - x2 sin | x2 cos + x1 sin] per head, halves of the head width
//   * SwiGLU:                            h = silu(gate) * up  on the fused [gate | up] projection, and its backward
// All HBM-bound row work: one 64-lane wave per row (norms), 16-byte accesses.  GEMMs and attention are the existing kernels
// (gemm.hip, attention_g.hip: H query heads on Hkv key/value heads of width 128).
#include "common.h"

namespace {

constexpr int RMS_MAXC = 8;          // f32x4 chunks per lane: d <= 64 * 4 * 8 = 2048 per pass; wider rows loop
constexpr int RMS_BWD_ROWS = 32;     // rows per workgroup in the backward (one dw atomic per column per workgroup)

// y (bf16) = w * x * rstd, rstd = rsqrt(mean(x^2) + eps); one wave per row
__global__ __launch_bounds__(256) void rms_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, bf16_t* __restrict__ y,
                                                      float* __restrict__ y32, float* __restrict__ rstd_out, int M, int d, float eps) {
    const int lane = threadIdx.x & 63, row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int nc = d >> 2;
    const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
    float ss = 0.f;
    for (int c = lane; c < nc; c += 64) {
        const f32x4 v = xr[c];
        ss += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
    }
    const float rs = rsqrtf(wave_sum(ss) / d + eps);
    if (lane == 0 && rstd_out) rstd_out[row] = rs;
    for (int c = lane; c < nc; c += 64) {
        const f32x4 v = xr[c], g = reinterpret_cast<const f32x4*>(w)[c];
        const f32x4 o = {v[0] * rs * g[0], v[1] * rs * g[1], v[2] * rs * g[2], v[3] * rs * g[3]};
        if (y) reinterpret_cast<u32x2*>(y + (size_t)row * d)[c] = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
        if (y32) reinterpret_cast<f32x4*>(y32 + (size_t)row * d)[c] = o;
    }
}

// dx (+)= rstd * (g - xh * mean(g xh)),  g = dy * w,  xh = x * rstd;   dw += sum_rows dy * xh;   dx_bf16 = bf16(dx)
// A workgroup walks RMS_BWD_ROWS rows (a wave per row); dw partial sums live in registers per PANEL of 64 * 4 * RMS_MAXC columns.
template <bool DY_F32>
__global__ __launch_bounds__(256) void rms_bwd_kernel(const void* __restrict__ dy, const float* __restrict__ x,
                                                      const float* __restrict__ w, const float* __restrict__ rstd,
                                                      float* __restrict__ dx, int dx_accumulate, bf16_t* __restrict__ dx_bf16,
                                                      float* __restrict__ dw, int M, int d, int blk0) {
    __shared__ float red[4][RMS_MAXC * 256];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int nc = d >> 2;
    const int row0 = (blockIdx.x + blk0) * RMS_BWD_ROWS, row_end = min(M, row0 + RMS_BWD_ROWS);      // (blk0: deterministic mode, one workgroup per launch)
    auto load_dy = [&](int row, int c) -> f32x4 {
        if (DY_F32) return reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(dy) + (size_t)row * d)[c];
        const u32x2 pk = reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(dy) + (size_t)row * d)[c];
        return f32x4{bf16lo(pk[0]), bf16hi(pk[0]), bf16lo(pk[1]), bf16hi(pk[1])};
    };
    // pass 1 per row: c2 = mean(g xh); pass 2 per panel: dx and the dw partials (the row's dy / x come from L2 the second time)
    float c2[RMS_BWD_ROWS / 4];
#pragma unroll
    for (int i = 0; i < RMS_BWD_ROWS / 4; ++i) {
        const int row = row0 + wv + 4 * i;
        float s = 0.f;
        if (row < row_end) {
            const float rs = rstd[row];
            const f32x4* xr = reinterpret_cast<const f32x4*>(x + (size_t)row * d);
            for (int c = lane; c < nc; c += 64) {
                const f32x4 v = xr[c], g = reinterpret_cast<const f32x4*>(w)[c], dyv = load_dy(row, c);
#pragma unroll
                for (int e = 0; e < 4; ++e) s += dyv[e] * g[e] * v[e] * rs;
            }
        }
        c2[i] = wave_sum(s) / d;
    }
    for (int p0 = 0; p0 < nc; p0 += 64 * RMS_MAXC) {
        f32x4 pg[RMS_MAXC];
#pragma unroll
        for (int j = 0; j < RMS_MAXC; ++j) pg[j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < RMS_BWD_ROWS / 4; ++i) {
            const int row = row0 + wv + 4 * i;
            if (row >= row_end) continue;
            const float rs = rstd[row];
#pragma unroll
            for (int j = 0; j < RMS_MAXC; ++j) {
                const int c = p0 + lane + 64 * j;
                if (c >= nc) continue;
                const f32x4 v = reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c], g = reinterpret_cast<const f32x4*>(w)[c];
                const f32x4 dyv = load_dy(row, c);
                f32x4* dxp = reinterpret_cast<f32x4*>(dx + (size_t)row * d) + c;
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float xh = v[e] * rs;
                    o[e] = rs * (dyv[e] * g[e] - xh * c2[i]);
                    pg[j][e] += dyv[e] * xh;
                }
                if (dx_accumulate) o += *dxp;
                *dxp = o;
                if (dx_bf16)
                    reinterpret_cast<u32x2*>(dx_bf16 + (size_t)row * d)[c] = u32x2{pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3])};
            }
        }
        if (dw) {
            __syncthreads();
#pragma unroll
            for (int j = 0; j < RMS_MAXC; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) red[wv][(lane + 64 * j) * 4 + e] = pg[j][e];
            __syncthreads();
            for (int t = threadIdx.x; t < RMS_MAXC * 256; t += 256) {
                const int col = p0 * 4 + t;
                if (col < d) atomicAdd(dw + col, red[0][t] + red[1][t] + red[2][t] + red[3][t]);
            }
        }
    }
}

// rotary embedding in place on `nh` heads of width hd starting at column col0 of the bf16 rows x [M][rs]; position of row m:
// pos[m] (packed rows) | *pos_ptr (decode step) | pos_offset + m % T.  cs = [positions][hd] fp32: cos in [0, hd/2), sin in [hd/2, hd).
// A thread owns 8 consecutive dims i and their partners i + hd/2.  inverse: rotate by -angle (the backward of the forward).
__global__ __launch_bounds__(256) void rope_kernel(bf16_t* __restrict__ x, int rs, int col0, int nh, int hd, const float* __restrict__ cs,
                                                   const int* __restrict__ pos, const int* __restrict__ pos_ptr, int pos_offset, int T,
                                                   int M, int inverse) {
    const int per_head = hd >> 4;                         // threads per head (hd/2 pairs, 8 per thread)
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)M * nh * per_head) return;
    const int m = (int)(i / (nh * per_head));
    const int r = (int)(i - (long)m * nh * per_head);
    const int h = r / per_head, j = (r - h * per_head) * 8;
    const int p = pos ? pos[m] : (pos_ptr ? *pos_ptr + pos_offset : pos_offset + m % T);
    const float* c = cs + (size_t)p * hd + j;
    const float* s = c + (hd >> 1);
    bf16_t* a = x + (size_t)m * rs + col0 + h * hd + j;
    bf16_t* b = a + (hd >> 1);
    const u32x4 va = *reinterpret_cast<const u32x4*>(a), vb = *reinterpret_cast<const u32x4*>(b);
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(c), c1 = *reinterpret_cast<const f32x4*>(c + 4);
    f32x4 s0 = *reinterpret_cast<const f32x4*>(s), s1 = *reinterpret_cast<const f32x4*>(s + 4);
    if (inverse) { s0 = -s0; s1 = -s1; }
    u32x4 oa, ob;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float cl = e < 2 ? c0[2 * e] : c1[2 * e - 4], ch = e < 2 ? c0[2 * e + 1] : c1[2 * e - 3];
        const float sl = e < 2 ? s0[2 * e] : s1[2 * e - 4], sh = e < 2 ? s0[2 * e + 1] : s1[2 * e - 3];
        const float al = bf16lo(va[e]), ah = bf16hi(va[e]), bl = bf16lo(vb[e]), bh = bf16hi(vb[e]);
        oa[e] = pack_bf16x2(al * cl - bl * sl, ah * ch - bh * sh);
        ob[e] = pack_bf16x2(bl * cl + al * sl, bh * ch + ah * sh);
    }
    *reinterpret_cast<u32x4*>(a) = oa;
    *reinterpret_cast<u32x4*>(b) = ob;
}

__device__ __forceinline__ float sigmoid_(float v) { return 1.f / (1.f + __expf(-v)); }

// h[m][n] = silu(gu[m][n]) * gu[m][ff + n]
__global__ __launch_bounds__(256) void swiglu_fwd_kernel(const bf16_t* __restrict__ gu, int ld, bf16_t* __restrict__ h, long n8, int ff8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const long m = i / ff8;
    const int n = (int)(i - m * ff8) * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(gu + (size_t)m * ld + n);
    const u32x4 u = *reinterpret_cast<const u32x4*>(gu + (size_t)m * ld + 8 * ff8 + n);
    u32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gl = bf16lo(g[e]), gh = bf16hi(g[e]);
        o[e] = pack_bf16x2(gl * sigmoid_(gl) * bf16lo(u[e]), gh * sigmoid_(gh) * bf16hi(u[e]));
    }
    *reinterpret_cast<u32x4*>(h + (size_t)m * 8 * ff8 + n) = o;
}

// dgu[m][n] = dh * up * (s + g s (1 - s)),  dgu[m][ff + n] = dh * g s     (s = sigmoid(g))
__global__ __launch_bounds__(256) void swiglu_bwd_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ gu, int ld,
                                                         bf16_t* __restrict__ dgu, long n8, int ff8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    const long m = i / ff8;
    const int n = (int)(i - m * ff8) * 8;
    const u32x4 g = *reinterpret_cast<const u32x4*>(gu + (size_t)m * ld + n);
    const u32x4 u = *reinterpret_cast<const u32x4*>(gu + (size_t)m * ld + 8 * ff8 + n);
    const u32x4 d = *reinterpret_cast<const u32x4*>(dh + (size_t)m * 8 * ff8 + n);
    u32x4 og, ou;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float gl = bf16lo(g[e]), gh = bf16hi(g[e]), sl = sigmoid_(gl), sh = sigmoid_(gh);
        const float dl = bf16lo(d[e]), dhh = bf16hi(d[e]);
        og[e] = pack_bf16x2(dl * bf16lo(u[e]) * (sl + gl * sl * (1.f - sl)), dhh * bf16hi(u[e]) * (sh + gh * sh * (1.f - sh)));
        ou[e] = pack_bf16x2(dl * gl * sl, dhh * gh * sh);
    }
    *reinterpret_cast<u32x4*>(dgu + (size_t)m * ld + n) = og;
    *reinterpret_cast<u32x4*>(dgu + (size_t)m * ld + 8 * ff8 + n) = ou;
}

}  // namespace

extern "C" int i2t_rmsnorm_fwd(void* stream, const float* x, const float* w, void* y, float* y_f32, float* rstd, int M, int d,
                               float eps) {
    I2T_REQUIRE(x && w && (y || y_f32) && M > 0 && d > 0 && d % 4 == 0, "i2t_rmsnorm_fwd: bad args (d=%d must be a multiple of 4)", d);
    hipLaunchKernelGGL(rms_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, (hipStream_t)stream, x, w, (bf16_t*)y, y_f32, rstd, M, d, eps);
    I2T_CHECK_LAUNCH("i2t_rmsnorm_fwd");
    return I2T_OK;
}

extern "C" int i2t_rmsnorm_bwd(void* stream, const void* dy, int dy_is_f32, const float* x, const float* w, const float* rstd,
                               float* dx, int dx_accumulate, void* dx_bf16, float* dw, int M, int d) {
    I2T_REQUIRE(dy && x && w && rstd && dx && M > 0 && d > 0 && d % 4 == 0, "i2t_rmsnorm_bwd: bad args (d=%d must be a multiple of 4)", d);
    const int grid = (M + RMS_BWD_ROWS - 1) / RMS_BWD_ROWS;
    const int per = i2t_det() ? 1 : grid;       // deterministic mode: the dw atomics land in workgroup order
    for (int b0 = 0; b0 < grid; b0 += per) {
        if (dy_is_f32)
            hipLaunchKernelGGL(rms_bwd_kernel<true>, dim3(per), dim3(256), 0, (hipStream_t)stream, dy, x, w, rstd, dx, dx_accumulate,
                               (bf16_t*)dx_bf16, dw, M, d, b0);
        else
            hipLaunchKernelGGL(rms_bwd_kernel<false>, dim3(per), dim3(256), 0, (hipStream_t)stream, dy, x, w, rstd, dx, dx_accumulate,
                               (bf16_t*)dx_bf16, dw, M, d, b0);
    }
    I2T_CHECK_LAUNCH("i2t_rmsnorm_bwd");
    return I2T_OK;
}

extern "C" int i2t_rope(void* stream, void* x, int rs, int col0, int n_heads, int hd, const float* cos_sin, int n_positions,
                        const int* pos, const int* pos_ptr, int pos_offset, int T, int M, int inverse) {
    I2T_REQUIRE(x && cos_sin && M > 0 && n_heads > 0 && hd >= 16 && hd % 16 == 0 && rs % 8 == 0 && col0 % 8 == 0 &&
                    col0 + n_heads * hd <= rs,
                "i2t_rope: bad args (hd=%d must be a multiple of 16, heads inside the row)", hd);
    I2T_REQUIRE(pos || pos_ptr || (T > 0 && pos_offset >= 0 && pos_offset + (M < T ? M : T) <= n_positions),
                "i2t_rope: positions %d..%d outside the table of %d", pos_offset, pos_offset + T - 1, n_positions);
    const long n = (long)M * n_heads * (hd >> 4);
    hipLaunchKernelGGL(rope_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (bf16_t*)x, rs, col0,
                       n_heads, hd, cos_sin, pos, pos_ptr, pos_offset, T > 0 ? T : 1, M, inverse);
    I2T_CHECK_LAUNCH("i2t_rope");
    return I2T_OK;
}

extern "C" int i2t_swiglu_fwd(void* stream, const void* gate_up, int ld, void* h, int M, int ff) {
    I2T_REQUIRE(gate_up && h && M > 0 && ff > 0 && ff % 8 == 0 && ld >= 2 * ff && ld % 8 == 0, "i2t_swiglu_fwd: bad args (ff=%d)", ff);
    const long n8 = (long)M * (ff / 8);
    hipLaunchKernelGGL(swiglu_fwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)gate_up, ld, (bf16_t*)h, n8, ff / 8);
    I2T_CHECK_LAUNCH("i2t_swiglu_fwd");
    return I2T_OK;
}

extern "C" int i2t_swiglu_bwd(void* stream, const void* dh, const void* gate_up, int ld, void* d_gate_up, int M, int ff) {
    I2T_REQUIRE(dh && gate_up && d_gate_up && M > 0 && ff > 0 && ff % 8 == 0 && ld >= 2 * ff && ld % 8 == 0,
                "i2t_swiglu_bwd: bad args (ff=%d)", ff);
    const long n8 = (long)M * (ff / 8);
    hipLaunchKernelGGL(swiglu_bwd_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh,
                       (const bf16_t*)gate_up, ld, (bf16_t*)d_gate_up, n8, ff / 8);
    I2T_CHECK_LAUNCH("i2t_swiglu_bwd");
    return I2T_OK;
}
