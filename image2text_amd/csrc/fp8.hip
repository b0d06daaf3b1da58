// fp8 (OCP e4m3) GEMM operand path for FROZEN weights (BASELINE.json configs[4]; the reference only names fp8 as an accelerate
// precision option, training_configs/local/nano.yaml:21): C = (A8 . B8^T) * sa[m] * sb[n] on the block-scaled MFMA
// v_mfma_scale_f32_16x16x128_f8f6f4 with unit block scales (E8M0 = 127) -- that form runs 2x the bf16 MFMA rate per clock (the
// non-scaled fp8 MFMAs only run at the bf16 rate, MI355X_MICROARCH.md, Matrix cores) -- and per-ROW fp32 scales applied to the
// fp32 accumulators in the epilogue: a row of the activations (dynamic, amax / 448 per GEMM call) and a row of the weight matrix
// (quantised once per parameter version).  A frozen weight needs no dW, so both GEMMs that touch it -- y = x W^T and dx = dy W --
// take fp8 operands; W is kept in both orientations ([N][K] and [K][N]), 1 byte each.
//
// Operand map: the scaled MFMA pairs byte j of lane l in A with byte j of lane l in B (probe: tools/probes/probe_mfma_fp8.cpp --
// any k assignment works as long as both operands use the same one), so lane (r = l & 15, q = l >> 4) simply takes the 32
// consecutive bytes k = 32 q .. 32 q + 31 of row r: two ds_read_b128.  C/D map as every 16x16 MFMA (col = l & 15, row = 4 q + e).
//
// Kernel: 128 x 128 x 128-byte tiles, 4 waves (2 x 2, 64 x 64 each = 16 accumulators), register-staged double buffer, LDS rows of
// 128 B with the 16-byte chunks XOR-swizzled by (row & 7).  First version: correctness and the 2x-rate instruction; the persistent
// 256^2 / LDS-DMA structure of gemm.hip is the next step.
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(8))) int i32x8;

constexpr int F8_BM = 128, F8_BN = 128, F8_BK = 128;       // BK in bytes = elements
constexpr float E4M3_MAX = 448.0f;

// ---- row quantisation: scale[m] = amax(row) / 448 (1 for an all-zero row), q = round_to_e4m3(x / scale); K padded with zero bytes
template <bool IN_F32>
__global__ __launch_bounds__(256) void quant_rows_kernel(const void* __restrict__ x, int ld, unsigned char* __restrict__ out, int ld_out,
                                                         float* __restrict__ scale, int M, int K) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= M) return;
    auto load4 = [&](int c, float (&v)[4]) {
        if (IN_F32) {
            const f32x4 t = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(x) + (size_t)row * ld + c);
            v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
        } else {
            const u32x2 t = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(x) + (size_t)row * ld + c);
            v[0] = bf16lo(t[0]); v[1] = bf16hi(t[0]); v[2] = bf16lo(t[1]); v[3] = bf16hi(t[1]);
        }
    };
    float amax = 0.f;
    for (int c = lane * 4; c < K; c += 256) {
        float v[4];
        load4(c, v);
        amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
    }
    amax = wave_max(amax);
    const float sc = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (lane == 0) scale[row] = sc;
    for (int c = lane * 4; c < ld_out; c += 256) {
        unsigned pk = 0;
        if (c < K) {
            float v[4];
            load4(c, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(v[e] / sc, -E4M3_MAX), E4M3_MAX);      // (a true division: x / scale, as a host reference computes it)
            pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
            pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], (int)pk, true);
        }
        *reinterpret_cast<unsigned*>(out + (size_t)row * ld_out + c) = pk;
    }
}

// ---- the same in ONE pass over the row (K % 8 == 0, K <= TPR x 96): TPR threads per row keep their 16-byte chunks in registers between the
// amax reduction and the conversion (the two-pass kernel above re-reads the row and moves 8 bytes per lane and trip: 2-3x off the HBM time
// of its 3 bytes per element at the decoder widths).  TPR = 64: a wave per row (K <= 6144); TPR = 256: a workgroup per row (K <= 24576).
template <bool IN_F32, int TPR>
__global__ __launch_bounds__(256) void quant_rows1_kernel(const void* __restrict__ x, int ld, unsigned char* __restrict__ out, int ld_out,
                                                          float* __restrict__ scale, int M, int K) {
    constexpr int MAXI = 12, RPB = 256 / TPR;
    __shared__ float red[16];
    const int t = threadIdx.x % TPR, row = blockIdx.x * RPB + threadIdx.x / TPR;
    const bool live = row < M;
    f32x4 v[MAXI][2];
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = (t + TPR * i) * 8;
        v[i][0] = v[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (live && c < K) {
            if (IN_F32) {
                const float* p = reinterpret_cast<const float*>(x) + (size_t)row * ld + c;
                v[i][0] = *reinterpret_cast<const f32x4*>(p);
                v[i][1] = *reinterpret_cast<const f32x4*>(p + 4);
            } else {
                const u32x4 w = *reinterpret_cast<const u32x4*>(reinterpret_cast<const bf16_t*>(x) + (size_t)row * ld + c);
                v[i][0] = f32x4{bf16lo(w[0]), bf16hi(w[0]), bf16lo(w[1]), bf16hi(w[1])};
                v[i][1] = f32x4{bf16lo(w[2]), bf16hi(w[2]), bf16lo(w[3]), bf16hi(w[3])};
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
                amax = fmaxf(fmaxf(amax, fmaxf(fabsf(v[i][h][0]), fabsf(v[i][h][1]))), fmaxf(fabsf(v[i][h][2]), fabsf(v[i][h][3])));
        }
    }
    if (TPR == 64) amax = wave_max(amax);
    else amax = block_max(amax, red);
    const float sc = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (!live) return;
    if (t == 0) scale[row] = sc;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = (t + TPR * i) * 8;
        if (c >= ld_out) continue;                 // (ld_out % 16 == 0; chunks in [K, ld_out) are the zero padding)
        unsigned pk[2] = {0u, 0u};
        if (c < K) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                float q[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) q[e] = fminf(fmaxf(v[i][h][e] / sc, -E4M3_MAX), E4M3_MAX);
                pk[h] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], 0, false);
                pk[h] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], (int)pk[h], true);
            }
        }
        *reinterpret_cast<u32x2*>(out + (size_t)row * ld_out + c) = u32x2{pk[0], pk[1]};
    }
}

// ---- transposed quantisation of a weight W bf16 [N][K]: out8 [K][Np] with a scale per k (row of W^T): the dx = dy . W operand.
// One workgroup per 64 k-columns: pass 1 column amax over all n, pass 2 convert and store 4 n at a time per k (tiny, once per version).
__global__ __launch_bounds__(256) void quant_cols_kernel(const bf16_t* __restrict__ w, int ld, unsigned char* __restrict__ out, int ld_out,
                                                         float* __restrict__ scale, int N, int K) {
    __shared__ float red[4][64];
    const int k = blockIdx.x * 64 + (threadIdx.x & 63), part = threadIdx.x >> 6;
    float amax = 0.f;
    if (k < K)
        for (int n = part; n < N; n += 4) amax = fmaxf(amax, fabsf(bf16_to_f32(w[(size_t)n * ld + k])));
    red[part][threadIdx.x & 63] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0][threadIdx.x & 63], red[1][threadIdx.x & 63]), fmaxf(red[2][threadIdx.x & 63], red[3][threadIdx.x & 63]));
    const float sc = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (k >= K) return;
    if (part == 0) scale[k] = sc;
    for (int n0 = part * 4; n0 < ld_out; n0 += 16) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (n0 + e < N) ? fminf(fmaxf(bf16_to_f32(w[(size_t)(n0 + e) * ld + k]) / sc, -E4M3_MAX), E4M3_MAX) : 0.f;
        unsigned pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], (int)pk, true);
        *reinterpret_cast<unsigned*>(out + (size_t)k * ld_out + n0) = pk;
    }
}

// ---- producers that emit the fp8 operand themselves (frozen Llama / Qwen2 blocks, engine_llama): the row is in registers anyway, so the
// amax, the scale and the conversion ride along and the bf16 copy + the quantisation pass over it (3 bytes per element each way) disappear.
// One workgroup per row; a thread owns chunks t + 256 i.  Scales and rounding as quant_rows (from the fp32 values, not from a bf16 copy).
__device__ __forceinline__ float sigmoid_(float v) { return 1.f / (1.f + __expf(-v)); }       // (as csrc/llama.hip)
__device__ __forceinline__ unsigned f8_pack4(const f32x4& v, float sc) {
    unsigned pk = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[0] / sc, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(v[1] / sc, -E4M3_MAX), E4M3_MAX), 0, false);
    return (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(fminf(fmaxf(v[2] / sc, -E4M3_MAX), E4M3_MAX), fminf(fmaxf(v[3] / sc, -E4M3_MAX), E4M3_MAX), (int)pk, true);
}
__device__ __forceinline__ float f8_amax4(float a, const f32x4& v) {
    return fmaxf(fmaxf(a, fmaxf(fabsf(v[0]), fabsf(v[1]))), fmaxf(fabsf(v[2]), fabsf(v[3])));
}

// RMSNorm: y = w x rsqrt(mean(x^2) + eps) -> e4m3 row + scale (+ rstd for the backward); d <= 8192
__global__ __launch_bounds__(256) void rms_fwd_fp8_kernel(const float* __restrict__ x, const float* __restrict__ w, unsigned char* __restrict__ y8,
                                                          int ld8, float* __restrict__ scale, float* __restrict__ rstd_out, int d, float eps,
                                                          bf16_t* __restrict__ y16) {
    constexpr int MAXI = 8;
    __shared__ float red[16];
    const int row = blockIdx.x, t = threadIdx.x, nc = d >> 2;
    f32x4 v[MAXI];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = t + 256 * i;
        v[i] = c < nc ? reinterpret_cast<const f32x4*>(x + (size_t)row * d)[c] : f32x4{0.f, 0.f, 0.f, 0.f};
        ss += v[i][0] * v[i][0] + v[i][1] * v[i][1] + v[i][2] * v[i][2] + v[i][3] * v[i][3];
    }
    const float rs = rsqrtf(block_sum(ss, red) / d + eps);
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = t + 256 * i;
        if (c < nc) {
            const f32x4 g = reinterpret_cast<const f32x4*>(w)[c];
            v[i] = f32x4{v[i][0] * rs * g[0], v[i][1] * rs * g[1], v[i][2] * rs * g[2], v[i][3] * rs * g[3]};
            amax = f8_amax4(amax, v[i]);
            if (y16) reinterpret_cast<u32x2*>(y16 + (size_t)row * d)[c] = u32x2{pack_bf16x2(v[i][0], v[i][1]), pack_bf16x2(v[i][2], v[i][3])};
        }
    }
    amax = block_max(amax, red);
    const float sc = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (t == 0) {
        scale[row] = sc;
        if (rstd_out) rstd_out[row] = rs;
    }
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = t + 256 * i;
        if (c * 4 < ld8) *reinterpret_cast<unsigned*>(y8 + (size_t)row * ld8 + c * 4) = c < nc ? f8_pack4(v[i], sc) : 0u;
    }
}

// SwiGLU forward: h = silu(gate) up -> e4m3 row + scale; ff <= 12288.  BWD: [d gate | d up] of the fused projection -> e4m3 row (2 ff) + scale
template <bool BWD>
__global__ __launch_bounds__(256) void swiglu_fp8_kernel(const bf16_t* __restrict__ dh, const bf16_t* __restrict__ gu, int ld,
                                                         unsigned char* __restrict__ out8, int ld8, float* __restrict__ scale, int ff,
                                                         bf16_t* __restrict__ out16, int ld16) {
    constexpr int MAXI = 6;
    __shared__ float red[16];
    const int row = blockIdx.x, t = threadIdx.x, n8 = ff >> 3;
    f32x4 a[MAXI][2], b[BWD ? MAXI : 1][2];          // forward: a = h; backward: a = d gate, b = d up
    float amax = 0.f;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = t + 256 * i;
        a[i][0] = a[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (BWD) b[i][0] = b[i][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (c >= n8) continue;
        const u32x4 g = *reinterpret_cast<const u32x4*>(gu + (size_t)row * ld + c * 8);
        const u32x4 u = *reinterpret_cast<const u32x4*>(gu + (size_t)row * ld + ff + c * 8);
        u32x4 dv = {0u, 0u, 0u, 0u};
        if (BWD) dv = *reinterpret_cast<const u32x4*>(dh + (size_t)row * ff + c * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gl = bf16lo(g[e]), gh = bf16hi(g[e]), sl = sigmoid_(gl), sh = sigmoid_(gh);
            const float ul = bf16lo(u[e]), uh = bf16hi(u[e]);
            if (!BWD) {
                a[i][e >> 1][(e & 1) * 2] = gl * sl * ul;
                a[i][e >> 1][(e & 1) * 2 + 1] = gh * sh * uh;
            } else {
                const float dl = bf16lo(dv[e]), dhh = bf16hi(dv[e]);
                a[i][e >> 1][(e & 1) * 2] = dl * ul * (sl + gl * sl * (1.f - sl));
                a[i][e >> 1][(e & 1) * 2 + 1] = dhh * uh * (sh + gh * sh * (1.f - sh));
                b[i][e >> 1][(e & 1) * 2] = dl * gl * sl;
                b[i][e >> 1][(e & 1) * 2 + 1] = dhh * gh * sh;
            }
        }
        amax = f8_amax4(f8_amax4(amax, a[i][0]), a[i][1]);
        if (BWD) amax = f8_amax4(f8_amax4(amax, b[i][0]), b[i][1]);
        if (out16) {         // (an adapter's rank GEMMs want the bf16 row as well: LoRA on a frozen fp8 base)
            *reinterpret_cast<u32x4*>(out16 + (size_t)row * ld16 + c * 8) =
                u32x4{pack_bf16x2(a[i][0][0], a[i][0][1]), pack_bf16x2(a[i][0][2], a[i][0][3]), pack_bf16x2(a[i][1][0], a[i][1][1]), pack_bf16x2(a[i][1][2], a[i][1][3])};
            if (BWD) *reinterpret_cast<u32x4*>(out16 + (size_t)row * ld16 + ff + c * 8) =
                u32x4{pack_bf16x2(b[i][0][0], b[i][0][1]), pack_bf16x2(b[i][0][2], b[i][0][3]), pack_bf16x2(b[i][1][0], b[i][1][1]), pack_bf16x2(b[i][1][2], b[i][1][3])};
        }
    }
    amax = block_max(amax, red);
    const float sc = amax > 0.f ? amax / E4M3_MAX : 1.0f;
    if (t == 0) scale[row] = sc;
    const int width = BWD ? 2 * ff : ff;
#pragma unroll
    for (int i = 0; i < MAXI; ++i) {
        const int c = t + 256 * i;
        if (c < n8) {
            *reinterpret_cast<u32x2*>(out8 + (size_t)row * ld8 + c * 8) = u32x2{f8_pack4(a[i][0], sc), f8_pack4(a[i][1], sc)};
            if (BWD) *reinterpret_cast<u32x2*>(out8 + (size_t)row * ld8 + ff + c * 8) = u32x2{f8_pack4(b[i][0], sc), f8_pack4(b[i][1], sc)};
        }
    }
    for (int c = width + t * 8; c < ld8; c += 2048) *reinterpret_cast<u32x2*>(out8 + (size_t)row * ld8 + c) = u32x2{0u, 0u};      // zero padding
}

struct F8Params {
    const unsigned char* A;      // [M][lda] e4m3
    const unsigned char* B;      // [N][ldb] e4m3
    const float* sa;             // [M]
    const float* sb;             // [N]
    void* C;
    const float* bias;           // [N] or null
    const float* residual;       // f32 [M][ldr] or null
    int M, N, K, lda, ldb, ldc, ldr, c_is_f32, act;
};

__device__ __forceinline__ void f8_stage_load(u32x4 (&r)[4], const unsigned char* base, int ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u, row = c >> 3, kc = c & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + row < nrows && k0 + kc * 16 < K) v = *reinterpret_cast<const u32x4*>(base + (size_t)(row0 + row) * ld + k0 + kc * 16);
        r[u] = v;
    }
}
__device__ __forceinline__ void f8_stage_store(const u32x4 (&r)[4], unsigned char* lds, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int c = tid + 256 * u, row = c >> 3, kc = c & 7;
        *reinterpret_cast<u32x4*>(lds + row * 128 + ((kc ^ (row & 7)) << 4)) = r[u];
    }
}
__device__ __forceinline__ i32x8 f8_frag(const unsigned char* lds, int row0, int lane) {
    const int r = row0 + (lane & 15), q = lane >> 4;
    const u32x4 lo = *reinterpret_cast<const u32x4*>(lds + r * 128 + (((2 * q) ^ (r & 7)) << 4));
    const u32x4 hi = *reinterpret_cast<const u32x4*>(lds + r * 128 + (((2 * q + 1) ^ (r & 7)) << 4));
    return i32x8{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
}

__global__ __launch_bounds__(256, 2) void gemm_fp8_kernel(F8Params p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2][2][F8_BM * 128];      // [buffer][A | B][row][128 B]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave >> 1, wn = wave & 1;
    // XCD-aware bijective remap of the 1-D grid, N fastest inside groups of 8 column tiles (the weight panel stays in L2)
    const int tiles_n = (p.N + F8_BN - 1) / F8_BN, nwg = gridDim.x, bid = blockIdx.x;
    const int xcd = bid & 7, q8 = nwg >> 3, r8 = nwg & 7;
    const int swz = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + (bid >> 3);
    const int tile_m = swz / tiles_n, tile_n = swz % tiles_n;
    const int m0 = tile_m * F8_BM, n0 = tile_n * F8_BN;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int nk = (p.K + F8_BK - 1) / F8_BK;
    u32x4 ra[4], rb[4];
    f8_stage_load(ra, p.A, p.lda, m0, p.M, 0, p.K, tid);
    f8_stage_load(rb, p.B, p.ldb, n0, p.N, 0, p.K, tid);
    f8_stage_store(ra, smem[0][0], tid);
    f8_stage_store(rb, smem[0][1], tid);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nk) {
            f8_stage_load(ra, p.A, p.lda, m0, p.M, (kt + 1) * F8_BK, p.K, tid);
            f8_stage_load(rb, p.B, p.ldb, n0, p.N, (kt + 1) * F8_BK, p.K, tid);
        }
        i32x8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = f8_frag(smem[cur][1], wn * 64 + j * 16, lane);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const i32x8 af = f8_frag(smem[cur][0], wm * 64 + i * 16, lane);
#pragma unroll
            for (int j = 0; j < 4; ++j)      // swapped issue (B rows as the MFMA's A operand): D[n][m] -> a lane owns 4 consecutive n of one row m
                acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(bf[j], af, acc[i][j], 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        }
        if (kt + 1 < nk) {
            f8_stage_store(ra, smem[cur ^ 1][0], tid);
            f8_stage_store(rb, smem[cur ^ 1][1], tid);
        }
        __syncthreads();
    }
    // epilogue: lane (li = lane & 15, g = lane >> 4) holds C[m = .. + li][n = .. + 4 g + e]: 4 consecutive columns -> 8 / 16-byte stores
    const int li = lane & 15, g = lane >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + wm * 64 + i * 16 + li;
        if (m >= p.M) continue;
        const float sa = p.sa[m];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int n = n0 + wn * 64 + j * 16 + 4 * g;
            if (n >= p.N) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int ne = min(n + e, p.N - 1);
                v[e] = acc[i][j][e] * sa * p.sb[ne] + (p.bias ? p.bias[ne] : 0.f);
                if (p.act == I2T_ACT_GELU_ERF) v[e] = gelu_erf(v[e]);
                else if (p.act == I2T_ACT_GELU) v[e] = gelu_tanh(v[e]);
            }
            if (n + 3 < p.N) {
                if (p.residual) {
                    const f32x4 r = *reinterpret_cast<const f32x4*>(p.residual + (size_t)m * p.ldr + n);
                    v[0] += r[0]; v[1] += r[1]; v[2] += r[2]; v[3] += r[3];
                }
                if (p.c_is_f32) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = f32x4{v[0], v[1], v[2], v[3]};
                else *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + (size_t)m * p.ldc + n) = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            } else {
                for (int e = 0; e < 4 && n + e < p.N; ++e) {
                    const float o = v[e] + (p.residual ? p.residual[(size_t)m * p.ldr + n + e] : 0.f);
                    if (p.c_is_f32) reinterpret_cast<float*>(p.C)[(size_t)m * p.ldc + n + e] = o;
                    else reinterpret_cast<bf16_t*>(p.C)[(size_t)m * p.ldc + n + e] = f32_to_bf16(o);
                }
            }
        }
    }
}

}  // namespace

extern "C" int i2t_quant_rows_fp8(void* stream, const void* x, int x_is_f32, int ld, void* out, int ld_out, float* scale, int M, int K) {
    I2T_REQUIRE(x && out && scale && M > 0 && K > 0 && K % 4 == 0 && ld % 4 == 0 && ld_out % 16 == 0 && ld_out >= K && ALIGNED16(x) && ALIGNED16(out),
                "i2t_quant_rows_fp8: bad args (K=%d %% 4, ld_out=%d %% 16 and >= K)", K, ld_out);
    hipStream_t s_ = (hipStream_t)stream;
    unsigned char* o_ = (unsigned char*)out;
    // (K < 2048: a row is a fraction of a wave's 12 register chunks -- the two-pass kernel is the faster one there: 36 vs 50 us at K = 768)
    const bool one_pass = K % 8 == 0 && K >= 2048 && ld % 8 == 0 && ld_out <= 24576 && !(getenv("I2T_FP8_QUANT1") && getenv("I2T_FP8_QUANT1")[0] == '0');
    if (one_pass && ld_out <= 6144) {          // a wave per row
        if (x_is_f32) hipLaunchKernelGGL((quant_rows1_kernel<true, 64>), dim3((M + 3) / 4), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
        else hipLaunchKernelGGL((quant_rows1_kernel<false, 64>), dim3((M + 3) / 4), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
    } else if (one_pass) {                     // a workgroup per row
        if (x_is_f32) hipLaunchKernelGGL((quant_rows1_kernel<true, 256>), dim3(M), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
        else hipLaunchKernelGGL((quant_rows1_kernel<false, 256>), dim3(M), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
    } else if (x_is_f32) hipLaunchKernelGGL(quant_rows_kernel<true>, dim3((M + 3) / 4), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
    else hipLaunchKernelGGL(quant_rows_kernel<false>, dim3((M + 3) / 4), dim3(256), 0, s_, x, ld, o_, ld_out, scale, M, K);
    I2T_CHECK_LAUNCH("i2t_quant_rows_fp8");
    return I2T_OK;
}

extern "C" int i2t_rmsnorm_fwd_fp8(void* stream, const float* x, const float* w, void* y8, int ld8, float* scale, float* rstd, int M, int d, float eps,
                                   void* y_bf16) {
    I2T_REQUIRE(x && w && y8 && scale && M > 0 && d > 0 && d % 4 == 0 && d <= 8192 && ld8 % 16 == 0 && ld8 >= d && ld8 <= 8192 && ALIGNED16(x) && ALIGNED16(y8),
                "i2t_rmsnorm_fwd_fp8: bad args (d=%d <= 8192, ld8=%d %% 16)", d, ld8);
    hipLaunchKernelGGL(rms_fwd_fp8_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, x, w, (unsigned char*)y8, ld8, scale, rstd, d, eps, (bf16_t*)y_bf16);
    I2T_CHECK_LAUNCH("i2t_rmsnorm_fwd_fp8");
    return I2T_OK;
}

extern "C" int i2t_swiglu_fwd_fp8(void* stream, const void* gate_up, int ld, void* h8, int ld8, float* scale, int M, int ff, void* h_bf16) {
    I2T_REQUIRE(gate_up && h8 && scale && M > 0 && ff > 0 && ff % 8 == 0 && ff <= 12288 && ld >= 2 * ff && ld % 8 == 0 && ld8 % 16 == 0 && ld8 >= ff &&
                    ALIGNED16(gate_up) && ALIGNED16(h8), "i2t_swiglu_fwd_fp8: bad args (ff=%d <= 12288)", ff);
    hipLaunchKernelGGL(swiglu_fp8_kernel<false>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)nullptr, (const bf16_t*)gate_up, ld,
                       (unsigned char*)h8, ld8, scale, ff, (bf16_t*)h_bf16, ff);
    I2T_CHECK_LAUNCH("i2t_swiglu_fwd_fp8");
    return I2T_OK;
}

extern "C" int i2t_swiglu_bwd_fp8(void* stream, const void* dh, const void* gate_up, int ld, void* dgu8, int ld8, float* scale, int M, int ff,
                                  void* dgu_bf16) {
    I2T_REQUIRE(dh && gate_up && dgu8 && scale && M > 0 && ff > 0 && ff % 8 == 0 && ff <= 12288 && ld >= 2 * ff && ld % 8 == 0 && ld8 % 16 == 0 &&
                    ld8 >= 2 * ff && ALIGNED16(gate_up) && ALIGNED16(dh) && ALIGNED16(dgu8), "i2t_swiglu_bwd_fp8: bad args (ff=%d <= 12288)", ff);
    hipLaunchKernelGGL(swiglu_fp8_kernel<true>, dim3(M), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dh, (const bf16_t*)gate_up, ld,
                       (unsigned char*)dgu8, ld8, scale, ff, (bf16_t*)dgu_bf16, 2 * ff);
    I2T_CHECK_LAUNCH("i2t_swiglu_bwd_fp8");
    return I2T_OK;
}

extern "C" int i2t_quant_cols_fp8(void* stream, const void* w, int ld, void* out, int ld_out, float* scale, int N, int K) {
    I2T_REQUIRE(w && out && scale && N > 0 && K > 0 && ld_out % 16 == 0 && ld_out >= N && ALIGNED16(out), "i2t_quant_cols_fp8: bad args");
    hipLaunchKernelGGL(quant_cols_kernel, dim3((K + 63) / 64), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)w, ld, (unsigned char*)out, ld_out, scale, N, K);
    I2T_CHECK_LAUNCH("i2t_quant_cols_fp8");
    return I2T_OK;
}

extern "C" int i2t_gemm_fp8(void* stream, const void* A8, int lda, const float* sa, const void* B8, int ldb, const float* sb, void* C, int ldc,
                            int c_is_f32, int M, int N, int K, const float* bias, int act, const float* residual, int ldr) {
    I2T_REQUIRE(A8 && B8 && sa && sb && C && M > 0 && N > 0 && K > 0, "i2t_gemm_fp8: bad args");
    I2T_REQUIRE(act == I2T_ACT_NONE || act == I2T_ACT_GELU || act == I2T_ACT_GELU_ERF, "i2t_gemm_fp8: act %d (0 none | 1 GELU tanh | 3 GELU erf)", act);
    I2T_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K && ALIGNED16(A8) && ALIGNED16(B8),
                "i2t_gemm_fp8: operands must be 16-byte aligned with leading dimensions %% 16 == 0 and >= K (zero-padded rows)");
    I2T_REQUIRE(ldc >= N && ldc % 4 == 0 && ALIGNED16(C) && (!residual || (ldr % 4 == 0 && ALIGNED16(residual))), "i2t_gemm_fp8: C / residual alignment");
    // enough 256 x 256 tiles and K % 256 == 0: the persistent LDS-DMA kernel of gemm.hip on fp8 operands (same results up to summation order)
    if (i2t_g256_fp8_try((hipStream_t)stream, A8, lda, sa, B8, ldb, sb, C, ldc, c_is_f32, M, N, K, bias, act, residual, ldr)) {
        I2T_CHECK_LAUNCH("i2t_gemm_fp8(256)");
        return I2T_OK;
    }
    F8Params p{(const unsigned char*)A8, (const unsigned char*)B8, sa, sb, C, bias, residual, M, N, K, lda, ldb, ldc, ldr, c_is_f32, act};
    const long tiles = (long)((M + F8_BM - 1) / F8_BM) * ((N + F8_BN - 1) / F8_BN);
    I2T_REQUIRE(tiles < 2147483647L, "i2t_gemm_fp8: grid too large");
    hipLaunchKernelGGL(gemm_fp8_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
    I2T_CHECK_LAUNCH("i2t_gemm_fp8");
    return I2T_OK;
}
