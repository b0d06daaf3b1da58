// Grouped small GEMMs for AdvancedPositionalBiasMLP (reference models/layers.py:617-638; decoder.py:231-232): every position p of
// the sequence owns a private MLP (d -> g1 -> .. -> d), so one layer of that module is T independent GEMMs -- group g = position,
// M_g = the sequences that reach position g (ragged with packed caption rows), N / K <= 1024 and mostly 32..128.  Rows arrive
// position-major (the caller gathers them once), weights of consecutive positions sit a constant stride apart in the parameter
// arena (named_parameters order), so a group is (row segment, base + g * stride): no pointer tables.
//
//   mode 0 (forward)   Y[rows, N]  = act(X[rows, K] . W_g[N, K]^T + b_g) (+ residual)       pre-activation kept in aux_out
//   mode 1 (backward)  dX[rows, K] = (dY[rows, N] . W_g[N, K]) * gelu'(aux_in) (+ residual)
//   mode 2 (backward)  dW_g[N, K] (+)= dY_g[rows, N]^T . X_g[rows, K]
//
// One workgroup = 4 waves = one 64 x 64 output tile; both operands are staged through LDS as [64][64 reduction] tiles (stride 144 B:
// conflict-free 16-byte row reads), transposing on the way in when the reduction index is not the contiguous one; MFMA 16x16x32
// bf16 with the operands swapped so that a lane ends up holding 4 consecutive output columns of one row (vector stores).
// These are latency-sized problems (a few MFLOP per group): the point is one launch per layer instead of one per position.
#include "common.h"

namespace {

constexpr int GS = 72;                          // LDS row stride in elements (144 B)

struct GGemm {
    const bf16_t* A; int lda;
    const bf16_t* B; int ldb; long b_gs;
    void* C; int ldc; long c_gs; int c_is_f32;
    const float* bias; long bias_gs;
    const bf16_t* aux_in; bf16_t* aux_out; int ld_aux;
    const float* residual; int ldr;
    const int* seg;                             // [n_groups + 1] row offsets, or null: every group owns rows [0, rows_fixed)
    const int* group_ptr;                       // decode step: the (single) group index is read from the device
    int rows_fixed, group0;
    int I, J, R;                                // tile-space extents: output rows / output cols / reduction (ragged one = per group)
    int act, accumulate;
};

// tile[p][q] = src[(row0 + p) * ld + col0 + q]  for p < 64, q < 64; rows >= nrows / cols >= ncols read as zero
__device__ __forceinline__ void stage(bf16_t* lds, const bf16_t* src, int ld, int row0, int nrows, int col0, int ncols, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + 256 * u, p = c >> 3, q = (c & 7) * 8;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + p < nrows && col0 + q < ncols) v = *reinterpret_cast<const u32x4*>(src + (size_t)(row0 + p) * ld + col0 + q);
        *reinterpret_cast<u32x4*>(lds + p * GS + q) = v;
    }
}
// tile[q][p] = src[(row0 + p) * ld + col0 + q]: the same 16-byte global loads, transposed on the LDS side
__device__ __forceinline__ void stage_t(bf16_t* lds, const bf16_t* src, int ld, int row0, int nrows, int col0, int ncols, int tid) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int c = tid + 256 * u, p = c & 63, q = (c >> 6) * 8;       // consecutive lanes -> consecutive p: LDS writes of one q spread over banks
        u32x4 v = {0u, 0u, 0u, 0u};
        if (row0 + p < nrows && col0 + q < ncols) v = *reinterpret_cast<const u32x4*>(src + (size_t)(row0 + p) * ld + col0 + q);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            lds[(q + 2 * e) * GS + p] = (bf16_t)(v[e] & 0xffffu);
            lds[(q + 2 * e + 1) * GS + p] = (bf16_t)(v[e] >> 16);
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void grouped_gemm_kernel(GGemm p) {
    __shared__ __attribute__((aligned(16))) bf16_t at[64 * GS];
    __shared__ __attribute__((aligned(16))) bf16_t bt[64 * GS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, g = lane >> 4, li = lane & 15;
    const int grp = p.group_ptr ? 0 : (int)blockIdx.z;
    const int wgrp = (p.group_ptr ? *p.group_ptr : grp) + p.group0;      // which position's weights
    const int row_lo = p.seg ? p.seg[grp] : 0;
    const int rows = p.seg ? p.seg[grp + 1] - row_lo : p.rows_fixed;
    const int i0 = blockIdx.x * 64, j0 = blockIdx.y * 64;
    const int I = MODE == 2 ? p.I : rows, R = MODE == 2 ? rows : p.R;
    if (i0 >= I) return;                                                 // workgroup-uniform
    const bf16_t* Wg = p.B + (size_t)wgrp * p.b_gs;
    f32x4 acc[4];
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) acc[jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int r0 = 0; r0 < R; r0 += 64) {
        __syncthreads();
        if (MODE == 0) {            // A = X rows (reduction k contiguous), B = W_g rows n (k contiguous)
            stage(at, p.A + (size_t)row_lo * p.lda, p.lda, i0, rows, r0, p.R, tid);
            stage(bt, Wg, p.ldb, j0, p.J, r0, p.R, tid);
        } else if (MODE == 1) {     // A = dY rows (reduction n contiguous), B = W_g[n][k]: k contiguous -> transpose
            stage(at, p.A + (size_t)row_lo * p.lda, p.lda, i0, rows, r0, p.R, tid);
            stage_t(bt, Wg, p.ldb, r0, p.R, j0, p.J, tid);
        } else {                    // A(i = n, r = row) = dY[row][n], B(r = row, j = k) = X[row][k]: both transposed
            stage_t(at, p.A + (size_t)row_lo * p.lda, p.lda, r0, rows, i0, p.I, tid);
            stage_t(bt, p.B + (size_t)row_lo * p.ldb, p.ldb, r0, rows, j0, p.J, tid);
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const bf16x8 af = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(at + (16 * w + li) * GS + ks * 32 + 8 * g));
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) {
                const bf16x8 bf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(bt + (16 * jt + li) * GS + ks * 32 + 8 * g));
                acc[jt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bf, af, acc[jt], 0, 0, 0);     // D[j = 4 g + r][i = li]
            }
        }
    }
    const int i = i0 + 16 * w + li;
    if (i >= I) return;
    const size_t crow = MODE == 2 ? (size_t)wgrp * p.c_gs + (size_t)i * p.ldc : (size_t)(row_lo + i) * p.ldc;
    const size_t xrow = (size_t)(row_lo + i);                            // aux / residual row (modes 0, 1)
#pragma unroll
    for (int jt = 0; jt < 4; ++jt) {
        const int j = j0 + 16 * jt + 4 * g;
        if (j >= p.J) continue;
        f32x4 v = acc[jt];
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + (size_t)wgrp * p.bias_gs + j);
        if (p.act == I2T_ACT_GELU) {
            if (p.aux_out) {
                u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                *reinterpret_cast<u32x2*>(p.aux_out + xrow * p.ld_aux + j) = pk;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = gelu_tanh(v[e]);
        } else if (p.act == I2T_ACT_DGELU) {
            const u32x2 a = *reinterpret_cast<const u32x2*>(p.aux_in + xrow * p.ld_aux + j);
            v[0] *= gelu_tanh_grad(bf16lo(a[0])); v[1] *= gelu_tanh_grad(bf16hi(a[0]));
            v[2] *= gelu_tanh_grad(bf16lo(a[1])); v[3] *= gelu_tanh_grad(bf16hi(a[1]));
        }
        if (p.residual) v += *reinterpret_cast<const f32x4*>(p.residual + xrow * p.ldr + j);
        if (p.c_is_f32) {
            float* c = reinterpret_cast<float*>(p.C) + crow + j;
            if (p.accumulate) v += *reinterpret_cast<const f32x4*>(c);
            *reinterpret_cast<f32x4*>(c) = v;
        } else {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(p.C) + crow + j) = pk;
        }
    }
}

// out[(g + group0) * out_gs + n] += sum over the rows of group g of X[row][n]   (bias gradients of one layer, all positions)
__global__ __launch_bounds__(256) void grouped_colsum_kernel(const bf16_t* __restrict__ X, int ld, const int* __restrict__ seg,
                                                             float* __restrict__ out, long out_gs, int group0, int N) {
    __shared__ float red[4][64];
    const int grp = blockIdx.y, n = blockIdx.x * 64 + (threadIdx.x & 63), ph = threadIdx.x >> 6;
    const int lo = seg[grp], hi = seg[grp + 1];
    float a = 0.f;
    if (n < N)
        for (int r = lo + ph; r < hi; r += 4) a += bf16_to_f32(X[(size_t)r * ld + n]);
    red[ph][threadIdx.x & 63] = a;
    __syncthreads();
    if (ph == 0 && n < N) out[(size_t)(grp + group0) * out_gs + n] += (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

}  // namespace

extern "C" int i2t_grouped_gemm(void* stream, int mode, const void* A, int lda, const void* B, int ldb, long b_group_stride, void* C, int ldc,
                                long c_group_stride, int c_is_f32, const float* bias, long bias_group_stride, int act, const void* aux_in,
                                void* aux_out, int ld_aux, const float* residual, int ldr, int accumulate, const int* seg, int n_groups,
                                int max_rows, const int* group_ptr, int group0, int N, int K) {
    I2T_REQUIRE(mode >= 0 && mode <= 2 && A && B && C && n_groups > 0 && max_rows > 0 && N > 0 && K > 0, "i2t_grouped_gemm: bad args");
    I2T_REQUIRE(N % 32 == 0 && K % 32 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0 && ALIGNED16(A) && ALIGNED16(B) && ALIGNED16(C),
                "i2t_grouped_gemm: N, K must be multiples of 32 and operands 16-byte aligned (N=%d K=%d)", N, K);
    I2T_REQUIRE(b_group_stride % 8 == 0 && c_group_stride % 4 == 0 && bias_group_stride % 4 == 0, "i2t_grouped_gemm: group strides must keep 16-byte alignment");
    I2T_REQUIRE(seg || n_groups == 1, "i2t_grouped_gemm: several groups need the row segment table");
    I2T_REQUIRE(!group_ptr || (n_groups == 1 && mode != 2), "i2t_grouped_gemm: a device-side group index selects ONE group (forward / dX)");
    I2T_REQUIRE(act == I2T_ACT_NONE || (act == I2T_ACT_GELU && mode == 0) || (act == I2T_ACT_DGELU && mode == 1 && aux_in),
                "i2t_grouped_gemm: activation %d not available in mode %d", act, mode);
    I2T_REQUIRE(!(accumulate && !c_is_f32) && !(mode == 2 && (bias || residual || act)), "i2t_grouped_gemm: unsupported epilogue");
    GGemm p{};
    p.A = (const bf16_t*)A; p.lda = lda; p.B = (const bf16_t*)B; p.ldb = ldb; p.b_gs = mode == 2 ? 0 : b_group_stride;
    p.C = C; p.ldc = ldc; p.c_gs = c_group_stride; p.c_is_f32 = c_is_f32; p.bias = bias; p.bias_gs = bias_group_stride;
    p.aux_in = (const bf16_t*)aux_in; p.aux_out = (bf16_t*)aux_out; p.ld_aux = ld_aux; p.residual = residual; p.ldr = ldr;
    p.seg = seg; p.group_ptr = group_ptr; p.rows_fixed = max_rows; p.group0 = group0; p.act = act; p.accumulate = accumulate;
    hipStream_t s = (hipStream_t)stream;
    const int row_tiles = (max_rows + 63) / 64;
    if (mode == 0) {
        p.I = max_rows; p.J = N; p.R = K;
        hipLaunchKernelGGL(grouped_gemm_kernel<0>, dim3(row_tiles, (N + 63) / 64, n_groups), dim3(256), 0, s, p);
    } else if (mode == 1) {
        p.I = max_rows; p.J = K; p.R = N;
        hipLaunchKernelGGL(grouped_gemm_kernel<1>, dim3(row_tiles, (K + 63) / 64, n_groups), dim3(256), 0, s, p);
    } else {
        p.I = N; p.J = K; p.R = max_rows;
        hipLaunchKernelGGL(grouped_gemm_kernel<2>, dim3((N + 63) / 64, (K + 63) / 64, n_groups), dim3(256), 0, s, p);
    }
    I2T_CHECK_LAUNCH("i2t_grouped_gemm");
    return I2T_OK;
}

extern "C" int i2t_grouped_colsum(void* stream, const void* X, int ld, const int* seg, int n_groups, float* out, long out_group_stride,
                                  int group0, int N) {
    I2T_REQUIRE(X && seg && out && n_groups > 0 && N > 0, "i2t_grouped_colsum: bad args");
    hipLaunchKernelGGL(grouped_colsum_kernel, dim3((N + 63) / 64, n_groups), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)X, ld, seg, out,
                       out_group_stride, group0, N);
    I2T_CHECK_LAUNCH("i2t_grouped_colsum");
    return I2T_OK;
}
