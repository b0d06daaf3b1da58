// ConvMLP feature extractor (6x6 'same' convolutions over 3 -> 8 -> 16 -> 32 channels at full resolution).
//
// v1 structure: direct convolution on the vector ALUs.  A workgroup owns a 32 x 32 output tile of one image and
// COUT_T output channels; the input patch (37 x 37 with halo) is staged into LDS four input channels at a time
// (GELU applied while staging when the producer stored pre-activations), each thread computes 4 consecutive pixels
// x COUT_T channels.  The weights are repacked to [cin][ky][kx][cout] so that the COUT_T weights of one tap are a
// wave-uniform contiguous run: the compiler fetches them with scalar loads and feeds them to v_fma as SGPR
// operands, i.e. the inner loop issues no LDS or vector-memory traffic for weights.
//   backward-data  = the same kernel on the flipped/transposed weights with the padding mirrored (3 before / 2
//                    after for k = 6), fused with the multiplication by GELU'(pre-activation);
//   backward-weight: a thread owns a (cin, ky, kx) tap and all COUT accumulators, the dY values of a pixel are
//                    wave-uniform (scalar loads), one atomic per weight per workgroup at the end.
#include "common.h"

namespace {

constexpr int TILE = 32;
constexpr int CI_CHUNK = 4;
constexpr int KMAX = 8;

// wr[ci][ky][kx][co] = w[co][ci][ky][kx]                          (forward)
// wr[co][ky][kx][ci] = w[co][ci][k-1-ky][k-1-kx]                  (backward-data: roles of ci/co swapped)
__global__ void repack_weights_kernel(const float* __restrict__ w, float* __restrict__ wr, int Cout, int Cin, int k,
                                      int flip) {
    const int n = Cout * Cin * k * k;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        int kx = i % k, ky = (i / k) % k, ci = (i / (k * k)) % Cin, co = i / (k * k * Cin);
        if (!flip) wr[((ci * k + ky) * k + kx) * Cout + co] = w[i];
        else wr[((co * k + (k - 1 - ky)) * k + (k - 1 - kx)) * Cin + ci] = w[i];
    }
}

template <int COUT_T, int K, bool IN_F32, bool IN_GELU, bool DGELU_OUT>
__global__ __launch_bounds__(256) void conv_direct_kernel(const void* __restrict__ xin, const float* __restrict__ wr,
                                                          const float* __restrict__ bias, bf16_t* __restrict__ y,
                                                          const bf16_t* __restrict__ pre, int Cin, int Cout, int H, int W,
                                                          int pad_before, int cout_groups) {
    constexpr int PW = TILE + K - 1;                       // patch width/height
    __shared__ float patch[CI_CHUNK][PW][PW + 1];
    const int tid = threadIdx.x;
    const int tx = tid & 7, ty = tid >> 3;
    const int x0 = blockIdx.x * TILE, y0 = blockIdx.y * TILE;
    const int b = blockIdx.z / cout_groups, cg = blockIdx.z % cout_groups;
    const int co0 = cg * COUT_T;
    const size_t plane = (size_t)H * W;

    float acc[4][COUT_T];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < COUT_T; ++c) acc[p][c] = 0.f;

    for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
        __syncthreads();
        const int nci = min(CI_CHUNK, Cin - ci0);
        for (int i = tid; i < nci * PW * PW; i += 256) {
            int c = i / (PW * PW), rem = i % (PW * PW);
            int py = rem / PW, px = rem % PW;
            int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                size_t off = ((size_t)b * Cin + ci0 + c) * plane + (size_t)gy * W + gx;
                v = IN_F32 ? reinterpret_cast<const float*>(xin)[off] : bf16_to_f32(reinterpret_cast<const bf16_t*>(xin)[off]);
                if (IN_GELU) v = gelu_tanh(v);
            }
            patch[c][py][px] = v;
        }
        __syncthreads();
        for (int c = 0; c < nci; ++c) {
            const float* wc = wr + (size_t)(ci0 + c) * K * K * Cout + co0;
#pragma unroll
            for (int ky = 0; ky < K; ++ky) {
                float in[4 + K - 1];
#pragma unroll
                for (int j = 0; j < 4 + K - 1; ++j) in[j] = patch[c][ty + ky][4 * tx + j];
#pragma unroll
                for (int kx = 0; kx < K; ++kx) {
                    const float* wt = wc + (ky * K + kx) * Cout;    // wave-uniform address -> scalar loads
#pragma unroll
                    for (int co = 0; co < COUT_T; ++co) {
                        const float wv = wt[co];
#pragma unroll
                        for (int p = 0; p < 4; ++p) acc[p][co] = fmaf(in[p + kx], wv, acc[p][co]);
                    }
                }
            }
        }
    }
    const int oy = y0 + ty, ox = x0 + 4 * tx;
    if (oy >= H || ox >= W) return;
#pragma unroll
    for (int co = 0; co < COUT_T; ++co) {
        const int c = co0 + co;
        if (c >= Cout) break;
        const float bv = bias ? bias[c] : 0.f;
        const size_t off = ((size_t)b * Cout + c) * plane + (size_t)oy * W + ox;
        float v[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) v[p] = acc[p][co] + bv;
        if (DGELU_OUT) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                if (ox + p < W) v[p] *= gelu_tanh_grad(bf16_to_f32(pre[off + p]));
        }
        if (ox + 3 < W && ((off & 3) == 0)) {
            u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
            *reinterpret_cast<u32x2*>(y + off) = pk;
        } else {
            for (int p = 0; p < 4 && ox + p < W; ++p) y[off + p] = f32_to_bf16(v[p]);
        }
    }
}

// dw[co][ci][ky][kx] += sum over a 32x32 tile of dy[co][y][x] * act[ci][y+ky-pad][x+kx-pad]; db[co] += sum dy
template <int COUT, int K, bool IN_F32, bool IN_GELU>
__global__ __launch_bounds__(256) void conv_bwd_weight_kernel(const bf16_t* __restrict__ dy, const void* __restrict__ xin,
                                                              float* __restrict__ dw, float* __restrict__ db, int Cin,
                                                              int H, int W, int pad_before, int bx0, int by0, int bz0) {
    constexpr int PW = TILE + K - 1;
    __shared__ float patch[CI_CHUNK][PW][PW + 1];
    __shared__ __attribute__((aligned(16))) float dyt[TILE * TILE][COUT];   // channel-fastest: 4 channels per LDS broadcast read
    const int tid = threadIdx.x;
    const int x0 = (blockIdx.x + bx0) * TILE, y0 = (blockIdx.y + by0) * TILE, b = blockIdx.z + bz0;      // (offsets: deterministic mode)
    const size_t plane = (size_t)H * W;
    const int taps_per_chunk = CI_CHUNK * K * K;             // 144 for k=6
    // stage dY tile (all COUT channels) once
    for (int i = tid; i < COUT * TILE * TILE; i += 256) {
        int c = i / (TILE * TILE), rem = i % (TILE * TILE);
        int py = rem / TILE, px = rem % TILE;
        int gy = y0 + py, gx = x0 + px;
        float v = 0.f;
        if (gy < H && gx < W) v = bf16_to_f32(dy[((size_t)b * COUT + c) * plane + (size_t)gy * W + gx]);
        dyt[rem][c] = v;
    }
    if (db) {
        __syncthreads();
        if (tid < COUT) {
            float s = 0.f;
            for (int p = 0; p < TILE * TILE; ++p) s += dyt[p][tid];
            atomicAdd(db + tid, s);
        }
    }
    for (int ci0 = 0; ci0 < Cin; ci0 += CI_CHUNK) {
        __syncthreads();
        const int nci = min(CI_CHUNK, Cin - ci0);
        for (int i = tid; i < nci * PW * PW; i += 256) {
            int c = i / (PW * PW), rem = i % (PW * PW);
            int py = rem / PW, px = rem % PW;
            int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
            float v = 0.f;
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                size_t off = ((size_t)b * Cin + ci0 + c) * plane + (size_t)gy * W + gx;
                v = IN_F32 ? reinterpret_cast<const float*>(xin)[off] : bf16_to_f32(reinterpret_cast<const bf16_t*>(xin)[off]);
                if (IN_GELU) v = gelu_tanh(v);
            }
            patch[c][py][px] = v;
        }
        __syncthreads();
        if (tid < nci * K * K) {
            const int c = tid / (K * K), ky = (tid / K) % K, kx = tid % K;
            float acc[COUT];
#pragma unroll
            for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
            for (int py = 0; py < TILE; ++py) {
                for (int px = 0; px < TILE; ++px) {
                    const float a = patch[c][py + ky][px + kx];
                    const f32x4* dv = reinterpret_cast<const f32x4*>(dyt[py * TILE + px]);   // LDS broadcast reads
#pragma unroll
                    for (int c4 = 0; c4 < COUT / 4; ++c4) {
                        const f32x4 d = dv[c4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) acc[c4 * 4 + e] = fmaf(a, d[e], acc[c4 * 4 + e]);
                    }
                }
            }
#pragma unroll
            for (int co = 0; co < COUT; ++co)
                atomicAdd(dw + (((size_t)co * Cin + ci0 + c) * K + ky) * K + kx, acc[co]);
        }
        (void)taps_per_chunk;
    }
}

template <int COUT_T, bool IN_F32, bool IN_GELU, bool DGELU>
int launch_direct(hipStream_t s, const void* x, const float* wr, const float* bias, bf16_t* y, const bf16_t* pre, int B,
                  int Cin, int Cout, int H, int W, int k, int pad_before) {
    const int groups = (Cout + COUT_T - 1) / COUT_T;
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B * groups);
    if (k == 6)
        hipLaunchKernelGGL((conv_direct_kernel<COUT_T, 6, IN_F32, IN_GELU, DGELU>), grid, dim3(256), 0, s, x, wr, bias, y, pre,
                           Cin, Cout, H, W, pad_before, groups);
    else if (k == 4)
        hipLaunchKernelGGL((conv_direct_kernel<COUT_T, 4, IN_F32, IN_GELU, DGELU>), grid, dim3(256), 0, s, x, wr, bias, y, pre,
                           Cin, Cout, H, W, pad_before, groups);
    else {
        i2t_set_error("conv: kernel size %d unsupported (4 or 6)", k);
        return I2T_EINVAL;
    }
    return I2T_OK;
}

template <bool IN_F32, bool IN_GELU, bool DGELU>
int dispatch_cout(hipStream_t s, const void* x, const float* wr, const float* bias, bf16_t* y, const bf16_t* pre, int B,
                  int Cin, int Cout, int H, int W, int k, int pad_before) {
    if (Cout <= 4) return launch_direct<4, IN_F32, IN_GELU, DGELU>(s, x, wr, bias, y, pre, B, Cin, Cout, H, W, k, pad_before);
    if (Cout <= 8) return launch_direct<8, IN_F32, IN_GELU, DGELU>(s, x, wr, bias, y, pre, B, Cin, Cout, H, W, k, pad_before);
    return launch_direct<16, IN_F32, IN_GELU, DGELU>(s, x, wr, bias, y, pre, B, Cin, Cout, H, W, k, pad_before);
}

}  // namespace

extern "C" int i2t_conv_fwd(void* stream, const void* x, int in_is_f32, int in_gelu, const float* w, const float* bias,
                            void* y, float* w_ws, int B, int Cin, int Cout, int H, int W, int k) {
    I2T_REQUIRE(x && w && y && w_ws && B > 0 && Cin > 0 && Cout > 0, "i2t_conv_fwd: bad args");
    I2T_REQUIRE(k == 4 || k == 6, "i2t_conv_fwd: kernel size %d unsupported (4 or 6)", k);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(repack_weights_kernel, dim3(32), dim3(256), 0, s, w, w_ws, Cout, Cin, k, 0);
    const int pad = (k - 1) / 2;
    int rc;
    if (in_is_f32 && !in_gelu) rc = dispatch_cout<true, false, false>(s, x, w_ws, bias, (bf16_t*)y, nullptr, B, Cin, Cout, H, W, k, pad);
    else if (!in_is_f32 && in_gelu) rc = dispatch_cout<false, true, false>(s, x, w_ws, bias, (bf16_t*)y, nullptr, B, Cin, Cout, H, W, k, pad);
    else if (!in_is_f32 && !in_gelu) rc = dispatch_cout<false, false, false>(s, x, w_ws, bias, (bf16_t*)y, nullptr, B, Cin, Cout, H, W, k, pad);
    else rc = dispatch_cout<true, true, false>(s, x, w_ws, bias, (bf16_t*)y, nullptr, B, Cin, Cout, H, W, k, pad);
    if (rc != I2T_OK) return rc;
    I2T_CHECK_LAUNCH("i2t_conv_fwd");
    return I2T_OK;
}

extern "C" int i2t_conv_bwd_data(void* stream, const void* dy, const float* w, const void* x_pre, int in_gelu, void* dx,
                                 float* w_ws, int B, int Cin, int Cout, int H, int W, int k) {
    I2T_REQUIRE(dy && w && dx && w_ws && B > 0, "i2t_conv_bwd_data: bad args");
    I2T_REQUIRE(!in_gelu || x_pre, "i2t_conv_bwd_data: in_gelu needs the stored pre-activation");
    I2T_REQUIRE(k == 4 || k == 6, "i2t_conv_bwd_data: kernel size %d unsupported (4 or 6)", k);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(repack_weights_kernel, dim3(32), dim3(256), 0, s, w, w_ws, Cout, Cin, k, 1);
    const int pad = k - 1 - (k - 1) / 2;   // mirrored padding
    int rc;
    // roles swap: the "input" is dY with Cout channels, the "output" is dX with Cin channels
    if (in_gelu) rc = dispatch_cout<false, false, true>(s, dy, w_ws, nullptr, (bf16_t*)dx, (const bf16_t*)x_pre, B, Cout, Cin, H, W, k, pad);
    else rc = dispatch_cout<false, false, false>(s, dy, w_ws, nullptr, (bf16_t*)dx, nullptr, B, Cout, Cin, H, W, k, pad);
    if (rc != I2T_OK) return rc;
    I2T_CHECK_LAUNCH("i2t_conv_bwd_data");
    return I2T_OK;
}

#define BWD_W_CASE(CO, KK)                                                                                              \
    if (Cout == CO && k == KK) {                                                                                        \
        if (in_is_f32 && !in_gelu)                                                                                      \
            hipLaunchKernelGGL((conv_bwd_weight_kernel<CO, KK, true, false>), g1, dim3(256), 0, s, (const bf16_t*)dy, x, dw, \
                               db, Cin, H, W, pad, bx, by, bz);                                                         \
        else if (!in_is_f32 && in_gelu)                                                                                 \
            hipLaunchKernelGGL((conv_bwd_weight_kernel<CO, KK, false, true>), g1, dim3(256), 0, s, (const bf16_t*)dy, x, dw, \
                               db, Cin, H, W, pad, bx, by, bz);                                                         \
        else                                                                                                            \
            hipLaunchKernelGGL((conv_bwd_weight_kernel<CO, KK, false, false>), g1, dim3(256), 0, s, (const bf16_t*)dy, x, dw, \
                               db, Cin, H, W, pad, bx, by, bz);                                                         \
        launched = true;                                                                                                \
    }

extern "C" int i2t_conv_bwd_weight(void* stream, const void* dy, const void* x, int in_is_f32, int in_gelu, float* dw,
                                   float* db, int B, int Cin, int Cout, int H, int W, int k) {
    I2T_REQUIRE(dy && x && dw && B > 0, "i2t_conv_bwd_weight: bad args");
    I2T_REQUIRE(!(in_is_f32 && in_gelu), "i2t_conv_bwd_weight: f32+gelu input combination unsupported");
    hipStream_t s = (hipStream_t)stream;
    dim3 grid((W + TILE - 1) / TILE, (H + TILE - 1) / TILE, B);
    const int pad = (k - 1) / 2;
    bool launched = false;
    // deterministic mode: the workgroups' atomics onto dw / db land in (image, tile row, tile column) order -- one launch each
    const bool det = i2t_det();
    const dim3 g1 = det ? dim3(1, 1, 1) : grid;
    for (int bz = 0; bz < (det ? (int)grid.z : 1); ++bz)
        for (int by = 0; by < (det ? (int)grid.y : 1); ++by)
            for (int bx = 0; bx < (det ? (int)grid.x : 1); ++bx) {
                BWD_W_CASE(4, 6) BWD_W_CASE(8, 6) BWD_W_CASE(16, 6) BWD_W_CASE(32, 6)
                BWD_W_CASE(4, 4) BWD_W_CASE(8, 4) BWD_W_CASE(16, 4) BWD_W_CASE(32, 4)
            }
    I2T_REQUIRE(launched, "i2t_conv_bwd_weight: Cout=%d k=%d unsupported (Cout in {4,8,16,32}, k in {4,6})", Cout, k);
    I2T_CHECK_LAUNCH("i2t_conv_bwd_weight");
    return I2T_OK;
}
