// Shared device/host helpers for the gfx950 kernels (wave = 64 lanes, MFMA 16x16x32 bf16, LDS 160 KiB/CU).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/i2t.h"

typedef unsigned short bf16_t;                                         // raw bf16 bits
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;             // one MFMA A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) float f32x4;               // one 16x16 accumulator (4 VGPRs)
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2;

#define WAVE 64

// ---- host-side error plumbing (never throw across the C ABI) ----
void i2t_set_error(const char* fmt, ...);
#define I2T_REQUIRE(cond, ...)                                   \
    do {                                                         \
        if (!(cond)) {                                           \
            i2t_set_error(__VA_ARGS__);                          \
            return I2T_EINVAL;                                   \
        }                                                        \
    } while (0)
#define I2T_CHECK_LAUNCH(name)                                                       \
    do {                                                                             \
        hipError_t e_ = hipGetLastError();                                           \
        if (e_ != hipSuccess) {                                                      \
            i2t_set_error("%s: launch failed: %s", name, hipGetErrorString(e_));     \
            return I2T_EHIP;                                                         \
        }                                                                            \
    } while (0)
#define ALIGNED16(p) ((((uintptr_t)(p)) & 15) == 0)
// Deterministic mode (i2t_set_deterministic / I2T_DETERMINISTIC=1): every reduction on the gradient path that normally combines
// workgroup partials with fp32 atomics (order-dependent in the last bit) runs in ONE fixed order instead -- a single workgroup for
// the grid-stride reductions, one K slice for the dW GEMMs, one launch per tile for the convolution weight gradients.  Slow, bit-
// reproducible: two backward passes of the same step are bit-equal (tests/test_round3_gpu.py).
bool i2t_det();
// gemm.hip: the persistent 256^2 kernel on fp8 operands (false = not eligible); used by fp8.hip
bool i2t_g256_fp8_try(hipStream_t s, const void* A8, int lda, const float* sa, const void* B8, int ldb, const float* sb, void* C, int ldc,
                      int c_is_f32, int M, int N, int K, const float* bias, int act, const float* residual, int ldr);

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) -- MI355X_MICROARCH.md correctness table
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    // one v_cvt_pk_bf16_f32 for the pair (two scalar casts compile to two of them plus an SDWA or)
    typedef __attribute__((ext_vector_type(2))) float f32x2_;
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_;
    const f32x2_ v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_));
}
__device__ __forceinline__ float bf16lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- GELU(tanh): 0.5 x (1 + tanh(u)), u = sqrt(2/pi) (x + 0.044715 x^3), evaluated as x * sigmoid(2u): one v_exp_f32 and
// one v_rcp_f32 (1 ulp each) instead of an IEEE division -- the epilogue VALU work of a K = 512 GEMM rivals its MFMA work,
// so every instruction here is paid per output element.
__device__ __forceinline__ float gelu_sigmoid_(float x, float x2) {      // sigmoid(2u)
    const float c0 = -2.0f * 0.7978845608028654f * 1.4426950408889634f, c1 = c0 * 0.044715f;
    const float e = __builtin_amdgcn_exp2f(x * (c0 + c1 * x2));          // exp(-2u); +inf for very negative x -> s = 0
    return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float gelu_tanh(float x) { return x * gelu_sigmoid_(x, x * x); }
__device__ __forceinline__ float gelu_tanh_grad(float x) {               // s + x s (1 - s) d(2u)/dx
    const float d0 = 2.0f * 0.7978845608028654f, d1 = d0 * 3.0f * 0.044715f;
    const float x2 = x * x, s = gelu_sigmoid_(x, x2);
    return s + (x * s) * (1.0f - s) * (d0 + d1 * x2);
}

// GELU and its derivative from ONE sigmoid (forward epilogues that keep the derivative for the backward pass: I2T_ACT_GELU_DOUT)
__device__ __forceinline__ void gelu_tanh_both(float x, float& h, float& dh) {
    const float d0 = 2.0f * 0.7978845608028654f, d1 = d0 * 3.0f * 0.044715f;
    const float x2 = x * x, s = gelu_sigmoid_(x, x2);
    h = x * s;
    dh = s + h * (1.0f - s) * (d0 + d1 * x2);
}

// ---- exact GELU (torch nn.GELU() default, torchvision's ViT MLP): x Phi(x), Phi(x) = (1 + erf(x / sqrt 2)) / 2; derivative
// Phi(x) + x phi(x).  erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, branch-free: one v_rcp_f32 + one v_exp_f32 -- libm's
// erff inlined sixteen times per thread spilled 341 registers in the generic epilogue); exp(-x^2/2) serves erf and phi alike.
__device__ __forceinline__ float erf_sqrt2_(float x, float& e) {       // erf(x / sqrt 2); e = exp(-x^2 / 2)
    const float z = fabsf(x) * 0.7071067811865476f;
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * z);
    e = __builtin_amdgcn_exp2f(-0.7213475204444817f * x * x);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    return copysignf(1.0f - poly * e, x);
}
__device__ __forceinline__ float gelu_erf(float x) {
    float e;
    return 0.5f * x * (1.0f + erf_sqrt2_(x, e));
}
__device__ __forceinline__ float gelu_erf_grad(float x) {
    float e;
    const float phi_cdf = 0.5f * (1.0f + erf_sqrt2_(x, e));
    return phi_cdf + x * 0.3989422804014327f * e;
}

// ---- counter-based dropout: one lowbias32 hash of (site key, idx >> 2) yields FOUR 8-bit uniforms; element idx of site
// `key` is kept iff byte (idx & 3) of the hash is >= thr8 (thr8 = round(p * 256), so the effective p is thr8 / 256 -- 0.1016
// for p = 0.1 -- and the host scales by 256 / (256 - thr8): the mask stays mean-preserving at the rate actually realised).
// The same function is evaluated in forward and backward: no mask is ever stored.  Four decisions per hash because
// v_mul_lo_u32 is quarter rate: at two per hash the hash was ~half of the attention kernels' VALU time (they are VALU-bound)
// and a per-element hash cost more than the MFMA loop of a K = 512 GEMM.  Single-multiply mixers were tried and fail the
// lag-correlation screen by hundreds of sigma (tests/test_host_cpu.py::test_dropout_rule_statistics holds the screen).
__device__ __forceinline__ unsigned dropout_hash(unsigned key, unsigned quad) {
    unsigned x = quad ^ key;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(unsigned key, unsigned idx, unsigned thr8) {
    const unsigned h = dropout_hash(key, idx >> 2);
    return ((h >> (8u * (idx & 3u))) & 0xffu) >= thr8;
}
// keep[r] for the 4 consecutive elements idx0 .. idx0 + 3, any alignment (2 hashes, one 64-bit funnel shift)
__device__ __forceinline__ void dropout_keep4(unsigned key, unsigned idx0, unsigned thr8, bool (&keep)[4]) {
    const unsigned q0 = idx0 >> 2;
    const unsigned long long w = ((unsigned long long)dropout_hash(key, q0 + 1) << 32) | dropout_hash(key, q0);
    const unsigned h = (unsigned)(w >> (8u * (idx0 & 3u)));
#pragma unroll
    for (int r = 0; r < 4; ++r) keep[r] = ((h >> (8 * r)) & 0xffu) >= thr8;
}

// idx0 % 4 == 0: the 4 elements are the 4 bytes of ONE hash (no per-lane alignment case)
__device__ __forceinline__ void dropout_keep4_even(unsigned key, unsigned idx0, unsigned thr8, bool (&keep)[4]) {
    const unsigned h = dropout_hash(key, idx0 >> 2);
#pragma unroll
    for (int r = 0; r < 4; ++r) keep[r] = ((h >> (8 * r)) & 0xffu) >= thr8;
}

// ---- wave reductions (64 lanes) ----
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// block-wide sum for blockDim.x <= 1024 (red = 16 floats of LDS scratch); result broadcast to all threads
__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int i = 0; i < nw; ++i) t += red[i];
    return t;
}
__device__ __forceinline__ float block_max(float v, float* red) {
    v = wave_max(v);
    int w = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float t = red[0];
    for (int i = 1; i < nw; ++i) t = fmaxf(t, red[i]);
    return t;
}

// LDS transposed read: 4 rows x 16 cols block of 16-bit elements per 16-lane group, delivered column-major
// (cdna_hip_programming.md T10).  EXEC must be all ones at the call site.
__device__ __forceinline__ s16x4 lds_read_tr16(const void* lds_ptr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds_ptr));
}
