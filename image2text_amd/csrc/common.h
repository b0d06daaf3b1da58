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

// ---- bf16 <-> f32 ----
__device__ __forceinline__ float bf16_to_f32(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ bf16_t f32_to_bf16(float f) {
    // plain cast: hipcc emits v_cvt_pk_bf16_f32 (RNE, NaN stays NaN) -- MI355X_MICROARCH.md correctness table
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ unsigned pack_bf16x2(float lo, float hi) {
    return (unsigned)f32_to_bf16(lo) | ((unsigned)f32_to_bf16(hi) << 16);
}
__device__ __forceinline__ float bf16lo(unsigned u) { return __uint_as_float(u << 16); }
__device__ __forceinline__ float bf16hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// ---- GELU(tanh) as torch computes it: 0.5 x (1 + tanh(sqrt(2/pi) (x + 0.044715 x^3))) ----
__device__ __forceinline__ float gelu_tanh(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float u = k0 * (x + k1 * x * x * x);
    // tanh(u) = 1 - 2/(1+exp(2u)); exp overflow -> inf -> t = 1, fine
    float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
    return 0.5f * x * (1.0f + t);
}
__device__ __forceinline__ float gelu_tanh_grad(float x) {
    const float k0 = 0.7978845608028654f, k1 = 0.044715f;
    float x2 = x * x;
    float u = k0 * (x + k1 * x * x2);
    float t = 1.0f - 2.0f / (1.0f + __expf(2.0f * u));
    float du = k0 * (1.0f + 3.0f * k1 * x2);
    return 0.5f * (1.0f + t) + 0.5f * x * (1.0f - t * t) * du;
}

// ---- counter-based dropout: keep element idx of site `key` iff hash(key, idx) >= thr (thr = p * 2^32); the same
// function is evaluated in forward and backward, so no mask is ever stored.  (lowbias32 mixer)
__device__ __forceinline__ unsigned dropout_hash(unsigned key, unsigned idx) {
    unsigned x = idx ^ key;
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bool dropout_keep(unsigned key, unsigned idx, unsigned thr) { return dropout_hash(key, idx) >= thr; }

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
