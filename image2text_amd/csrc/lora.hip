// LoRA side branches (reference models/utils.py:46-65 -> peft LoraModel: y = base(x) + lora_B(lora_A(dropout(x))) * alpha / r).  The
// rank-r products are GEMMs (i2t_gemm_bf16 with the rank padded to 64) and the input dropout is i2t_dropout_apply; what is left is
// the one elementwise step that a fused GEMM epilogue does for the un-adapted layer: the GELU derivative behind mlp.c_proj when its
// input gradient is the fp32 sum of the base and the adapter paths.
#include "common.h"

namespace {

// out (bf16) = dh (fp32) * gelu_tanh'(pre (bf16))
__global__ __launch_bounds__(256) void dgelu_mul_kernel(const float* __restrict__ dh, const bf16_t* __restrict__ pre, bf16_t* __restrict__ out,
                                                        long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(dh)[i];
        const u32x2 p = reinterpret_cast<const u32x2*>(pre)[i];
        const float o0 = g[0] * gelu_tanh_grad(bf16lo(p[0])), o1 = g[1] * gelu_tanh_grad(bf16hi(p[0]));
        const float o2 = g[2] * gelu_tanh_grad(bf16lo(p[1])), o3 = g[3] * gelu_tanh_grad(bf16hi(p[1]));
        reinterpret_cast<u32x2*>(out)[i] = u32x2{pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
    }
}

// the same behind an exact (erf) GELU: transformers' Falcon MLP (engine_llama.py falcon blocks)
__global__ __launch_bounds__(256) void dgelu_erf_mul_kernel(const float* __restrict__ dh, const bf16_t* __restrict__ pre, bf16_t* __restrict__ out,
                                                            long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 g = reinterpret_cast<const f32x4*>(dh)[i];
        const u32x2 p = reinterpret_cast<const u32x2*>(pre)[i];
        const float o0 = g[0] * gelu_erf_grad(bf16lo(p[0])), o1 = g[1] * gelu_erf_grad(bf16hi(p[0]));
        const float o2 = g[2] * gelu_erf_grad(bf16lo(p[1])), o3 = g[3] * gelu_erf_grad(bf16hi(p[1]));
        reinterpret_cast<u32x2*>(out)[i] = u32x2{pack_bf16x2(o0, o1), pack_bf16x2(o2, o3)};
    }
}

// out = gelu(pre), both bf16 (tanh or exact form): the activation behind a projection whose GEMM has no activation epilogue with a
// pre-activation output -- the fp8 classes (engine_llama's Falcon MLP on a frozen base, csrc/fp8.hip)
template <bool ERF>
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const bf16_t* __restrict__ pre, bf16_t* __restrict__ out, long n8) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const u32x4 p = reinterpret_cast<const u32x4*>(pre)[i];
        u32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float a = bf16lo(p[e]), b = bf16hi(p[e]);
            o[e] = ERF ? pack_bf16x2(gelu_erf(a), gelu_erf(b)) : pack_bf16x2(gelu_tanh(a), gelu_tanh(b));
        }
        reinterpret_cast<u32x4*>(out)[i] = o;
    }
}

// One pass over the adapted layer's input x [M][K] (bf16): copy it into the first K columns of the K-concatenated operand
// xcat [M][ldc] and, when the adapter has input dropout, write the masked copy xd [M][K] next to it (mask index r * K + c: the index
// space of i2t_dropout_apply and of the residual + dropout GEMM epilogue that applies the same mask in backward).
__global__ __launch_bounds__(256) void lora_stage_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ xcat, int ldc,
                                                         bf16_t* __restrict__ xd, long n8, int k8, unsigned key, unsigned thr, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long)gridDim.x * 256) {
        const long r = i / k8;
        const int c = (int)(i - r * k8) * 8;
        const u32x4 v = *reinterpret_cast<const u32x4*>(x + i * 8);
        *reinterpret_cast<u32x4*>(xcat + (size_t)r * ldc + c) = v;
        if (xd) {
            bool k0[4], k1[4];
            dropout_keep4(key, (unsigned)(i * 8), thr, k0);
            dropout_keep4(key, (unsigned)(i * 8 + 4), thr, k1);
            u32x4 o;
            o[0] = pack_bf16x2(k0[0] ? bf16lo(v[0]) * scale : 0.f, k0[1] ? bf16hi(v[0]) * scale : 0.f);
            o[1] = pack_bf16x2(k0[2] ? bf16lo(v[1]) * scale : 0.f, k0[3] ? bf16hi(v[1]) * scale : 0.f);
            o[2] = pack_bf16x2(k1[0] ? bf16lo(v[2]) * scale : 0.f, k1[1] ? bf16hi(v[2]) * scale : 0.f);
            o[3] = pack_bf16x2(k1[2] ? bf16lo(v[3]) * scale : 0.f, k1[3] ? bf16hi(v[3]) * scale : 0.f);
            *reinterpret_cast<u32x4*>(xd + i * 8) = o;
        }
    }
}

}  // namespace

extern "C" int i2t_dgelu_mul(void* stream, const float* dh, const void* pre, void* out, long n) {
    I2T_REQUIRE(dh && pre && out && n > 0 && n % 4 == 0 && ALIGNED16(dh), "i2t_dgelu_mul: bad args (n=%ld must be a multiple of 4)", n);
    const long n4 = n >> 2;
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(dgelu_mul_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, dh,
                       (const bf16_t*)pre, (bf16_t*)out, n4);
    I2T_CHECK_LAUNCH("i2t_dgelu_mul");
    return I2T_OK;
}

extern "C" int i2t_dgelu_erf_mul(void* stream, const float* dh, const void* pre, void* out, long n) {
    I2T_REQUIRE(dh && pre && out && n > 0 && n % 4 == 0 && ALIGNED16(dh), "i2t_dgelu_erf_mul: bad args (n=%ld must be a multiple of 4)", n);
    const long n4 = n >> 2;
    const long blocks = (n4 + 255) / 256;
    hipLaunchKernelGGL(dgelu_erf_mul_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream, dh,
                       (const bf16_t*)pre, (bf16_t*)out, n4);
    I2T_CHECK_LAUNCH("i2t_dgelu_erf_mul");
    return I2T_OK;
}

extern "C" int i2t_gelu_fwd(void* stream, const void* pre, void* out, long n, int erf) {
    I2T_REQUIRE(pre && out && n > 0 && n % 8 == 0 && ALIGNED16(pre) && ALIGNED16(out), "i2t_gelu_fwd: bad args (n=%ld must be a multiple of 8)", n);
    const long n8 = n >> 3;
    const long blocks = (n8 + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    if (erf) hipLaunchKernelGGL(gelu_fwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pre, (bf16_t*)out, n8);
    else hipLaunchKernelGGL(gelu_fwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, (const bf16_t*)pre, (bf16_t*)out, n8);
    I2T_CHECK_LAUNCH("i2t_gelu_fwd");
    return I2T_OK;
}

extern "C" int i2t_lora_stage(void* stream, const void* x, void* xcat, int ldc, void* xd, long M, int K, unsigned drop_key,
                              unsigned drop_thr, float drop_scale) {
    I2T_REQUIRE(x && xcat && M > 0 && K > 0 && K % 8 == 0 && ldc % 8 == 0 && ldc >= K && ALIGNED16(x) && ALIGNED16(xcat) &&
                    (!xd || ALIGNED16(xd)) && M * K < (1L << 32),
                "i2t_lora_stage: bad args (K=%d and ldc=%d must be multiples of 8, M*K < 2^32)", K, ldc);
    const long n8 = M * (K / 8);
    const long blocks = (n8 + 255) / 256;
    hipLaunchKernelGGL(lora_stage_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)x, (bf16_t*)xcat, ldc, (bf16_t*)xd, n8, K / 8, drop_key, drop_thr, drop_scale);
    I2T_CHECK_LAUNCH("i2t_lora_stage");
    return I2T_OK;
}
