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
