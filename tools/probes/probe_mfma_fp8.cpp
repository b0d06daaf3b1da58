// Which (row, k) does byte j of lane l hold in the A / B operands of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 data?
// The guide gives the bf16 maps only ("other dtypes: check the map with exact integer data").  Exact small-integer data, unit
// block scales (E8M0 127); two hypotheses for the k index are tried against a host reference.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/probe_mfma_fp8.cpp -o /tmp/probe_fp8 && /tmp/probe_fp8
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(const uint8_t* A, const uint8_t* B, float* D, int hyp) {   // A [16][128], B [16][128] (n-major: B[n][k]) e4m3 bytes
    const int l = threadIdx.x, r = l & 15, q = l >> 4;
    i32x8 a, b;
    uint8_t* pa = (uint8_t*)&a;
    uint8_t* pb = (uint8_t*)&b;
    for (int j = 0; j < 32; ++j) {
        const int kk = hyp == 0 ? 32 * q + j : (16 * q + (j & 15) + 64 * (j >> 4));
        pa[j] = A[r * 128 + kk];
        pb[j] = B[r * 128 + kk];
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    for (int e = 0; e < 4; ++e) D[(4 * q + e) * 16 + r] = c[e];               // C/D: col = lane & 15, row = 4 (lane >> 4) + e
}

static uint8_t e4m3(int v) {       // exact encodings of the integers -8 .. 8 (OCP e4m3fn: bias 7)
    if (v == 0) return 0;
    const uint8_t s = v < 0 ? 0x80 : 0;
    int a = v < 0 ? -v : v, e = 0;
    while ((1 << (e + 1)) <= a) ++e;                   // a in [2^e, 2^(e+1))
    const int m = ((a << 3) >> e) & 7;                 // 3 mantissa bits (exact for a <= 15)
    return s | ((e + 7) << 3) | m;
}

int main() {
    std::vector<uint8_t> A(16 * 128), B(16 * 128);
    std::vector<int> Ai(16 * 128), Bi(16 * 128);
    unsigned st = 12345;
    for (int i = 0; i < 16 * 128; ++i) {
        st = st * 1664525u + 1013904223u; Ai[i] = (int)((st >> 16) % 9) - 4;
        st = st * 1664525u + 1013904223u; Bi[i] = (int)((st >> 16) % 9) - 4;
        A[i] = e4m3(Ai[i]); B[i] = e4m3(Bi[i]);
    }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dD, 256 * 4);
    hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
    for (int hyp = 0; hyp < 2; ++hyp) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int m = 0; m < 16; ++m)
            for (int n = 0; n < 16; ++n) {
                long ref = 0;
                for (int kk = 0; kk < 128; ++kk) ref += (long)Ai[m * 128 + kk] * Bi[n * 128 + kk];
                if ((float)ref != D[m * 16 + n]) ++bad;
            }
        printf("hypothesis %d (%s): %d of 256 outputs wrong; D[0][0..3] = %g %g %g %g\n", hyp,
               hyp == 0 ? "k = 32 (lane >> 4) + j" : "k = 16 (lane >> 4) + (j & 15) + 64 (j >> 4)", bad, D[0], D[1], D[2], D[3]);
    }
    return 0;
}
