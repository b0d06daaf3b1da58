// 6x6 'same' convolutions of the ConvMLP feature extractor as implicit GEMMs on the MFMA units (gfx950).
//
// Why a new layout: the MFMA A/B fragments want 8 consecutive reduction indices per lane as ONE aligned 16-byte LDS
// read.  With channels-last (NHWC) activations the 8 consecutive indices are 8 input channels of one tap at one
// pixel -- always 16-byte aligned -- whereas with NCHW they would be 8 consecutive x positions at an arbitrary
// (2-byte aligned) offset.  So the intermediate pre-activations live in HBM as NHWC bf16; the first layer converts
// the fp32 NCHW image while staging, and the last layer writes NCHW (its output IS the flat-patch operand of the
// projector GEMM, reference encoder.py:166).
//
//   forward / backward-data (one kernel, weights flipped + roles swapped for backward-data):
//       D[co][pixel] = sum_{tap, ci} W[co][tap][ci] * patch[pixel + tap][ci]
//       workgroup = 16 x 16 output pixels; wave w owns 4 pixel rows; MFMA 16x16x32 with A = weights (rows = co),
//       B = patch (cols = 16 consecutive x); k-step = 4 (tap, 8-channel chunk) groups; weights (<= 36 KiB) and the
//       21 x 21 halo patch sit in LDS; GELU is applied while staging, bias / GELU' in the epilogue.
//   backward-weight:  dW[co][tap][ci] = sum_pixels dY[pixel][co] * act[pixel + tap][ci]
//       reduction over pixels = the ROW index of both LDS tiles -> both fragments come from transposed LDS reads
//       (ds_read_b64_tr_b16, any 4-row set); a workgroup sweeps a row of 14 tiles keeping dW in registers, then adds
//       its partial with contiguous float atomics into a [co][tap][ci] scratch that a tiny kernel folds into dW.
#include "common.h"
#include <utility>

namespace {

template <int N, class F, int... Is>
__device__ __forceinline__ void static_for_impl_c(F&& f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for_c(F&& f) { static_for_impl_c<N>(static_cast<F&&>(f), std::make_integer_sequence<int, N>{}); }

constexpr int TS = 16;            // output tile edge
constexpr int KS = 6;             // kernel size
constexpr int PW = TS + KS - 1;   // 21: patch edge
constexpr int NTAP = KS * KS;
#ifdef I2T_CONV_NOGELU          // timing experiment (tools/build_variant.sh): the staging GELU left out -- WRONG results, prices its VALU share
#define CONV_GELU(x) (x)
#else
#define CONV_GELU(x) gelu_tanh(x)
#endif

enum { SRC_NCHW_F32 = 0, SRC_NCHW_BF16 = 1, SRC_NHWC_BF16 = 2 };

// wr[co][tap][ci_p] (bf16, zero padded) from w[Cout][Cin][6][6] (f32).
//   flip = 0: forward            co = output channel of w, ci = input channel of w
//   flip = 1: backward-data      roles swapped: rows = Cin of w, reduction = Cout of w, taps mirrored
__global__ void conv_repack_kernel(const float* __restrict__ w, bf16_t* __restrict__ wr, int Cout, int Cin, int rows_p,
                                   int red_p, int flip) {
    const int n = rows_p * NTAP * red_p;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int c = i % red_p, tap = (i / red_p) % NTAP, r = i / (red_p * NTAP);
        const int ky = tap / KS, kx = tap % KS;
        float v = 0.f;
        if (!flip) {
            if (r < Cout && c < Cin) v = w[((r * Cin + c) * KS + ky) * KS + kx];
        } else {
            if (r < Cin && c < Cout) v = w[((c * Cin + r) * KS + (KS - 1 - ky)) * KS + (KS - 1 - kx)];
        }
        wr[i] = f32_to_bf16(v);
    }
}

// stage the (PW x PW) halo patch around tile (y0, x0) into LDS as [py][px][CP] bf16 (zero outside the image / pads)
template <int CP, int SRC, bool IN_GELU, int NTH = 256>
__device__ __forceinline__ void stage_patch(bf16_t* __restrict__ pl, const void* __restrict__ src, int b, int C, int H, int W,
                                            int y0, int x0, int pad_before, int tid) {
    if (SRC == SRC_NHWC_BF16) {
        constexpr int CH = CP / 8;                                   // 16-byte chunks per pixel
        const bf16_t* s = reinterpret_cast<const bf16_t*>(src) + (size_t)b * H * W * CP;
        for (int i = tid; i < PW * PW * CH; i += NTH) {
            const int ch = i % CH, px = (i / CH) % PW, py = i / (CH * PW);
            const int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gy >= 0 && gy < H && gx >= 0 && gx < W) {
                v = *reinterpret_cast<const u32x4*>(s + ((size_t)gy * W + gx) * CP + ch * 8);
                if (IN_GELU) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(CONV_GELU(bf16lo(v[e])), CONV_GELU(bf16hi(v[e])));
                }
            }
            *reinterpret_cast<u32x4*>(pl + (py * PW + px) * CP + ch * 8) = v;
        }
    } else {
        const size_t plane = (size_t)H * W;
        for (int i = tid; i < CP * PW * PW; i += NTH) {
            const int px = i % PW, py = (i / PW) % PW, c = i / (PW * PW);
            const int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
            float v = 0.f;
            if (c < C && gy >= 0 && gy < H && gx >= 0 && gx < W) {
                const size_t off = ((size_t)b * C + c) * plane + (size_t)gy * W + gx;
                v = (SRC == SRC_NCHW_F32) ? reinterpret_cast<const float*>(src)[off]
                                          : bf16_to_f32(reinterpret_cast<const bf16_t*>(src)[off]);
                if (IN_GELU) v = CONV_GELU(v);
            }
            pl[(py * PW + px) * CP + c] = f32_to_bf16(v);
        }
    }
}

// NHWC source: the same staging in two halves, so that the next tile's patch is in flight (registers) while the current
// tile is being computed -- a tile is only ~1 us of MFMA work, less than the latency of its own halo gather.
template <int CP, int NTH = 256>
struct PatchRegs {
    static constexpr int N = (PW * PW * (CP / 8) + NTH - 1) / NTH;
    u32x4 v[N];
};
template <int CP, bool IN_GELU, int NTH = 256>
__device__ __forceinline__ void patch_load(PatchRegs<CP, NTH>& pr, const void* __restrict__ src, int b, int H, int W, int y0, int x0,
                                           int pad_before, int tid) {
    constexpr int CH = CP / 8;
    const bf16_t* s = reinterpret_cast<const bf16_t*>(src) + (size_t)b * H * W * CP;
#pragma unroll
    for (int u = 0; u < PatchRegs<CP, NTH>::N; ++u) {
        const int i = tid + NTH * u;
        const int ch = i % CH, px = (i / CH) % PW, py = i / (CH * PW);
        const int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (i < PW * PW * CH && gy >= 0 && gy < H && gx >= 0 && gx < W)
            v = *reinterpret_cast<const u32x4*>(s + ((size_t)gy * W + gx) * CP + ch * 8);
        pr.v[u] = v;
    }
}
// CP = 32 (64-byte pixels): 16 consecutive pixels of one 8-channel chunk -- what a B-fragment read touches -- sit 64 B apart, four
// lanes of a ds_read_b128 lane group per 16-byte bank slot pair-wise (2-way conflict on 9 of the 15 reads per k-step group: the
// backward-data kernel of the last layer is LDS-bound).  Chunk ch of patch pixel px is therefore kept at ch ^ (((px >> 2) & 1) << 1):
// conflict-free for every horizontal tap offset (enumerated over the four lane groups x six kx).
template <int CP>
__device__ __forceinline__ int patch_chunk(int px, int ch) { return CP == 32 ? (ch ^ (((px >> 2) & 1) << 1)) : ch; }

template <int CP, bool IN_GELU, int NTH = 256>
__device__ __forceinline__ void patch_store(const PatchRegs<CP, NTH>& pr, bf16_t* __restrict__ pl, int tid) {
#pragma unroll
    for (int u = 0; u < PatchRegs<CP, NTH>::N; ++u) {
        const int i = tid + NTH * u;
        if (i >= PW * PW * (CP / 8)) continue;
        u32x4 v = pr.v[u];
        if (IN_GELU) {        // gelu(0) = 0: the zero halo stays zero
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = pack_bf16x2(CONV_GELU(bf16lo(v[e])), CONV_GELU(bf16hi(v[e])));
        }
        if (CP == 32) {
            const int ch = i % 4, pix = i / 4;                       // [py][px][ch']: the pixel's chunks permuted by its px
            *reinterpret_cast<u32x4*>(pl + (size_t)pix * 32 + patch_chunk<CP>(pix % PW, ch) * 8) = v;
        } else {
            *reinterpret_cast<u32x4*>(pl + (size_t)i * 8) = v;      // [py][px][ch] is exactly chunk order i
        }
    }
}

// fp32 NCHW source with <= 4 channels and CP == 8 (the image under the first layer): one pixel per thread slot, its channels
// gathered from the planes and written as ONE 16-byte [8 ch] chunk (the generic planar staging walks all CP channel slots
// -- 5 of 8 are padding -- with 2-byte LDS stores, 14 rounds per tile); in two halves like the NHWC patch so that the next
// tile's pixels are in flight while the current tile is computed.
struct PatchRegsImg {
    static constexpr int N = (PW * PW + 255) / 256;
    float v[N][4];
};
__device__ __forceinline__ void img_patch_load(PatchRegsImg& pr, const void* __restrict__ src, int b, int C, int H, int W, int y0,
                                               int x0, int pad_before, int tid) {
    const size_t plane = (size_t)H * W;
    const float* s = reinterpret_cast<const float*>(src) + (size_t)b * C * plane;
#pragma unroll
    for (int u = 0; u < PatchRegsImg::N; ++u) {
        const int i = tid + 256 * u;
        const int px = i % PW, py = i / PW;
        const int gy = y0 + py - pad_before, gx = x0 + px - pad_before;
        const bool ok = i < PW * PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
#pragma unroll
        for (int c = 0; c < 4; ++c) pr.v[u][c] = (ok && c < C) ? s[c * plane + (size_t)gy * W + gx] : 0.f;
    }
}
__device__ __forceinline__ void img_patch_store(const PatchRegsImg& pr, bf16_t* __restrict__ pl, int tid) {
#pragma unroll
    for (int u = 0; u < PatchRegsImg::N; ++u) {
        const int i = tid + 256 * u;
        if (i >= PW * PW) continue;
        const u32x4 v = {pack_bf16x2(pr.v[u][0], pr.v[u][1]), pack_bf16x2(pr.v[u][2], pr.v[u][3]), 0u, 0u};
        *reinterpret_cast<u32x4*>(pl + (size_t)i * 8) = v;            // [py][px][8 ch]
    }
}

// ------------------------------------------------------------------------------------------------ fwd / bwd-data
// NW = waves per workgroup (4: a wave owns 4 pixel rows of the 16 x 16 tile; 8: 2 rows).  The 32-channel-input form keeps 64 KiB of
// LDS (36 KiB of weights + 28 KiB of patch): two workgroups per CU whatever their size, so it runs 8-wave workgroups -- 4 waves per SIMD
// instead of 2 (PMC, round 4: at 2 waves per SIMD its waves were parked 56 % of their cycles, matrix pipe 32 % busy).
// GRP (NCHW output only): the packed outputs of GRP consecutive tiles of the sweep are held in registers and stored together, so that the
// GRP x 32-byte pieces of one (channel, row) -- one 128-byte line at GRP = 4 -- leave the wave in consecutive instructions and meet in the L2's
// write path.  (Stored tile by tile, the neighbouring pieces of a line arrive a tile time (~1 us) apart and the last layer's forward spent
// 1.2 us per image on its 3.2 MB of output -- 2.7 TB/s; the NHWC form, whose two 32-byte halves of a pixel are consecutive stores: 5.8.)
template <int CP, int NT, int SRC, bool IN_GELU, bool DGELU, bool DST_NCHW, int NW = 4, int GRP = 1>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(NW == 8 ? 4 : 3))) void conv_mfma_kernel(const void* __restrict__ src, const bf16_t* __restrict__ wr,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ dst,
                                                        const bf16_t* __restrict__ pre, int Cin, int Cout, int H, int W,
                                                        int pad_before, int tiles_x) {
    constexpr int WROW = NTAP * CP + 8;                       // +16 B pad: conflict-free 16-lane weight reads
    constexpr int COP = NT * 16;                              // padded output channels
    __shared__ __attribute__((aligned(16))) bf16_t wl[COP * WROW];
    __shared__ __attribute__((aligned(16))) bf16_t pl[PW * PW * CP];
    constexpr int NTH = 64 * NW, RW = TS / NW;                // threads per workgroup, pixel rows per wave
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int b = blockIdx.y;
    const int y0 = blockIdx.x * TS;
    // A workgroup sweeps one row of tiles: the weights (up to 36 KiB) are staged ONCE per 14 tiles -- per tile they cost
    // about as much LDS-fill time as the tile's MFMA work -- and the NCHW rows of neighbouring tiles are written by the
    // same workgroup back to back (32-byte pieces that the L2 merges into full lines).
    for (int i = tid; i < COP * NTAP * CP / 8; i += NTH) {    // weights: contiguous [co][tap][ci] -> padded rows
        const int co = i / (NTAP * CP / 8), rest = i % (NTAP * CP / 8);
        *reinterpret_cast<u32x4*>(wl + co * WROW + rest * 8) = *reinterpret_cast<const u32x4*>(wr + (size_t)i * 8);
    }
    // bias of the lane's output channels: once per workgroup (NCHW output: channel 16 t + li; NHWC: channels 16 t + 4 g .. + 3)
    float bvn[NT];
    f32x4 bvh[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        bvn[t] = (bias && 16 * t + li < Cout) ? bias[16 * t + li] : 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) bvh[t][e] = (bias && 16 * t + 4 * g + e < Cout) ? bias[16 * t + 4 * g + e] : 0.f;
    }
    PatchRegs<CP, NTH> pr;
    PatchRegsImg pi;
    constexpr bool IMG_OK = SRC == SRC_NCHW_F32 && CP == 8 && !IN_GELU && NW == 4;
    const bool img = IMG_OK && Cin <= 4;                      // workgroup-uniform
    if (SRC == SRC_NHWC_BF16) patch_load<CP, IN_GELU, NTH>(pr, src, b, H, W, y0, 0, pad_before, tid);
    else if (img) img_patch_load(pi, src, b, Cin, H, W, y0, 0, pad_before, tid);
    u32x2 hold[GRP][NT][TS / NW];                             // (DST_NCHW) packed outputs of the group's tiles
    auto do_tile = [&](int tx, auto SUB_) {
    constexpr int SUB = decltype(SUB_)::value;
    const int x0 = tx * TS;
    __syncthreads();                                          // the previous tile's patch reads are done
    if (SRC == SRC_NHWC_BF16) patch_store<CP, IN_GELU, NTH>(pr, pl, tid);
    else if (img) img_patch_store(pi, pl, tid);
    else stage_patch<CP, SRC, IN_GELU, NTH>(pl, src, b, Cin, H, W, y0, x0, pad_before, tid);
    __syncthreads();
    if (tx + 1 < tiles_x) {
        if (SRC == SRC_NHWC_BF16) patch_load<CP, IN_GELU, NTH>(pr, src, b, H, W, y0, x0 + TS, pad_before, tid);
        else if (img) img_patch_load(pi, src, b, Cin, H, W, y0, x0 + TS, pad_before, tid);
    }

    // wave-uniform: the tile lies inside the image and every padded output channel exists
    const bool full = (y0 + TS <= H) && (x0 + TS <= W) && (Cout == COP) && (!DST_NCHW || (W & 3) == 0);
    u32x2 prq[RW][NT];         // GELU' inputs (DGELU): requested now, used after the MFMA loop
    if constexpr (DGELU) {
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int oy = min(y0 + RW * w + r, H - 1), oxc = min(x0 + li, W - 1), c0 = min(16 * t + 4 * g, Cout - 4);
#ifdef I2T_CONV_DBG_NOPRE
                prq[r][t] = u32x2{0x3f803f80u, 0x3f803f80u};
#else
                prq[r][t] = *reinterpret_cast<const u32x2*>(pre + (((size_t)b * H + oy) * W + oxc) * Cout + c0);
#endif
            }
    }
    f32x4 acc[RW][NT];
#pragma unroll
    for (int r = 0; r < RW; ++r)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[r][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CH = CP / 8;
    constexpr int NSTEP = NTAP * CH / 4;                      // k-steps of 4 (tap, chunk) groups
#ifdef I2T_CONV_DBG_NOMMA
    if constexpr (false) {
#else
    if constexpr (CP >= 16) {
#endif
        // 16 / 32 input channels: one MFMA k-step covers 2 / 1 horizontal taps of ONE kernel row, so for a fixed
        // horizontal tap (pair) the 4 output rows x 6 vertical taps of a wave read only 9 distinct patch-row fragments:
        // keep them in registers and walk the vertical taps -- (9 + 6 NT) LDS reads per 24 NT MFMAs instead of 30 NT
        // (the k-step-major loop re-read every patch fragment once per vertical tap; the kernel is LDS-read bound).
        constexpr int TPM = 32 / CP;                          // horizontal taps per MFMA k-step
        const int kxo = TPM == 2 ? (g >> 1) : 0, ch = TPM == 2 ? (g & 1) : g;
#pragma unroll 1
        for (int m = 0; m < KS / TPM; ++m) {
            const int kx = m * TPM + kxo;
            bf16x8 pf[RW + KS - 1];                           // rows RW w .. RW w + RW + 4 of the halo patch
            const int chs = (SRC == SRC_NHWC_BF16) ? patch_chunk<CP>(li + kx, ch) : ch;      // (the planar staging keeps chunk order)
#pragma unroll
            for (int y = 0; y < RW + KS - 1; ++y)
                pf[y] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pl + ((RW * w + y) * PW + li + kx) * CP + chs * 8));
#pragma unroll
            for (int ky = 0; ky < KS; ++ky) {
                const int kg = (ky * KS + kx) * CH + ch;
                bf16x8 fw[NT];
#pragma unroll
                for (int t = 0; t < NT; ++t)
                    fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + (16 * t + li) * WROW + kg * 8));
#pragma unroll
                for (int r = 0; r < RW; ++r)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        // DST_NCHW: patch as the A operand -> acc = D^T[pixel 4 g + e][co 16 t + li]: a lane's 4 values are 4 consecutive x of
                        // ONE channel = one 8-byte NCHW store (un-swapped, co in the registers, NCHW took four 2-byte stores per lane)
                        acc[r][t] = DST_NCHW ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(pf[r + ky], fw[t], acc[r][t], 0, 0, 0)
                                             : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t], pf[r + ky], acc[r][t], 0, 0, 0);
            }
        }
    } else
#ifdef I2T_CONV_DBG_NOMMA
    if (pl[tid] == 0x1234)
#endif
#pragma unroll 2
    for (int s = 0; s < NSTEP; ++s) {
        const int kg = 4 * s + g;
        const int tap = kg / CH, ch = kg % CH;
        const int ky = tap / KS, kx = tap % KS;
        bf16x8 fw[NT], fp[RW];
#pragma unroll
        for (int t = 0; t < NT; ++t)
            fw[t] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(wl + (16 * t + li) * WROW + kg * 8));
#pragma unroll
        for (int r = 0; r < RW; ++r)
            fp[r] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(pl + ((RW * w + r + ky) * PW + li + kx) * CP + ch * 8));
#pragma unroll
        for (int r = 0; r < RW; ++r)
#pragma unroll
            for (int t = 0; t < NT; ++t)
                acc[r][t] = DST_NCHW ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp[r], fw[t], acc[r][t], 0, 0, 0)
                                     : __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[t], fp[r], acc[r][t], 0, 0, 0);
    }
    // ---- epilogue.  Everything the stores depend on is in registers by now (bias: loaded once before the sweep; GELU' inputs: requested
    // before the MFMA loop), and a tile that lies inside the image / channel range takes a branch-free copy: the stores of a tile then
    // issue back to back.  (As first written -- a bias load and per-lane range branches between the stores -- hipcc put s_waitcnt vmcnt(0)
    // in front of every store: each waited for the previous one's write acknowledgement, 1.2 us per image in the last layer's forward.)
    if constexpr (DST_NCHW) {
        // D^T[pixel x0 + 4 g + e][co = 16 t + li] of output row y0 + RW w + r: 8 bytes per lane, 32 contiguous bytes per (channel, row);
        // packed here, stored by flush() once the group's tiles are done
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < RW; ++r) {
                const float bv = bvn[t];
                hold[SUB][t][r] = u32x2{pack_bf16x2(acc[r][t][0] + bv, acc[r][t][1] + bv), pack_bf16x2(acc[r][t][2] + bv, acc[r][t][3] + bv)};
            }
        return;
    }
    // D[co = 16 t + 4 g + e][pixel = li] of output row y0 + RW w + r
    const int ox = x0 + li;
#pragma unroll
    for (int r = 0; r < RW; ++r) {
        const int oy = y0 + RW * w + r;
        if (!full && (oy >= H || ox >= W)) continue;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c0 = 16 * t + 4 * g;
            if (!full && c0 >= Cout) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = acc[r][t][e] + bvh[t][e];
            if (DGELU) {       // pre-activation of the layer below, NHWC with Cout channels (loaded before the MFMA loop)
                const u32x2 pk = prq[r][t];
                v[0] *= gelu_tanh_grad(bf16lo(pk[0]));
                v[1] *= gelu_tanh_grad(bf16hi(pk[0]));
                v[2] *= gelu_tanh_grad(bf16lo(pk[1]));
                v[3] *= gelu_tanh_grad(bf16hi(pk[1]));
            }
            const u32x2 pk = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
#ifdef I2T_CONV_DBG_NOSTORE
            if (pk[0] == 0x12345678u)
#endif
            *reinterpret_cast<u32x2*>(dst + (((size_t)b * H + oy) * W + ox) * Cout + c0) = pk;
        }
    }
    };   // do_tile
    // (DST_NCHW) the group's outputs: for every (channel, row) the tiles' 32-byte pieces in consecutive store instructions
    auto flush = [&](int tx0, int count) {
        if constexpr (DST_NCHW) {
            const bool inside = (y0 + TS <= H) && ((tx0 + count) * TS <= W) && (Cout == COP) && ((W & 3) == 0);      // wave-uniform
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int r = 0; r < TS / NW; ++r) {
                    const int co = 16 * t + li, oy = y0 + (TS / NW) * w + r;
                    bf16_t* o = dst + (((size_t)b * Cout + co) * H + oy) * W + tx0 * TS + 4 * g;
                    if (inside) {
#pragma unroll
                        for (int sub = 0; sub < GRP; ++sub)
                            if (sub < count) {
#ifdef I2T_CONV_DBG_NOSTORE
                                if (hold[sub][t][r][0] == 0x12345678u)
#endif
                                *reinterpret_cast<u32x2*>(o + sub * TS) = hold[sub][t][r];
                            }
                    } else if (co < Cout && oy < H) {
#pragma unroll
                        for (int sub = 0; sub < GRP; ++sub) {
                            if (sub >= count) continue;
                            const int oxq = (tx0 + sub) * TS + 4 * g;
                            const unsigned lo = hold[sub][t][r][0], hi = hold[sub][t][r][1];
                            const bf16_t v4[4] = {(bf16_t)(lo & 0xffffu), (bf16_t)(lo >> 16), (bf16_t)(hi & 0xffffu), (bf16_t)(hi >> 16)};
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (oxq + e < W) o[sub * TS + e] = v4[e];
                        }
                    }
                }
        }
    };
    for (int tx0 = 0; tx0 < tiles_x; tx0 += GRP) {
        const int count = min(GRP, tiles_x - tx0);                // block-uniform
        static_for_c<GRP>([&](auto S_) {
            if (decltype(S_)::value < count) do_tile(tx0 + decltype(S_)::value, S_);
        });
        flush(tx0, count);
    }
}

// ------------------------------------------------------------------------------------------------ bwd-weight
// scratch layout: [COP][NTAP][CP] f32 (this launch's partial dW), db scratch [COP]
template <int CP, int COP, int DY_SRC, int ACT_SRC, bool ACT_GELU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void conv_mfma_bwd_weight_kernel(const void* __restrict__ dy, const void* __restrict__ act,
                                                                   float* __restrict__ scratch, float* __restrict__ db,
                                                                   int Cin, int Cout, int H, int W, int pad_before,
                                                                   int tiles_x, int by0, int bz0) {
    constexpr int NT = COP / 16;                              // co tiles
    constexpr int NN = (CP == 16) ? NTAP : NTAP / 2;           // n-tiles of 16 (tap x ci) columns
    constexpr int NPW = (NN + 3) / 4;                          // n-tiles per wave
    __shared__ __attribute__((aligned(16))) bf16_t dl[TS * TS * COP];     // dY tile  [pixel][co]
    __shared__ __attribute__((aligned(16))) bf16_t pl[PW * PW * CP];      // act patch [py][px][ci]
    __shared__ float dbs[4][COP];                             // per-wave bias partials, summed in wave order (no LDS atomics: fixed order)
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int g = lane >> 4, li = lane & 15;
    const int q = li >> 2, p = li & 3;
    const int b = blockIdx.y + bz0, ty = blockIdx.x + by0;   // one workgroup sweeps the tile row ty of image b (offsets: deterministic mode)
    const int y0 = ty * TS;
    f32x4 acc[NPW][NT];
#pragma unroll
    for (int n = 0; n < NPW; ++n)
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[n][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (tid < COP) dbs[0][tid] = dbs[1][tid] = dbs[2][tid] = dbs[3][tid] = 0.f;
    float dbp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) dbp[t] = 0.f;

    // the fp32 image under the first layer: pixel-chunk staging, next tile's pixels in flight during the MFMAs
    constexpr bool IMG_OK = ACT_SRC == SRC_NCHW_F32 && CP == 8 && !ACT_GELU;
    const bool img = IMG_OK && Cin <= 4;
    PatchRegsImg pi;
    if (img) img_patch_load(pi, act, b, Cin, H, W, y0, 0, pad_before, tid);
    // NHWC sources (the layers above the first): the next tile's dY tile and activation patch are in flight in registers while this
    // tile is computed (a tile is < 1 us of MFMA work: staged synchronously, every tile paid its own global-load latency at a barrier)
    // (16-channel activations: the 32 extra registers cost the kernel its fourth wave per SIMD and it ran 3 % slower -- 8-channel layers only)
    constexpr bool PRE_DY = DY_SRC == SRC_NHWC_BF16 && CP == 8, PRE_ACT = ACT_SRC == SRC_NHWC_BF16 && CP == 8;
    constexpr int DCH = COP / 8, NDY = TS * TS * DCH / 256;
    u32x4 dyr[PRE_DY ? NDY : 1];
    PatchRegs<CP> pr;
    auto dy_load = [&](int x0) {
        const bf16_t* s_ = reinterpret_cast<const bf16_t*>(dy) + (size_t)b * H * W * Cout;
#pragma unroll
        for (int u = 0; u < NDY; ++u) {
            const int i = tid + 256 * u;
            const int ch = i % DCH, px = (i / DCH) % TS, py = i / (DCH * TS);
            const int gy = y0 + py, gx = x0 + px;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (gy < H && gx < W && ch * 8 < Cout) v = *reinterpret_cast<const u32x4*>(s_ + ((size_t)gy * W + gx) * Cout + ch * 8);
            dyr[u] = v;
        }
    };
    if constexpr (PRE_DY) dy_load(0);
    if constexpr (PRE_ACT) patch_load<CP, ACT_GELU>(pr, act, b, H, W, y0, 0, pad_before, tid);
    for (int tx = 0; tx < tiles_x; ++tx) {
        const int x0 = tx * TS;
        __syncthreads();
        // dY tile as [pixel][COP] (pads zero); pad_before = 0: no halo
        if constexpr (PRE_DY) {
#pragma unroll
            for (int u = 0; u < NDY; ++u) *reinterpret_cast<u32x4*>(dl + (size_t)(tid + 256 * u) * 8) = dyr[u];      // [py][px][ch] is chunk order
        } else if constexpr (DY_SRC == SRC_NHWC_BF16) {
            const bf16_t* s_ = reinterpret_cast<const bf16_t*>(dy) + (size_t)b * H * W * Cout;
            for (int i = tid; i < TS * TS * DCH; i += 256) {
                const int ch = i % DCH, px = (i / DCH) % TS, py = i / (DCH * TS);
                const int gy = y0 + py, gx = x0 + px;
                u32x4 v = {0u, 0u, 0u, 0u};
                if (gy < H && gx < W && ch * 8 < Cout) v = *reinterpret_cast<const u32x4*>(s_ + ((size_t)gy * W + gx) * Cout + ch * 8);
                *reinterpret_cast<u32x4*>(dl + (py * TS + px) * COP + ch * 8) = v;
            }
        } else {
            const size_t plane = (size_t)H * W;
            for (int i = tid; i < COP * TS * TS; i += 256) {
                const int px = i % TS, py = (i / TS) % TS, c = i / (TS * TS);
                const int gy = y0 + py, gx = x0 + px;
                bf16_t v = 0;
                if (c < Cout && gy < H && gx < W) v = reinterpret_cast<const bf16_t*>(dy)[((size_t)b * Cout + c) * plane + (size_t)gy * W + gx];
                dl[(py * TS + px) * COP + c] = v;
            }
        }
        if (img) img_patch_store(pi, pl, tid);
        else if constexpr (PRE_ACT) patch_store<CP, ACT_GELU>(pr, pl, tid);
        else stage_patch<CP, ACT_SRC, ACT_GELU>(pl, act, b, Cin, H, W, y0, x0, pad_before, tid);
        __syncthreads();
        if (tx + 1 < tiles_x) {
            if (img) img_patch_load(pi, act, b, Cin, H, W, y0, x0 + TS, pad_before, tid);
            if constexpr (PRE_DY) dy_load(x0 + TS);
            if constexpr (PRE_ACT) patch_load<CP, ACT_GELU>(pr, act, b, H, W, y0, x0 + TS, pad_before, tid);
        }
        // 8 k-steps of 32 pixels = 2 tile rows x 16 x; k-slot (g, j) <-> pixel (row 2 s + (j >> 2), x = 4 g + (j & 3))
#pragma unroll 1
        for (int s = 0; s < 8; ++s) {
            bf16x8 fa[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const bf16_t* a0 = dl + ((2 * s) * TS + 4 * g + q) * COP + 16 * t + 4 * p;
                s16x4 lo = lds_read_tr16(a0);
                s16x4 hi = lds_read_tr16(a0 + TS * COP);
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                fa[t] = __builtin_bit_cast(bf16x8, v);
            }
            // bias gradient = column sums of dY: lane (g, li) already holds dY[8 pixels of chunk g][co = 16 t + li] in fa[t]
            // (every wave loads the same A fragments), so the waves take turns adding them up -- the separate pass over the
            // LDS tile (COP threads x 256 serial reads per tile) cost 15-30 % of this kernel
            if (db && w == (s & 3)) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int j = 0; j < 8; ++j) dbp[t] += (float)fa[t][j];
            }
#pragma unroll
            for (int n = 0; n < NPW; ++n) {
                const int nt = min(w * NPW + n, NN - 1);     // clamped: out-of-range slots recompute the last tile (discarded)
                int ky, kx, coff;
                if (CP == 16) {
                    ky = nt / KS; kx = nt % KS; coff = 4 * p;
                } else {                                     // CP == 8: 16 columns = taps (kx, kx+1) x 8 channels
                    ky = nt / (KS / 2); kx = 2 * (nt % (KS / 2)) + (p >> 1); coff = 4 * (p & 1);
                }
                const bf16_t* b0 = pl + ((2 * s + ky) * PW + 4 * g + q + kx) * CP + coff;
                s16x4 lo = lds_read_tr16(b0);
                s16x4 hi = lds_read_tr16(b0 + PW * CP);
                s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                const bf16x8 fb = __builtin_bit_cast(bf16x8, v);
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[n][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[t], fb, acc[n][t], 0, 0, 0);
            }
        }
    }
    // D[co = 16 t + 4 g + e][column li of n-tile]: add into scratch[co][tap][ci]
#pragma unroll
    for (int n = 0; n < NPW; ++n) {
        const int nt = w * NPW + n;
        if (nt >= NN) continue;
        int tap, ci;
        if (CP == 16) { tap = nt; ci = li; }
        else { tap = (nt / (KS / 2)) * KS + 2 * (nt % (KS / 2)) + (li >> 3); ci = li & 7; }
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int co = 16 * t + 4 * g + e;
                atomicAdd(scratch + ((size_t)co * NTAP + tap) * CP + ci, acc[n][t][e]);
            }
    }
    if (db) {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float v = dbp[t];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            if (g == 0) dbs[w][16 * t + li] = v;               // this wave's slot
        }
    }
    __syncthreads();
    if (db && tid < Cout) atomicAdd(db + tid, (dbs[0][tid] + dbs[1][tid]) + (dbs[2][tid] + dbs[3][tid]));
}

// dw[co][ci][ky][kx] += scratch[co][tap][ci]
__global__ void conv_fold_dw_kernel(const float* __restrict__ scratch, float* __restrict__ dw, int Cout, int Cin, int CP) {
    const int n = Cout * Cin * NTAP;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const int tap = i % NTAP, ci = (i / NTAP) % Cin, co = i / (NTAP * Cin);
        dw[i] += scratch[((size_t)co * NTAP + tap) * CP + ci];
    }
}

int pad8(int c) { return c <= 8 ? 8 : (c <= 16 ? 16 : 32); }

// dst[b][y][x][c] = src[b][c][y][x] (bf16): one workgroup per (64-pixel run of a row, b); coalesced both ways through LDS.
// The projector's dX GEMM produces the last conv layer's gradient as flat NCHW patches; both backward kernels of that layer
// are ~3x faster on channels-last input (16-byte staging instead of 2-byte scatter), so it is transposed once.
// C = 32, 16-byte accesses on both sides: thread (c, 8-pixel group) loads 16 B of one channel row, the LDS image is
// [c][64 px] with a 33-dword row stride (written as 4 dwords, read back as 8 conflict-free 16-bit reads per thread), thread
// (px, 8-channel group) stores 16 B -- a wave writes 1 KiB of consecutive NHWC bytes.
__global__ __launch_bounds__(256) void nchw_to_nhwc32_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int H, int W) {
    __shared__ unsigned tile[32 * 33];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64, t = threadIdx.x;
    const size_t plane = (size_t)H * W;
    {
        const int c = t >> 3, pg = t & 7;
        u32x4 v = {0u, 0u, 0u, 0u};
        if (x0 + pg * 8 < W) v = *reinterpret_cast<const u32x4*>(src + ((size_t)b * 32 + c) * plane + (size_t)y * W + x0 + pg * 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) tile[c * 33 + pg * 4 + e] = v[e];
    }
    __syncthreads();
    const int px = t >> 2, cg = t & 3;
    if (x0 + px >= W) return;
    const bf16_t* th = reinterpret_cast<const bf16_t*>(tile);
    unsigned o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned lo = th[((cg * 8 + 2 * j) * 33) * 2 + px], hi = th[((cg * 8 + 2 * j + 1) * 33) * 2 + px];
        o[j] = lo | (hi << 16);
    }
    *reinterpret_cast<u32x4*>(dst + (((size_t)b * H + y) * W + x0 + px) * 32 + cg * 8) = u32x4{o[0], o[1], o[2], o[3]};
}

__global__ __launch_bounds__(256) void nchw_to_nhwc_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, int C, int H,
                                                           int W) {
    __shared__ bf16_t tile[64][33];
    const int b = blockIdx.z, y = blockIdx.y, x0 = blockIdx.x * 64;
    const size_t plane = (size_t)H * W;
    for (int i = threadIdx.x; i < C * 64; i += 256) {
        const int c = i >> 6, px = i & 63;
        tile[px][c] = (x0 + px < W) ? src[((size_t)b * C + c) * plane + (size_t)y * W + x0 + px] : (bf16_t)0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
        const int px = i / C, c = i % C;
        if (x0 + px < W) dst[(((size_t)b * H + y) * W + x0 + px) * C + c] = tile[px][c];
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
#ifndef CONV_NCHW_GRP
#define CONV_NCHW_GRP 4
#endif
#define LAUNCH_CONV(CP_, NT_, SRC_, G_, D_, N_)                                                                                             \
    hipLaunchKernelGGL((conv_mfma_kernel<CP_, NT_, SRC_, G_, D_, N_, (CP_ == 32 || (N_ && CP_ == 16) ? 8 : 4), ((N_ && CP_ >= 16) ? CONV_NCHW_GRP : 1)>), grid,         \
                       dim3((CP_ == 32 || (N_ && CP_ == 16)) ? 512 : 256), 0, s, src,                                                  \
                       (const bf16_t*)wr, bias, (bf16_t*)dst, (const bf16_t*)pre, Cin, Cout, H, W, pad_before, tiles_x)

static int conv_mfma_dispatch(hipStream_t s, const void* src, int src_layout, int in_gelu, const void* wr, const float* bias,
                              void* dst, int dst_nchw, const void* pre, int B, int Cin, int Cout, int H, int W,
                              int pad_before) {
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    dim3 grid(tiles_y, B);
    const int cp = pad8(Cin), nt = Cout <= 16 ? 1 : 2;
    const bool dg = pre != nullptr;
    // the combinations the feature extractor uses (forward: f32 NCHW image | NHWC+GELU; backward-data: NCHW | NHWC dY)
    if (src_layout == SRC_NCHW_F32 && cp == 8 && !in_gelu && !dg && !dst_nchw) { if (nt == 1) LAUNCH_CONV(8, 1, SRC_NCHW_F32, false, false, false); else LAUNCH_CONV(8, 2, SRC_NCHW_F32, false, false, false); }
    else if (src_layout == SRC_NCHW_F32 && cp == 8 && !in_gelu && !dg && dst_nchw) { if (nt == 1) LAUNCH_CONV(8, 1, SRC_NCHW_F32, false, false, true); else LAUNCH_CONV(8, 2, SRC_NCHW_F32, false, false, true); }
    else if (src_layout == SRC_NHWC_BF16 && in_gelu && !dg && !dst_nchw) {
        if (cp == 8 && nt == 1) LAUNCH_CONV(8, 1, SRC_NHWC_BF16, true, false, false);
        else if (cp == 8) LAUNCH_CONV(8, 2, SRC_NHWC_BF16, true, false, false);
        else if (cp == 16 && nt == 1) LAUNCH_CONV(16, 1, SRC_NHWC_BF16, true, false, false);
        else if (cp == 16) LAUNCH_CONV(16, 2, SRC_NHWC_BF16, true, false, false);
        else if (nt == 1) LAUNCH_CONV(32, 1, SRC_NHWC_BF16, true, false, false);
        else LAUNCH_CONV(32, 2, SRC_NHWC_BF16, true, false, false);
    } else if (src_layout == SRC_NHWC_BF16 && in_gelu && !dg && dst_nchw) {
        if (cp == 8 && nt == 1) LAUNCH_CONV(8, 1, SRC_NHWC_BF16, true, false, true);
        else if (cp == 8) LAUNCH_CONV(8, 2, SRC_NHWC_BF16, true, false, true);
        else if (cp == 16 && nt == 1) LAUNCH_CONV(16, 1, SRC_NHWC_BF16, true, false, true);
        else if (cp == 16) LAUNCH_CONV(16, 2, SRC_NHWC_BF16, true, false, true);
        else if (nt == 1) LAUNCH_CONV(32, 1, SRC_NHWC_BF16, true, false, true);
        else LAUNCH_CONV(32, 2, SRC_NHWC_BF16, true, false, true);
    } else if (dg && !in_gelu && !dst_nchw && (src_layout == SRC_NCHW_BF16 || src_layout == SRC_NHWC_BF16)) {
        if (src_layout == SRC_NCHW_BF16) {
            if (cp == 8) LAUNCH_CONV(8, 1, SRC_NCHW_BF16, false, true, false);
            else if (cp == 16) LAUNCH_CONV(16, 1, SRC_NCHW_BF16, false, true, false);
            else LAUNCH_CONV(32, 1, SRC_NCHW_BF16, false, true, false);
        } else {
            if (cp == 8) LAUNCH_CONV(8, 1, SRC_NHWC_BF16, false, true, false);
            else if (cp == 16) LAUNCH_CONV(16, 1, SRC_NHWC_BF16, false, true, false);
            else LAUNCH_CONV(32, 1, SRC_NHWC_BF16, false, true, false);
        }
        if (nt != 1) { i2t_set_error("conv6 bwd-data: Cin=%d > 16 unsupported", Cout); return I2T_EINVAL; }
    } else {
        i2t_set_error("conv6 mfma: unsupported layout combination (src=%d gelu=%d dgelu=%d nchw_out=%d)", src_layout, in_gelu, (int)dg, dst_nchw);
        return I2T_EINVAL;
    }
    return I2T_OK;
}

extern "C" int i2t_nchw_to_nhwc_bf16(void* stream, const void* src, void* dst, int B, int C, int H, int W) {
    I2T_REQUIRE(src && dst && B > 0 && C > 0 && C <= 32 && H > 0 && W > 0, "i2t_nchw_to_nhwc_bf16: bad args (C <= 32)");
    if (C == 32 && W % 8 == 0 && ALIGNED16(src) && ALIGNED16(dst)) {
        hipLaunchKernelGGL(nchw_to_nhwc32_kernel, dim3((W + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                           (bf16_t*)dst, H, W);
        I2T_CHECK_LAUNCH("i2t_nchw_to_nhwc_bf16");
        return I2T_OK;
    }
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((W + 63) / 64, H, B), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src,
                       (bf16_t*)dst, C, H, W);
    I2T_CHECK_LAUNCH("i2t_nchw_to_nhwc_bf16");
    return I2T_OK;
}

extern "C" int i2t_conv6_fwd(void* stream, const void* x, int x_layout, int in_gelu, const float* w, const float* bias,
                             void* y, int y_nchw, void* w_ws, int B, int Cin, int Cout, int H, int W) {
    I2T_REQUIRE(x && w && y && w_ws && B > 0 && Cin > 0 && Cout > 0 && Cin <= 32 && Cout <= 32, "i2t_conv6_fwd: bad args");
    I2T_REQUIRE(x_layout != SRC_NHWC_BF16 || Cin == pad8(Cin), "i2t_conv6_fwd: NHWC input needs 8/16/32 channels");
    I2T_REQUIRE(y_nchw || Cout % 8 == 0, "i2t_conv6_fwd: NHWC output needs Cout %% 8 == 0");
    hipStream_t s = (hipStream_t)stream;
    const int cp = pad8(Cin), cop = Cout <= 16 ? 16 : 32;
    hipLaunchKernelGGL(conv_repack_kernel, dim3(64), dim3(256), 0, s, w, (bf16_t*)w_ws, Cout, Cin, cop, cp, 0);
    int rc = conv_mfma_dispatch(s, x, x_layout, in_gelu, w_ws, bias, y, y_nchw, nullptr, B, Cin, Cout, H, W, (KS - 1) / 2);
    if (rc != I2T_OK) return rc;
    I2T_CHECK_LAUNCH("i2t_conv6_fwd");
    return I2T_OK;
}

extern "C" int i2t_conv6_bwd_data(void* stream, const void* dy, int dy_layout, const float* w, const void* x_pre, void* dx,
                                  void* w_ws, int B, int Cin, int Cout, int H, int W) {
    I2T_REQUIRE(dy && w && x_pre && dx && w_ws && B > 0 && Cin % 8 == 0 && Cin <= 16 && Cout <= 32, "i2t_conv6_bwd_data: bad args");
    I2T_REQUIRE(dy_layout == SRC_NCHW_BF16 || (dy_layout == SRC_NHWC_BF16 && Cout == pad8(Cout)), "i2t_conv6_bwd_data: dy layout");
    hipStream_t s = (hipStream_t)stream;
    const int cp = pad8(Cout);                 // reduction channels = Cout of the forward conv
    hipLaunchKernelGGL(conv_repack_kernel, dim3(64), dim3(256), 0, s, w, (bf16_t*)w_ws, Cout, Cin, 16, cp, 1);
    // roles swapped: "Cin" of the kernel = Cout, "Cout" of the kernel = Cin; mirrored padding (3 before)
    int rc = conv_mfma_dispatch(s, dy, dy_layout, 0, w_ws, nullptr, dx, 0, x_pre, B, Cout, Cin, H, W, KS - 1 - (KS - 1) / 2);
    if (rc != I2T_OK) return rc;
    I2T_CHECK_LAUNCH("i2t_conv6_bwd_data");
    return I2T_OK;
}

#define LAUNCH_BW(CP_, COP_, DS_, AS_, G_)                                                                              \
    for (int bz = 0; bz < (det ? B : 1); ++bz)                                                                         \
        for (int by = 0; by < (det ? tiles_y : 1); ++by)                                                               \
            hipLaunchKernelGGL((conv_mfma_bwd_weight_kernel<CP_, COP_, DS_, AS_, G_>), (det ? dim3(1, 1) : grid), dim3(256), 0, s, dy, x, \
                               scratch, db, Cin, Cout, H, W, (KS - 1) / 2, tiles_x, by, bz)

extern "C" int i2t_conv6_bwd_weight(void* stream, const void* dy, int dy_layout, const void* x, int x_layout, int in_gelu,
                                    float* dw, float* db, float* scratch, int B, int Cin, int Cout, int H, int W) {
    I2T_REQUIRE(dy && x && dw && scratch && B > 0 && Cin <= 16 && Cout <= 32, "i2t_conv6_bwd_weight: bad args");
    hipStream_t s = (hipStream_t)stream;
    const int cp = pad8(Cin), cop = Cout <= 16 ? 16 : 32;
    hipError_t e = hipMemsetAsync(scratch, 0, sizeof(float) * (size_t)cop * NTAP * cp, s);
    if (e != hipSuccess) { i2t_set_error("i2t_conv6_bwd_weight: memset: %s", hipGetErrorString(e)); return I2T_EHIP; }
    const int tiles_x = (W + TS - 1) / TS, tiles_y = (H + TS - 1) / TS;
    dim3 grid(tiles_y, B);
    const bool det = i2t_det();      // deterministic mode: the scratch / db atomics land in (image, tile row) order, one launch each
    bool ok = true;
    if (x_layout == SRC_NCHW_F32 && !in_gelu && cp == 8) {
        if (dy_layout == SRC_NHWC_BF16 && cop == 16) LAUNCH_BW(8, 16, SRC_NHWC_BF16, SRC_NCHW_F32, false);
        else if (dy_layout == SRC_NHWC_BF16) LAUNCH_BW(8, 32, SRC_NHWC_BF16, SRC_NCHW_F32, false);
        else if (cop == 16) LAUNCH_BW(8, 16, SRC_NCHW_BF16, SRC_NCHW_F32, false);
        else LAUNCH_BW(8, 32, SRC_NCHW_BF16, SRC_NCHW_F32, false);
    } else if (x_layout == SRC_NHWC_BF16 && in_gelu && Cin == cp) {
        if (cp == 8) {
            if (dy_layout == SRC_NHWC_BF16 && cop == 16) LAUNCH_BW(8, 16, SRC_NHWC_BF16, SRC_NHWC_BF16, true);
            else if (dy_layout == SRC_NHWC_BF16) LAUNCH_BW(8, 32, SRC_NHWC_BF16, SRC_NHWC_BF16, true);
            else if (cop == 16) LAUNCH_BW(8, 16, SRC_NCHW_BF16, SRC_NHWC_BF16, true);
            else LAUNCH_BW(8, 32, SRC_NCHW_BF16, SRC_NHWC_BF16, true);
        } else {
            if (dy_layout == SRC_NHWC_BF16 && cop == 16) LAUNCH_BW(16, 16, SRC_NHWC_BF16, SRC_NHWC_BF16, true);
            else if (dy_layout == SRC_NHWC_BF16) LAUNCH_BW(16, 32, SRC_NHWC_BF16, SRC_NHWC_BF16, true);
            else if (cop == 16) LAUNCH_BW(16, 16, SRC_NCHW_BF16, SRC_NHWC_BF16, true);
            else LAUNCH_BW(16, 32, SRC_NCHW_BF16, SRC_NHWC_BF16, true);
        }
    } else {
        ok = false;
    }
    I2T_REQUIRE(ok, "i2t_conv6_bwd_weight: unsupported layout combination (x_layout=%d gelu=%d Cin=%d)", x_layout, in_gelu, Cin);
    I2T_REQUIRE(dy_layout != SRC_NHWC_BF16 || Cout % 8 == 0, "i2t_conv6_bwd_weight: NHWC dy needs Cout %% 8 == 0");
    hipLaunchKernelGGL(conv_fold_dw_kernel, dim3(32), dim3(256), 0, s, scratch, dw, Cout, Cin, cp);
    I2T_CHECK_LAUNCH("i2t_conv6_bwd_weight");
    return I2T_OK;
}
