// Pieces shared by the head_dim-64 attention kernels (attention.hip) and the grouped-query / any-head-dim ones (attention_g.hip).
#pragma once
#include "common.h"

namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

struct AttnPtr {
    const bf16_t* p;
    long bs;   // batch stride (elements)
    int rs;    // row stride (elements)
};

// Packed variable-length batches: sequence b owns rows [cu[b], cu[b+1]) of a [total, width] tensor (batch stride unused).
// cu_q / cu_k may be given independently (cross-attention: packed queries against fixed-length memories).
// With cu_q the per-row statistics (lse, delta) are laid out [H][total_q].
struct VarLen {
    const int* cu_q;
    const int* cu_k;
    int total_q;
    int nseq;                 // B: sequences in the batch (the 1-D grid decode needs it)
};

__device__ __forceinline__ bf16x8 pack_frag(const f32x4& a, const f32x4& b) {
    u32x4 v = {pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(b[0], b[1]), pack_bf16x2(b[2], b[3])};
    return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ float quad_max(float v) {   // across the 4 lanes that share a query column
    v = fmaxf(v, __shfl_xor(v, 16, 64));
    return fmaxf(v, __shfl_xor(v, 32, 64));
}
__device__ __forceinline__ float quad_sum(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
}

// Workgroup -> (tile, head, sequence) for a 1-D grid of ntile * H * B workgroups.  The tiles of one (head, sequence) pair
// re-read that pair's K and V (forward, dQ) or Q and dO (dK/dV) -- 5 times at T = 260.  Workgroups are dealt round-robin
// over the 8 XCDs by their linear id, so with the tile index fastest those 5 land on 5 different L2s and every re-read
// crosses the fabric (PMC: 6.3 GB per encoder layer forward against 2.2 GB algorithmic at B = 2048).  Here ids l, l + 8,
// l + 16, ... walk the tiles of ONE pair, i.e. one XCD serves all tiles of a pair back to back and its L2 absorbs the
// re-reads: FETCH_SIZE of the three kernels fell from 1840 to 906 MB per launch.  Their time did not move (VALU-bound, see
// below) -- kept for the fabric / HBM headroom it leaves to whatever runs beside them (the gradient exchange).  Placement
// affects speed only: any mapping gives the same result.
__device__ __forceinline__ void attn_block_coords(int ntile, int H, int B, int& tile, int& h, int& b) {
    const int L = blockIdx.x, P = H * B, full = (P >> 3) * 8 * ntile;
    int pair;
    if (L < full) {
        const int grp = L / (8 * ntile), r = L - grp * 8 * ntile;
        pair = grp * 8 + (r & 7);
        tile = r >> 3;
    } else {
        const int r = L - full;
        pair = (P >> 3) * 8 + r / ntile;
        tile = r % ntile;
    }
    h = pair % H;
    b = pair / H;
}

}  // namespace
