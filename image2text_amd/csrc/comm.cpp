// C-ABI communicator of the data-parallel gradient exchange (i2t_comm_{unique_id,init,allreduce,destroy}):
// one RCCL communicator per process over xGMI, all-reduce of the flat fp32 gradient arena in place.  RCCL is bound at run time
// (dlsym on the copy already in the process -- torch ships one -- else dlopen): the library carries no link-time dependency on
// it, so it loads and every other entry point works on a box without RCCL.
//   wire = 0: fp32 on the wire (the reference semantics: mean of per-shard gradients, exact up to summation order)
//   wire = 1: bf16 on the wire (half the bytes: 647 -> 324 MB per step at nano-224): the caller hands a bf16 staging buffer; the
//             fp32 arena is rounded into it, reduced, and widened back (opt-in: one extra rounding of every gradient element)
#include "common.h"
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <string.h>

namespace {

struct Rccl {
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r;
    static bool tried = false;
    if (tried) return r;
    tried = true;
    void* h = nullptr;
    if (dlsym(RTLD_DEFAULT, "ncclAllReduce")) h = RTLD_DEFAULT;
    const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
    for (int pass = 0; pass < 2 && !h; ++pass)          // first a copy that is already loaded, then a fresh one
        for (const char* n : names)
            if (!h) h = dlopen(n, RTLD_NOW | RTLD_GLOBAL | (pass == 0 ? RTLD_NOLOAD : 0));
    if (!h) return r;
    r.GetUniqueId = (decltype(r.GetUniqueId))dlsym(h, "ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))dlsym(h, "ncclCommInitRank");
    r.AllReduce = (decltype(r.AllReduce))dlsym(h, "ncclAllReduce");
    r.CommDestroy = (decltype(r.CommDestroy))dlsym(h, "ncclCommDestroy");
    r.GetErrorString = (decltype(r.GetErrorString))dlsym(h, "ncclGetErrorString");
    r.ok = r.GetUniqueId && r.CommInitRank && r.AllReduce && r.CommDestroy;
    return r;
}

struct Comm {
    ncclComm_t comm;
    int world, rank;
};

#define RCCL_TRY(call, what)                                                                        \
    do {                                                                                            \
        ncclResult_t r_ = (call);                                                                   \
        if (r_ != ncclSuccess) {                                                                    \
            i2t_set_error("%s: %s", what, rccl().GetErrorString ? rccl().GetErrorString(r_) : "RCCL error"); \
            return I2T_EHIP;                                                                        \
        }                                                                                           \
    } while (0)

// fp32 -> bf16 (round to nearest even) and back, with the 1 / world of the mean folded into the widening pass
__global__ __launch_bounds__(256) void narrow_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, long n4) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const f32x4 v = reinterpret_cast<const f32x4*>(src)[i];
        reinterpret_cast<u32x2*>(dst)[i] = u32x2{pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
    }
}
__global__ __launch_bounds__(256) void widen_kernel(const bf16_t* __restrict__ src, float* __restrict__ dst, long n4, float scale) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        const u32x2 v = reinterpret_cast<const u32x2*>(src)[i];
        reinterpret_cast<f32x4*>(dst)[i] = f32x4{bf16lo(v[0]), bf16hi(v[0]), bf16lo(v[1]), bf16hi(v[1])} * scale;
    }
}

}  // namespace

extern "C" int i2t_comm_available(void) { return rccl().ok ? 1 : 0; }

extern "C" int i2t_comm_unique_id(void* id_out, int bytes) {
    I2T_REQUIRE(id_out && bytes >= (int)sizeof(ncclUniqueId), "i2t_comm_unique_id: need a %d-byte buffer", (int)sizeof(ncclUniqueId));
    I2T_REQUIRE(rccl().ok, "i2t_comm_unique_id: RCCL is not available in this process");
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, sizeof(id));
    return I2T_OK;
}

extern "C" int i2t_comm_init(const void* id, int world, int rank, void** comm_out) {
    I2T_REQUIRE(id && comm_out && world >= 1 && rank >= 0 && rank < world, "i2t_comm_init: bad args (world=%d rank=%d)", world, rank);
    I2T_REQUIRE(rccl().ok, "i2t_comm_init: RCCL is not available in this process");
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof(uid));
    Comm* c = new Comm{nullptr, world, rank};
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);      // (uses the calling thread's current HIP device)
    if (r != ncclSuccess) {
        i2t_set_error("ncclCommInitRank: %s", rccl().GetErrorString ? rccl().GetErrorString(r) : "RCCL error");
        delete c;
        return I2T_EHIP;
    }
    *comm_out = c;
    return I2T_OK;
}

extern "C" int i2t_comm_allreduce(void* comm, void* stream, float* buf, long count, int mean, void* bf16_staging) {
    I2T_REQUIRE(comm && buf && count > 0, "i2t_comm_allreduce: bad args");
    Comm* c = (Comm*)comm;
    hipStream_t s = (hipStream_t)stream;
    if (!bf16_staging) {
        RCCL_TRY(rccl().AllReduce(buf, buf, (size_t)count, ncclFloat32, mean ? ncclAvg : ncclSum, c->comm, s), "ncclAllReduce(f32)");
        return I2T_OK;
    }
    I2T_REQUIRE(count % 4 == 0 && ALIGNED16(buf) && (((uintptr_t)bf16_staging) & 7) == 0, "i2t_comm_allreduce: the bf16 wire form needs count %% 4 == 0");
    const long n4 = count >> 2;
    const long blocks = (n4 + 255) / 256;
    const unsigned grid = (unsigned)(blocks < 65536 ? blocks : 65536);
    hipLaunchKernelGGL(narrow_kernel, dim3(grid), dim3(256), 0, s, buf, (bf16_t*)bf16_staging, n4);
    I2T_CHECK_LAUNCH("i2t_comm_allreduce(narrow)");
    RCCL_TRY(rccl().AllReduce(bf16_staging, bf16_staging, (size_t)count, ncclBfloat16, ncclSum, c->comm, s), "ncclAllReduce(bf16)");
    hipLaunchKernelGGL(widen_kernel, dim3(grid), dim3(256), 0, s, (const bf16_t*)bf16_staging, buf, n4, mean ? 1.0f / (float)c->world : 1.0f);
    I2T_CHECK_LAUNCH("i2t_comm_allreduce(widen)");
    return I2T_OK;
}

extern "C" int i2t_comm_destroy(void* comm) {
    if (!comm) return I2T_OK;
    Comm* c = (Comm*)comm;
    if (rccl().ok && c->comm) rccl().CommDestroy(c->comm);
    delete c;
    return I2T_OK;
}

// bytes of scratch a call of the named entry point needs beyond its operands (SURVEY.md 8(b) lists it; the hot path's kernels
// take their scratch as explicit operands, so this is a lookup for callers that size buffers generically)
extern "C" int i2t_workspace_bytes(const char* entry, long M, long N, long K, long* bytes_out) {
    I2T_REQUIRE(entry && bytes_out, "i2t_workspace_bytes: null argument");
    long b = 0;
    if (!strcmp(entry, "i2t_gemm_bf16_ws")) {          // deterministic split-K planes: (tiles x slices) fp32 planes of 128 x 128
        const long tiles = ((M + 127) / 128) * ((N + 127) / 128);
        long slices = 1;
        while (tiles * slices < 256 && (K / 64) / (slices * 2) >= 4 && slices < 64) slices *= 2;
        b = tiles * slices * 128 * 128 * 4;
    } else if (!strcmp(entry, "i2t_comm_allreduce")) b = M * 2;          // the bf16 wire form's staging buffer: M = element count
    else if (!strcmp(entry, "i2t_attention_bwd")) b = M * 4;             // delta_ws: M = B * H * Tq floats
    else if (!strcmp(entry, "i2t_conv6_bwd_weight")) b = 32 * 36 * 16 * 4;
    else if (strncmp(entry, "i2t_", 4)) {
        i2t_set_error("i2t_workspace_bytes: unknown entry point %s", entry);
        return I2T_EINVAL;
    }
    *bytes_out = b;
    return I2T_OK;
}
