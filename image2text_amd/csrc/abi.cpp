// C-ABI plumbing that is not a kernel: version, per-thread error text, hipGraph capture/replay.
#include "common.h"
#include <string.h>
#include <stdlib.h>

static thread_local char g_err[512] = "";

void i2t_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int i2t_abi_version(void) { return I2T_ABI_VERSION; }

static int g_det = -1;                       // -1: not decided yet -> the environment decides at the first query
bool i2t_det() {
    if (g_det < 0) {
        const char* e = getenv("I2T_DETERMINISTIC");
        g_det = (e && e[0] && e[0] != '0') ? 1 : 0;
    }
    return g_det != 0;
}
extern "C" int i2t_set_deterministic(int on) {
    g_det = on ? 1 : 0;
    return I2T_OK;
}
extern "C" int i2t_deterministic(void) { return i2t_det() ? 1 : 0; }

extern "C" int i2t_last_error(char* buf, size_t n) {
    if (buf && n) {
        strncpy(buf, g_err, n - 1);
        buf[n - 1] = 0;
    }
    return (int)strlen(g_err);
}

#define HIP_TRY(call, what)                                                     \
    do {                                                                        \
        hipError_t e_ = (call);                                                 \
        if (e_ != hipSuccess) {                                                 \
            i2t_set_error("%s: %s", what, hipGetErrorString(e_));               \
            return I2T_EHIP;                                                    \
        }                                                                       \
    } while (0)

extern "C" int i2t_graph_capture_begin(void* stream) {
    HIP_TRY(hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal), "hipStreamBeginCapture");
    return I2T_OK;
}

extern "C" int i2t_graph_capture_end(void* stream, void** graph_exec_out) {
    I2T_REQUIRE(graph_exec_out, "i2t_graph_capture_end: null out pointer");
    hipGraph_t graph = nullptr;
    HIP_TRY(hipStreamEndCapture((hipStream_t)stream, &graph), "hipStreamEndCapture");
    hipGraphExec_t exec = nullptr;
    hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    if (e != hipSuccess) {
        i2t_set_error("hipGraphInstantiate: %s", hipGetErrorString(e));
        return I2T_EHIP;
    }
    *graph_exec_out = (void*)exec;
    return I2T_OK;
}

extern "C" int i2t_graph_launch(void* graph_exec, void* stream) {
    I2T_REQUIRE(graph_exec, "i2t_graph_launch: null graph");
    HIP_TRY(hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream), "hipGraphLaunch");
    return I2T_OK;
}

extern "C" int i2t_graph_destroy(void* graph_exec) {
    if (graph_exec) HIP_TRY(hipGraphExecDestroy((hipGraphExec_t)graph_exec), "hipGraphExecDestroy");
    return I2T_OK;
}
