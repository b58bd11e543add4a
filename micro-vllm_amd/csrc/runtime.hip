// Library identity, error reporting, device query and hipGraph capture/replay.
#include "common.h"

#include <string.h>

namespace mv {

static thread_local char g_last_err[256] = "";

void set_last_hip_error(hipError_t e) {
    const char* s = hipGetErrorString(e);
    strncpy(g_last_err, s ? s : "unknown", sizeof(g_last_err) - 1);
    g_last_err[sizeof(g_last_err) - 1] = 0;
}

void set_last_error_text(const char* text) {
    strncpy(g_last_err, text ? text : "", sizeof(g_last_err) - 1);
    g_last_err[sizeof(g_last_err) - 1] = 0;
}

int num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (cached[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            n <= 0)
            n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}

int ensure_dynamic_lds(const void* kernel, size_t lds, LdsOptIn& done) {
    if (lds <= 64 * 1024) return MV_OK;
    int dev = 0;
    const bool known = hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64;
    if (known && done.bytes[dev] >= lds) return MV_OK;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) {
        set_last_hip_error(e);
        return MV_ERR_LAUNCH;
    }
    if (known) done.bytes[dev] = lds;
    return MV_OK;
}

}  // namespace mv

extern "C" {

int mv_version(void) { return 100; }

const char* mv_status_string(int status) {
    switch (status) {
        case MV_OK: return "ok";
        case MV_ERR_INVALID_ARG: return "invalid argument";
        case MV_ERR_UNSUPPORTED: return "unsupported shape";
        case MV_ERR_LAUNCH: return "kernel launch failed";
        case MV_ERR_WORKSPACE: return "workspace too small";
        case MV_ERR_GRAPH: return "graph capture failed";
        default: return "unknown status";
    }
}

const char* mv_last_hip_error(void) { return mv::g_last_err; }

int mv_graph_begin_capture(mv_stream_t stream) {
    // thread-local mode: other threads (e.g. a data loader) may keep calling HIP
    hipError_t e = hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal);
    if (e != hipSuccess) {
        mv::set_last_hip_error(e);
        return MV_ERR_GRAPH;
    }
    return MV_OK;
}

int mv_graph_end_capture(mv_stream_t stream, void** graph_exec_out) {
    if (!graph_exec_out) return MV_ERR_INVALID_ARG;
    hipGraph_t g = nullptr;
    hipError_t e = hipStreamEndCapture((hipStream_t)stream, &g);
    if (e != hipSuccess || !g) {
        mv::set_last_hip_error(e);
        return MV_ERR_GRAPH;
    }
    hipGraphExec_t ex = nullptr;
    e = hipGraphInstantiate(&ex, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (e != hipSuccess) {
        mv::set_last_hip_error(e);
        return MV_ERR_GRAPH;
    }
    *graph_exec_out = (void*)ex;
    return MV_OK;
}

int mv_graph_launch(void* graph_exec, mv_stream_t stream) {
    if (!graph_exec) return MV_ERR_INVALID_ARG;
    hipError_t e = hipGraphLaunch((hipGraphExec_t)graph_exec, (hipStream_t)stream);
    if (e != hipSuccess) {
        mv::set_last_hip_error(e);
        return MV_ERR_GRAPH;
    }
    return MV_OK;
}

int mv_graph_destroy(void* graph_exec) {
    if (!graph_exec) return MV_OK;
    hipError_t e = hipGraphExecDestroy((hipGraphExec_t)graph_exec);
    if (e != hipSuccess) {
        mv::set_last_hip_error(e);
        return MV_ERR_GRAPH;
    }
    return MV_OK;
}

}  // extern "C"
