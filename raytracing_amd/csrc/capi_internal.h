// Internals shared by the device-side translation units of the C ABI (capi_device.hip,
// capi_comm.hip): the opaque handle types and the helpers every entry point uses.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <unordered_map>
#include <vector>
#include "capi_common.h"

#define HIP_TRY(expr)                                                                                         \
    do {                                                                                                      \
        hipError_t _e = (expr);                                                                               \
        if(_e != hipSuccess) return rta::fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e));      \
    } while(0)

struct pt_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 0;
    void *scratch = nullptr;  // this device's copy of the file-scope scratch array (pt_pass_intersect_noalloc)
    uint64_t next_handle = 1;
    std::unordered_map<uint64_t, pt_tex *> textures;
    // Draws run on streams of their own (pt_pass::lanes).  Each leaves its completion event
    // here; everything issued on `stream` first makes `stream` wait for them (ctx_join).
    std::vector<hipEvent_t> pending;
    // live passes: a buffer or texture that is released is unbound from them first, so that a
    // later draw reports the missing binding instead of reading freed memory
    std::vector<pt_pass *> passes;
};

// Orders the context's main stream behind every draw issued so far.
inline int ctx_join(pt_ctx *ctx) {
    for(hipEvent_t e : ctx->pending) {
        hipError_t r = hipStreamWaitEvent(ctx->stream, e, 0);
        if(r != hipSuccess) return rta::fail(PT_ERR_HIP, std::string("hipStreamWaitEvent: ") + hipGetErrorString(r));
    }
    ctx->pending.clear();
    return PT_OK;
}
#define JOIN_TRY(ctx)                       \
    do {                                    \
        int _rc = ctx_join(ctx);            \
        if(_rc) return _rc;                 \
    } while(0)

// Reports (once) a traversal-stack overflow of any pass of the context; to be called where the
// device has just been waited for.
int ctx_check_overflow(pt_ctx *ctx);

struct pt_buf {
    pt_ctx *ctx = nullptr;
    void *dptr = nullptr;
    size_t bytes = 0;
    // last bytes uploaded from the host, if shadow_valid: a cache of the device copy for host-side readers
    // (the traversal re-layout), given back when large once nothing is going to read it
    mutable std::vector<uint8_t> shadow;
    mutable bool shadow_valid = false;
    uint64_t version = 0;  // bumped by every content change
    bool owned = true;     // false: a view into memory its owner frees (the outputs of a pass)
};

struct pt_tex {
    pt_ctx *ctx = nullptr;
    void *dptr = nullptr;
    int width = 0, height = 0, channels = 0, format = 0;
    uint64_t handle = 0;
};

