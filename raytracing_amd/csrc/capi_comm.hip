// Multi-GPU half of the C ABI: the one exchange step of the path, over RCCL.
//
// The reference is single-GPU.  What shards is the per-pixel mean over the samples of a frame
// (shader/pathtracing.glsl:657-670): every rank holds the whole scene and draws its slice of the
// global sample indices with pt_params.accum_mode = 1 (sums); ONE all-reduce(sum) over the nine
// accumulated floats per pixel and a multiplication by 1 / SPP finish the frame, and the seven
// last-write floats per pixel come from the rank that drew the last sample (one broadcast).
// The pass keeps each group in one allocation, so neither collective needs a staging copy.
//
// RCCL is opened at run time (dlopen) by the first pt_comm_* call: single-GPU users never load
// it, and a process that already holds an RCCL (torch's, same soname) shares that one.
#include <dlfcn.h>
#include <cstring>
#include <mutex>
#include <rccl/rccl.h>
#include "capi_internal.h"

using namespace rta;

namespace {

struct RcclApi {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

RcclApi &rccl() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
        for(const char *n : names) {
            api.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
            if(api.handle) break;
        }
        if(!api.handle) {
            const char *e = dlerror();
            api.error = std::string("cannot open librccl.so.1: ") + (e ? e : "unknown error");
            return;
        }
        bool ok = true;
        auto sym = [&](const char *name) {
            void *p = dlsym(api.handle, name);
            if(!p) {
                ok = false;
                api.error = std::string("librccl has no symbol ") + name;
            }
            return p;
        };
        api.GetUniqueId = reinterpret_cast<decltype(api.GetUniqueId)>(sym("ncclGetUniqueId"));
        api.CommInitRank = reinterpret_cast<decltype(api.CommInitRank)>(sym("ncclCommInitRank"));
        api.CommInitAll = reinterpret_cast<decltype(api.CommInitAll)>(sym("ncclCommInitAll"));
        api.CommDestroy = reinterpret_cast<decltype(api.CommDestroy)>(sym("ncclCommDestroy"));
        api.AllReduce = reinterpret_cast<decltype(api.AllReduce)>(sym("ncclAllReduce"));
        api.Broadcast = reinterpret_cast<decltype(api.Broadcast)>(sym("ncclBroadcast"));
        api.GroupStart = reinterpret_cast<decltype(api.GroupStart)>(sym("ncclGroupStart"));
        api.GroupEnd = reinterpret_cast<decltype(api.GroupEnd)>(sym("ncclGroupEnd"));
        api.GetErrorString = reinterpret_cast<decltype(api.GetErrorString)>(sym("ncclGetErrorString"));
        if(!ok) {
            dlclose(api.handle);
            api.handle = nullptr;
        }
    });
    return api;
}

#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _r = (expr);                                                                          \
        if(_r != ncclSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + rccl().GetErrorString(_r)); \
    } while(0)

int need_rccl() {
    RcclApi &a = rccl();
    if(!a.handle) return fail(PT_ERR_UNSUPPORTED, "multi-GPU entry points need RCCL: " + a.error);
    return PT_OK;
}

__global__ __launch_bounds__(256) void scale_kernel(float4 *x, size_t n4, float *tail, int n_tail, float w) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for(size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        float4 v = x[i];
        v.x *= w; v.y *= w; v.z *= w; v.w *= w;
        x[i] = v;
    }
    if(blockIdx.x == 0 && (int)threadIdx.x < n_tail) tail[threadIdx.x] *= w;
}

}  // namespace

struct pt_comm {
    pt_ctx *ctx = nullptr;
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1;
};

// defined in capi_device.hip: the two output groups of a pass and what a combine must mark
int pass_output_groups(pt_pass *pass, pt_ctx **ctx, float **accum, float **lastwrite, size_t *npix);

namespace {

// the two collectives of one pass, queued on its context's main stream (inside a group or not)
int queue_combine(pt_pass *pass, pt_comm *comm, uint32_t total_spp, int last_rank, bool scale_now) {
    pt_ctx *ctx = nullptr;
    float *accum = nullptr, *last = nullptr;
    size_t npix = 0;
    int rc = pass_output_groups(pass, &ctx, &accum, &last, &npix);
    if(rc) return rc;
    if(ctx != comm->ctx) return fail(PT_ERR_INVALID, "pt_pass_allreduce: pass and communicator belong to different contexts");
    HIP_TRY(hipSetDevice(ctx->device));
    if(!scale_now) {
        JOIN_TRY(ctx);  // behind every draw of the frame
        RCCL_TRY(rccl().AllReduce(accum, accum, npix * 9, ncclFloat32, ncclSum, comm->comm, ctx->stream));
        RCCL_TRY(rccl().Broadcast(last, last, npix * 7, ncclFloat32, last_rank, comm->comm, ctx->stream));
    } else {
        const float w = (float)(1.0 / (double)total_spp);
        const size_t n = npix * 9, n4 = n / 4;
        size_t blocks = (n4 + 255) / 256;
        if(blocks > 8192) blocks = 8192;
        if(blocks < 1) blocks = 1;
        hipLaunchKernelGGL(scale_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<float4 *>(accum), n4,
                           accum + n4 * 4, (int)(n - n4 * 4), w);
        HIP_TRY(hipGetLastError());
    }
    return PT_OK;
}

int check_combine_args(const char *fn, uint32_t total_spp, int last_rank, int world) {
    if(total_spp == 0) return fail(PT_ERR_INVALID, std::string(fn) + ": total_spp must be at least 1");
    if(last_rank < 0 || last_rank >= world) return fail(PT_ERR_INVALID, std::string(fn) + ": last_rank is not a rank of the communicator");
    return PT_OK;
}

}  // namespace

extern "C" {

int pt_comm_unique_id(void *id_out) {
    if(!id_out) return fail(PT_ERR_INVALID, "pt_comm_unique_id: null argument");
    int rc = need_rccl();
    if(rc) return rc;
    static_assert(sizeof(ncclUniqueId) == PT_COMM_ID_BYTES, "PT_COMM_ID_BYTES must be the size of ncclUniqueId");
    ncclUniqueId id;
    RCCL_TRY(rccl().GetUniqueId(&id));
    memcpy(id_out, &id, sizeof id);
    return PT_OK;
}

int pt_comm_create(pt_ctx *ctx, const void *id, int rank, int world, pt_comm **out) {
    if(!ctx || !id || !out) return fail(PT_ERR_INVALID, "pt_comm_create: null argument");
    if(world < 1 || rank < 0 || rank >= world) return fail(PT_ERR_INVALID, "pt_comm_create: rank must be in [0, world)");
    int rc = need_rccl();
    if(rc) return rc;
    HIP_TRY(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    pt_comm *c = new pt_comm;
    c->ctx = ctx; c->rank = rank; c->world = world;
    ncclResult_t r = rccl().CommInitRank(&c->comm, world, uid, rank);
    if(r != ncclSuccess) {
        delete c;
        return fail(PT_ERR_HIP, std::string("ncclCommInitRank: ") + rccl().GetErrorString(r));
    }
    *out = c;
    return PT_OK;
}

int pt_comm_create_all(pt_ctx *const *ctxs, int n, pt_comm **out) {
    if(!ctxs || !out || n < 1) return fail(PT_ERR_INVALID, "pt_comm_create_all: null argument or n < 1");
    int rc = need_rccl();
    if(rc) return rc;
    std::vector<int> devs(n);
    for(int k = 0; k < n; k++) {
        if(!ctxs[k]) return fail(PT_ERR_INVALID, "pt_comm_create_all: null context");
        devs[k] = ctxs[k]->device;
        for(int j = 0; j < k; j++)
            if(devs[j] == devs[k]) return fail(PT_ERR_INVALID, "pt_comm_create_all: two contexts on one GPU (one rank per GPU)");
    }
    std::vector<ncclComm_t> comms(n);
    RCCL_TRY(rccl().CommInitAll(comms.data(), n, devs.data()));
    for(int k = 0; k < n; k++) {
        pt_comm *c = new pt_comm;
        c->ctx = ctxs[k]; c->rank = k; c->world = n; c->comm = comms[k];
        out[k] = c;
    }
    return PT_OK;
}

int pt_comm_destroy(pt_comm *comm) {
    if(!comm) return PT_OK;
    (void)hipSetDevice(comm->ctx->device);
    (void)ctx_join(comm->ctx);
    (void)hipStreamSynchronize(comm->ctx->stream);
    if(comm->comm && rccl().handle) (void)rccl().CommDestroy(comm->comm);
    delete comm;
    return PT_OK;
}

int pt_comm_rank(const pt_comm *comm) { return comm ? comm->rank : -1; }
int pt_comm_world(const pt_comm *comm) { return comm ? comm->world : 0; }

int pt_pass_allreduce(pt_pass *pass, pt_comm *comm, uint32_t total_spp, int last_rank) {
    if(!pass || !comm) return fail(PT_ERR_INVALID, "pt_pass_allreduce: null argument");
    int rc = check_combine_args("pt_pass_allreduce", total_spp, last_rank, comm->world);
    if(rc) return rc;
    // (two collectives on one stream; grouping them lets RCCL queue both before either runs)
    RCCL_TRY(rccl().GroupStart());
    rc = queue_combine(pass, comm, total_spp, last_rank, false);
    ncclResult_t r = rccl().GroupEnd();
    if(rc) return rc;
    if(r != ncclSuccess) return fail(PT_ERR_HIP, std::string("ncclGroupEnd: ") + rccl().GetErrorString(r));
    return queue_combine(pass, comm, total_spp, last_rank, true);
}

int pt_pass_allreduce_all(pt_pass *const *passes, pt_comm *const *comms, int n, uint32_t total_spp, int last_rank) {
    if(!passes || !comms || n < 1) return fail(PT_ERR_INVALID, "pt_pass_allreduce_all: null argument or n < 1");
    for(int k = 0; k < n; k++) {
        if(!passes[k] || !comms[k]) return fail(PT_ERR_INVALID, "pt_pass_allreduce_all: null pass or communicator");
        if(comms[k]->world != n || comms[k]->rank != k) return fail(PT_ERR_INVALID, "pt_pass_allreduce_all: comms must be the n communicators of pt_comm_create_all, in rank order");
    }
    int rc = check_combine_args("pt_pass_allreduce_all", total_spp, last_rank, n);
    if(rc) return rc;
    // one thread drives all ranks: the collectives of all of them must sit in one group
    RCCL_TRY(rccl().GroupStart());
    for(int k = 0; k < n && !rc; k++) rc = queue_combine(passes[k], comms[k], total_spp, last_rank, false);
    ncclResult_t r = rccl().GroupEnd();
    if(rc) return rc;
    if(r != ncclSuccess) return fail(PT_ERR_HIP, std::string("ncclGroupEnd: ") + rccl().GetErrorString(r));
    for(int k = 0; k < n && !rc; k++) rc = queue_combine(passes[k], comms[k], total_spp, last_rank, true);
    return rc;
}

}  // extern "C"
