// rccl.hip — the native collective of the row-sharded sparse models: RCCL all-reduce over xGMI, issued by the library
// itself on its own stream (north star: "an RCCL all-reduce over xGMI of the m x m and m-vector partials").
//
// libgpx.so has NO link-time dependency on librccl: ncclAllReduce is resolved at run time, first among the symbols the
// process has already loaded (a host program linked against RCCL, or PyTorch's bundled copy — the communicator the
// caller hands over must belong to THAT library), then from librccl.so.1.  The communicator is created and owned by the
// caller (one process per GPU, ncclCommInitRank); the library only enqueues collectives on it.
#include "ctx.h"
#include <dlfcn.h>
#include <mutex>

namespace {

// rccl.h: ncclResult_t ncclAllReduce(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t)
using allreduce_fn = int (*)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclDouble = 8;   // rccl.h:467  ncclFloat64 = ncclDouble = 8
constexpr int kNcclSum = 0;      // rccl.h:448  ncclSum = 0

std::mutex g_mu;
allreduce_fn g_allreduce = nullptr;
const char* g_source = "";

int resolve() {
    std::lock_guard<std::mutex> lk(g_mu);
    if (g_allreduce) return GPX_OK;
    void* sym = dlsym(RTLD_DEFAULT, "ncclAllReduce");
    g_source = "already loaded in the process";
    if (!sym) {
        for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_GLOBAL)) {
                sym = dlsym(h, "ncclAllReduce");
                if (sym) {
                    g_source = name;
                    break;
                }
            }
        }
    }
    if (!sym) {
        fprintf(stderr, "gpx: ncclAllReduce not found (no RCCL in the process and librccl.so.1 cannot be loaded: %s)\n", dlerror());
        return GPX_ERR_UNSUPPORTED;
    }
    g_allreduce = reinterpret_cast<allreduce_fn>(sym);
    return GPX_OK;
}

}  // namespace

int gpx_rccl_allreduce(gpx_ctx* c, double* buf, long count, hipStream_t st) {
    if (!g_allreduce || !c->rccl_comm) return GPX_ERR_STATE;
    size_t slot = 0;
    const bool timed = c->rccl_timing;
    if (timed) {
        if (c->rccl_used + 2 > c->rccl_ev.size()) {
            const size_t old = c->rccl_ev.size();
            c->rccl_ev.resize(old + 64, nullptr);
            for (size_t i = old; i < c->rccl_ev.size(); ++i) HIP_TRY(hipEventCreate(&c->rccl_ev[i]));
        }
        slot = c->rccl_used;
        c->rccl_used += 2;
        HIP_TRY(hipEventRecord(c->rccl_ev[slot], st));
    }
    const int r = g_allreduce(buf, buf, (size_t)count, kNcclDouble, kNcclSum, c->rccl_comm, st);
    if (r != 0) {
        fprintf(stderr, "gpx: ncclAllReduce of %ld doubles failed with ncclResult_t %d\n", count, r);
        return GPX_ERR_STATE;
    }
    if (timed) HIP_TRY(hipEventRecord(c->rccl_ev[slot + 1], st));
    return GPX_OK;
}

void gpx_rccl_release(gpx_ctx* c) {
    for (hipEvent_t e : c->rccl_ev)
        if (e) (void)hipEventDestroy(e);
    c->rccl_ev.clear();
    c->rccl_used = 0;
}

extern "C" {

int gpx_set_allreduce_rccl(gpx_ctx* c, void* nccl_comm) {
    if (!c) return GPX_ERR_ARG;
    if (nccl_comm) {
        const int rc = resolve();
        if (rc) return rc;
    }
    c->rccl_comm = nccl_comm;
    c->allreduce = nullptr;
    c->allreduce_stream = nullptr;
    c->allreduce_user = nullptr;
    return GPX_OK;
}

int gpx_rccl_timing(gpx_ctx* c, int on) {
    if (!c) return GPX_ERR_ARG;
    c->rccl_timing = on != 0;
    c->rccl_used = 0;
    return GPX_OK;
}

int gpx_rccl_elapsed(gpx_ctx* c, double* total_ms, int64_t* collectives) {
    if (!c || !total_ms || !collectives) return GPX_ERR_ARG;
    HIP_TRY(hipSetDevice(c->dev));
    double ms = 0;
    for (size_t i = 0; i + 1 < c->rccl_used; i += 2) {
        HIP_TRY(hipEventSynchronize(c->rccl_ev[i + 1]));
        float t = 0;
        HIP_TRY(hipEventElapsedTime(&t, c->rccl_ev[i], c->rccl_ev[i + 1]));
        ms += t;
    }
    *total_ms = ms;
    *collectives = (int64_t)(c->rccl_used / 2);
    c->rccl_used = 0;
    return GPX_OK;
}

const char* gpx_rccl_source(void) { return g_allreduce ? g_source : "not resolved"; }

}  // extern "C"
