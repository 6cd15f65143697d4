// The two exchange steps of the path for callers that are not Python (SURVEY.md §8b "comm"): FedAvg over the clients'
// communicator and the two-party "open" of the SPDZ / FSS protocols, on an RCCL communicator the CALLER created.
//
// RCCL is not a link-time dependency of this library: ncclAllReduce is looked up in the running process (the host
// application — PyTorch or a C++ runtime — has loaded its RCCL already; two RCCL instances in one process must be
// avoided), with `librccl.so` as the fallback.  The Python host code of this repository reaches the same collectives
// through torch.distributed (primia_amd/fed.py, primia_amd/secure.py DistOpener).
#include <dlfcn.h>

#include "common.h"

namespace primia {

typedef int (*nccl_allreduce_fn)(const void*, void*, size_t, int, int, void*, hipStream_t);
constexpr int kNcclInt64 = 4, kNcclFloat32 = 7, kNcclSum = 0;   // ncclDataType_t / ncclRedOp_t (nccl.h)

static nccl_allreduce_fn resolve_allreduce() {
    static nccl_allreduce_fn fn = [] {
        void* s = dlsym(RTLD_DEFAULT, "ncclAllReduce");
        if (!s) {
            void* h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
            if (h) s = dlsym(h, "ncclAllReduce");
        }
        return (nccl_allreduce_fn)s;
    }();
    return fn;
}

}  // namespace primia

using namespace primia;

extern "C" {

int primia_comm_available(void) { return resolve_allreduce() ? 1 : 0; }

int primia_open2(int64_t* buf, int64_t n, void* comm, primia_stream_t st) {
    if (n == 0) return PRIMIA_OK;
    PRIMIA_REQUIRE(buf && n > 0 && comm);
    nccl_allreduce_fn ar = resolve_allreduce();
    if (!ar) return PRIMIA_ERR_UNSUPPORTED;
    return ar(buf, buf, (size_t)n, kNcclInt64, kNcclSum, comm, (hipStream_t)st) == 0 ? PRIMIA_OK : PRIMIA_ERR_LAUNCH;
}

int primia_fedavg_allreduce(float* flat, int64_t n, float weight, int nclients, int secure, int precision_fractional,
                            int64_t* scratch, void* comm, primia_stream_t st) {
    PRIMIA_REQUIRE(flat && n > 0 && nclients > 0 && comm && (!secure || scratch) && precision_fractional >= 0 &&
                   precision_fractional <= 18);
    nccl_allreduce_fn ar = resolve_allreduce();
    if (!ar) return PRIMIA_ERR_UNSUPPORTED;
    int rc;
    if (weight >= 0.f && (rc = primia_scale(flat, n, weight, st)) != PRIMIA_OK) return rc;
    if (secure) {
        float scale = 1.f;
        for (int i = 0; i < precision_fractional; ++i) scale *= 10.f;
        if ((rc = primia_fx_encode(flat, scratch, n, scale, st)) != PRIMIA_OK) return rc;
        if (ar(scratch, scratch, (size_t)n, kNcclInt64, kNcclSum, comm, (hipStream_t)st) != 0) return PRIMIA_ERR_LAUNCH;
        if ((rc = primia_fx_decode(scratch, flat, n, scale, st)) != PRIMIA_OK) return rc;
    } else if (ar(flat, flat, (size_t)n, kNcclFloat32, kNcclSum, comm, (hipStream_t)st) != 0) {
        return PRIMIA_ERR_LAUNCH;
    }
    if (weight < 0.f) return primia_divide(flat, n, (float)nclients, st);
    return PRIMIA_OK;
}

}  // extern "C"
