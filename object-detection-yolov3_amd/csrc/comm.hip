// RCCL gradient exchange behind the C ABI (y3_comm_init / y3_allreduce_sum_f32 / y3_comm_destroy).
//
// Replaces what tf.distribute.MirroredStrategy does inside apply_gradients (train.py:38-39, model.py:500,510-515): a SUM
// all-reduce of the gradient arena over the GPUs of one node.  The Python host (yolo3/parallel.py) issues the same
// collective through torch.distributed (backend "nccl" = RCCL); these entry points are for callers without torch: one
// process per GPU, rank 0 creates the 128-byte unique id and hands it to the others out of band (file, MPI, socket).
// librccl.so is opened lazily with dlopen, so the library loads -- and every other entry point works -- where RCCL is
// not installed.
#include "common.h"
#include <dlfcn.h>
#include <string.h>
#include <rccl/rccl.h>

namespace {
struct Rccl {
    void* handle = nullptr;
    ncclResult_t (*get_unique_id)(ncclUniqueId*) = nullptr;
    ncclResult_t (*comm_init_rank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*all_reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
    ncclResult_t (*comm_count)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*get_version)(int*) = nullptr;
    const char* (*error_string)(ncclResult_t) = nullptr;
};
Rccl* rccl() {
    static Rccl r;
    static bool tried = false;
    if (!tried) {
        tried = true;
        const char* names[] = {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names)
            if ((r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
        if (r.handle) {
            r.get_unique_id = (decltype(r.get_unique_id))dlsym(r.handle, "ncclGetUniqueId");
            r.comm_init_rank = (decltype(r.comm_init_rank))dlsym(r.handle, "ncclCommInitRank");
            r.all_reduce = (decltype(r.all_reduce))dlsym(r.handle, "ncclAllReduce");
            r.comm_destroy = (decltype(r.comm_destroy))dlsym(r.handle, "ncclCommDestroy");
            r.error_string = (decltype(r.error_string))dlsym(r.handle, "ncclGetErrorString");
            r.comm_count = (decltype(r.comm_count))dlsym(r.handle, "ncclCommCount");
            r.get_version = (decltype(r.get_version))dlsym(r.handle, "ncclGetVersion");
        }
    }
    const bool ok = r.handle && r.get_unique_id && r.comm_init_rank && r.all_reduce && r.comm_destroy;
    return ok ? &r : nullptr;
}
int fail(Rccl* r, const char* what, ncclResult_t rc) {
    y3_set_error("%s: %s", what, (r && r->error_string) ? r->error_string(rc) : "RCCL error");
    return Y3_ELAUNCH;
}
}  // namespace

extern "C" int y3_comm_unique_id(void* id128) {
    Y3_CHECK_ARG(id128, "comm_unique_id: null pointer");
    Rccl* r = rccl();
    Y3_CHECK_ARG(r, "comm_unique_id: librccl.so not found");
    static_assert(sizeof(ncclUniqueId) == 128, "unique id size");
    const ncclResult_t rc = r->get_unique_id((ncclUniqueId*)id128);
    return rc == ncclSuccess ? Y3_OK : fail(r, "ncclGetUniqueId", rc);
}

extern "C" int y3_comm_init(const void* id128, int nranks, int rank, void** comm) {
    Y3_CHECK_ARG(id128 && comm && nranks > 0 && rank >= 0 && rank < nranks, "comm_init: bad args (rank %d of %d)", rank, nranks);
    Rccl* r = rccl();
    Y3_CHECK_ARG(r, "comm_init: librccl.so not found");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof(id));
    ncclComm_t c = nullptr;
    const ncclResult_t rc = r->comm_init_rank(&c, nranks, id, rank);   // uses the calling thread's current HIP device
    if (rc != ncclSuccess) return fail(r, "ncclCommInitRank", rc);
    *comm = (void*)c;
    return Y3_OK;
}

extern "C" int y3_allreduce_sum_f32(void* comm, float* buf, size_t count, y3_stream_t stream) {
    Y3_CHECK_ARG(comm && (buf || count == 0), "allreduce_sum_f32: null pointer");
    if (count == 0) return Y3_OK;
    Rccl* r = rccl();
    Y3_CHECK_ARG(r, "allreduce_sum_f32: librccl.so not found");
    const ncclResult_t rc = r->all_reduce(buf, buf, count, ncclFloat32, ncclSum, (ncclComm_t)comm, (hipStream_t)stream);
    return rc == ncclSuccess ? Y3_OK : fail(r, "ncclAllReduce", rc);
}

extern "C" int y3_comm_info(void* comm, int* nranks, int* version) {
    Y3_CHECK_ARG(comm && nranks && version, "comm_info: null pointer");
    Rccl* r = rccl();
    Y3_CHECK_ARG(r, "comm_info: librccl.so not found");
    *nranks = -1;
    *version = -1;
    if (r->comm_count) {
        const ncclResult_t rc = r->comm_count((ncclComm_t)comm, nranks);
        if (rc != ncclSuccess) return fail(r, "ncclCommCount", rc);
    }
    if (r->get_version) {
        const ncclResult_t rc = r->get_version(version);
        if (rc != ncclSuccess) return fail(r, "ncclGetVersion", rc);
    }
    return Y3_OK;
}

extern "C" int y3_comm_destroy(void* comm) {
    if (!comm) return Y3_OK;
    Rccl* r = rccl();
    Y3_CHECK_ARG(r, "comm_destroy: librccl.so not found");
    const ncclResult_t rc = r->comm_destroy((ncclComm_t)comm);
    return rc == ncclSuccess ? Y3_OK : fail(r, "ncclCommDestroy", rc);
}
