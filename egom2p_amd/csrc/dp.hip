// Data-parallel gradient exchange behind the C-ABI ("dp_allreduce_begin / wait wrapping RCCL").
//
// Replaces what torch.nn.parallel.DistributedDataParallel does for the reference (run_training_egom2p.py:514-515 wrap, :723
// no_sync; egom2p/utils/dist.py:78-100 process-group set-up): one communicator per process (one process per GPU), gradient
// buckets summed in place over xGMI on a communication stream the caller owns, so the exchange overlaps the rest of the
// backward; the compute stream joins with ego_dp_wait.  Two exchange algorithms per bucket: RCCL's all-reduce, or an in-place
// reduce-scatter + all-gather (every rank sums one 1/world shard; the shards travel over all seven links at once).
//
// RCCL is reached through dlopen / dlsym: the library has no link-time dependency on it, and a process that already holds an
// RCCL (PyTorch bundles one) shares that copy instead of loading a second.  No global state beyond the resolved entry points.
#include "common.h"
#include "egom2p_hip.h"
#include <dlfcn.h>
#include <string.h>

namespace {

// the few RCCL declarations used (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat = 7, ncclSum = 0)
typedef struct { char internal[128]; } rccl_uid;
typedef void* rccl_comm;
typedef int (*fn_get_uid)(rccl_uid*);
typedef int (*fn_init_rank)(rccl_comm*, int, rccl_uid, int);
typedef int (*fn_destroy)(rccl_comm);
typedef int (*fn_allreduce)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_reduce_scatter)(const void*, void*, size_t, int, int, rccl_comm, hipStream_t);
typedef int (*fn_allgather)(const void*, void*, size_t, int, rccl_comm, hipStream_t);
typedef int (*fn_group)(void);
constexpr int RCCL_FLOAT = 7, RCCL_SUM = 0;

struct Rccl {
    void* h = nullptr;
    fn_get_uid get_uid = nullptr; fn_init_rank init_rank = nullptr; fn_destroy destroy = nullptr;
    fn_allreduce allreduce = nullptr; fn_reduce_scatter reduce_scatter = nullptr; fn_allgather allgather = nullptr;
    fn_group group_start = nullptr, group_end = nullptr;
    bool ok = false;
};

Rccl& rccl() {
    static Rccl r = [] {
        Rccl x;
        const char* names[] = {"librccl.so", "librccl.so.1"};
        for (const char* n : names) if ((x.h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;      // a copy the process already holds
        if (!x.h) for (const char* n : names) if ((x.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
        if (!x.h) x.h = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
        if (!x.h) return x;
        x.get_uid = (fn_get_uid)dlsym(x.h, "ncclGetUniqueId");
        x.init_rank = (fn_init_rank)dlsym(x.h, "ncclCommInitRank");
        x.destroy = (fn_destroy)dlsym(x.h, "ncclCommDestroy");
        x.allreduce = (fn_allreduce)dlsym(x.h, "ncclAllReduce");
        x.reduce_scatter = (fn_reduce_scatter)dlsym(x.h, "ncclReduceScatter");
        x.allgather = (fn_allgather)dlsym(x.h, "ncclAllGather");
        x.group_start = (fn_group)dlsym(x.h, "ncclGroupStart");
        x.group_end = (fn_group)dlsym(x.h, "ncclGroupEnd");
        x.ok = x.get_uid && x.init_rank && x.destroy && x.allreduce && x.reduce_scatter && x.allgather && x.group_start && x.group_end;
        return x;
    }();
    return r;
}

}  // namespace

struct ego_dp_comm { rccl_comm comm; int rank, world; hipEvent_t ev; };

extern "C" int ego_dp_unique_id(void* out128) {
    if (!out128 || !rccl().ok) return EGO_ERR_ARG;
    rccl_uid id;
    if (rccl().get_uid(&id) != 0) return EGO_ERR_LAUNCH;
    memcpy(out128, id.internal, 128);
    return EGO_OK;
}

extern "C" int ego_dp_comm_create(const void* unique_id128, int rank, int world, ego_dp_comm** out) {
    if (!unique_id128 || !out || world < 1 || rank < 0 || rank >= world || !rccl().ok) return EGO_ERR_ARG;
    rccl_uid id;
    memcpy(id.internal, unique_id128, 128);
    ego_dp_comm* c = new ego_dp_comm{nullptr, rank, world, nullptr};
    if (rccl().init_rank(&c->comm, world, id, rank) != 0) { delete c; return EGO_ERR_LAUNCH; }
    if (hipEventCreateWithFlags(&c->ev, hipEventDisableTiming) != hipSuccess) { rccl().destroy(c->comm); delete c; return EGO_ERR_LAUNCH; }
    *out = c;
    return EGO_OK;
}

extern "C" int ego_dp_comm_destroy(ego_dp_comm* c) {
    if (!c) return EGO_OK;
    (void)hipEventDestroy(c->ev);
    const int rc = rccl().destroy(c->comm);
    delete c;
    return rc == 0 ? EGO_OK : EGO_ERR_LAUNCH;
}

// buf[0, count) (fp32, device) is summed over the ranks in place.  The exchange is enqueued on comm_stream behind everything
// compute_stream holds at the time of the call (one event), and returns at once.  algo 0: all-reduce; 1: in-place
// reduce-scatter + all-gather (the count % world leftover elements take an all-reduce).
extern "C" int ego_dp_allreduce_begin(ego_dp_comm* c, float* buf, long count, int algo, hipStream_t compute_stream,
                                      hipStream_t comm_stream) {
    if (!c || !buf || count < 0 || (algo != 0 && algo != 1)) return EGO_ERR_ARG;
    if (count == 0) return EGO_OK;
    if (hipEventRecord(c->ev, compute_stream) != hipSuccess || hipStreamWaitEvent(comm_stream, c->ev, 0) != hipSuccess) return EGO_ERR_LAUNCH;
    Rccl& r = rccl();
    const long shard = count / c->world;
    int rc = 0;
    if (algo == 1 && shard > 0) {
        float* mine = buf + (long)c->rank * shard;
        rc |= r.reduce_scatter(buf, mine, (size_t)shard, RCCL_FLOAT, RCCL_SUM, c->comm, comm_stream);
        rc |= r.allgather(mine, buf, (size_t)shard, RCCL_FLOAT, c->comm, comm_stream);
        const long rest = count - shard * c->world;
        if (rest > 0) rc |= r.allreduce(buf + shard * c->world, buf + shard * c->world, (size_t)rest, RCCL_FLOAT, RCCL_SUM, c->comm, comm_stream);
    } else {
        rc |= r.allreduce(buf, buf, (size_t)count, RCCL_FLOAT, RCCL_SUM, c->comm, comm_stream);
    }
    return rc == 0 ? EGO_OK : EGO_ERR_LAUNCH;
}

// compute_stream waits (on the device, the host does not block) for everything enqueued on comm_stream so far
extern "C" int ego_dp_wait(ego_dp_comm* c, hipStream_t compute_stream, hipStream_t comm_stream) {
    if (!c) return EGO_ERR_ARG;
    if (hipEventRecord(c->ev, comm_stream) != hipSuccess || hipStreamWaitEvent(compute_stream, c->ev, 0) != hipSuccess) return EGO_ERR_LAUNCH;
    return EGO_OK;
}
