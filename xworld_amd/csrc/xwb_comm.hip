// xwb_comm.hip -- the multi-GPU exchange of libxwb.so: RCCL directly, below any Python (include/xwb.h, "multi-GPU").
//
// The reference scales out with one OS process per environment behind a TCP server (examples/demo_interface.cpp:67-95,
// simulator_interface.cpp:170-313: every process ships its StatePacket to the trainer).  Here one xwb_sim per GPU holds a
// shard of the batch (contiguous global env ids, xwb_config.env_gid0) and the per-step exchange is one RCCL operation per
// shard over xGMI: an all-gather of (reward, game_over) -- 8 bytes per env --, and the north star's "one contiguous
// observation tensor": every remote shard's frames sent straight into its slice of the root's tensor (ncclSend / ncclRecv
// inside one group: each remote GPU has ONE direct xGMI link to the root, so this is link-bound by construction and is
// issued on the communicator's own stream, beside the next step's kernels).
//
// RCCL is resolved at run time (dlopen of the librccl.so.1 already in the process -- PyTorch ships its own -- else the
// system's): libxwb.so itself has no RCCL dependency, and a communicator the caller created with that same RCCL can be
// adopted as it is.
#include "../../include/xwb.h"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <dlfcn.h>

#include <cstring>
#include <functional>
#include <map>
#include <string>
#include <vector>

extern "C" __attribute__((visibility("hidden"))) int xwb_internal_fail(int code, const char *msg);     // xwb_create.hip: sets xwb_last_error on this thread
// xwb_verbs.hip: the last step's rows of the results ring, `beside` ordered behind that step's kernel (1: through its epoch, 0: an event)
extern "C" __attribute__((visibility("hidden"))) int xwb_internal_last_results(xwb_sim *s, void *beside, void *step_stream, const float **rows, int32_t *n);

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetVersion)(int *) = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
    std::string error;
};

// resolved once per process; the function-local static's initialiser runs exactly once even when several threads create
// their first communicator together (one thread per GPU is the pattern this API targets)
Rccl *rccl() {
    static Rccl r = [] {
        Rccl t;
        // the RCCL this process already runs (a communicator handed to xwb_comm_adopt belongs to it), else the system's
        const char *names[] = {"librccl.so.1", "librccl.so"};
        for (const char *n : names) if (!t.handle) t.handle = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
        for (const char *n : names) if (!t.handle) t.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        if (!t.handle) { const char *e = dlerror(); t.error = std::string("librccl.so.1 not found: ") + (e ? e : ""); return t; }
        bool ok = true;
        auto sym = [&](const char *name) { void *p = dlsym(t.handle, name); if (!p) { ok = false; t.error = std::string("RCCL lacks ") + name; } return p; };
#define XWB_RCCL_SYM(field, name) t.field = reinterpret_cast<decltype(t.field)>(sym(name))
        XWB_RCCL_SYM(GetVersion, "ncclGetVersion"); XWB_RCCL_SYM(GetUniqueId, "ncclGetUniqueId"); XWB_RCCL_SYM(CommInitRank, "ncclCommInitRank");
        XWB_RCCL_SYM(CommDestroy, "ncclCommDestroy"); XWB_RCCL_SYM(CommCount, "ncclCommCount"); XWB_RCCL_SYM(CommUserRank, "ncclCommUserRank");
        XWB_RCCL_SYM(GroupStart, "ncclGroupStart"); XWB_RCCL_SYM(GroupEnd, "ncclGroupEnd"); XWB_RCCL_SYM(Send, "ncclSend"); XWB_RCCL_SYM(Recv, "ncclRecv");
        XWB_RCCL_SYM(AllGather, "ncclAllGather"); XWB_RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef XWB_RCCL_SYM
        if (!ok) t.handle = nullptr;
        return t;
    }();
    return &r;
}

int fail(int code, const std::string &msg) { return xwb_internal_fail(code, msg.c_str()); }

#define RCCL_TRY(expr)                                                                                     \
    do {                                                                                                   \
        ncclResult_t _e = (expr);                                                                          \
        if (_e != ncclSuccess) return fail(XWB_ERR_HIP, std::string(#expr) + ": " + R->GetErrorString(_e)); \
    } while (0)
#define HIP_TRY(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t _e = (expr);                                                                            \
        if (_e != hipSuccess) return fail(XWB_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

struct DeviceGuard {
    int prev = -1;
    bool changed = false;
    explicit DeviceGuard(int dev) {
        if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; }
        if (prev != dev) changed = hipSetDevice(dev) == hipSuccess;
    }
    ~DeviceGuard() { if (changed && prev >= 0) (void)hipSetDevice(prev); }
};

}  // namespace

struct xwb_comm {
    ncclComm_t comm = nullptr;
    bool owned = false;                    // created by xwb_comm_init_rank (destroyed with the object)
    int world = 1, rank = 0, device = 0;
    hipStream_t stream = nullptr;          // the exchange runs here, beside the caller's stream
    hipEvent_t ready = nullptr, done = nullptr;
    int in_flight = 0;                     // begins since the last end (several shards may share one communicator)
    int group_depth = 0;                   // xwb_comm_group_start calls not yet ended: RCCL enqueues their operations at the
    std::vector<std::function<int()>> after_group;   // outermost end, so what must FOLLOW a transfer on `stream` waits here
    hipEvent_t marks[XWB_COMM_MARKS] = {};
    bool mark_set[XWB_COMM_MARKS] = {};
    // xwb_gather_grids_begin: per batch that gathers through this communicator (one, unless shards share a rank), two staging
    // slabs that alternate (the draw state a shard sends; on the root the gathered state its render reads);
    // done[k] = the transfer / render that last read slab k
    struct Slabs {
        void *slab[2] = {nullptr, nullptr};
        size_t bytes[2] = {0, 0};
        hipEvent_t done[2] = {nullptr, nullptr};
        bool busy[2] = {false, false};
        bool queued[2] = {false, false};      // begun inside an open group: its done[k] is only recorded when that group ends
        int next = 0;
    };
    std::map<const xwb_sim *, Slabs> slabs;
};

// runs `f` behind the transfers begun so far on c->stream: now, or when the caller's outermost group ends
static int after_transfers(xwb_comm *c, std::function<int()> f) {
    if (c->group_depth > 0) { c->after_group.push_back(std::move(f)); return XWB_OK; }
    return f();
}

extern "C" {

int xwb_comm_version(int32_t *version) {
    if (!version) return fail(XWB_ERR_ARG, "NULL argument");
    Rccl *R = rccl();
    if (!R->handle) return fail(XWB_ERR_STATE, R->error);
    int v = 0;
    RCCL_TRY(R->GetVersion(&v));
    *version = v;
    return XWB_OK;
}

int xwb_comm_unique_id(uint8_t out[XWB_COMM_ID_BYTES]) {
    if (!out) return fail(XWB_ERR_ARG, "NULL argument");
    Rccl *R = rccl();
    if (!R->handle) return fail(XWB_ERR_STATE, R->error);
    static_assert(sizeof(ncclUniqueId) == XWB_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    RCCL_TRY(R->GetUniqueId(&id));
    memcpy(out, &id, sizeof id);
    return XWB_OK;
}

static int finish_comm(xwb_comm *c) {
    DeviceGuard g(c->device);
    {   // the communicator's stream comes from the HIGH-priority pool of hardware queues: streams of one priority share a few
        // hardware queues in creation order, and a transfer that waits for an event at the head of the queue a batch's map
        // generator uses holds the generator back until the render it should run beside is over ("queues")
        int least = 0, greatest = 0;
        if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) { (void)hipGetLastError(); greatest = 0; }
        HIP_TRY(hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, greatest));
    }
    HIP_TRY(hipEventCreateWithFlags(&c->ready, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&c->done, hipEventDisableTiming));
    for (auto &m : c->marks) HIP_TRY(hipEventCreateWithFlags(&m, hipEventDisableTiming));
    return XWB_OK;
}

int xwb_comm_init_rank(const uint8_t id_bytes[XWB_COMM_ID_BYTES], int32_t world, int32_t rank, int32_t device, xwb_comm **out) {
    if (!id_bytes || !out) return fail(XWB_ERR_ARG, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(XWB_ERR_ARG, "need 0 <= rank < world");
    Rccl *R = rccl();
    if (!R->handle) return fail(XWB_ERR_STATE, R->error);
    DeviceGuard g(device);
    ncclUniqueId id;
    memcpy(&id, id_bytes, sizeof id);
    xwb_comm *c = new xwb_comm();
    c->world = world; c->rank = rank; c->device = device; c->owned = true;
    ncclResult_t e = R->CommInitRank(&c->comm, world, id, rank);
    if (e != ncclSuccess) { delete c; return fail(XWB_ERR_HIP, std::string("ncclCommInitRank: ") + R->GetErrorString(e)); }
    const int rc = finish_comm(c);
    if (rc) { xwb_comm_destroy(c); return rc; }
    *out = c;
    return XWB_OK;
}

int xwb_comm_adopt(void *nccl_comm, int32_t device, xwb_comm **out) {
    if (!nccl_comm || !out) return fail(XWB_ERR_ARG, "NULL argument");
    Rccl *R = rccl();
    if (!R->handle) return fail(XWB_ERR_STATE, R->error);
    xwb_comm *c = new xwb_comm();
    c->comm = static_cast<ncclComm_t>(nccl_comm); c->device = device; c->owned = false;
    ncclResult_t e = R->CommCount(c->comm, &c->world);
    if (e == ncclSuccess) e = R->CommUserRank(c->comm, &c->rank);
    if (e != ncclSuccess) { delete c; return fail(XWB_ERR_HIP, std::string("ncclCommCount / ncclCommUserRank: ") + R->GetErrorString(e)); }
    const int rc = finish_comm(c);
    if (rc) { xwb_comm_destroy(c); return rc; }
    *out = c;
    return XWB_OK;
}

int xwb_comm_destroy(xwb_comm *c) {
    if (!c) return XWB_OK;
    Rccl *R = rccl();
    DeviceGuard g(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->owned && c->comm && R->handle) (void)R->CommDestroy(c->comm);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->ready) (void)hipEventDestroy(c->ready);
    if (c->done) (void)hipEventDestroy(c->done);
    for (auto m : c->marks) if (m) (void)hipEventDestroy(m);
    for (auto &kv : c->slabs)
        for (int k = 0; k < 2; ++k) {
            if (kv.second.done[k]) (void)hipEventDestroy(kv.second.done[k]);
            if (kv.second.slab[k]) (void)hipFree(kv.second.slab[k]);
        }
    delete c;
    return XWB_OK;
}

int xwb_comm_info(const xwb_comm *c, int32_t *world, int32_t *rank) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    if (world) *world = c->world;
    if (rank) *rank = c->rank;
    return XWB_OK;
}

int xwb_comm_group_start(xwb_comm *c) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    Rccl *R = rccl();
    RCCL_TRY(R->GroupStart());
    c->group_depth += 1;
    return XWB_OK;
}

int xwb_comm_group_end(xwb_comm *c) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    if (c->group_depth < 1) return fail(XWB_ERR_STATE, "xwb_comm_group_end without xwb_comm_group_start");
    Rccl *R = rccl();
    c->group_depth -= 1;
    ncclResult_t e = R->GroupEnd();
    if (c->group_depth > 0) { if (e != ncclSuccess) return fail(XWB_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(e)); return XWB_OK; }
    std::vector<std::function<int()>> todo;
    todo.swap(c->after_group);
    if (e != ncclSuccess) return fail(XWB_ERR_HIP, std::string("ncclGroupEnd: ") + R->GetErrorString(e));
    DeviceGuard g(c->device);
    for (auto &f : todo) { const int rc = f(); if (rc) return rc; }      // the transfers are on c->stream now
    return XWB_OK;
}

int xwb_comm_mark(xwb_comm *c, int32_t slot) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    if (slot < 0 || slot >= XWB_COMM_MARKS) return fail(XWB_ERR_ARG, "mark slot out of range");
    DeviceGuard g(c->device);
    return after_transfers(c, [c, slot]() -> int {
        HIP_TRY(hipEventRecord(c->marks[slot], c->stream));
        c->mark_set[slot] = true;
        return XWB_OK;
    });
}

int xwb_comm_wait(xwb_comm *c, int32_t slot, void *stream) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    if (slot < 0 || slot >= XWB_COMM_MARKS) return fail(XWB_ERR_ARG, "mark slot out of range");
    if (c->group_depth > 0) return fail(XWB_ERR_STATE, "xwb_comm_wait inside an open group: its marks are not recorded yet");
    if (!c->mark_set[slot]) return XWB_OK;
    DeviceGuard g(c->device);
    HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->marks[slot], 0));
    return XWB_OK;
}

// The layout of a gather: shard i holds counts[i] envs and lives on communicator rank peers[i] (NULL: rank i).
static int check_layout(const xwb_comm *c, const int32_t *counts, const int32_t *peers, int32_t n_shards, int32_t shard) {
    if (!c || !counts) return fail(XWB_ERR_ARG, "NULL argument");
    if (n_shards < 1 || shard < 0 || shard >= n_shards) return fail(XWB_ERR_ARG, "need 0 <= shard < n_shards");
    for (int i = 0; i < n_shards; ++i) {
        if (counts[i] < 0) return fail(XWB_ERR_ARG, "negative shard size");
        const int p = peers ? peers[i] : i;
        if (p < 0 || p >= c->world) return fail(XWB_ERR_ARG, "a shard's peer is not a rank of the communicator");
    }
    return XWB_OK;
}

int xwb_gather_results(xwb_comm *c, const float *packed_dev, float *all_dev, const int32_t *counts, const int32_t *peers,
                       int32_t n_shards, int32_t shard, void *stream) {
    int rc = check_layout(c, counts, peers, n_shards, shard);
    if (rc) return rc;
    if (!packed_dev || !all_dev) return fail(XWB_ERR_ARG, "NULL argument");
    Rccl *R = rccl();
    DeviceGuard g(c->device);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    bool equal = !peers && n_shards == c->world;
    for (int i = 1; i < n_shards; ++i) equal = equal && counts[i] == counts[0];
    if (equal) {                                        // the plainest collective there is: 8 bytes per env
        RCCL_TRY(R->AllGather(packed_dev, all_dev, (size_t)counts[0] * 2, ncclFloat32, c->comm, st));
        return XWB_OK;
    }
    // ragged shards / several shards per rank, one group.  Both sides apply the same two rules, so every send has its receive:
    //   shard a SENDS its rows to the rank of every shard b that lives on another rank, iff counts[a] > 0;
    //   shard b RECEIVES the rows of every shard a that lives on another rank, iff counts[a] > 0
    // (an empty shard sends nothing and still receives; rows of shards on the caller's own rank are never sent -- those shards
    // share the caller's all_dev, see xwb.h).  Between two ranks RCCL matches sends and receives in issue order: shards that
    // share a rank call in ascending shard order.
    std::vector<size_t> off(n_shards + 1, 0);
    for (int i = 0; i < n_shards; ++i) off[i + 1] = off[i] + (size_t)counts[i];
    const int me = peers ? peers[shard] : shard;
    if (counts[shard] > 0)
        HIP_TRY(hipMemcpyAsync(all_dev + off[shard] * 2, packed_dev, (size_t)counts[shard] * 8, hipMemcpyDeviceToDevice, st));
    RCCL_TRY(R->GroupStart());
    for (int i = 0; i < n_shards; ++i) {
        const int p = peers ? peers[i] : i;
        if (i == shard || p == me) continue;
        ncclResult_t e = ncclSuccess;
        if (counts[i] > 0) e = R->Recv(all_dev + off[i] * 2, (size_t)counts[i] * 2, ncclFloat32, p, c->comm, st);
        if (e == ncclSuccess && counts[shard] > 0) e = R->Send(packed_dev, (size_t)counts[shard] * 2, ncclFloat32, p, c->comm, st);
        if (e != ncclSuccess) { (void)R->GroupEnd(); return fail(XWB_ERR_HIP, std::string("ncclSend / ncclRecv: ") + R->GetErrorString(e)); }
    }
    RCCL_TRY(R->GroupEnd());
    return XWB_OK;
}

int xwb_gather_results_beside(xwb_sim *sim, xwb_comm *c, float *all_dev, const int32_t *counts, const int32_t *peers, int32_t n_shards,
                              int32_t shard, void *stream, int32_t *by_epoch) {
    int rc = check_layout(c, counts, peers, n_shards, shard);
    if (rc) return rc;
    if (!sim || !all_dev) return fail(XWB_ERR_ARG, "NULL argument");
    if (c->group_depth > 0) return fail(XWB_ERR_STATE, "xwb_gather_results_beside inside an open group");
    DeviceGuard g(c->device);
    const float *rows = nullptr;
    int32_t n = 0;
    // validate before anything is enqueued: a refused call must leave no wait kernel / event behind on either stream
    if ((rc = xwb_num_envs(sim, &n))) return rc;
    if (n != counts[shard]) return fail(XWB_ERR_ARG, "counts[shard] is not this batch's num_envs");
    rc = xwb_internal_last_results(sim, c->stream, stream, &rows, &n);
    if (rc < 0) return rc;
    if (by_epoch) *by_epoch = rc;
    rc = xwb_gather_results(c, rows, all_dev, counts, peers, n_shards, shard, c->stream);
    if (rc == XWB_OK) c->in_flight += 1;                     // (an exchange that was refused is not one xwb_gather_screens_end may wait for)
    return rc;
}

// what the two gathers of frames share: arguments, the hand-over from `stream` to the communicator's stream
struct GatherCtx {
    int32_t n = 0;
    size_t bpe = 0;
    void *obs = nullptr;
    int root_peer = 0, me = 0;
    hipStream_t st = nullptr;
};

static int gather_prologue(xwb_sim *sim, xwb_comm *c, void *dst_dev, const int32_t *counts, const int32_t *peers, int32_t n_shards,
                           int32_t shard, int32_t root_shard, void *stream, GatherCtx *x) {
    int rc = check_layout(c, counts, peers, n_shards, shard);
    if (rc) return rc;
    if (!sim) return fail(XWB_ERR_ARG, "NULL argument");
    if (root_shard < 0 || root_shard >= n_shards) return fail(XWB_ERR_ARG, "root_shard out of range");
    if ((rc = xwb_obs_dev(sim, &x->obs, &x->bpe)) || (rc = xwb_num_envs(sim, &x->n))) return rc;
    if (x->n != counts[shard]) return fail(XWB_ERR_ARG, "counts[shard] is not this batch's num_envs");
    if (shard == root_shard && !dst_dev) return fail(XWB_ERR_ARG, "the root needs the destination tensor");
    x->root_peer = peers ? peers[root_shard] : root_shard;
    x->me = peers ? peers[shard] : shard;
    x->st = reinterpret_cast<hipStream_t>(stream);
    // A shard on the root's own rank reaches the root by a send to that rank, which only a receive of the SAME group matches
    bool self_exchange = shard != root_shard && x->me == x->root_peer;
    if (shard == root_shard)
        for (int i = 0; i < n_shards; ++i) self_exchange = self_exchange || (i != root_shard && counts[i] > 0 && (peers ? peers[i] : i) == x->root_peer);
    if (self_exchange && c->group_depth < 1)
        return fail(XWB_ERR_STATE, "shards that share the root's rank gather inside xwb_comm_group_start / _end");
    return XWB_OK;
}

int xwb_gather_screens_begin(xwb_sim *sim, xwb_comm *c, void *dst_dev, const int32_t *counts, const int32_t *peers, int32_t n_shards,
                             int32_t shard, int32_t root_shard, void *stream) {
    GatherCtx x;
    int rc = gather_prologue(sim, c, dst_dev, counts, peers, n_shards, shard, root_shard, stream, &x);
    if (rc) return rc;
    Rccl *R = rccl();
    DeviceGuard g(c->device);
    const int32_t n = x.n;
    const size_t bpe = x.bpe;
    // the frames are complete in `stream` order; the transfer runs on the communicator's stream from there on
    HIP_TRY(hipEventRecord(c->ready, x.st));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    std::vector<size_t> off(n_shards + 1, 0);
    for (int i = 0; i < n_shards; ++i) off[i + 1] = off[i] + (size_t)counts[i] * bpe;
    uint8_t *dst = static_cast<uint8_t *>(dst_dev);
    if (shard == root_shard) {
        if (dst + off[shard] != x.obs)                // (bound into its slice with xwb_bind_obs: nothing to copy)
            HIP_TRY(hipMemcpyAsync(dst + off[shard], x.obs, (size_t)n * bpe, hipMemcpyDeviceToDevice, c->stream));
        RCCL_TRY(R->GroupStart());
        for (int i = 0; i < n_shards; ++i) {
            if (i == root_shard || counts[i] == 0) continue;
            const int p = peers ? peers[i] : i;
            ncclResult_t e = R->Recv(dst + off[i], (size_t)counts[i] * bpe, ncclUint8, p, c->comm, c->stream);
            if (e != ncclSuccess) { (void)R->GroupEnd(); return fail(XWB_ERR_HIP, std::string("ncclRecv: ") + R->GetErrorString(e)); }
        }
        RCCL_TRY(R->GroupEnd());
    } else if (n > 0) {
        RCCL_TRY(R->Send(x.obs, (size_t)n * bpe, ncclUint8, x.root_peer, c->comm, c->stream));
    }
    c->in_flight += 1;          // (the completion event is recorded by xwb_gather_screens_end: inside a caller's group the
    return XWB_OK;              //  operations above are only enqueued when the outermost group ends)
}

int xwb_gather_grids_begin(xwb_sim *sim, xwb_comm *c, void *dst_dev, const int32_t *counts, const int32_t *peers, int32_t n_shards,
                           int32_t shard, int32_t root_shard, void *stream) {
    GatherCtx x;
    int rc = gather_prologue(sim, c, dst_dev, counts, peers, n_shards, shard, root_shard, stream, &x);
    if (rc) return rc;
    double X = 0, Y = 0;
    if ((rc = xwb_get_world_dimensions(sim, &X, &Y, nullptr))) return rc;      // (fails for the games that have no grid)
    const size_t cells = (size_t)X * (size_t)Y;
    Rccl *R = rccl();
    DeviceGuard g(c->device);
    const bool root = shard == root_shard;
    size_t total = 0, first = 0;
    for (int i = 0; i < n_shards; ++i) { if (i < shard) first += (size_t)counts[i]; total += (size_t)counts[i]; }
    // slab layout: cell codes uint16[envs][cells], then one flag byte per env; the root's slab holds every shard's rows
    const size_t envs = root ? total : (size_t)x.n;
    const size_t grid_bytes = (envs * cells * 2 + 15) & ~(size_t)15, need = grid_bytes + ((envs + 15) & ~(size_t)15);
    xwb_comm::Slabs &sl = c->slabs[sim];
    const int k = sl.next;                    // (committed below, once nothing can fail any more before the transfers are posted)
    // a third begin for one batch inside ONE open group would pack into a slab whose transfer is still only queued -- its
    // completion event is not recorded yet, so waiting for it would wait for nothing
    if (sl.queued[k])
        return fail(XWB_ERR_STATE, "more than two gathers of one batch inside one open group: close the group first");
    if (!sl.done[k]) HIP_TRY(hipEventCreateWithFlags(&sl.done[k], hipEventDisableTiming));
    if (sl.bytes[k] < need) {
        if (c->group_depth > 0 && sl.busy[k])
            return fail(XWB_ERR_STATE, "the gather's layout grew inside an open group while a transfer of this batch is still queued");
        HIP_TRY(hipStreamSynchronize(c->stream));                              // (grows on first use / when the layout grows)
        if (sl.slab[k]) HIP_TRY(hipFree(sl.slab[k]));
        sl.slab[k] = nullptr; sl.bytes[k] = 0; sl.busy[k] = false;
        HIP_TRY(hipMalloc(&sl.slab[k], need));
        sl.bytes[k] = need;
    }
    // the transfer (root: the render) that last read this slab is over before this step's state is packed into it
    if (sl.busy[k]) HIP_TRY(hipStreamWaitEvent(x.st, sl.done[k], 0));
    uint16_t *grids = static_cast<uint16_t *>(sl.slab[k]);
    uint8_t *flags = static_cast<uint8_t *>(sl.slab[k]) + grid_bytes;
    const size_t row0 = root ? first : 0;
    // (a pack that is refused -- e.g. context > 1 with two draws since the last pack -- leaves the slab rotation as it was and
    // posts nothing; the OTHER ranks cannot know: like any collective, a gather one rank drops out of has to be abandoned by all)
    if (x.n > 0 && (rc = xwb_xw_pack_grids(sim, grids + row0 * cells, flags + row0, x.st))) return rc;
    sl.next = k ^ 1;
    HIP_TRY(hipEventRecord(c->ready, x.st));
    HIP_TRY(hipStreamWaitEvent(c->stream, c->ready, 0));
    if (root) {
        RCCL_TRY(R->GroupStart());
        size_t row = 0;
        for (int i = 0; i < n_shards; ++i) {
            const size_t cnt = (size_t)counts[i];
            if (i != root_shard && cnt > 0) {
                const int p = peers ? peers[i] : i;
                ncclResult_t e = R->Recv(grids + row * cells, cnt * cells * 2, ncclUint8, p, c->comm, c->stream);
                if (e == ncclSuccess) e = R->Recv(flags + row, cnt, ncclUint8, p, c->comm, c->stream);
                if (e != ncclSuccess) { (void)R->GroupEnd(); return fail(XWB_ERR_HIP, std::string("ncclRecv: ") + R->GetErrorString(e)); }
            }
            row += cnt;
        }
        RCCL_TRY(R->GroupEnd());
    } else if (x.n > 0) {
        RCCL_TRY(R->GroupStart());
        ncclResult_t e = R->Send(grids, (size_t)x.n * cells * 2, ncclUint8, x.root_peer, c->comm, c->stream);
        if (e == ncclSuccess) e = R->Send(flags, (size_t)x.n, ncclUint8, x.root_peer, c->comm, c->stream);
        if (e != ncclSuccess) { (void)R->GroupEnd(); return fail(XWB_ERR_HIP, std::string("ncclSend: ") + R->GetErrorString(e)); }
        RCCL_TRY(R->GroupEnd());
    }
    c->in_flight += 1;
    sl.busy[k] = true;
    sl.queued[k] = c->group_depth > 0;
    hipEvent_t slab_done = sl.done[k];
    // behind the transfers: the root draws the whole batch from the gathered state; the slab is free again after that
    return after_transfers(c, [=]() -> int {
        auto it = c->slabs.find(sim);
        if (it != c->slabs.end()) it->second.queued[k] = false;
        if (root && total > 0) {
            const int rr = xwb_xw_render_grids(sim, grids, flags, (int32_t)total, dst_dev, c->stream);
            if (rr) return rr;
        }
        HIP_TRY(hipEventRecord(slab_done, c->stream));
        return XWB_OK;
    });
}

int xwb_comm_release_sim(xwb_comm *c, const xwb_sim *sim) {
    if (!c || !sim) return fail(XWB_ERR_ARG, "NULL argument");
    if (c->group_depth > 0) return fail(XWB_ERR_STATE, "xwb_comm_release_sim inside an open group: its transfers are not enqueued yet");
    auto it = c->slabs.find(sim);
    if (it == c->slabs.end()) return XWB_OK;
    DeviceGuard g(c->device);
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (int k = 0; k < 2; ++k) {
        if (it->second.done[k]) (void)hipEventDestroy(it->second.done[k]);
        if (it->second.slab[k]) (void)hipFree(it->second.slab[k]);
    }
    c->slabs.erase(it);
    return XWB_OK;
}

int xwb_gather_screens_end(xwb_comm *c, void *stream) {
    if (!c) return fail(XWB_ERR_ARG, "NULL argument");
    if (c->group_depth > 0) return fail(XWB_ERR_STATE, "xwb_gather_screens_end inside an open group: the transfers are not enqueued yet");
    if (!c->in_flight) return XWB_OK;
    DeviceGuard g(c->device);
    HIP_TRY(hipEventRecord(c->done, c->stream));
    HIP_TRY(hipStreamWaitEvent(reinterpret_cast<hipStream_t>(stream), c->done, 0));
    c->in_flight = 0;
    return XWB_OK;
}

}  // extern "C"
