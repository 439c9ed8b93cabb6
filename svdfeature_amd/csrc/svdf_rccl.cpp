// svdf_rccl.cpp -- the exchange of the one-process-per-GPU ranks issued from C++ straight into RCCL: the same steps as
// multi_gpu.HipShard runs through torch.distributed, without a Python call or a c10d work object per collective.
//
// Why: at N = 8 a rank's share of a pass is ~3 ms of GPU work cut into 64 stratum steps, each followed by a point-to-point hand-over of one item
// block.  The host thread must ENQUEUE a step faster than the GPU runs it; through torch a hand-over is a batch_isend_irecv (two P2POps, a
// coalescing manager, a work object, a stream wait) plus ~10 ctypes calls, which is the same order as the 47 us the step takes on the GPU.  Here
// a hand-over is one C call: copy-out kernel, two event waits, ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd on a side stream, one event.
//
//   svdf_rccl_unique_id(out[128])                 rank 0: ncclGetUniqueId; the bytes travel through the caller's process group / store once
//   svdf_rccl_init(t, id, rank, world)            ncclCommInitRank on the trainer's device, side stream + events
//   all-reduce step, per window:                  svdf_train_dataset; svdf_rccl_window_allreduce (pack into the rank's wire buffer, ncclAllReduce in
//                                                 place on the trainer's stream, k_delta_addto)
//   stratified hand-over, per stratum step:       svdf_rccl_block_handoff(t, dst, src, slot, in_block, nblocks): the ACTIVE item block (svdf_item_delta_select)
//                                                 goes to rank dst while block in_block arrives from rank src into inbox slot `slot`;
//                                                 svdf_rccl_block_arrive(t, slot): the trainer's stream waits for that transfer and puts the block in place
// Ordering is by HIP events only (the host never waits): a send buffer is reused after the transfer that read it, an inbox slot after the copy
// that emptied it.  librccl.so is resolved at run time (inside a torch process: the library torch already loaded).
// Replaces, like the rest of section 6, what ONE process does in /root/reference/svd_feature.cpp:220-248 (the round loop) on N ranks.
#include <dlfcn.h>

#include <cstring>
#include <string>
#include <vector>

#include "svdf_engine.h"
#include "svdf_kernels.h"

namespace svdf {

#define HIPCHECK(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)
static inline void check(bool ok, const char *msg) { if (!ok) fail(msg); }

namespace {
struct RcclApi {   // rccl.h: ncclUniqueId = 128 bytes; ncclHalf = 6, ncclFloat = 7, ncclSum = 0
    void *lib = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, const void *, int) = nullptr;   // the id is passed BY VALUE in rccl.h (a 128-byte struct): see init()
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    void *init_rank_sym = nullptr;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        GetUniqueId = (int (*)(void *))dlsym(lib, "ncclGetUniqueId");
        init_rank_sym = dlsym(lib, "ncclCommInitRank");
        AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
        Send = (int (*)(const void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclSend");
        Recv = (int (*)(void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclRecv");
        GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
        GetErrorString = (const char *(*)(int))dlsym(lib, "ncclGetErrorString");
        return GetUniqueId && init_rank_sym && AllReduce && Send && Recv && GroupStart && GroupEnd && CommDestroy;
    }
    void ok(int rc, const char *what) const {
        if (rc != 0) fail(std::string("svdfeature_amd: RCCL ") + what + " failed: " + (GetErrorString ? GetErrorString(rc) : "error"));
    }
};
struct UniqueId { char bytes[128]; };   // ncclUniqueId
RcclApi &api() { static RcclApi a; return a; }
}  // namespace

struct RcclState {
    void *comm = nullptr;
    int rank = -1, world = 0;
    hipStream_t xfer = nullptr;
    hipEvent_t out_ready[2] = {nullptr, nullptr}, moved[2] = {nullptr, nullptr}, consumed[2] = {nullptr, nullptr};
    bool moved_set[2] = {false, false}, consumed_set[2] = {false, false}, in_flight[2] = {false, false};
    DevBuf<float> out[2], in[2];
    DevBuf<char> wire;
    int64_t handoffs = 0, allreduces = 0;
};
void RcclDeleter::operator()(RcclState *s) const {
    if (!s) return;
    if (s->xfer) (void)hipStreamSynchronize(s->xfer);
    if (s->comm && api().CommDestroy) (void)api().CommDestroy(s->comm);
    for (int j = 0; j < 2; j++) {
        if (s->out_ready[j]) (void)hipEventDestroy(s->out_ready[j]);
        if (s->moved[j]) (void)hipEventDestroy(s->moved[j]);
        if (s->consumed[j]) (void)hipEventDestroy(s->consumed[j]);
    }
    if (s->xfer) (void)hipStreamDestroy(s->xfer);
    delete s;
}

void rccl_unique_id(unsigned char *out128) {
    if (!api().load()) fail("svdfeature_amd: librccl.so could not be loaded");
    UniqueId id;
    memset(&id, 0, sizeof(id));
    api().ok(api().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(out128, id.bytes, 128);
}

void Engine::rccl_init(const unsigned char *id128, int rank, int world) {
    need_device("rccl_init");
    check(world >= 1 && rank >= 0 && rank < world, "svdf_rccl_init: rank / world out of range");
    check(!multi_, "svdf_rccl_*: the per-rank exchange of the one-process-per-GPU scheme; an amd:gpus handle exchanges by itself");
    if (!api().load()) fail("svdfeature_amd: librccl.so could not be loaded");
    flush();
    rccl_.reset(new RcclState());
    RcclState &S = *rccl_;
    S.rank = rank; S.world = world;
    HIPCHECK(hipSetDevice(device_));
    UniqueId id;
    memcpy(id.bytes, id128, 128);
    // ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId commId, int rank): the id travels by value
    auto init = reinterpret_cast<int (*)(void **, int, UniqueId, int)>(api().init_rank_sym);
    api().ok(init(&S.comm, world, id, rank), "ncclCommInitRank");
    {   // the transfers ride a HIGH-priority stream: their few workgroups must not queue behind the training kernels that fill the chip
        int lo = 0, hi = 0;
        HIPCHECK(hipDeviceGetStreamPriorityRange(&lo, &hi));
        HIPCHECK(hipStreamCreateWithPriority(&S.xfer, hipStreamNonBlocking, hi));
    }
    for (int j = 0; j < 2; j++) {
        HIPCHECK(hipEventCreateWithFlags(&S.out_ready[j], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&S.moved[j], hipEventDisableTiming));
        HIPCHECK(hipEventCreateWithFlags(&S.consumed[j], hipEventDisableTiming));
    }
}
void Engine::rccl_check(const char *what) {
    if (!rccl_) fail(std::string("svdf_rccl_") + what + ": call svdf_rccl_init first");
}

// the all-reduce step's exchange for the window trained last: per-item sums into the rank's wire buffer, ncclAllReduce (SUM) in place on the
// trainer's stream, replicated ranges += the sum
void Engine::rccl_window_allreduce(Dataset *ds, int half) {
    rccl_check("window_allreduce");
    need_device("rccl");
    RcclState &S = *rccl_;
    int64_t count = 0;
    window_delta_pack(ds, nullptr, half, &count);
    S.wire.reserve((size_t)count * (half ? 2 : 4) + 256);
    window_delta_pack(ds, S.wire.p, half, nullptr);
    api().ok(api().AllReduce(S.wire.p, S.wire.p, (size_t)count, half ? 6 : 7, 0, S.comm, stream_), "ncclAllReduce");
    window_delta_apply(S.wire.p, half);
    S.allreduces++;
}

// Stratified hand-over.  The caller has selected the block that LEAVES (svdf_item_delta_select); in_block / nblocks name the block that arrives.
void Engine::rccl_block_handoff(int dst, int src, int slot, int in_block, int nblocks) {
    rccl_check("block_handoff");
    need_device("rccl");
    RcclState &S = *rccl_;
    check(dst >= 0 && dst < S.world && src >= 0 && src < S.world && (slot == 0 || slot == 1), "svdf_rccl_block_handoff: bad rank / slot");
    check(!S.in_flight[slot], "svdf_rccl_block_handoff: the block that arrived in this inbox slot has not been put in place yet (two hand-overs in flight at most)");
    int64_t n_out = 0, n_in = 0;
    item_block_copy(nullptr, 0, &n_out);
    {
        const int keep_part = delta_part_, keep_n = delta_nparts_;
        item_delta_select(in_block, nblocks);
        item_block_copy(nullptr, 0, &n_in);
        item_delta_select(keep_part, keep_n);
    }
    S.out[slot].reserve((size_t)std::max<int64_t>(n_out, 1));
    S.in[slot].reserve((size_t)std::max<int64_t>(n_in, 1));
    // the send buffer of this slot is free once the transfer that read it is over; the trainer's stream copies the block out behind that
    if (S.moved_set[slot]) HIPCHECK(hipStreamWaitEvent(stream_, S.moved[slot], 0));
    item_block_copy(S.out[slot].p, 0, nullptr);
    HIPCHECK(hipEventRecord(S.out_ready[slot], stream_));
    // the transfer: behind the copy-out, and behind the copy that emptied this inbox slot the last time
    HIPCHECK(hipStreamWaitEvent(S.xfer, S.out_ready[slot], 0));
    if (S.consumed_set[slot]) HIPCHECK(hipStreamWaitEvent(S.xfer, S.consumed[slot], 0));
    api().ok(api().GroupStart(), "ncclGroupStart");
    const int rs = api().Send(S.out[slot].p, (size_t)n_out, 7, dst, S.comm, S.xfer);
    const int rr = rs == 0 ? api().Recv(S.in[slot].p, (size_t)n_in, 7, src, S.comm, S.xfer) : 0;
    const int re = api().GroupEnd();   // the group is closed whatever happened inside it
    api().ok(rs, "ncclSend");
    api().ok(rr, "ncclRecv");
    api().ok(re, "ncclGroupEnd");
    HIPCHECK(hipEventRecord(S.moved[slot], S.xfer));
    S.moved_set[slot] = true;
    S.in_flight[slot] = true;
    S.handoffs++;
}
// the block of inbox slot `slot` into place: the caller has selected the partition it belongs to
void Engine::rccl_block_arrive(int slot) {
    rccl_check("block_arrive");
    need_device("rccl");
    RcclState &S = *rccl_;
    check((slot == 0 || slot == 1) && S.in_flight[slot], "svdf_rccl_block_arrive: no hand-over is in flight for this slot");
    HIPCHECK(hipStreamWaitEvent(stream_, S.moved[slot], 0));
    item_block_copy(S.in[slot].p, 1, nullptr);
    HIPCHECK(hipEventRecord(S.consumed[slot], stream_));
    S.consumed_set[slot] = true;
    S.in_flight[slot] = false;
}
int64_t Engine::rccl_counter(int what) const { return !rccl_ ? -1 : what == 0 ? rccl_->handoffs : rccl_->allreduces; }
void Engine::rccl_close() {
    if (!rccl_) return;
    HIPCHECK(hipStreamSynchronize(stream_));
    HIPCHECK(hipStreamSynchronize(rccl_->xfer));
    rccl_.reset();
}

}  // namespace svdf
