// svdf_ipc.cpp -- cross-PROCESS direct exchange of the window-minibatch step: one process per GPU (bench.py --gpus N,
// torch.distributed), but the item-side sums travel through IPC-mapped device buffers instead of RCCL's ring.
//
//   setup    every rank allocates its wire buffer + one flag page and exports both (hipIpcGetMemHandle); the 2 x 64 bytes per rank are
//            exchanged by the caller (all_gather of bytes) and opened with hipIpcOpenMemHandle: every rank then addresses every wire buffer
//            and flag page -- over xGMI on distinct devices, plain device memory when the ranks share a device (the one-GPU tests);
//   pack     k_window_items / k_wunit_sum into the rank's OWN wire buffer, then the rank's sequence number into word (0, me) of every page;
//   reduce   wait until all words (0, r) of the own page carry the sequence number, then k_delta_reduce_gather on slice `me` of ALL wire
//            buffers (sum in rank order in fp32, stored back into slice `me` of all buffers: reduce-scatter + all-gather over every link at
//            once -- the kernel of the amd:gpus handle, svdf_multi.cpp), then the sequence number into word (1, me) of every page;
//   apply    wait for all words (1, r), then k_delta_addto from the own wire buffer.
// Everything is enqueued on the trainer's stream; the host never waits inside a pass.  A wire buffer is not written again before every
// rank has finished reading it: the next pack comes after this window's apply, which waited for every rank's phase-1 word.
// Replaces, like the rest of section 6, what ONE process does in /root/reference/svd_feature.cpp:220-248 (the round loop) on N ranks.
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

struct IpcState {
    int rank = -1, world = 0;
    size_t wire_bytes = 0, block_floats = 0;
    void *wire = nullptr;             // own wire buffer (hipMalloc), IPC-exported
    unsigned *page = nullptr;         // own flag page: [phase 0..5][src rank 0..15][32 words], IPC-exported
                                      // phase 0 / 1: window packed / reduced; 2 / 3: block arrived in inbox slot 0 / 1; 4 / 5: inbox slot 0 / 1 of the
                                      // rank I send to has been consumed (its acknowledgement, so that a slot is never overwritten early)
    unsigned slot_seq[2] = {0u, 0u};  // block sequence number of my last send into slot 0 / 1 of the destination's inbox
    float *inbox = nullptr;           // own block inbox: 2 slots of block_floats (stratified hand-over), part of the wire allocation
    std::vector<void *> peer_wire;    // [world]: peers' wire buffers as mapped here (own entry = wire)
    std::vector<unsigned *> peer_page;
    std::vector<bool> opened;
    unsigned seq = 0, seq_block = 0;
    unsigned *err_host = nullptr;     // pinned, device-visible: raised by a wait kernel that hit its spin limit
    unsigned *err_dev = nullptr;
    unsigned long long spin_limit = 200000000ull;   // ~ several seconds of s_sleep(32) polls
};
void IpcDeleter::operator()(IpcState *s) const {
    if (!s) return;
    for (int r = 0; r < s->world; r++) {
        if (r == s->rank || !s->opened[(size_t)r]) continue;
        (void)hipIpcCloseMemHandle(s->peer_wire[(size_t)r]);
        (void)hipIpcCloseMemHandle(s->peer_page[(size_t)r]);
    }
    if (s->wire) (void)hipFree(s->wire);
    if (s->page) (void)hipFree(s->page);
    if (s->err_host) (void)hipHostFree(s->err_host);
    delete s;
}

static const size_t PAGE_WORDS = 6 * 16 * 32;

// handles_out: 2 x 64 bytes (wire buffer, flag page).  block_floats: size of the largest item block handed over by the stratified
// schedule (0 = none); the inbox (two slots) sits behind the wire bytes in the same allocation.
void Engine::ipc_setup(int rank, int world, int64_t wire_bytes, int64_t block_floats, unsigned char *handles_out) {
    need_device("ipc_setup");
    check(world >= 1 && world <= 16 && rank >= 0 && rank < world, "svdf_ipc_setup: rank / world out of range (at most 16 ranks)");
    check(!multi_, "svdf_ipc_*: the per-rank exchange of the one-process-per-GPU scheme; an amd:gpus handle exchanges by itself");
    check(wire_bytes > 0 && block_floats >= 0, "svdf_ipc_setup: bad sizes");
    flush();
    ipc_.reset(new IpcState());
    IpcState &S = *ipc_;
    S.rank = rank; S.world = world;
    if (ipc_spin_limit_ > 0) S.spin_limit = (unsigned long long)ipc_spin_limit_;
    S.wire_bytes = ((size_t)wire_bytes + 255) & ~(size_t)255;
    S.block_floats = (size_t)block_floats;
    HIPCHECK(hipMalloc(&S.wire, S.wire_bytes + 2 * S.block_floats * sizeof(float) + 256));
    // the flag page is polled by this device while PEERS store into it: allocated uncached (no L2 line of it can go stale under a remote
    // store), fine-grained as the second choice, plain device memory last (the ranks of the one-GPU tests share a device: any of them works)
    if (hipExtMallocWithFlags((void **)&S.page, PAGE_WORDS * sizeof(unsigned), hipDeviceMallocUncached) != hipSuccess) {
        (void)hipGetLastError();
        if (hipExtMallocWithFlags((void **)&S.page, PAGE_WORDS * sizeof(unsigned), hipDeviceMallocFinegrained) != hipSuccess) {
            (void)hipGetLastError();
            HIPCHECK(hipMalloc((void **)&S.page, PAGE_WORDS * sizeof(unsigned)));
        }
    }
    HIPCHECK(hipMemset(S.page, 0, PAGE_WORDS * sizeof(unsigned)));
    S.inbox = reinterpret_cast<float *>(reinterpret_cast<char *>(S.wire) + S.wire_bytes);
    HIPCHECK(hipHostMalloc((void **)&S.err_host, sizeof(unsigned), hipHostMallocMapped));
    *S.err_host = 0u;
    HIPCHECK(hipHostGetDevicePointer((void **)&S.err_dev, S.err_host, 0));
    hipIpcMemHandle_t hw, hp;
    static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
    HIPCHECK(hipIpcGetMemHandle(&hw, S.wire));
    HIPCHECK(hipIpcGetMemHandle(&hp, S.page));
    memcpy(handles_out, &hw, 64);
    memcpy(handles_out + 64, &hp, 64);
    S.peer_wire.assign((size_t)world, nullptr);
    S.peer_page.assign((size_t)world, nullptr);
    S.opened.assign((size_t)world, false);
    S.peer_wire[(size_t)rank] = S.wire;
    S.peer_page[(size_t)rank] = S.page;
    HIPCHECK(hipDeviceSynchronize());
}
// all_handles: world x (2 x 64) bytes in rank order, as gathered by the caller
void Engine::ipc_connect(const unsigned char *all_handles) {
    check(ipc_ != nullptr, "svdf_ipc_connect: call svdf_ipc_setup first");
    need_device("ipc_connect");
    IpcState &S = *ipc_;
    for (int r = 0; r < S.world; r++) {
        if (r == S.rank) continue;
        hipIpcMemHandle_t hw, hp;
        memcpy(&hw, all_handles + (size_t)r * 128, 64);
        memcpy(&hp, all_handles + (size_t)r * 128 + 64, 64);
        void *w = nullptr, *p = nullptr;
        HIPCHECK(hipIpcOpenMemHandle(&w, hw, hipIpcMemLazyEnablePeerAccess));
        HIPCHECK(hipIpcOpenMemHandle(&p, hp, hipIpcMemLazyEnablePeerAccess));
        S.peer_wire[(size_t)r] = w;
        S.peer_page[(size_t)r] = reinterpret_cast<unsigned *>(p);
        S.opened[(size_t)r] = true;
    }
}
void Engine::ipc_check(const char *what) {
    check(ipc_ != nullptr, "svdf_ipc_*: call svdf_ipc_setup / svdf_ipc_connect first");
    for (int r = 0; r < ipc_->world; r++) check(ipc_->peer_wire[(size_t)r] != nullptr, "svdf_ipc_*: svdf_ipc_connect has not been called");
    if (*ipc_->err_host != 0u)
        fail(std::string("svdf_ipc_") + what + ": a wait on rank " + std::to_string((int)*ipc_->err_host - 1) + "'s flag hit its spin limit (that rank died or never reached the exchange)");
}
void Engine::ipc_window_pack(Dataset *ds, int half) {
    ipc_check("window_pack");
    need_device("ipc");
    IpcState &S = *ipc_;
    int64_t count = 0;
    window_delta_pack(ds, nullptr, half, &count);
    check((size_t)count * (half ? 2 : 4) <= S.wire_bytes, "svdf_ipc_window_pack: the wire buffer of svdf_ipc_setup is too small for this exchange");
    window_delta_pack(ds, S.wire, half, nullptr);
    S.seq++;
    launch_ipc_signal(S.peer_page.data(), S.world, 0, S.rank, S.seq, S.err_dev, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
void Engine::ipc_window_reduce(int half) {
    ipc_check("window_reduce");
    need_device("ipc");
    IpcState &S = *ipc_;
    const int64_t count = delta_ranges().off[delta_ranges().n];
    launch_ipc_wait(S.page, 0, S.world, S.seq, S.err_dev, S.spin_limit, stream_);
    launch_delta_reduce_gather(S.peer_wire.data(), S.world, count * S.rank / S.world, count * (S.rank + 1) / S.world, half, stream_, S.err_dev);
    launch_ipc_signal(S.peer_page.data(), S.world, 1, S.rank, S.seq, S.err_dev, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_ += 3;
}
void Engine::ipc_window_apply(int half) {
    ipc_check("window_apply");
    need_device("ipc");
    IpcState &S = *ipc_;
    launch_ipc_wait(S.page, 1, S.world, S.seq, S.err_dev, S.spin_limit, stream_);
    window_delta_apply(S.wire, half);
    n_launches_++;
}
// Stratified hand-over: the active item block (svdf_item_delta_select) is copied out and stored straight into slot `slot` of rank dst's
// inbox, then word (2 + slot, me) of dst's page gets the block sequence number.
void Engine::ipc_block_send(int dst, int slot) {
    ipc_check("block_send");
    need_device("ipc");
    IpcState &S = *ipc_;
    check(dst >= 0 && dst < S.world && (slot == 0 || slot == 1), "svdf_ipc_block_send: bad destination / slot");
    int64_t n = 0;
    item_block_copy(nullptr, 0, &n);
    check((size_t)n <= S.block_floats, "svdf_ipc_block_send: the inbox of svdf_ipc_setup is too small for this block");
    w_out_.reserve((size_t)n);
    item_block_copy(w_out_.p, 0, nullptr);
    // the slot is free once the destination acknowledged my previous block in it (ranks of a ring can drift apart by more than two steps)
    if (S.slot_seq[slot] != 0u) launch_ipc_wait(S.page + (size_t)dst * 32, 4 + slot, 1, S.slot_seq[slot], S.err_dev, S.spin_limit, stream_);
    float *peer_inbox = reinterpret_cast<float *>(reinterpret_cast<char *>(S.peer_wire[(size_t)dst]) + S.wire_bytes) + (size_t)slot * S.block_floats;
    launch_ipc_copy(peer_inbox, w_out_.p, n, S.err_dev, stream_);
    S.seq_block++;
    S.slot_seq[slot] = S.seq_block;
    unsigned *page[1] = {S.peer_page[(size_t)dst]};
    launch_ipc_signal(page, 1, 2 + slot, S.rank, S.seq_block, S.err_dev, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_ += 2;
}
// waits for the block rank src sent into inbox slot `slot` with block sequence number `seq`, puts it in place (active partition) and
// acknowledges the slot to the sender
void Engine::ipc_block_recv(int src, int slot, unsigned seq) {
    ipc_check("block_recv");
    need_device("ipc");
    IpcState &S = *ipc_;
    check(src >= 0 && src < S.world && (slot == 0 || slot == 1), "svdf_ipc_block_recv: bad source / slot");
    // one-word waits: the wait kernel polls words 0..n-1 of a phase, so it is pointed at src's word with n = 1
    launch_ipc_wait(S.page + (size_t)src * 32, 2 + slot, 1, seq, S.err_dev, S.spin_limit, stream_);
    item_block_copy(S.inbox + (size_t)slot * S.block_floats, 1, nullptr);
    unsigned *page[1] = {S.peer_page[(size_t)src]};
    launch_ipc_signal(page, 1, 4 + slot, S.rank, seq, S.err_dev, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_ += 2;
}
void Engine::ipc_set_spin_limit(long polls) { check(polls >= 1, "ipc_spin_limit must be positive"); ipc_spin_limit_ = polls; if (ipc_) ipc_->spin_limit = (unsigned long long)polls; }
int Engine::ipc_status() const { return ipc_ ? (int)*ipc_->err_host : 0; }
void Engine::ipc_close() {
    if (!ipc_) return;
    HIPCHECK(hipStreamSynchronize(stream_));
    const unsigned err = *ipc_->err_host;   // a wait that timed out in the LAST window has no later svdf_ipc_* call to report it
    ipc_.reset();
    if (err != 0u)
        fail("svdf_ipc_close: a wait on rank " + std::to_string((int)err - 1) + "'s flag hit its spin limit during the pass (that rank died or never reached the exchange); "
             "this rank's model holds an incomplete exchange");
}
void Engine::ipc_fail_if_dead(const char *where) {
    if (ipc_ && *ipc_->err_host != 0u)
        fail(std::string(where) + ": the IPC exchange is dead -- a wait on rank " + std::to_string((int)*ipc_->err_host - 1) +
             "'s flag hit its spin limit (svdf_ipc_status); this rank's model holds an incomplete exchange");
}

}  // namespace svdf
