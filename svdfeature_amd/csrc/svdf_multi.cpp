// svdf_multi.cpp -- N GPUs behind ONE trainer handle, no Python: config key "amd:gpus = N" (ignored by the reference like any
// unknown key).  SURVEY.md 8e, the C++ form of svdfeature_amd/multi_gpu.py:
//   * rank d owns the users with id % N == d; instances follow their (first) user id, so user rows are touched by one rank;
//   * item-side parameters (W_item, i_bias, g_bias [, W_ufeedback]) are replicated; a staging window is one exchange window:
//     every rank runs its exact conflict-free SGD on its share of the window (one host thread per rank for scheduling and
//     launching), then the item-side deltas are packed (svdf_item_delta_pack), summed over the ranks and unpacked;
//   * the sum is an RCCL all-reduce over xGMI -- ncclCommInitAll in this process, ncclAllReduce per rank inside a group,
//     librccl.so resolved at run time so that a single-GPU process never loads it -- when every rank sits on a device of its
//     own; with fewer visible devices than ranks the ranks share devices ("virtual ranks": the same algorithm, deltas summed by
//     a kernel), which is also how the path is tested on a one-GPU box;
//   * predictions go to the owner of the instance's user, model files and views gather the owners' user rows.
// With N = 1 nothing here runs.  With N > 1 the result is window-synchronous SGD on the item side: the accuracy contract
// (|dRMSE| <= 1e-4) instead of bit parity, exactly like the torch.distributed path of bench.py.
#include <dlfcn.h>

#include <algorithm>
#include <cstring>
#include <thread>

#include "svdf_engine.h"
#include "svdf_kernels.h"

namespace svdf {

#define MCHECK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)

static inline void check(bool ok, const char *msg) { if (!ok) fail(msg); }

// ---- RCCL, resolved at run time (rccl.h: ncclCommInitAll :236, ncclAllReduce :611; ncclHalf = 6, ncclFloat = 7, ncclSum = 0)
struct Rccl {
    void *lib = nullptr;
    int (*CommInitAll)(void **, int, const int *) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    std::vector<void *> comms;
    bool load() {
        if (lib) return true;
        for (const char *name : {"librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so"}) {
            lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
            if (lib) break;
        }
        if (!lib) return false;
        CommInitAll = (int (*)(void **, int, const int *))dlsym(lib, "ncclCommInitAll");
        AllReduce = (int (*)(const void *, void *, size_t, int, int, void *, hipStream_t))dlsym(lib, "ncclAllReduce");
        GroupStart = (int (*)())dlsym(lib, "ncclGroupStart");
        GroupEnd = (int (*)())dlsym(lib, "ncclGroupEnd");
        CommDestroy = (int (*)(void *))dlsym(lib, "ncclCommDestroy");
        return CommInitAll && AllReduce && GroupStart && GroupEnd && CommDestroy;
    }
    ~Rccl() {
        if (CommDestroy) for (void *c : comms) if (c) CommDestroy(c);
    }
};

struct MultiState {
    std::vector<std::unique_ptr<Engine>> peers;   // ranks 1 .. N-1 (rank 0 is the handle itself)
    std::vector<int> device;                       // device of every rank
    std::vector<std::unique_ptr<DevBuf<char>>> wire;   // per rank: packed deltas in the wire format
    bool distinct_devices = false, snapshot_taken = false;
    Rccl rccl;
    bool rccl_ready = false;
    int64_t exchanges = 0;
};

void MultiDeleter::operator()(MultiState *m) const { delete m; }

Engine *Engine::rank_engine(int d) { return d == 0 ? this : multi_->peers[(size_t)d - 1].get(); }

// creates ranks 1..N-1 (same type, same configuration), places them on devices
void Engine::multi_setup() {
    if (gpus_ <= 1 || multi_ || is_peer_ || host_only_) return;
    check(!user_group(), "svdfeature_amd: amd:gpus > 1 is implemented for the random-order format (format_type = 0); user-group data shards through svdfeature_amd.multi_gpu");
    check(mp_.common_latent_space == 0, "svdfeature_amd: amd:gpus > 1 needs separate user and item spaces");
    int ndev = 0;
    MCHECK(hipGetDeviceCount(&ndev));
    multi_.reset(new MultiState());
    MultiState &M = *multi_;
    M.device.resize((size_t)gpus_);
    M.distinct_devices = ndev >= gpus_;
    for (int d = 0; d < gpus_; d++) M.device[(size_t)d] = M.distinct_devices ? (device_ + d) % ndev : device_;   // virtual ranks share this handle's device
    for (int d = 1; d < gpus_; d++) {
        std::unique_ptr<Engine> e(new Engine(mtype_, M.device[(size_t)d]));
        e->is_peer_ = true;
        for (const auto &kv : param_log_) e->set_param(kv.first.c_str(), kv.second.c_str());
        M.peers.push_back(std::move(e));
    }
    for (int d = 0; d < gpus_; d++) M.wire.emplace_back(new DevBuf<char>());
    // one exchange window = one staging window: about 64 / 42 / 32 updates per item (2 / 3-4 / more ranks), the calibration of
    // DESIGN.md section 6; an explicit stage_window knob or amd:window key wins
    if (!window_set_) {
        const long per_item = gpus_ <= 2 ? 64 : (gpus_ <= 4 ? 42 : 32);
        stage_window_ = std::max<long>(1024, per_item * (long)std::max(mp_.num_item, 1));
    }
    MCHECK(hipSetDevice(device_));
}

// ranks 1..N-1 start from the model of rank 0 (one rand_init / one model file, like the reference's single trainer)
void Engine::multi_copy_model_to_peers() {
    if (!multi_) return;
    check(host_model_valid_, "multi-GPU: no host model to hand to the other ranks");
    for (auto &p : multi_->peers) {
        p->mp_ = mp_;
        p->alloc_host_model();
        p->hW_ = hW_; p->hbias_ = hbias_; p->hg_ = hg_;
        p->host_model_valid_ = true;
    }
}

static inline int owner_of_row(const HostCSR &src, long r, int n) {
    const int *p = &src.row_ptr[(size_t)3 * r];
    return p[2] > p[1] ? (int)(src.feat_index[(size_t)p[1]] % (unsigned)n) : 0;
}

// the staged rows as exchange windows over all ranks (called instead of flush_csr on the handle): at most stage_window_ rows
// per window, so one large update_csr_batch call is cut the same way a stream of single instances would be
void Engine::multi_flush(HostCSR &src) {
    MultiState &M = *multi_;
    const long n = src.num_row();
    if (n == 0) return;
    const int N = gpus_;
    if (!M.snapshot_taken) {
        for (int d = 0; d < N; d++) { Engine *e = rank_engine(d); MCHECK(hipSetDevice(e->device_)); e->item_delta_begin_local(); }
        M.snapshot_taken = true;
    }
    for (long w0 = 0; w0 < n; w0 += stage_window_) {
        const long w1 = std::min(n, w0 + stage_window_);
        std::vector<HostCSR> part((size_t)N);
        for (long r = w0; r < w1; r++) {
            const int d = owner_of_row(src, r, N);
            const int *p = &src.row_ptr[(size_t)3 * r];
            HostCSR &o = part[(size_t)d];
            const int base = o.row_ptr.back() - p[0];
            o.row_label.push_back(src.row_label[(size_t)r]);
            o.row_ptr.push_back(p[1] + base); o.row_ptr.push_back(p[2] + base); o.row_ptr.push_back(p[3] + base);
            o.feat_index.insert(o.feat_index.end(), src.feat_index.begin() + p[0], src.feat_index.begin() + p[3]);
            o.feat_value.insert(o.feat_value.end(), src.feat_value.begin() + p[0], src.feat_value.begin() + p[3]);
        }
        // every rank schedules and launches its share on a host thread of its own
        std::vector<std::string> errors((size_t)N);
        std::vector<std::thread> th;
        for (int d = 0; d < N; d++) {
            th.emplace_back([&, d]() {
                try {
                    Engine *e = rank_engine(d);
                    (void)hipSetDevice(e->device_);
                    e->flush_csr(part[(size_t)d]);
                } catch (const std::exception &ex) { errors[(size_t)d] = ex.what(); }
            });
        }
        for (auto &t : th) t.join();
        for (auto &m : errors) if (!m.empty()) fail(m);
        multi_exchange();
    }
    MCHECK(hipSetDevice(device_));
    n_instances_ += 0;   // every rank counts its own share; svdf_counter(0) on the handle reports rank 0's
    src.clear();
}

void Engine::multi_exchange() {
    MultiState &M = *multi_;
    const int N = gpus_;
    const int half = delta_half_ ? 1 : 0;
    int64_t count = 0;
    item_delta_pack(nullptr, half, &count);
    const size_t bytes = (size_t)count * (half ? 2 : 4);
    for (int d = 0; d < N; d++) {
        Engine *e = rank_engine(d);
        MCHECK(hipSetDevice(e->device_));
        M.wire[(size_t)d]->reserve(bytes);
        e->item_delta_pack(M.wire[(size_t)d]->p, half, nullptr);
    }
    bool reduced = false;
    if (M.distinct_devices) {
        if (!M.rccl_ready && M.rccl.load()) {
            M.rccl.comms.assign((size_t)N, nullptr);
            if (M.rccl.CommInitAll(M.rccl.comms.data(), N, M.device.data()) == 0) M.rccl_ready = true;
        }
        if (M.rccl_ready) {
            M.rccl.GroupStart();
            for (int d = 0; d < N; d++) {
                Engine *e = rank_engine(d);
                M.rccl.AllReduce(M.wire[(size_t)d]->p, M.wire[(size_t)d]->p, (size_t)count, half ? 6 : 7, 0, M.rccl.comms[(size_t)d], e->stream_);
            }
            if (M.rccl.GroupEnd() != 0) fail("svdfeature_amd: RCCL all-reduce of the item-side deltas failed");
            reduced = true;
        }
    }
    if (!reduced) {
        // shared devices (virtual ranks) or no RCCL: bring the buffers to rank 0's device, sum them there, hand the sum back
        for (int d = 0; d < N; d++) { Engine *e = rank_engine(d); MCHECK(hipSetDevice(e->device_)); MCHECK(hipStreamSynchronize(e->stream_)); }
        MCHECK(hipSetDevice(device_));
        std::vector<const void *> srcs;
        std::vector<std::unique_ptr<DevBuf<char>>> staged;
        for (int d = 0; d < N; d++) staged.emplace_back(new DevBuf<char>());
        for (int d = 0; d < N; d++) {
            if (M.device[(size_t)d] == device_) { srcs.push_back(M.wire[(size_t)d]->p); continue; }
            staged[(size_t)d]->reserve(bytes);
            MCHECK(hipMemcpyPeerAsync(staged[(size_t)d]->p, device_, M.wire[(size_t)d]->p, M.device[(size_t)d], bytes, stream_));
            srcs.push_back(staged[(size_t)d]->p);
        }
        launch_delta_sum(srcs.data(), N, M.wire[0]->p, count, half, stream_);
        MCHECK(hipStreamSynchronize(stream_));
        for (int d = 1; d < N; d++) {
            if (M.device[(size_t)d] == device_) MCHECK(hipMemcpyAsync(M.wire[(size_t)d]->p, M.wire[0]->p, bytes, hipMemcpyDeviceToDevice, stream_));
            else MCHECK(hipMemcpyPeerAsync(M.wire[(size_t)d]->p, M.device[(size_t)d], M.wire[0]->p, device_, bytes, stream_));
        }
        MCHECK(hipStreamSynchronize(stream_));
    }
    for (int d = 0; d < N; d++) {
        Engine *e = rank_engine(d);
        MCHECK(hipSetDevice(e->device_));
        e->item_delta_unpack(M.wire[(size_t)d]->p, half, 1);
    }
    for (int d = 0; d < N; d++) { Engine *e = rank_engine(d); MCHECK(hipSetDevice(e->device_)); MCHECK(hipStreamSynchronize(e->stream_)); }
    MCHECK(hipSetDevice(device_));
    M.exchanges++;
}

// predictions: every row is scored by the owner of its user
void Engine::multi_predict(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out) {
    const int N = gpus_;
    HostCSR all;
    stage_rows_into(all, num_row, row_label, row_ptr, feat_index, feat_value);
    std::vector<std::vector<int>> rows((size_t)N);
    for (long r = 0; r < num_row; r++) rows[(size_t)owner_of_row(all, r, N)].push_back((int)r);
    for (int d = 0; d < N; d++) {
        if (rows[(size_t)d].empty()) continue;
        HostCSR sub;
        for (int r : rows[(size_t)d]) {
            const int *p = &all.row_ptr[(size_t)3 * r];
            const int base = sub.row_ptr.back() - p[0];
            sub.row_label.push_back(all.row_label[(size_t)r]);
            sub.row_ptr.push_back(p[1] + base); sub.row_ptr.push_back(p[2] + base); sub.row_ptr.push_back(p[3] + base);
            sub.feat_index.insert(sub.feat_index.end(), all.feat_index.begin() + p[0], all.feat_index.begin() + p[3]);
            sub.feat_value.insert(sub.feat_value.end(), all.feat_value.begin() + p[0], all.feat_value.begin() + p[3]);
        }
        std::vector<float> res(rows[(size_t)d].size());
        Engine *e = rank_engine(d);
        MCHECK(hipSetDevice(e->device_));
        e->predict_csr_batch_local((int)res.size(), sub.row_label.data(), sub.row_ptr.data(), sub.feat_index.data(), sub.feat_value.data(), res.data());
        for (size_t j = 0; j < res.size(); j++) out[rows[(size_t)d][j]] = res[j];
    }
    MCHECK(hipSetDevice(device_));
}

// the owners' user rows into rank 0's host model (after download_model on rank 0)
void Engine::multi_gather_user_rows() {
    const int N = gpus_;
    for (int d = 1; d < N; d++) {
        Engine *e = rank_engine(d);
        MCHECK(hipSetDevice(e->device_));
        e->download_model();
        for (long u = d; u < mp_.num_user; u += N) {
            memcpy(&hW_[((size_t)user_off_ + (size_t)u) * pitch_], &e->hW_[((size_t)user_off_ + (size_t)u) * pitch_], (size_t)pitch_ * sizeof(float));
            hbias_[(size_t)user_off_ + (size_t)u] = e->hbias_[(size_t)user_off_ + (size_t)u];
        }
        e->hW_.clear(); e->hW_.shrink_to_fit(); e->hbias_.clear(); e->hg_.clear(); e->host_model_valid_ = false;
    }
    MCHECK(hipSetDevice(device_));
}

int64_t Engine::multi_counter(int what) const {
    if (!multi_) return 0;
    if (what == 0) return multi_->exchanges;
    if (what == 1) return multi_->rccl_ready ? 1 : 0;
    if (what == 2) return multi_->distinct_devices ? 1 : 0;
    return -1;
}

}  // namespace svdf
