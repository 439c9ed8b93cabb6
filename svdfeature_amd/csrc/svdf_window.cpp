// svdf_window.cpp -- part of the host engine (class Engine, svdf_engine.h): window-minibatch data sets of one exchange window and the item-side delta entry points (N > 1 ranks)
// Reference citations are relative to /root/reference.
#include "svdf_engine.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <atomic>
#include <thread>

#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {

// contributes in /root/reference/solvers/base-solver/apex_svd_base.h:383-427 being applied at once by "applied at the window's end".
WindowSchedule Engine::window_view(const Dataset *ds) const {
    const bool pairs = ds->win_item1.p != nullptr && ds->fused.max_ni == 2;
    return WindowSchedule{ds->win_urec.p, ds->num_units, ds->item.p, pairs ? nullptr : ds->label.p, ds->win_slot.p, ds->unit_values ? nullptr : ds->uval.p,
                          (ds->unit_values && !pairs) ? nullptr : ds->ival.p, ds->win_iptr.p, d_contrib_.p, d_cbias_.p,
                          pairs ? ds->win_item1.p : nullptr, pairs ? ds->win_slot1.p : nullptr, pairs ? ds->win_ival1.p : nullptr, contrib_bf16_ ? 1 : 0,
                          0, nullptr};   // (ordered sub-steps of hot items: set by wseq_train for the windows that have any)
}
Dataset *Engine::dataset_window_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; an amd:gpus handle builds them itself from svdf_dataset_from_triples");
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    window_build(ds.get(), n, user, item, label);
    return ds.release();
}
// rank pairs (user, positive item, negative item) of one exchange window: the instance PairwiseRankGenerator emits for two plain rows
// (apex_svd_data.cpp:828-860, :905-911: label 1, user:1, the two items in index order with the negative's sign flipped), BASELINE
// configs[4].  Two contribution slots per pair.
Dataset *Engine::dataset_window_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; shard rank pairs through svdfeature_amd.multi_gpu");
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    window_build(ds.get(), n, user, pos, nullptr, neg);
    return ds.release();
}
// (re)fills ds in place: the staged path of an amd:gpus handle rebuilds one window data set per rank every window.
// neg != nullptr: rank pairs, `item` holds the positive items and the labels are 1.
void Engine::window_build_header(Dataset *ds, long n, bool pairs) {
    check(!user_group() && mtype_.extend_type == 0, "window data sets: random-order trainers only");
    check(basic_fast_path_allowed(), "window data sets: no side tables, relaxed ids, lazy decay or shared latent space; num_factor <= 256");
    check(n >= 0 && n < (1L << 30), "window data sets: at most 2^30-1 instances per window");
    if (window_trained_ == ds) window_trained_ = nullptr;
    ds->num_row = n; ds->kind = 5;
    ds->win_slots = pairs ? 2 * n : n;
    ds->fused.max_ni = pairs ? 2 : 1;
}
void Engine::window_build(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg) {
    check(!user_group() && mtype_.extend_type == 0, "window data sets: random-order trainers only");
    check(basic_fast_path_allowed(), "window data sets: no side tables, relaxed ids, lazy decay or shared latent space; num_factor <= 256");
    check(n >= 0 && n < (1L << 30), "window data sets: at most 2^30-1 instances per window");
    const long NU = mp_.num_user, NI = mp_.num_item;
    const bool pairs = neg != nullptr;
    window_build_header(ds, n, pairs);
    if (window_build_device(ds, n, user, item, label, neg)) return;
    std::vector<int> ucnt((size_t)NU, 0), iptr((size_t)NI + 1, 0);
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)NU) fail("user feature index exceed bound");
        if (item[r] >= (unsigned)NI) fail("item feature index exceed bound");
        ucnt[user[r]]++;
        iptr[(size_t)item[r] + 1]++;
        if (pairs) {
            if (neg[r] >= (unsigned)NI) fail("item feature index exceed bound");
            if (neg[r] == item[r]) fail("rank pair: positive and negative item must differ");
            iptr[(size_t)neg[r] + 1]++;
        }
    }
    for (long i = 0; i < NI; i++) iptr[(size_t)i + 1] += iptr[(size_t)i];
    // users in launch order: by instance count, descending (the lane groups of a wave then run the same number of iterations),
    // ties by user id; a user's instances are contiguous in that order
    int maxc = 0;
    long nact = 0;
    for (long u = 0; u < NU; u++) { maxc = std::max(maxc, ucnt[(size_t)u]); nact += ucnt[(size_t)u] > 0; }
    std::vector<long> start((size_t)maxc + 2, 0);
    for (long u = 0; u < NU; u++) if (ucnt[(size_t)u] > 0) start[(size_t)ucnt[(size_t)u]]++;
    { long acc = 0; for (int c = maxc; c >= 1; c--) { const long m = start[(size_t)c]; start[(size_t)c] = acc; acc += m; } }
    std::vector<WinUser> urec((size_t)nact);
    for (long u = 0; u < NU; u++) {
        const int c = ucnt[(size_t)u];
        if (c > 0) urec[(size_t)start[(size_t)c]++] = WinUser{(unsigned)u, 0, c, 0};
    }
    std::vector<int> ubegin((size_t)NU, 0);
    { long acc = 0; for (long j = 0; j < nact; j++) { urec[(size_t)j].begin = (int)acc; ubegin[urec[(size_t)j].user] = (int)acc; acc += urec[(size_t)j].count; } }
    std::vector<unsigned> w_item((size_t)n), w_item1(pairs ? (size_t)n : 0);
    std::vector<float> w_label(pairs ? 0 : (size_t)n), w_v0(pairs ? (size_t)n : 0), w_v1(pairs ? (size_t)n : 0);
    std::vector<int> w_slot((size_t)n), w_slot1(pairs ? (size_t)n : 0), icur(iptr.begin(), iptr.end() - 1);
    for (long r = 0; r < n; r++) {   // file order: a user's instances and an item's slots both keep it
        const int at = ubegin[user[r]]++;
        if (!pairs) {
            w_item[(size_t)at] = item[r];
            w_label[(size_t)at] = label[r];
            w_slot[(size_t)at] = icur[item[r]]++;
        } else {   // entry 0 = the lower item id (the merged row is index sorted), the negative's sign flipped
            const bool pf = item[r] < neg[r];
            const unsigned lo = pf ? item[r] : neg[r], hi = pf ? neg[r] : item[r];
            w_item[(size_t)at] = lo; w_item1[(size_t)at] = hi;
            w_v0[(size_t)at] = pf ? 1.0f : -1.0f; w_v1[(size_t)at] = pf ? -1.0f : 1.0f;
            w_slot[(size_t)at] = icur[lo]++; w_slot1[(size_t)at] = icur[hi]++;
        }
    }
    ds->win_urec.upload(urec.data(), (size_t)nact, stream_);
    ds->item.upload(w_item.data(), (size_t)n, stream_);
    ds->win_slot.upload(w_slot.data(), (size_t)n, stream_);
    ds->win_iptr.upload(iptr.data(), (size_t)NI + 1, stream_);
    if (!pairs) {
        ds->label.upload(w_label.data(), (size_t)n, stream_);
        ds->win_item1.release();
    } else {
        ds->win_item1.upload(w_item1.data(), (size_t)n, stream_);
        ds->win_slot1.upload(w_slot1.data(), (size_t)n, stream_);
        ds->ival.upload(w_v0.data(), (size_t)n, stream_);
        ds->win_ival1.upload(w_v1.data(), (size_t)n, stream_);
    }
    HIPCHECK(hipStreamSynchronize(stream_));   // the host columns go out of scope
    ds->sched_signature = schedule_signature();   // a data set refilled in place (the staged path of an amd:gpus handle) is valid under the CURRENT configuration
    ds->win_item_lo = NI; ds->win_item_hi = -1;    // the item ids the window touches: window_delta_apply_local checks them against the active block
    for (long i = 0; i < NI; i++) if (iptr[(size_t)i + 1] > iptr[(size_t)i]) { if (ds->win_item_lo == NI) ds->win_item_lo = i; ds->win_item_hi = i; }
    ds->unit_values = true;
    ds->num_units = nact;
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = n;
    const long nrow_touched = pairs ? 3 : 2, nb = (mp_.no_user_bias ? 0 : 1) + (pairs ? 2 : 1);
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * nrow_touched + 8 * nb + 16 + 8 * nrow_touched);   // SURVEY 8(d4), what the reference's step moves per instance
}
// The same arrays from the device (svdf_k_wbuild.hip): the window's columns go up as they are, three stable sorts and two scans regroup them
// in HBM.  false = not taken (host-only handle, knob device_window = 0, an empty window): the host builder above runs.
bool Engine::window_build_device(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg) {
    if (host_only_ || !device_window_ || n <= 0) return false;
    need_device("dataset");
    const bool pairs = neg != nullptr;
    wb_user_.upload(user, (size_t)n, stream_);
    wb_item_.upload(item, (size_t)n, stream_);
    if (pairs) wb_neg_.upload(neg, (size_t)n, stream_); else wb_label_.upload(label, (size_t)n, stream_);
    window_build_resident(ds, n, wb_user_.p, wb_item_.p, pairs ? nullptr : wb_label_.p, pairs ? wb_neg_.p : nullptr);
    return true;
}
// the columns already in HBM (a whole data set handed over in one copy, its windows built from slices: wseq_from_triples / _pairs).
// pairs: d_label == nullptr, d_neg != nullptr.  The caller has set the window's header fields (window_build).
void Engine::window_build_resident(Dataset *ds, long n, const unsigned *d_user, const unsigned *d_item, const float *d_label, const unsigned *d_neg) {
    const long NU = mp_.num_user, NI = mp_.num_item;
    const bool pairs = d_neg != nullptr;
    const long E = pairs ? 2 * n : n;
    wb_k0_.reserve((size_t)E); wb_k1_.reserve((size_t)E); wb_v0_.reserve((size_t)E); wb_v1_.reserve((size_t)E);
    wb_inst_.reserve((size_t)n); wb_slot_e_.reserve((size_t)E); wb_head_.reserve((size_t)n); wb_mark_.reserve((size_t)n);
    wb_run_user_.reserve((size_t)n); wb_run_start_.reserve((size_t)n); wb_run_begin_.reserve((size_t)n);
    wb_state_.reserve(8);
    const size_t tb = wbuild_tmp_bytes(E);
    wb_tmp_.reserve(std::max<size_t>(tb, 1));
    ds->win_urec.reserve((size_t)std::min<long>(n, std::max<long>(NU, 1)));
    ds->item.reserve((size_t)n); ds->win_slot.reserve((size_t)n); ds->win_iptr.reserve((size_t)NI + 1);
    if (!pairs) { ds->label.reserve((size_t)n); ds->win_item1.release(); }
    else { ds->win_item1.reserve((size_t)n); ds->win_slot1.reserve((size_t)n); ds->ival.reserve((size_t)n); ds->win_ival1.reserve((size_t)n); }
    WBuildIn in{n, pairs ? 1 : 0, d_user, d_item, d_neg, d_label, NU, NI};
    WBuildBuffers B{wb_k0_.p, wb_k1_.p, wb_v0_.p, wb_v1_.p, wb_inst_.p, wb_slot_e_.p, wb_head_.p, wb_mark_.p, wb_run_user_.p, wb_run_start_.p, wb_run_begin_.p,
                    wb_tmp_.p, tb, wb_state_.p};
    WBuildOut out{ds->win_urec.p, ds->item.p, pairs ? ds->win_item1.p : nullptr, pairs ? nullptr : ds->label.p, pairs ? ds->ival.p : nullptr,
                  pairs ? ds->win_ival1.p : nullptr, ds->win_slot.p, pairs ? ds->win_slot1.p : nullptr, ds->win_iptr.p};
    long nact = 0, lo = 0, hi = -1;
    try {
        device_window_build(in, B, out, &nact, &lo, &hi, stream_);
    } catch (const std::runtime_error &e) {
        fail(e.what());
    }
    ds->sched_signature = schedule_signature();
    ds->win_item_lo = lo; ds->win_item_hi = hi;
    ds->unit_values = true;
    ds->num_units = nact;
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = n;
    const long nrow_touched = pairs ? 3 : 2, nb = (mp_.no_user_bias ? 0 : 1) + (pairs ? 2 : 1);
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * nrow_touched + 8 * nb + 16 + 8 * nrow_touched);   // SURVEY 8(d4), as in the host builder
}
void Engine::window_delta_pack(Dataset *ds, void *device_dst, int half, int64_t *count) {
    check(trainer_ready_, "window_delta: init_trainer has not been called");
    check(ds && ds->owner == this && (ds->kind == 5 || ds->kind == 7), "window_delta_pack: not a window data set of this trainer");
    if (ds->kind == 7) {   // user units: the whole replicated side in one piece, [feedback rows | item rows | their biases | global biases]
        check(delta_nparts_ == 1, "window_delta_pack: user-unit window data sets exchange the replicated side in one piece");
        const DeltaRanges R7 = delta_ranges();
        const long T = (user_group() ? (long)num_fb_rows() : 0) + (long)mp_.num_item;
        check(R7.off[R7.n] == T * (pitch_ + 1) + (long)mp_.num_global, "window_delta_pack: unexpected layout of the replicated ranges");
        if (count) *count = R7.off[R7.n];
        if (!device_dst) return;
        need_device("window_delta");
        check(window_trained_ == ds, "window_delta_pack: train this window data set first (svdf_train_dataset)");
        wunit_sum(ds, device_dst, half);
        HIPCHECK(hipGetLastError());
        n_launches_++;
        return;
    }
    check(!relaxed() && g_stride_ == 1 && user_off_ == 0, "window_delta_pack: random-order trainers without relaxed ids only");
    const long ni = mp_.num_item;
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    const long nglobal = (delta_nparts_ == 1 || delta_part_ == 0) ? (long)mp_.num_global : 0;
    const DeltaRanges R = delta_ranges();
    check(R.off[R.n] == (hi - lo) * (pitch_ + 1) + nglobal, "window_delta_pack: unexpected layout of the replicated ranges");
    if (count) *count = R.off[R.n];
    if (!device_dst) return;
    need_device("window_delta");
    check(window_trained_ == ds, "window_delta_pack: train this window data set first (svdf_train_dataset)");
    launch_window_items(window_view(ds), pitch_, mp_.num_factor, lo, hi, nglobal, device_dst, half, stream_, ds->win_slots);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
// Stratified schedule: the rank owns the active item block (svdf_item_delta_select) exclusively while it trains
// a stratum, so the window's per-item sums go straight into the model -- no wire buffer, no sum over ranks.
void Engine::window_delta_apply_local(Dataset *ds) {
    check(ds && ds->owner == this && (ds->kind == 5 || ds->kind == 7), "window_delta_apply_local: not a window data set of this trainer");
    need_device("window_delta");
    check(window_trained_ == ds, "window_delta_apply_local: train this window data set first (svdf_train_dataset)");
    if (ds->kind == 7) {   // user units: every per-target sum of the window, added in place (one rank holds the whole replicated side)
        check(delta_nparts_ == 1, "window_delta_apply_local: user-unit window data sets apply the replicated side in one piece");
        wunit_sum(ds, nullptr, 0);
        HIPCHECK(hipGetLastError());
        n_launches_++;
        window_trained_ = nullptr;   // the sums are in the model: applying them twice would be a silent error
        return;
    }
    check(!relaxed() && g_stride_ == 1 && user_off_ == 0, "window_delta_apply_local: random-order trainers without relaxed ids only");
    const long ni = mp_.num_item;
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    check(ds->win_item_hi < 0 || (ds->win_item_lo >= lo && ds->win_item_hi < hi),
          "window_delta_apply_local: the window holds instances of items outside the active item block (svdf_item_delta_select): their updates would be lost");
    launch_window_items_local(window_view(ds), pitch_, mp_.num_factor, lo, hi, dW_.p + (size_t)item_off_ * pitch_, dbias_.p + item_off_, stream_, ds->win_slots);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
void Engine::item_block_copy(float *device_buf, int set, int64_t *count) {
    check(trainer_ready_, "item_block: init_trainer has not been called");
    const DeltaRanges R = delta_ranges();
    if (count) *count = R.off[R.n];
    if (!device_buf) return;
    need_device("item_block");
    flush();
    launch_ranges_copy(R, device_buf, set, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
// One stratum step of the stratified schedule in ONE call: every window data set trained and summed in place into the active
// item block, then (device_out != nullptr) the block copied out for its hand-over -- the sequence multi_gpu.StratifiedTrainer issued as
// 4 + 4 W calls.  The host thread has ~40 us per step at N = 8; the calls were a quarter of that.
void Engine::stratum_step(Dataset *const *ds, int n, int block, int nblocks, float *device_out) {
    check(n >= 0 && (n == 0 || ds != nullptr), "stratum_step: bad window list");
    for (int w = 0; w < n; w++) {
        item_delta_select(0, 1);
        train_dataset(ds[w]);
        item_delta_select(block, nblocks);
        window_delta_apply_local(ds[w]);
    }
    if (device_out) {
        item_delta_select(block, nblocks);
        item_block_copy(device_out, 0, nullptr);
    }
    item_delta_select(0, 1);
}
void Engine::item_block_set_at(int block, int nblocks, const float *device_src) {
    item_delta_select(block, nblocks);
    item_block_copy(const_cast<float *>(device_src), 1, nullptr);
    item_delta_select(0, 1);
}
void Engine::window_delta_apply(const void *device_src, int half) {
    need_device("window_delta");
    flush();
    launch_delta_addto(delta_ranges(), device_src, half, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}

// =============================================================================== item-side delta (multi-GPU)
std::vector<Engine::Range> Engine::shared_ranges() {
    check(mp_.common_latent_space == 0, "svdfeature_amd: user sharding needs separate user and item spaces");
    std::vector<Range> r;
    if (user_off_ > 0) r.push_back({dW_.p, (long)user_off_ * pitch_});
    r.push_back({dW_.p + (size_t)item_off_ * pitch_, (long)(n_uiset_ - item_off_) * pitch_});
    if (user_off_ > 0) r.push_back({dbias_.p, (long)user_off_});
    r.push_back({dbias_.p + item_off_, (long)(n_uiset_ - item_off_)});
    if (mp_.num_global > 0) r.push_back({dg_.p, (long)mp_.num_global * g_stride_});   // (padding floats stay 0: zero deltas)
    return r;
}
void Engine::item_delta_begin() {
    check(trainer_ready_, "item_delta: init_trainer has not been called");
    need_device("item_delta");
    flush();
    item_delta_begin_local();
}
void Engine::item_delta_begin_local() {
    need_device("item_delta");
    auto rg = shared_ranges();
    long total = 0;
    for (auto &x : rg) total += x.n;
    d_snap_.reserve((size_t)total);
    d_delta_.reserve((size_t)total);
    long off = 0;
    for (auto &x : rg) {
        HIPCHECK(hipMemcpyAsync(d_snap_.p + off, x.base, (size_t)x.n * sizeof(float), hipMemcpyDeviceToDevice, stream_));
        off += x.n;
    }
}
void *Engine::item_delta_buffer(int64_t *count) {
    need_device("item_delta");
    flush();
    auto rg = shared_ranges();
    long off = 0;
    for (auto &x : rg) {
        launch_delta_sub(x.base, d_snap_.p + off, d_delta_.p + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
    if (count) *count = off;
    return d_delta_.p;
}
void Engine::item_delta_apply() {
    need_device("item_delta");
    auto rg = shared_ranges();
    long off = 0;
    for (auto &x : rg) {
        launch_delta_add(x.base, d_snap_.p + off, d_delta_.p + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
}

void Engine::item_delta_into(float *device_dst, int64_t *count) {
    need_device("item_delta");
    flush();
    long off = 0;
    for (auto &x : shared_ranges()) {
        launch_delta_sub(x.base, d_snap_.p + off, device_dst + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
    if (count) *count = off;
}
void Engine::item_delta_apply_from(const float *device_src) {
    need_device("item_delta");
    long off = 0;
    for (auto &x : shared_ranges()) {
        launch_delta_add(x.base, d_snap_.p + off, device_src + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
}
// Replicated ranges of the ACTIVE exchange partition (svdf_item_delta_select): the item rows are cut into nparts id ranges so
// that a window's exchange can be split into pieces that overlap with training on the other pieces' instances; everything that
// is not indexed by item id (feedback rows, global biases) travels with partition 0.  snap_off addresses the full snapshot.
DeltaRanges Engine::delta_ranges() {
    DeltaRanges R;
    memset(&R, 0, sizeof(R));
    auto rg = shared_ranges();
    check(rg.size() <= SVDF_MAX_DELTA_RANGES, "item_delta: too many replicated ranges");
    // shared_ranges(): [W_fb] W_item [bias_fb] bias_item [g_bias]; item ranges are the ones starting at item_off_
    const long ni = (long)(n_uiset_ - item_off_);
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    long off = 0, snap = 0;
    int n = 0;
    for (size_t q = 0; q < rg.size(); q++) {
        const bool is_w_item = rg[q].base == dW_.p + (size_t)item_off_ * pitch_;
        const bool is_b_item = rg[q].base == dbias_.p + item_off_;
        if (delta_nparts_ > 1 && (is_w_item || is_b_item)) {
            const long unit = is_w_item ? pitch_ : 1;
            R.base[n] = rg[q].base + lo * unit; R.off[n] = off; R.snap_off[n] = snap + lo * unit;
            off += (hi - lo) * unit; n++;
        } else if (delta_nparts_ == 1 || delta_part_ == 0) {
            R.base[n] = rg[q].base; R.off[n] = off; R.snap_off[n] = snap;
            off += rg[q].n; n++;
        }
        snap += rg[q].n;
    }
    R.n = n;
    for (int q = n; q <= SVDF_MAX_DELTA_RANGES; q++) R.off[q] = off;
    return R;
}
void Engine::item_delta_select(int part, int nparts) {
    check(nparts >= 1 && part >= 0 && part < nparts, "item_delta_select: bad partition");
    delta_part_ = part; delta_nparts_ = nparts;
}
void Engine::item_delta_pack(void *device_dst, int half, int64_t *count) {
    check(trainer_ready_, "item_delta: init_trainer has not been called");
    if (device_dst) need_device("item_delta");
    const DeltaRanges R = delta_ranges();
    if (count) *count = R.off[R.n];
    if (!device_dst) return;   // size query
    check(d_snap_.p != nullptr && (long)d_snap_.cap >= R.off[R.n], "item_delta: call item_delta_begin first");
    flush();
    launch_delta_pack(R, d_snap_.p, device_dst, half, stream_);
    HIPCHECK(hipGetLastError());
}
void Engine::item_delta_unpack(const void *device_src, int half, int refresh_snapshot) {
    need_device("item_delta");
    const DeltaRanges R = delta_ranges();
    check(d_snap_.p != nullptr && (long)d_snap_.cap >= R.off[R.n], "item_delta: call item_delta_begin first");
    launch_delta_unpack(R, d_snap_.p, device_src, half, refresh_snapshot, stream_);
    HIPCHECK(hipGetLastError());
}
void Engine::item_delta_copy(float *device_dst, const float *device_src) {
    need_device("item_delta");
    long total = 0;
    for (auto &x : shared_ranges()) total += x.n;
    check(d_delta_.p != nullptr && (long)d_delta_.cap >= total, "item_delta: call item_delta_begin / item_delta_buffer first");
    if (device_dst) HIPCHECK(hipMemcpyAsync(device_dst, d_delta_.p, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    if (device_src) HIPCHECK(hipMemcpyAsync(d_delta_.p, device_src, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
}

}  // namespace svdf
