// svdf_ranker.cpp -- host side of the device ranker: ISVDRanker (apex_svd.h:160-197) as implemented by SVDFeatureRanker
// (solvers/base-solver/apex_svd_base.h:597-813), behind the svdf_ranker_* entry points of include/svdfeature_amd.h.
//
// The protocol is a stream of tagged lines (svdranker_tag, apex_svd.h:115-152): ITEM lines build the candidate set once, then
// per user section USER / POS / BAN / SPEC lines and a PROCESS line that returns either the top_k candidates or the rank
// positions of the positive samples.  Lines are staged on the host; the device work happens at PROCESS:
//   k_rank_items (once per new candidates), k_rank_user, k_rank_spec, k_rank_score, k_rank_positions  (svdf_k_rank.hip)
// Ordering is index work and has to be the reference's: scores are computed in its fp32 order on the device; where several
// ranked candidates tie with a requested position (or inside the top_k prefix) the order is whatever std::sort makes of
// the reference's entry vector, so those rare sections are finished by running exactly that sort on the host.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <future>
#include <mutex>
#include <thread>
#include <cmath>
#include <cstring>
#include <memory>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_stdsort.h"

namespace svdf {

#define RCHECK(call)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)
static inline void rcheck(bool ok, const char *msg) { if (!ok) fail(msg); }

Ranker::Ranker(TypeParam mtype, int device) : eng_(new Engine(TypeParam{mtype.format_type, mtype.active_type, 0, 0}, device)) { pin_malloc_threshold(); }
Ranker::~Ranker() {
    (void)hipStreamSynchronize(eng_->stream_);
    if (sort_tmp_) (void)hipFree(sort_tmp_);
    if (tie_pin_) (void)hipHostFree(tie_pin_);
    if (tie_stream_) (void)hipStreamDestroy(tie_stream_);
    for (RankSlot &S : slots_) {
        if (S.pin) (void)hipHostFree(S.pin);
        if (S.back) (void)hipHostFree(S.back);
        if (S.ev) (void)hipEventDestroy(S.ev);
    }
}

void Ranker::set_param(const char *name, const char *val) {   // :656-660
    if (!strcmp(name, "amd:rank_tile")) tile_enabled_ = atoi(val) != 0;   // extension key: 0 = one scoring pass per section
    if (!strcmp(name, "feature_user") || !strcmp(name, "feature_item")) eng_->set_param(name, val);
    if (!strcmp(name, "top_k")) top_k_ = atoi(val);
}
void Ranker::load_model(FILE *fi) { eng_->load_model(fi); }   // :662-664
void Ranker::init_ranker(int num_item_set) {                   // :666-685
    rcheck(num_item_set >= 0, "init_ranker: negative item set size");
    eng_->init_trainer();   // side tables, model upload
    num_item_set_ = num_item_set;
    num_item_processed_ = 0;
    items_on_device_ = 0;
    const size_t pitch = (size_t)eng_->pitch_;
    d_ifactors_.reserve((size_t)std::max(num_item_set, 1) * pitch);
    d_ift_.reserve((size_t)std::max(num_item_set, 1) * pitch);   // the same matrix chunk-major, what the scoring pass streams
    d_ibias_.reserve((size_t)std::max(num_item_set, 1));
    d_tag_.reserve((size_t)std::max(num_item_set, 1));
    d_tu_.reserve(pitch + 4);
    d_fb_.reserve(pitch + 4);
    RCHECK(hipMemsetAsync(d_fb_.p, 0, (pitch + 4) * sizeof(float), eng_->stream_));
    if (eng_->user_group() && eng_->mp_.num_user > 0) {
        // tmp_ufeedback = clone( model.W_user[0] ) (:680-682) is a COPY of user row 0 (CloneSolver,
        // apex_tensor_func_decl_common.h:265-274): what user sections see until the first block arrives
        const DevParams &P = eng_->params();
        RCHECK(hipMemcpyAsync(d_fb_.p, P.W + (size_t)P.user_off * (size_t)P.pitch, (size_t)eng_->mp_.num_factor * sizeof(float), hipMemcpyDeviceToDevice,
                              eng_->stream_));
    }
    RCHECK(hipMemsetAsync(d_tag_.p, 0, (size_t)std::max(num_item_set, 1), eng_->stream_));
    d_banmask_.reserve((size_t)std::max(num_item_set, 1));
    RCHECK(hipMemsetAsync(d_banmask_.p, 0, (size_t)std::max(num_item_set, 1) * sizeof(unsigned), eng_->stream_));
    d_tu_tile_.reserve((size_t)RANK_TILE * pitch + 256);   // chunk-major user factors of a tile (+ one group the scoring pass prefetches past the end)
    tile_.clear();
    tile_prev_ban_.clear();
    tag_.assign((size_t)num_item_set, 0);
    tagged_.clear();
    dev_tagged_.clear();
    n_banned_ = 0;
    items_.clear();
    init_end_ = true;
    user_open_ = false;
}

void Ranker::stage(HostCSR &dst, int ng, int nu, int ni, const unsigned *index, const float *value) {
    const int b = dst.row_ptr.back();
    dst.row_label.push_back(0.0f);
    dst.row_ptr.push_back(b + ng);
    dst.row_ptr.push_back(b + ng + nu);
    dst.row_ptr.push_back(b + ng + nu + ni);
    dst.feat_index.insert(dst.feat_index.end(), index, index + ng + nu + ni);
    dst.feat_value.insert(dst.feat_value.end(), value, value + ng + nu + ni);
}
void Ranker::check_item_side(int ng, int nu, int ni, const unsigned *index) {   // asserts of prepare_ifactor (:684,696)
    for (int j = 0; j < ni; j++) rcheck(index[ng + nu + j] < (unsigned)eng_->mp_.num_item, "item feature index exceed setting");
    for (int j = 0; j < ng; j++) rcheck(index[j] < (unsigned)eng_->mp_.num_global, "global feature index exceed setting");
}

long Ranker::process(float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out, long cap) {   // proc (:786-796)
    rcheck(init_end_, "ranker: init_ranker has not been called");
    const int tag = (int)label;
    const unsigned *iu = index + ng;
    switch (tag) {
    case 0: {   // ITEM_TAG: proc_item (:702-707)
        rcheck(num_item_processed_ + 1 <= num_item_set_, "item instance exceed specified item set size");
        check_item_side(ng, nu, ni, index);
        stage(items_, ng, nu, ni, index, value);
        num_item_processed_++;
        return 0;
    }
    case 2: {   // USER_TAG: proc_user (:709-728)
        for (int j = 0; j < nu; j++) rcheck(iu[j] < (unsigned)eng_->mp_.num_user, "user feature index exceed bound");
        user_idx_.assign(iu, iu + nu);
        user_val_.assign(value + ng, value + ng + nu);
        pos_item_.clear();
        for (int idx : tagged_) tag_[(size_t)idx] = 0;   // every candidate is back to "ranked, not positive" (:727)
        tagged_.clear();
        n_banned_ = 0;
        spec_.clear();
        spec_idx_.clear();
        user_open_ = true;
        return 0;
    }
    case 1: case -1: {   // POS_SAMPLE / BAN_SAMPLE: proc_tag (:729-738)
        for (int j = 0; j < nu; j++) {
            const int idx = (int)iu[j];
            rcheck(idx < num_item_processed_, "sample item index exceed bound");
            rcheck(tag_[(size_t)idx] == 0, "each pos sample item can not occur in baned sample list");
            tag_[(size_t)idx] = (signed char)tag;
            tagged_.push_back(idx);
            if (tag == 1) pos_item_.push_back(idx); else n_banned_++;
        }
        return 0;
    }
    case 3: {   // SPEC_SAMPLE: proc_spec (:739-747); a later special sample of the same candidate replaces the earlier one
        rcheck(nu == 1, "must specify item index of sample in user feature field\n");
        const int idx = (int)iu[0];
        rcheck(idx < num_item_processed_, "sample item index exceed bound");
        check_item_side(ng, nu, ni, index);
        for (size_t j = 0; j < spec_idx_.size(); j++)
            if (spec_idx_[j] == idx) { spec_idx_[j] = -1; }   // superseded
        stage(spec_, ng, nu, ni, index, value);
        spec_idx_.push_back(idx);
        return 0;
    }
    case 4: return rank(out, cap);   // PROCESS_TAG: proc_rank (:748-785)
    default: return 0;
    }
}

long Ranker::process_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value, int *out, long cap) {   // :797-812
    rcheck(init_end_, "ranker: init_ranker has not been called");
    if (tag == TAG_DEFAULT || tag == TAG_START) {
        rcheck(eng_->user_group(), "ranker: user-grouped input needs a user-group model (format_type = 1)");
        for (int j = 0; j < nfb; j++) rcheck(ifb[j] < (unsigned)eng_->mp_.num_ufeedback, "ufeedback id exceed bound");
        const DevParams &P = eng_->params();
        w_fbidx_.upload(ifb, (size_t)nfb, eng_->stream_);
        w_fbval_.upload(vfb, (size_t)nfb, eng_->stream_);
        launch_rank_feedback(P, w_fbidx_.p, w_fbval_.p, nfb, d_fb_.p, eng_->stream_);
        RCHECK(hipStreamSynchronize(eng_->stream_));   // the staging buffers are reused by the next block
    }
    long total = 0;
    for (int r = 0; r < num_row; r++) {
        const int p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
        const long got = process(row_label[r], p1 - p0, p2 - p1, p3 - p2, feat_index + p0, feat_value + p0, out ? out + total : nullptr, cap - total);
        total += got;
    }
    return total;
}

struct RankEntry {   // SVDFeatureRanker::Entry (:617-624)
    int iid;
    float score;
    bool operator<(const RankEntry &p) const { return score > p.score; }
};

// A PROCESS_TAG line = one section: enqueue() puts its work on the stream (one pinned upload, k_rank_user, the scoring pass, one
// small readback) into one of RANK_SLOTS slots, resolve() waits for a slot's event and turns the readback into results.
// process() resolves at once; process_rows() -- a whole input of the rank task (svd_feature_infer.cpp:347-375) in one call --
// keeps up to RANK_SLOTS sections in flight, so the per-section cost is the enqueue, not a launch + sync round trip.
void Ranker::slot_reserve_pin(RankSlot &S, size_t words) {
    if (words + 4 <= S.pin_words) return;
    if (S.pin) (void)hipHostFree(S.pin);
    S.pin = nullptr;
    S.pin_words = 2 * (words + 4) + 1024;
    RCHECK(hipHostMalloc(reinterpret_cast<void **>(&S.pin), S.pin_words * sizeof(unsigned), hipHostMallocDefault));
    S.d_stage.reserve(S.pin_words);
}
void Ranker::slot_reserve_back(RankSlot &S, size_t words) {
    if (words <= S.back_words) return;
    if (S.back) (void)hipHostFree(S.back);
    S.back = nullptr;
    S.back_words = 2 * words + 256;
    RCHECK(hipHostMalloc(reinterpret_cast<void **>(&S.back), S.back_words * sizeof(unsigned), hipHostMallocDefault));
}

// the staged tile -> ONE pinned upload, the opening kernel (user factors, positives' scores, ban bits, counters), ONE pass over the
// candidate matrix for all of its sections, ONE readback of the counters; resolved section by section, in order
void Ranker::flush_tile() {
    if (tile_.empty()) return;
    struct Tm { int64_t &acc; std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
                ~Tm() { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); } } tm_{flush_ns_};
    const DevParams &P = eng_->params();
    hipStream_t st = eng_->stream_;
    const long n = tile_n_, cap_items = (long)std::max(num_item_set_, 1);
    if (pending_.size() == RANK_SLOTS) resolve();
    const int slot = next_slot_;
    next_slot_ = (next_slot_ + 1) % RANK_SLOTS;
    RankSlot &L = slots_[slot];
    if (!L.ev) RCHECK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));
    RankTile T;
    memset(&T, 0, sizeof(T));
    T.nsec = (int)tile_.size();
    size_t words = 0;
    int totpos = 0;
    for (int u = 0; u < T.nsec; u++) {
        const TileSec &S = tile_[(size_t)u];
        T.off[u] = (int)words; T.nu[u] = (int)S.user_idx.size(); T.npos[u] = (int)S.pos_item.size(); T.nban[u] = (int)S.banned.size(); T.pos0[u] = totpos;
        words += 2 * S.user_idx.size() + S.pos_item.size() + S.banned.size();
        totpos += T.npos[u];
    }
    const size_t prev_off = words;
    words += tile_prev_ban_.size();
    slot_reserve_pin(L, words);
    {
        unsigned *w = L.pin;
        for (const TileSec &S : tile_) {
            memcpy(w, S.user_idx.data(), S.user_idx.size() * 4); w += S.user_idx.size();
            memcpy(w, S.user_val.data(), S.user_val.size() * 4); w += S.user_val.size();
            memcpy(w, S.pos_item.data(), S.pos_item.size() * 4); w += S.pos_item.size();
            memcpy(w, S.banned.data(), S.banned.size() * 4); w += S.banned.size();
        }
        memcpy(w, tile_prev_ban_.data(), tile_prev_ban_.size() * 4);
    }
    if (words) RCHECK(hipMemcpyAsync(L.d_stage.p, L.pin, words * sizeof(unsigned), hipMemcpyHostToDevice, st));
    L.d_cnt.reserve((size_t)2 * std::max(totpos, 1));
    L.d_flag.reserve(RANK_TILE);
    L.d_ps.reserve((size_t)std::max(totpos, 1));
    L.d_score.reserve((size_t)RANK_TILE * (size_t)cap_items);   // scores (positions mode) or sort keys (top_k) of the tile's sections
    unsigned *d_out = reinterpret_cast<unsigned *>(L.d_score.p);
    // top_k: short prefixes are selected from the keys and per-wave minima of the scoring pass (k_rank_tile_select); long ones by the radix
    // selection, whose work areas (histograms + state of every section) are cleared by the opening kernel
    int take_max = 0;
    for (const TileSec &S : tile_) take_max = std::max(take_max, S.take);
    const bool top = top_k_ > 0;
    const bool quick = top && rank_tile_select_applies(n, cap_items, take_max);
    const size_t sel_ww = (size_t)rank_select_work_words(), sel_rc = (size_t)rank_select_cap();
    const size_t sel_out_stride = (size_t)2 * ((size_t)std::max(top_k_, 0) + 1) + 1;
    const size_t nmin = (size_t)rank_tile_minima(n);
    if (top) d_sel_.reserve((size_t)RANK_TILE * (sel_ww + 2 * sel_rc + sel_out_stride + nmin));
    unsigned *work = d_sel_.p, *ck = work + RANK_TILE * sel_ww, *cv = ck + RANK_TILE * sel_rc, *res = cv + RANK_TILE * sel_rc, *wmin = res + RANK_TILE * sel_out_stride;
    launch_rank_tile_open(P, L.d_stage.p, T, eng_->user_group() ? d_fb_.p : nullptr, d_tu_tile_.p, d_banmask_.p, L.d_stage.p + prev_off, (int)tile_prev_ban_.size(),
                          L.d_cnt.p, L.d_flag.p, cap_items, d_ift_.p, d_ibias_.p, L.d_ps.p, (top && !quick) ? d_sel_.p : nullptr,
                          (top && !quick) ? (long)((size_t)T.nsec * sel_ww) : 0L, st);
    launch_rank_score_tile(P, n, cap_items, d_tu_tile_.p, d_ift_.p, d_ibias_.p, d_banmask_.p, d_out, L.d_stage.p, T, L.d_ps.p, L.d_cnt.p, top ? 1 : 0,
                           top ? wmin : nullptr, L.d_flag.p, st);
    RCHECK(hipGetLastError());
    RankPending Q;
    Q.slot = slot; Q.n = n; Q.nsec = T.nsec;
    if (top) {
        RselSecs Ks;
        memset(&Ks, 0, sizeof(Ks));
        for (int u = 0; u < T.nsec; u++) Ks.K1[u] = (unsigned)tile_[(size_t)u].take;
        if (quick) launch_rank_tile_select(n, cap_items, T.nsec, d_out, wmin, Ks, res, (long)sel_out_stride, L.d_flag.p, st);
        else launch_rank_select_tile(n, cap_items, T.nsec, d_out, Ks, work, ck, cv, res, (long)sel_out_stride, L.d_flag.p, st);
        RCHECK(hipGetLastError());
        slot_reserve_back(L, (size_t)T.nsec * sel_out_stride);
        RCHECK(hipMemcpyAsync(L.back, res, (size_t)T.nsec * sel_out_stride * sizeof(unsigned), hipMemcpyDeviceToHost, st));
        Q.out_stride = (long)sel_out_stride;
        for (int u = 0; u < T.nsec; u++) Q.tile_take.push_back(tile_[(size_t)u].take);
    } else {
        slot_reserve_back(L, (size_t)2 * totpos);
        RCHECK(hipMemcpyAsync(L.back, L.d_cnt.p, (size_t)2 * totpos * sizeof(int), hipMemcpyDeviceToHost, st));
    }
    RCHECK(hipEventRecord(L.ev, st));
    tile_prev_ban_.clear();
    for (TileSec &S : tile_) {
        for (int idx : S.banned) tile_prev_ban_.push_back(idx);
        Q.tile_pos.push_back(std::move(S.pos_item));
        Q.tile_ban.push_back(std::move(S.banned));
    }
    std::sort(tile_prev_ban_.begin(), tile_prev_ban_.end());
    tile_prev_ban_.erase(std::unique(tile_prev_ban_.begin(), tile_prev_ban_.end()), tile_prev_ban_.end());
    n_sections_ += T.nsec;
    n_tiles_++;
    tile_.clear();
    pending_.push_back(std::move(Q));
}

void Ranker::enqueue() {
    rcheck(user_open_, "ranker: PROCESS_TAG without a USER_TAG section");
    const DevParams &P = eng_->params();
    hipStream_t st = eng_->stream_;
    const long n = num_item_processed_;
    // process_rows, positions mode, no special sample, positives given, nothing new among the candidates: the section joins the tile
    {
        bool no_spec = true;
        for (int v : spec_idx_) no_spec = no_spec && v < 0;
        // ... or top_k mode with a prefix short enough for the radix selection (the positives play no part in a top_k answer, :768-775)
        const long nranked_t = n - n_banned_;
        const long take_t = top_k_ > 0 ? std::min<long>(nranked_t, (long)top_k_ + 1) : 0;
        const bool tile_pos = top_k_ <= 0 && !pos_item_.empty();
        const bool tile_top = top_k_ > 0 && nranked_t >= (long)top_k_ && take_t <= rank_select_cap() / 2;
        const bool tileable = deferred_ && tile_enabled_ && (tile_pos || tile_top) && no_spec && n > 0 && items_on_device_ == n;
        if (!tile_.empty() && (!tileable || tile_n_ != n)) flush_tile();
        if (tileable) {
            TileSec S;
            S.user_idx = user_idx_; S.user_val = user_val_;
            if (tile_pos) S.pos_item = pos_item_;
            S.take = (int)take_t;
            for (int idx : tagged_) if (tag_[(size_t)idx] == -1) S.banned.push_back(idx);
            tile_n_ = n;
            tile_.push_back(std::move(S));
            if ((int)tile_.size() == RANK_TILE) flush_tile();
            return;
        }
    }
    // candidates that arrived since the last section
    if (items_on_device_ < n) {
        w_label_.upload(items_.row_label.data(), items_.row_label.size(), st);
        w_ptr_.upload(items_.row_ptr.data(), items_.row_ptr.size(), st);
        w_index_.upload(items_.feat_index.data(), items_.feat_index.size(), st);
        w_value_.upload(items_.feat_value.data(), items_.feat_value.size(), st);
        DevCSR D{w_label_.p, w_ptr_.p, w_index_.p, w_value_.p};
        launch_rank_items(P, D, items_on_device_, n, d_ifactors_.p, d_ibias_.p, st);
        launch_rank_transpose(P, items_on_device_, n, (long)std::max(num_item_set_, 1), d_ifactors_.p, d_ift_.p, st);
        RCHECK(hipStreamSynchronize(st));
        items_on_device_ = n;
    }
    // special samples: only the last one per candidate counts (an assignment, :746)
    HostCSR live;
    std::vector<int> live_idx;
    for (size_t j = 0; j < spec_idx_.size(); j++) {
        if (spec_idx_[j] < 0) continue;
        const int *p = &spec_.row_ptr[3 * j];
        stage(live, p[1] - p[0], p[2] - p[1], p[3] - p[2], spec_.feat_index.data() + p[0], spec_.feat_value.data() + p[0]);
        live_idx.push_back(spec_idx_[j]);
    }
    const bool fresh = live_idx.empty();   // no special sample this section: k_rank_score writes 0 + (bias + dot) without reading item_score
    const long cap_items = (long)std::max(num_item_set_, 1);
    const int nu = (int)user_idx_.size(), npos = (int)pos_item_.size();
    const long nranked = n - n_banned_;
    if (top_k_ > 0) rcheck(nranked >= (long)top_k_, "k can not exceed candidate size");

    if (pending_.size() == RANK_SLOTS) resolve();
    const int slot = next_slot_;
    next_slot_ = (next_slot_ + 1) % RANK_SLOTS;
    RankSlot &L = slots_[slot];
    if (!L.ev) RCHECK(hipEventCreateWithFlags(&L.ev, hipEventDisableTiming));

    // one pinned upload per section: the user's rows, the positives, and the tag changes against the device's tag array
    const RankSection S{nu, npos, (int)dev_tagged_.size(), (int)tagged_.size()};
    const size_t words = (size_t)2 * nu + npos + S.nprev + 2 * (size_t)S.nnew;
    slot_reserve_pin(L, words);
    {
        unsigned *w = L.pin;
        memcpy(w, user_idx_.data(), (size_t)nu * 4); w += nu;
        memcpy(w, user_val_.data(), (size_t)nu * 4); w += nu;
        memcpy(w, pos_item_.data(), (size_t)npos * 4); w += npos;
        memcpy(w, dev_tagged_.data(), (size_t)S.nprev * 4); w += S.nprev;
        memcpy(w, tagged_.data(), (size_t)S.nnew * 4); w += S.nnew;
        for (int idx : tagged_) *w++ = (unsigned)(int)tag_[(size_t)idx];
    }
    if (words) RCHECK(hipMemcpyAsync(L.d_stage.p, L.pin, words * sizeof(unsigned), hipMemcpyHostToDevice, st));
    L.d_cnt.reserve((size_t)2 * std::max(npos, 1));
    L.d_flag.reserve(1);
    L.d_ps.reserve((size_t)std::max(npos, 1));
    L.d_score.reserve((size_t)cap_items);
    // positions mode without special samples: the positives' scores are known before the scoring pass, which then counts
    const bool fused_positions = top_k_ <= 0 && fresh && npos > 0 && n > 0;
    // top_k with a short prefix: radix selection (its work area is zeroed by k_rank_user, its first pass rides on the scoring pass)
    const size_t take = top_k_ > 0 ? (size_t)std::min<long>(nranked, (long)top_k_ + 1) : 0;
    const bool select = top_k_ > 0 && n > 0 && (long)take <= rank_select_cap() / 2;
    if (select) d_sel_.reserve((size_t)rank_select_work_words() + 4 * (size_t)rank_select_cap() + 4);
    launch_rank_user(P, L.d_stage.p, S, eng_->user_group() ? d_fb_.p : nullptr, d_tu_.p, d_tag_.p, L.d_cnt.p, L.d_flag.p, cap_items, d_ift_.p, d_ibias_.p,
                     fused_positions ? L.d_ps.p : nullptr, select ? d_sel_.p : nullptr, select ? (int)rank_select_work_words() : 0, st);
    dev_tagged_ = tagged_;
    const int *d_pos = reinterpret_cast<const int *>(L.d_stage.p + 2 * nu);
    n_sections_++;

    RankPending Q;
    Q.slot = slot; Q.n = n; Q.npos = 0; Q.take = 0;
    if (n > 0) {
        if (!fresh) {
            s_label_.upload(live.row_label.data(), live.row_label.size(), st);
            s_ptr_.upload(live.row_ptr.data(), live.row_ptr.size(), st);
            s_index_.upload(live.feat_index.data(), live.feat_index.size(), st);
            s_value_.upload(live.feat_value.data(), live.feat_value.size(), st);
            s_idx_.upload(live_idx.data(), live_idx.size(), st);
            DevCSR D{s_label_.p, s_ptr_.p, s_index_.p, s_value_.p};
            RCHECK(hipMemsetAsync(L.d_score.p, 0, (size_t)n * sizeof(float), st));   // item_score = 0 (:726)
            launch_rank_spec(P, D, (long)live_idx.size(), s_idx_.p, d_tu_.p, L.d_score.p, st);
            RCHECK(hipStreamSynchronize(st));   // the host staging vectors go out of use
        }
        RankFused F{0, d_pos, L.d_ps.p, npos, L.d_cnt.p, L.d_cnt.p + npos, nullptr, nullptr, L.d_flag.p, nullptr};
        if (top_k_ > 0) {
            d_keys_.reserve((size_t)2 * cap_items);
            d_vals_.reserve((size_t)2 * cap_items);
            F.mode = 2; F.keys = d_keys_.p; F.vals = d_vals_.p;
            if (select) F.hist1 = d_sel_.p;
        } else if (fused_positions) {
            F.mode = 1;
        }
        launch_rank_score(P, n, cap_items, d_tu_.p, d_ift_.p, d_ibias_.p, d_tag_.p, L.d_score.p, fresh ? 1 : 0, F, st);
        if (top_k_ > 0) {
            // top_k: order-preserving score keys sorted / selected on the device, only the first top_k+1 (key, candidate) pairs come
            // back (radix selection, svdf_k_rank.hip; a full rocPRIM sort when the prefix is long); the reference's std::sort
            // (:767) decides only when scores tie inside that prefix or a score is NaN
            slot_reserve_back(L, 2 * take + 1);
            if (select) {
                unsigned *work = d_sel_.p, *ck = work + rank_select_work_words(), *cv = ck + rank_select_cap(), *res = cv + rank_select_cap();
                launch_rank_select(n, d_keys_.p, d_vals_.p, (unsigned)take, work, ck, cv, res, L.d_flag.p, st);
                RCHECK(hipMemcpyAsync(L.back, res, (2 * take + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
            } else {
                device_sort_pairs_u32(d_keys_.p, d_keys_.p + n, d_vals_.p, d_vals_.p + n, n, &sort_tmp_, &sort_tmp_bytes_, st);
                RCHECK(hipMemcpyAsync(L.back, d_keys_.p + n, take * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                RCHECK(hipMemcpyAsync(L.back + take, d_vals_.p + n, take * sizeof(unsigned), hipMemcpyDeviceToHost, st));
                RCHECK(hipMemcpyAsync(L.back + 2 * take, L.d_flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, st));
            }
            Q.take = (int)take;
        } else if (npos > 0) {
            if (!fused_positions) launch_rank_positions(n, L.d_score.p, d_tag_.p, d_pos, npos, L.d_cnt.p, L.d_cnt.p + npos, st);
            slot_reserve_back(L, (size_t)2 * npos);
            RCHECK(hipMemcpyAsync(L.back, L.d_cnt.p, (size_t)2 * npos * sizeof(int), hipMemcpyDeviceToHost, st));
            Q.npos = npos;
        }
    } else {
        rcheck(top_k_ <= 0, "k can not exceed candidate size");
    }
    RCHECK(hipEventRecord(L.ev, st));
    if (Q.npos > 0) Q.pos_item = pos_item_;
    for (int idx : tagged_) if (tag_[(size_t)idx] == -1) Q.banned.push_back(idx);
    pending_.push_back(std::move(Q));
}

// The reference's own ordering step (:767) for a section whose scores tie: std::sort over the entry vector (ranked candidates
// in index order).  100 K entries take ~8 ms on one core, two hundred sections' worth of device time: it runs on a helper
// thread over a private copy of the section's scores while the pipeline goes on; results are still reported in section order.
// helper threads for the reference's tie-breaking sort: a quarter of the host's hardware threads, at least 8, at most 32 (each sorts
// a private 100 K-entry copy for ~8 ms; the sections they belong to are reported in order when they finish)
static size_t host_sort_threads() {
    static const size_t n = std::min<size_t>(32, std::max<size_t>(8, std::thread::hardware_concurrency() / 4));
    return n;
}
// k_rank_score_tile<., 1> keeps the sort KEY of a candidate where the positions pass keeps its score: the key back to a score the
// reference's comparator cannot tell from the original (-0 comes back as +0, every NaN as one NaN; banned candidates carry no score)
static void keys_to_scores(std::vector<float> &v) {
    for (float &f : v) {
        unsigned key;
        memcpy(&key, &f, 4);
        unsigned bits;
        if (key >= 0xFFFFFFFEu) bits = 0x7FC00000u;
        else { const unsigned up = ~key; bits = (up & 0x80000000u) ? (up & 0x7FFFFFFFu) : ~up; }
        memcpy(&f, &bits, 4);
    }
}
// std::sort over the reference's entry vector (:767; Entry::operator<: score > p.score) executed by several threads.  libstdc++'s
// introsort (restated in svdf_stdsort.h, checked against the library's own std::sort) recurses into DISJOINT sub-ranges after every
// partition, so the partitions may run in any order and concurrently: the ranges a partition splits off go to a shared queue when they are
// large.  Its final insertion sort is stable and never moves an element across a partition cut, so it runs per region between the cuts
// that were shared.  Same comparisons on the same ranges: the permutation of tied scores is the library's.  a[]: candidate ids.
struct ByScoreDesc {
    const float *score;
    bool operator()(int a, int b) const { return score[a] > score[b]; }
};
// helper threads of the sorts, started once (starting a thread inside a process that holds a GPU context measured ~170 us here: a sort
// that starts its own helpers is slower than the sequential one)
class SortPool {
  public:
    static SortPool &get() { static SortPool p; return p; }
    void submit(std::function<void()> f) {
        { std::lock_guard<std::mutex> lk(mu_); q_.push_back(std::move(f)); }
        cv_.notify_one();
    }
    int size() const { return (int)th_.size(); }
  private:
    SortPool() {
        const unsigned hw = std::thread::hardware_concurrency();
        const int n = (int)std::min<unsigned>(32u, std::max<unsigned>(4u, hw / 4u));
        for (int t = 0; t < n; t++) th_.emplace_back([this] { run(); });
    }
    ~SortPool() {
        { std::lock_guard<std::mutex> lk(mu_); stop_ = true; }
        cv_.notify_all();
        for (std::thread &t : th_) t.join();
    }
    void run() {
        for (;;) {
            std::function<void()> f;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return stop_ || !q_.empty(); });
                if (q_.empty()) return;
                f = std::move(q_.front());
                q_.pop_front();
            }
            f();
        }
    }
    std::vector<std::thread> th_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<std::function<void()>> q_;
    bool stop_ = false;
};

namespace {
struct SortFrame { int *first, *last; int depth; };
// what the threads of one sort share; helpers hold it by shared_ptr (one that is scheduled after the sort has finished finds nothing to do)
struct SortJob {
    ByScoreDesc comp;
    std::mutex mu;
    std::condition_variable cv;
    std::vector<SortFrame> queue;
    std::vector<int *> cuts;
    long open_frames = 0;          // frames queued or being worked on
    std::vector<int *> bounds;     // phase 2: regions of the final insertion sort
    std::atomic<size_t> next{0}, done{0};
    static constexpr long GRAIN = 2048;
    void partitions() {            // phase 1: __introsort_loop over the queued frames
        using namespace stdsort;
        std::vector<SortFrame> local;
        for (;;) {
            SortFrame f;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return !queue.empty() || open_frames == 0; });
                if (queue.empty()) return;
                f = queue.back();
                queue.pop_back();
            }
            local.push_back(f);
            while (!local.empty()) {
                SortFrame g = local.back();
                local.pop_back();
                while (g.last - g.first > 16) {
                    if (g.depth == 0) { heap_sort(g.first, g.last, comp); break; }
                    --g.depth;
                    int *mid = g.first + (g.last - g.first) / 2;
                    move_median_to_first(g.first, g.first + 1, mid, g.last - 1, comp);
                    int *cut = unguarded_partition(g.first + 1, g.last, g.first, comp);
                    const SortFrame right{cut, g.last, g.depth};
                    if (right.last - right.first >= GRAIN && cut - g.first >= GRAIN) {
                        std::lock_guard<std::mutex> lk(mu);
                        queue.push_back(right);
                        cuts.push_back(cut);
                        open_frames++;
                        cv.notify_one();
                    } else if (right.last - right.first > 16) {
                        local.push_back(right);
                    }
                    g.last = cut;
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (--open_frames == 0) cv.notify_all();
            }
        }
    }
    void regions() {               // phase 2: __final_insertion_sort, region by region
        for (;;) {
            const size_t r = next.fetch_add(1);
            if (r + 1 >= bounds.size()) return;
            stdsort::insertion_sort(bounds[r], bounds[r + 1], comp);
            if (done.fetch_add(1) + 2 == bounds.size()) { std::lock_guard<std::mutex> lk(mu); cv.notify_all(); }
        }
    }
};
}  // namespace

void host_parallel_sort_scores(const float *score, int *a, long n, int nthreads) {
    if (n <= 1) return;
    auto job = std::make_shared<SortJob>();
    job->comp = ByScoreDesc{score};
    job->queue.push_back(SortFrame{a, a + n, stdsort::floor_log2(n) * 2});
    job->open_frames = 1;
    const int T = nthreads <= 1 ? 1 : (int)std::max<long>(1, std::min<long>(std::min<long>(nthreads, SortPool::get().size() + 1), n / SortJob::GRAIN));
    for (int t = 1; t < T; t++) SortPool::get().submit([job] { job->partitions(); });
    job->partitions();   // (returns when no frame is queued or being worked on: every partition is done)
    {
        std::unique_lock<std::mutex> lk(job->mu);
        job->cv.wait(lk, [&] { return job->open_frames == 0; });
    }
    // every region start is a partition cut: nothing to its left sorts behind anything in it, and the insertion sort is stable
    std::vector<int *> &cuts = job->cuts;
    std::sort(cuts.begin(), cuts.end());
    cuts.erase(std::unique(cuts.begin(), cuts.end()), cuts.end());
    job->bounds.push_back(a);
    for (int *c : cuts) if (c > a && c < a + n) job->bounds.push_back(c);
    job->bounds.push_back(a + n);
    const size_t nreg = job->bounds.size() - 1;
    const int T2 = (int)std::max<size_t>(1, std::min<size_t>((size_t)T, nreg));
    for (int t = 1; t < T2; t++) SortPool::get().submit([job] { job->regions(); });
    job->regions();
    std::unique_lock<std::mutex> lk(job->mu);
    job->cv.wait(lk, [&] { return job->done.load() == nreg; });
}

static int rank_sort_threads() {
    static const int v = [] { const char *e = getenv("SVDF_RANK_SORT_THREADS"); const int x = e ? atoi(e) : 0; return x > 0 ? x : 8; }();
    return v;
}
// Work areas of a tied section's sort (100 K candidates: 1.3 MB), recycled.  Freed and re-allocated per section they are mmap / munmap
// calls, and every munmap of a process that holds a GPU context runs the driver's MMU notifier, which stalls the device queues: a
// process_rows call of 38 tiles with 7 tied sections took 5 ms or 40 ms depending on what malloc did with the blocks
// (MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ = never: always 5 ms).
struct TieWork {
    std::vector<float> score;
    std::vector<int> entry, where;
    std::vector<char> banned;
};
static std::mutex g_tiework_mu;
static std::vector<std::unique_ptr<TieWork>> g_tiework_free;
static std::unique_ptr<TieWork> tiework_get() {
    std::lock_guard<std::mutex> lk(g_tiework_mu);
    if (g_tiework_free.empty()) return std::unique_ptr<TieWork>(new TieWork());
    std::unique_ptr<TieWork> w = std::move(g_tiework_free.back());
    g_tiework_free.pop_back();
    return w;
}
static void tiework_put(std::unique_ptr<TieWork> w) {
    std::lock_guard<std::mutex> lk(g_tiework_mu);
    if (g_tiework_free.size() < 64) g_tiework_free.push_back(std::move(w));
}

static std::vector<int> host_sort_section(std::unique_ptr<TieWork> w, std::vector<int> banned_idx, std::vector<int> pos_item, int top_k) {
    const std::vector<float> &score = w->score;
    const long n = (long)score.size();
    std::vector<char> &banned = w->banned;
    banned.assign((size_t)n, 0);
    for (int idx : banned_idx) banned[(size_t)idx] = 1;
    std::vector<int> &entry = w->entry;   // the ranked candidates in index order (the reference's entry vector, :749-766)
    entry.clear();
    entry.reserve((size_t)n);
    for (long i = 0; i < n; i++)
        if (!banned[(size_t)i]) entry.push_back((int)i);
    host_parallel_sort_scores(score.data(), entry.data(), (long)entry.size(), rank_sort_threads());
    std::vector<int> out;
    if (top_k > 0) {
        for (int k = 0; k < top_k; k++) out.push_back(entry[(size_t)k]);
    } else {
        std::vector<int> &where = w->where;
        where.assign((size_t)n, 0);
        for (size_t i = 0; i < entry.size(); i++) where[(size_t)entry[i]] = (int)i;
        for (int p : pos_item) out.push_back(where[(size_t)p]);
    }
    tiework_put(std::move(w));
    return out;
}

// a section's sort on a thread of the pool (it takes part in its own partitions, so it finishes even when every other helper is busy)
static std::future<std::vector<int>> sort_section_async(std::unique_ptr<TieWork> w, std::vector<int> banned, std::vector<int> pos_item, int top_k) {
    std::shared_ptr<TieWork> hold(w.release());   // (a copyable holder for the task; handed on as the unique owner when it runs)
    auto task = std::make_shared<std::packaged_task<std::vector<int>()>>(
        [hold, banned = std::move(banned), pos_item = std::move(pos_item), top_k]() mutable {
            std::unique_ptr<TieWork> mine(new TieWork());
            std::swap(*mine, *hold);
            return host_sort_section(std::move(mine), std::move(banned), std::move(pos_item), top_k);
        });
    std::future<std::vector<int>> fut = task->get_future();
    if (rank_sort_threads() <= 1) std::thread([task] { (*task)(); }).detach();   // (measurements: no pool at all)
    else SortPool::get().submit([task] { (*task)(); });
    return fut;
}

// scores of a tied section for the host's sort.  The slot's event has passed, so its scores are final: they are copied on a stream of their
// own into a pinned buffer -- the pipeline's stream keeps running (a synchronise there drains every tile in flight), and a pageable
// destination would have the runtime pin and unpin its pages around the copy
std::unique_ptr<TieWork> Ranker::tie_scores(const float *d_src, long n) {
    if (!tie_stream_) RCHECK(hipStreamCreateWithFlags(&tie_stream_, hipStreamNonBlocking));
    if ((size_t)n > tie_pin_floats_) {
        if (tie_pin_) (void)hipHostFree(tie_pin_);
        tie_pin_ = nullptr;
        tie_pin_floats_ = (size_t)n + 1024;
        RCHECK(hipHostMalloc(reinterpret_cast<void **>(&tie_pin_), tie_pin_floats_ * sizeof(float), hipHostMallocDefault));
    }
    RCHECK(hipMemcpyAsync(tie_pin_, d_src, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, tie_stream_));
    RCHECK(hipStreamSynchronize(tie_stream_));
    std::unique_ptr<TieWork> w = tiework_get();
    w->score.assign(tie_pin_, tie_pin_ + n);
    return w;
}

// oldest section in flight -> its results (or the future of its host sort) appended to the result chunks
void Ranker::resolve() {
    RankPending Q = std::move(pending_.front());
    pending_.pop_front();
    RankSlot &L = slots_[Q.slot];
    {
        const auto t0_ = std::chrono::steady_clock::now();
        // poll: a blocking wait sleeps on an interrupt, and the wake-up measured up to ~0.4 ms per tile here (a process_rows call of 38 tiles
        // took 5 ms or 40 ms depending on it); the pipeline is a few tiles deep, so the event is usually a few microseconds away
        for (;;) {
            const hipError_t query = hipEventQuery(L.ev);
            if (query == hipSuccess) break;
            if (query != hipErrorNotReady) RCHECK(query);   // (RCHECK declares an e_ of its own)
            __builtin_ia32_pause();
        }
        event_wait_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
    }
    const long n = Q.n;
    RankChunk C;
    auto host_sort = [&]() {
        std::unique_ptr<TieWork> score = tie_scores(L.d_score.p, n);   // the slot's item_score is free again after this
        n_host_sorts_++;
        size_t running = 0;
        for (RankChunk &c : chunks_) running += c.pending ? 1 : 0;
        if (running >= host_sort_threads())   // bound the helper threads: wait for the oldest unfinished sort
            for (RankChunk &c : chunks_) if (c.pending) { c.vals = c.fut.get(); c.pending = false; break; }
        C.pending = true;
        C.fut = sort_section_async(std::move(score), Q.banned, Q.pos_item, top_k_);
    };
    if (Q.nsec > 0) {   // a tile: its sections in order, each with its own counters; a tie sends that one section to the host's sort
        const int *cnt = reinterpret_cast<const int *>(L.back);
        const long cap_items = (long)std::max(num_item_set_, 1);
        int p0 = 0;
        for (int u = 0; u < Q.nsec; u++) {
            const int npos = (int)Q.tile_pos[(size_t)u].size();
            RankChunk Cu;
            bool ties = false;
            if (!Q.tile_take.empty()) {   // top_k: the selected prefix decides unless scores tie inside it, a score is NaN, or the selection overflowed
                const size_t take = (size_t)Q.tile_take[(size_t)u];
                const unsigned *hk = L.back + (size_t)u * (size_t)Q.out_stride, *hv = hk + take;
                ties = hk[2 * take] != 0;
                for (size_t j = 0; j + 1 < take; j++) ties = ties || hk[j] == hk[j + 1];
                if (!ties) for (int k = 0; k < top_k_; k++) Cu.vals.push_back((int)hv[(size_t)k]);
            }
            for (int j = 0; j < npos; j++) ties = ties || cnt[(size_t)2 * p0 + npos + j] != 0;
            if (ties) {
                const auto t0_ = std::chrono::steady_clock::now();
                std::unique_ptr<TieWork> score = tie_scores(L.d_score.p + (size_t)u * cap_items, n);
                tie_copy_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
                if (!Q.tile_take.empty()) keys_to_scores(score->score);   // top_k tiles hold sort keys
                n_host_sorts_++;
                size_t running = 0;
                for (RankChunk &c : chunks_) running += c.pending ? 1 : 0;
                if (running >= host_sort_threads())
                    for (RankChunk &c : chunks_) if (c.pending) { c.vals = c.fut.get(); c.pending = false; break; }
                Cu.pending = true;
                Cu.fut = sort_section_async(std::move(score), Q.tile_ban[(size_t)u], Q.tile_pos[(size_t)u], top_k_);
            } else {
                for (int j = 0; j < npos; j++) Cu.vals.push_back(cnt[(size_t)2 * p0 + j]);
            }
            chunks_.push_back(std::move(Cu));
            p0 += npos;
        }
        return;
    }
    if (Q.take > 0) {
        const size_t take = (size_t)Q.take;
        const unsigned *hk = L.back, *hv = L.back + take;
        bool exact = L.back[2 * take] == 0;
        for (size_t j = 0; j + 1 < take; j++) if (hk[j] == hk[j + 1]) exact = false;
        if (exact) for (int k = 0; k < top_k_; k++) C.vals.push_back((int)hv[(size_t)k]);
        else host_sort();
    } else if (Q.npos > 0) {
        const int npos = Q.npos;
        const int *cnt = reinterpret_cast<const int *>(L.back);
        bool ties = false;
        for (int j = 0; j < npos; j++) ties = ties || cnt[(size_t)npos + j] != 0;
        if (ties) host_sort();   // positions inside a group of equal scores: the reference's sort decides
        else for (int j = 0; j < npos; j++) C.vals.push_back(cnt[(size_t)j]);
    }
    if (C.pending || !C.vals.empty()) chunks_.push_back(std::move(C));
}

// results of the resolved sections, in section order, to the caller's buffer
void Ranker::flush_chunks() {
    while (!chunks_.empty()) {
        RankChunk &c = chunks_.front();
        if (c.pending) {
            const auto t0_ = std::chrono::steady_clock::now();
            c.vals = c.fut.get();
            c.pending = false;
            tie_wait_ns_ += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0_).count();
        }
        for (int v : c.vals) { if (out_n_ < out_cap_ && out_ptr_) out_ptr_[out_n_] = v; out_n_++; }
        chunks_.pop_front();
    }
}

void Ranker::drain_quietly() {   // after an error: nothing of the sections in flight is reported
    tile_.clear();
    (void)hipStreamSynchronize(eng_->stream_);
    pending_.clear();
    for (RankChunk &c : chunks_) if (c.pending) { try { (void)c.fut.get(); } catch (...) {} }
    chunks_.clear();
}

long Ranker::rank(int *out, long cap) {
    if (!deferred_) { out_ptr_ = out; out_cap_ = cap; out_n_ = 0; }
    enqueue();
    if (deferred_) return 0;
    while (!pending_.empty()) resolve();
    flush_chunks();
    return out_n_;
}

long Ranker::process_rows(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, int *out, long cap) {
    rcheck(init_end_, "ranker: init_ranker has not been called");
    out_ptr_ = out; out_cap_ = cap; out_n_ = 0;
    deferred_ = true;
    try {
        for (int r = 0; r < num_row; r++) {
            const int p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
            process(row_label[r], p1 - p0, p2 - p1, p3 - p2, feat_index + p0, feat_value + p0, nullptr, 0);
        }
        flush_tile();
        while (!pending_.empty()) resolve();
        flush_chunks();
    } catch (...) {
        deferred_ = false;
        drain_quietly();
        throw;
    }
    deferred_ = false;
    return out_n_;
}

}  // namespace svdf
