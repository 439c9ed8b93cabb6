// svdf_sched.cpp -- part of the host engine (class Engine, svdf_engine.h): conflict-free level scheduling on the host and the glue to the device scheduler, few-row / user-unit schedules
// Reference citations are relative to /root/reference.
#include <cstdio>
#include <cstdlib>
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

// =============================================================================== scheduler
void build_schedule(const std::vector<int> &levels, int base, Schedule &out) {
    const long n = (long)levels.size();
    int maxl = base;
    for (long t = 0; t < n; t++) maxl = std::max(maxl, levels[(size_t)t]);
    const int nl = maxl - base;                 // number of batches; batch j holds level base+1+j
    std::vector<long> cnt((size_t)nl + 1, 0);
    for (long t = 0; t < n; t++) cnt[(size_t)(levels[(size_t)t] - base)]++;
    out.level_ptr.assign((size_t)nl + 1, 0);
    out.max_level_size = 0;
    long acc = 0;
    for (int j = 0; j < nl; j++) {
        out.level_ptr[(size_t)j] = acc;
        acc += cnt[(size_t)j + 1];
        out.max_level_size = std::max(out.max_level_size, cnt[(size_t)j + 1]);
    }
    out.level_ptr[(size_t)nl] = acc;
    std::vector<long> cursor(out.level_ptr.begin(), out.level_ptr.end());
    out.order.resize((size_t)n);
    for (long t = 0; t < n; t++) out.order[(size_t)cursor[(size_t)(levels[(size_t)t] - base - 1)]++] = (int)t;   // counting sort: stable
}
// =============================================================================== scheduling helpers
int Engine::level_of_row(const unsigned *ig, int ng, const unsigned *iu, int nu, const unsigned *ii, int ni, int lvl) {
    const int *last = tracker_.last.data();
    const size_t goff = (size_t)n_uiset_;
    if (!relax_global_) for (int j = 0; j < ng; j++) lvl = std::max(lvl, last[goff + ig[j]]);
    for (int j = 0; j < nu; j++) {
        const unsigned uid = iu[j];
        if (uid >= relax_user_from_) continue;   // shared id in relaxed mode: not a scheduling resource
        lvl = std::max(lvl, last[user_off_ + uid]);
        if (uid < feat_user_.num_row())
            for (unsigned c = feat_user_.row_ptr[uid]; c < feat_user_.row_ptr[uid + 1]; c++) lvl = std::max(lvl, last[user_off_ + feat_user_.index[c]]);
    }
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        if (iid >= relax_item_from_) continue;
        lvl = std::max(lvl, last[item_off_ + iid]);
        if (iid < feat_item_.num_row())
            for (unsigned c = feat_item_.row_ptr[iid]; c < feat_item_.row_ptr[iid + 1]; c++) lvl = std::max(lvl, last[item_off_ + feat_item_.index[c]]);
    }
    return lvl;
}
void Engine::touch_row(const unsigned *ig, int ng, const unsigned *iu, int nu, const unsigned *ii, int ni, int lvl) {
    int *last = tracker_.last.data();
    const size_t goff = (size_t)n_uiset_;
    if (!relax_global_) for (int j = 0; j < ng; j++) last[goff + ig[j]] = lvl;
    for (int j = 0; j < nu; j++) {
        const unsigned uid = iu[j];
        if (uid >= relax_user_from_) continue;
        last[user_off_ + uid] = lvl;
        if (uid < feat_user_.num_row())
            for (unsigned c = feat_user_.row_ptr[uid]; c < feat_user_.row_ptr[uid + 1]; c++) last[user_off_ + feat_user_.index[c]] = lvl;
    }
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        if (iid >= relax_item_from_) continue;
        last[item_off_ + iid] = lvl;
        if (iid < feat_item_.num_row())
            for (unsigned c = feat_item_.row_ptr[iid]; c < feat_item_.row_ptr[iid + 1]; c++) last[item_off_ + feat_item_.index[c]] = lvl;
    }
}

// Instances of one batch commute, so their order inside the batch is free: sorting a batch by item id (or
// user id) makes neighbouring lane groups touch neighbouring factor rows (DRAM page / TLB locality) without
// changing a single bit of the result.
void sort_batches(Schedule &sched, const unsigned *key) {
    const size_t nl = sched.num_levels();
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    auto work = [&](size_t a, size_t b) {
        for (size_t l = a; l < b; l++)
            std::sort(sched.order.begin() + sched.level_ptr[l], sched.order.begin() + sched.level_ptr[l + 1],
                      [key](int x, int y) { return key[x] < key[y] || (key[x] == key[y] && x < y); });
    };
    if (nl < 64 || hw == 1) { work(0, nl); return; }
    std::vector<std::thread> th;
    const size_t chunk = (nl + hw - 1) / hw;
    for (unsigned t = 0; t < hw; t++) {
        const size_t a = t * chunk, b = std::min(nl, a + chunk);
        if (a >= b) break;
        th.emplace_back(work, a, b);
    }
    for (auto &x : th) x.join();
}

void Engine::flush_csr(HostCSR &src) {
    const long n = src.num_row();
    if (n == 0) return;
    if (punit_flush(src)) return;   // a window of user-grouped rank pairs: user-run units (svdf_punit.cpp)
    need_device("update");
    const DevParams &P = params();
    tracker_.resize(num_resources() + 1);
    const int base = tracker_.base;
    std::vector<int> levels((size_t)n);
    bool basic = basic_fast_path_allowed();
    if (basic) {
        for (long r = 0; r < n && basic; r++) {
            const int *p = &src.row_ptr[(size_t)3 * r];
            basic = (p[1] == p[0]) && (p[2] == p[1] + 1) && (p[3] == p[2] + 1);
        }
    }
    if (basic && device_sched_ && n >= device_sched_min_) {
        // a window of plain (user, item) instances: columns up, levels built on the GPU (svdf_k_sched.hip).  Launches of
        // successive windows are ordered on one stream, so a window is scheduled on its own -- no level state carried over
        std::vector<unsigned> cu((size_t)n), ci((size_t)n);
        std::vector<float> ua((size_t)n), ia((size_t)n);
        bool unit = true;
        for (long r = 0; r < n; r++) {
            cu[(size_t)r] = src.feat_index[(size_t)2 * r]; ci[(size_t)r] = src.feat_index[(size_t)2 * r + 1];
            ua[(size_t)r] = src.feat_value[(size_t)2 * r]; ia[(size_t)r] = src.feat_value[(size_t)2 * r + 1];
            unit = unit && ua[(size_t)r] == 1.0f && ia[(size_t)r] == 1.0f;
        }
        Dataset &wd = w_dataset_;
        const int res_col[2] = {0, 1};
        const unsigned off[2] = {0u, (unsigned)mp_.num_user}, limit[2] = {(unsigned)mp_.num_user, (unsigned)mp_.num_item};
        const char *msg[2] = {"user feature index exceed bound", "item feature index exceed bound"};
        const int sort_col = sort_batches_ == 1 ? 1 : (sort_batches_ == 2 ? 0 : -1);
        std::vector<FCol> fc{FCol{src.row_label.data(), &w_label_}};
        if (!unit) { fc.push_back(FCol{ua.data(), &w_uval_}); fc.push_back(FCol{ia.data(), &w_ival_}); }
        schedule_columns_on_device(&wd, n, 2, res_col, off, limit, msg, sort_col, sort_col >= 0 ? limit[sort_col] : 0u,
                                   {UCol{cu.data(), &w_user_}, UCol{ci.data(), &w_item_}}, fc);
        BasicSchedule S{w_user_.p, w_item_.p, w_label_.p, unit ? nullptr : w_uval_.p, unit ? nullptr : w_ival_.p};
        const Schedule &sc = wd.sched;
        for (size_t l = 0; l < sc.num_levels(); l++) {
            launch_basicmf(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
            n_launches_++; n_kind_[0]++;
        }
        HIPCHECK(hipGetLastError());
        n_batches_ += (int64_t)sc.num_levels();
        n_instances_ += n;
        sample_counter_ += (unsigned)n;
        n_flushes_++;
        src.clear();
        return;
    }
    int *last = tracker_.last.data();
    for (long r = 0; r < n; r++) {
        const int *p = &src.row_ptr[(size_t)3 * r];
        const unsigned *idx = src.feat_index.data();
        int lvl = level_of_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], base) + 1;
        touch_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], lvl);
        levels[(size_t)r] = lvl;
    }
    (void)last;
    Schedule sched;
    build_schedule(levels, base, sched);
    tracker_.base = base + (int)sched.num_levels();
    if (basic) {
        std::vector<unsigned> su((size_t)n), si((size_t)n);
        std::vector<float> sl((size_t)n), sua((size_t)n), sia((size_t)n);
        bool unit = true;
        for (long s = 0; s < n; s++) {
            const long r = sched.order[(size_t)s];
            su[(size_t)s] = src.feat_index[(size_t)2 * r];
            si[(size_t)s] = src.feat_index[(size_t)2 * r + 1];
            sl[(size_t)s] = src.row_label[(size_t)r];
            sua[(size_t)s] = src.feat_value[(size_t)2 * r];
            sia[(size_t)s] = src.feat_value[(size_t)2 * r + 1];
            unit = unit && sua[(size_t)s] == 1.0f && sia[(size_t)s] == 1.0f;
        }
        w_user_.upload(su.data(), (size_t)n, stream_);
        w_item_.upload(si.data(), (size_t)n, stream_);
        w_label_.upload(sl.data(), (size_t)n, stream_);
        BasicSchedule S{w_user_.p, w_item_.p, w_label_.p, nullptr, nullptr};
        if (!unit) {
            w_uval_.upload(sua.data(), (size_t)n, stream_);
            w_ival_.upload(sia.data(), (size_t)n, stream_);
            S.uval = w_uval_.p; S.ival = w_ival_.p;
        }
        HIPCHECK(hipStreamSynchronize(stream_));
        for (size_t l = 0; l < sched.num_levels(); l++) {
            launch_basicmf(P, S, sched.level_ptr[l], sched.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
            n_launches_++; n_kind_[0]++;
        }
    } else if (fused_allowed() && fused_shape_ok(n, src.row_ptr.data(), src.feat_index.data(), w_fused_host_)) {
        fill_fused(n, src.row_label.data(), src.row_ptr.data(), src.feat_index.data(), src.feat_value.data(),
                   sched.order.data(), w_fused_host_);
        w_fused_.upload(w_fused_host_, stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
        const FusedSchedule S = w_fused_.view();
        for (size_t l = 0; l < sched.num_levels(); l++) {
            launch_fused(P, S, w_fused_.max_nu, w_fused_.max_ni, sched.level_ptr[l], sched.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
            n_launches_++; n_kind_[2]++;
        }
    } else {
        check(!relaxed(), "svdfeature_amd: relaxed shared ids need few-row instances (at most 2 user and 2 item ids, no side tables)");
        w_label_.upload(src.row_label.data(), (size_t)n, stream_);
        w_ptr_.upload(src.row_ptr.data(), src.row_ptr.size(), stream_);
        w_index_.upload(src.feat_index.data(), src.feat_index.size(), stream_);
        w_value_.upload(src.feat_value.data(), src.feat_value.size(), stream_);
        w_order_.upload(sched.order.data(), (size_t)n, stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
        DevCSR D{w_label_.p, w_ptr_.p, w_index_.p, w_value_.p};
        for (size_t l = 0; l < sched.num_levels(); l++) {
            launch_general(P, D, w_order_.p, sched.level_ptr[l], sched.level_ptr[l + 1], sample_counter_, stream_);
            n_launches_++; n_kind_[1]++;
        }
    }
    HIPCHECK(hipGetLastError());
    n_batches_ += (int64_t)sched.num_levels();
    n_instances_ += n;
    sample_counter_ += (unsigned)n;
    n_flushes_++;
    src.clear();
}

// ---- few-row fused path -------------------------------------------------------------------------
bool Engine::fused_allowed() const {
    return use_fused_ && !lazy_decay() && mp_.num_factor <= max_fast_path_factor() && (!user_group() || rows_as_instances_) && mp_.common_latent_space == 0 && feat_user_.num_row() == 0 && feat_item_.num_row() == 0;
}
bool Engine::fused_allowed_for_rows() const {
    return use_fused_ && !lazy_decay() && !relaxed() && mp_.num_factor <= max_fast_path_factor() && mp_.common_latent_space == 0 &&
           feat_user_.num_row() == 0 && feat_item_.num_row() == 0;
}
// every instance has <= 2 user ids, <= 2 item ids and no id twice in a section (ptr is int or int64)
template <typename PtrT>
bool Engine::fused_shape_ok(long n, const PtrT *row_ptr, const unsigned *idx, FusedHost &out) {
    int mu = 1, mi = 1;
    bool has_g = false;
    for (long r = 0; r < n; r++) {
        const PtrT *p = row_ptr + 3 * r;
        const int ng = (int)(p[1] - p[0]), nu = (int)(p[2] - p[1]), ni = (int)(p[3] - p[2]);
        if (nu > 2 || ni > 2) return false;
        if (nu == 2 && idx[p[1]] == idx[p[1] + 1]) return false;
        if (ni == 2 && idx[p[2]] == idx[p[2] + 1]) return false;
        if (ng > 0) has_g = true;   // global ids may repeat: the kernel walks them through memory in order
        mu = std::max(mu, nu);
        mi = std::max(mi, ni);
    }
    out.max_nu = mu; out.max_ni = mi; out.has_g = has_g;
    return true;
}
template <typename PtrT>
void Engine::fill_fused(long n, const float *row_label, const PtrT *row_ptr, const unsigned *idx, const float *val, const int *order,
                        FusedHost &out) {
    out.label.resize((size_t)n);
    for (int a = 0; a < 2; a++) {
        const bool on_u = a < out.max_nu, on_i = a < out.max_ni;
        out.uidx[a].assign(on_u ? (size_t)n : 0, SLOT_ABSENT); out.uval[a].assign(on_u ? (size_t)n : 0, 0.0f);
        out.iidx[a].assign(on_i ? (size_t)n : 0, SLOT_ABSENT); out.ival[a].assign(on_i ? (size_t)n : 0, 0.0f);
    }
    out.gptr.clear(); out.gidx.clear(); out.gval.clear();
    if (out.has_g) out.gptr.assign((size_t)n + 1, 0);
    for (long s = 0; s < n; s++) {
        const long r = order[s];
        const PtrT *p = row_ptr + 3 * r;
        out.label[(size_t)s] = row_label[r];
        for (PtrT j = p[1]; j < p[2]; j++) { out.uidx[j - p[1]][(size_t)s] = idx[j]; out.uval[j - p[1]][(size_t)s] = val[j]; }
        for (PtrT j = p[2]; j < p[3]; j++) { out.iidx[j - p[2]][(size_t)s] = idx[j]; out.ival[j - p[2]][(size_t)s] = val[j]; }
        if (out.has_g) {
            for (PtrT j = p[0]; j < p[1]; j++) { out.gidx.push_back(idx[j]); out.gval.push_back(val[j]); }
            out.gptr[(size_t)s + 1] = (int)out.gidx.size();
        }
    }
    // inline slots: the global ids of an instance next to its user / item slots, when all instances fit
    out.inline_g = out.has_g;
    for (long s = 0; s < n && out.inline_g; s++) {
        const int b = out.gptr[(size_t)s], e = out.gptr[(size_t)s + 1];
        if (e - b > 4) out.inline_g = false;
        for (int x = b; x < e && out.inline_g; x++)
            for (int y = x + 1; y < e; y++) if (out.gidx[(size_t)x] == out.gidx[(size_t)y]) out.inline_g = false;
    }
    for (int j = 0; j < 4; j++) { out.gsi[j].clear(); out.gsv[j].clear(); }
    if (out.inline_g) {
        for (int j = 0; j < 4; j++) { out.gsi[j].assign((size_t)n, (unsigned)SLOT_ABSENT); out.gsv[j].assign((size_t)n, 0.0f); }
        for (long s = 0; s < n; s++) {
            const int b = out.gptr[(size_t)s], e = out.gptr[(size_t)s + 1];
            for (int x = b; x < e; x++) { out.gsi[x - b][(size_t)s] = out.gidx[(size_t)x]; out.gsv[x - b][(size_t)s] = out.gval[(size_t)x]; }
        }
    }
}
void FusedDev::upload(const FusedHost &h, hipStream_t st) {
    max_nu = h.max_nu; max_ni = h.max_ni; has_g = h.has_g;
    label.upload(h.label.data(), h.label.size(), st);
    for (int a = 0; a < 2; a++) {
        uidx[a].upload(h.uidx[a].data(), h.uidx[a].size(), st); uval[a].upload(h.uval[a].data(), h.uval[a].size(), st);
        iidx[a].upload(h.iidx[a].data(), h.iidx[a].size(), st); ival[a].upload(h.ival[a].data(), h.ival[a].size(), st);
    }
    if (has_g) {
        gptr.upload(h.gptr.data(), h.gptr.size(), st);
        gidx.upload(h.gidx.data(), h.gidx.size(), st);
        gval.upload(h.gval.data(), h.gval.size(), st);
    }
    inline_g = h.inline_g;
    if (inline_g)
        for (int j = 0; j < 4; j++) { gsi[j].upload(h.gsi[j].data(), h.gsi[j].size(), st); gsv[j].upload(h.gsv[j].data(), h.gsv[j].size(), st); }
}
FusedSchedule FusedDev::view() const {
    FusedSchedule S;
    S.label = label.p;
    for (int a = 0; a < 2; a++) {
        // unused slots alias slot 0 so that the kernel never dereferences a null pointer for NU/NI = 2 variants
        S.uidx[a] = a < max_nu ? uidx[a].p : uidx[0].p; S.uval[a] = a < max_nu ? uval[a].p : uval[0].p;
        S.iidx[a] = a < max_ni ? iidx[a].p : iidx[0].p; S.ival[a] = a < max_ni ? ival[a].p : ival[0].p;
    }
    S.gptr = has_g ? gptr.p : nullptr;
    S.gidx = has_g ? gidx.p : nullptr;
    S.gval = has_g ? gval.p : nullptr;
    for (int j = 0; j < 4; j++) { S.gsi[j] = inline_g ? gsi[j].p : nullptr; S.gsv[j] = inline_g ? gsv[j].p : nullptr; }
    return S;
}

void Engine::schedule_units(int base, Schedule &sched, std::vector<DevUnit> &du) {
    const long nu = (long)staged_units_.size();
    tracker_.resize(num_resources() + 1);
    const size_t state_res = num_resources();
    std::vector<int> levels((size_t)nu);
    du.resize((size_t)nu);
    int *last = tracker_.last.data();
    const unsigned *idx = staged_.feat_index.data();
    const bool simple_ok = feat_user_.num_row() == 0 && feat_item_.num_row() == 0 && mp_.common_latent_space == 0 && mp_.common_feedback_space == 0;
    // per-call epoch: the stamp of unit t is stamp_epoch_ + t, so marks left by an earlier call (unit indices restart at 0
    // on every flush / dataset build) can never look like "seen in this unit"
    if (simple_ok && stamp_.size() < (size_t)n_uiset_) stamp_.assign((size_t)n_uiset_, -1);
    const int64_t epoch = stamp_epoch_;
    stamp_epoch_ += nu;
    if (relaxed())
        check((relax_item_from_ == 0u || relax_item_from_ == 0xFFFFFFFFu) && relax_user_from_ == 0xFFFFFFFFu,
              "svdfeature_amd: on user-group data the relaxed mode is amd:relax_item_from = 0 (all item rows) and / or amd:relax_feedback = 1");
    staged_fresh_.assign((size_t)staged_.num_row(), 0);
    any_fresh_ = false;
    simple_unit_values_ = true;
    const float *val = staged_.feat_value.data();
    for (long t = 0; t < nu; t++) {
        const HostUnit &u = staged_units_[(size_t)t];
        int lvl = base;
        bool simple = simple_ok && u.row_end > u.row_begin;
        const unsigned uid0 = simple ? idx[staged_.row_ptr[(size_t)3 * u.row_begin]] : 0;
        for (int r = u.row_begin; r < u.row_end; r++) {
            const int *p = &staged_.row_ptr[(size_t)3 * r];
            lvl = level_of_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], lvl);
            if (simple) {
                simple = (p[1] == p[0]) && (p[2] == p[1] + 1) && (p[3] == p[2] + 1) && idx[p[1]] == uid0;
                if (simple) {
                    if (val[p[1]] != 1.0f || val[p[2]] != 1.0f) simple_unit_values_ = false;
                    const unsigned row = item_off_ + idx[p[2]];
                    if (stamp_[row] == epoch + t) { staged_fresh_[(size_t)r] = 1; any_fresh_ = true; }   // the same item again: read at use
                    stamp_[row] = epoch + t;
                }
            }
        }
        for (int j = u.fb_begin; j < u.fb_end; j++) {
            const unsigned row = fb_off_ + staged_fb_index_[(size_t)j];
            if (!relax_feedback_) lvl = std::max(lvl, last[row]);
            if (simple) {
                if (stamp_[row] == epoch + t) simple = false;       // a feedback id listed twice
                stamp_[row] = epoch + t;
            }
        }
        if (u.flags & (UNIT_LOAD | UNIT_SAVE)) lvl = std::max(lvl, last[state_res]);
        lvl += 1;
        for (int r = u.row_begin; r < u.row_end; r++) {
            const int *p = &staged_.row_ptr[(size_t)3 * r];
            touch_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], lvl);
        }
        if (!relax_feedback_) for (int j = u.fb_begin; j < u.fb_end; j++) last[fb_off_ + staged_fb_index_[(size_t)j]] = lvl;
        if (u.flags & (UNIT_LOAD | UNIT_SAVE)) last[state_res] = lvl;
        levels[(size_t)t] = lvl;
        const bool fast_unit = simple && use_simple_units_ && !lazy_decay() && mp_.num_factor <= max_fast_path_factor();
        if (relaxed() && !fast_unit && u.row_end > u.row_begin)
            fail("svdfeature_amd: relaxed shared ids on user-group data need simple units (one user id per row, rows of one item, distinct feedback ids)");
        du[(size_t)t] = DevUnit{u.fb_begin, u.fb_end, u.row_begin, u.row_end, u.flags | (fast_unit ? UNIT_SIMPLE : 0)};
    }
    build_schedule(levels, base, sched);
    // inside a batch the fast-path users go first: they are launched as one wave per user (k_svdpp_wave), the rest as
    // lane groups (k_svdpp); units of a batch are independent, so the split changes nothing but the launch shape
    sched.level_mid.resize(sched.num_levels());
    for (size_t l = 0; l < sched.num_levels(); l++) {
        int *b = sched.order.data() + sched.level_ptr[l], *e = sched.order.data() + sched.level_ptr[l + 1];
        int *m = std::stable_partition(b, e, [&](int t) { return (du[(size_t)t].flags & UNIT_SIMPLE) != 0; });
        sched.level_mid[l] = sched.level_ptr[l] + (long)(m - b);
    }
}
void Engine::upload_unit_arrays(UnitDev &d) {
    d.label.upload(staged_.row_label.data(), staged_.row_label.size(), stream_);
    d.ptr.upload(staged_.row_ptr.data(), staged_.row_ptr.size(), stream_);
    d.index.upload(staged_.feat_index.data(), staged_.feat_index.size(), stream_);
    d.value.upload(staged_.feat_value.data(), staged_.feat_value.size(), stream_);
    d.fbidx.upload(staged_fb_index_.data(), staged_fb_index_.size(), stream_);
    d.fbval.upload(staged_fb_value_.data(), staged_fb_value_.size(), stream_);
}
void Engine::upload_units(UnitDev &d, const Schedule &sched, const std::vector<DevUnit> &du, bool scheduled_on_device) {
    if (!scheduled_on_device) {
        upload_unit_arrays(d);
        d.order.upload(sched.order.data(), sched.order.size(), stream_);
    }
    d.units.upload(du.data(), du.size(), stream_);
    {   // launch records of the wave-per-user kernel: schedule order, first row entry and user id inline
        std::vector<DevUnitX> xu(sched.order.size());
        for (size_t s = 0; s < sched.order.size(); s++) {
            const DevUnit &u = du[(size_t)sched.order[s]];
            DevUnitX x{u, 0, 0u, 0};
            if (u.row_end > u.row_begin) {
                x.e0 = staged_.row_ptr[3 * (size_t)u.row_begin];
                if ((u.flags & UNIT_SIMPLE) && staged_.row_ptr[3 * (size_t)u.row_begin + 2] > staged_.row_ptr[3 * (size_t)u.row_begin + 1])
                    x.user = staged_.feat_index[(size_t)staged_.row_ptr[3 * (size_t)u.row_begin + 1]];
            }
            xu[s] = x;
        }
        d.xunits.upload(xu.data(), xu.size(), stream_);
    }
    d.unit_values = simple_unit_values_;
    d.has_fresh = any_fresh_;
    if (any_fresh_ && !scheduled_on_device) d.fresh.upload(staged_fresh_.data(), staged_fresh_.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
}

// The same schedule built on the device (svdf_k_sched.hip: device_schedule_units) for resident user-group data sets: the staged arrays go
// up first, levels / order / the fast-path flags / row_fresh are formed in HBM, the host gets back the order and one byte per unit.
// 1 M users x 100 rows: 4.2 s of host scan -> see DESIGN.md 4e.  Falls back (returns false) for what the device form does not cover:
// relaxed ids, side tables, shared latent spaces (their units touch child rows).
bool Engine::schedule_units_on_device(UnitDev &d, Schedule &sched, std::vector<DevUnit> &du) {
    const long nu = (long)staged_units_.size();
    const bool plain = feat_user_.num_row() == 0 && feat_item_.num_row() == 0 && mp_.common_latent_space == 0 && mp_.common_feedback_space == 0;
    if (!device_sched_ || !plain || relaxed() || nu == 0 || staged_.num_row() < device_sched_min_) return false;
    if (staged_.feat_index.size() + staged_fb_index_.size() + (size_t)nu >= 0xFFFFFF00ull) return false;
    static_assert(sizeof(HostUnit) == sizeof(DevUnit), "HostUnit and DevUnit share their layout");
    upload_unit_arrays(d);
    d.units.upload(reinterpret_cast<const DevUnit *>(staged_units_.data()), (size_t)nu, stream_);
    d.order.reserve((size_t)nu);
    d.fresh.reserve((size_t)std::max<long>(staged_.num_row(), 1));
    DevBuf<unsigned char> simple;
    simple.reserve((size_t)nu);
    UnitSchedIn in{};
    in.n = nu; in.nrow = staged_.num_row();
    in.units = d.units.p; in.row_ptr = d.ptr.p; in.index = d.index.p; in.value = d.value.p; in.fb_index = d.fbidx.p;
    in.goff = (unsigned)n_uiset_; in.user_off = user_off_; in.item_off = item_off_; in.fb_off = fb_off_;
    in.num_item = (unsigned)mp_.num_item; in.num_fb = (unsigned)num_fb_rows(); in.state_res = (unsigned)num_resources();
    in.simple_ok = 1;
    in.fast_ok = (use_simple_units_ && !lazy_decay() && mp_.num_factor <= max_fast_path_factor()) ? 1 : 0;
    UnitSchedOut out;
    // the scratch of the peel is ~24 bytes per row entry (7 GB for 100 M rows): when it cannot be had, the host scan builds the same schedule
    try { device_schedule_units(in, d.order.p, simple.p, d.fresh.p, out, stream_); }
    catch (const std::exception &e) {
        (void)hipGetLastError();
        if (getenv("SVDF_SCHED_TRACE")) fprintf(stderr, "[svdfeature_amd] device unit schedule not built (%s): host scan instead\n", e.what());
        return false;
    }
    sched.level_ptr = out.level_ptr;
    sched.max_level_size = out.max_level_size;
    sched.order.resize((size_t)nu);
    std::vector<unsigned char> hs((size_t)nu);
    HIPCHECK(hipMemcpyAsync(sched.order.data(), d.order.p, (size_t)nu * sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipMemcpyAsync(hs.data(), simple.p, (size_t)nu, hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    du.resize((size_t)nu);
    for (long t = 0; t < nu; t++) {
        const HostUnit &u = staged_units_[(size_t)t];
        du[(size_t)t] = DevUnit{u.fb_begin, u.fb_end, u.row_begin, u.row_end, u.flags | (hs[(size_t)t] ? UNIT_SIMPLE : 0)};
    }
    sched.level_mid.resize(sched.num_levels());
    for (size_t l = 0; l < sched.num_levels(); l++) {   // the final sort put a level's fast-path units first
        long m = sched.level_ptr[l];
        while (m < sched.level_ptr[l + 1] && hs[(size_t)sched.order[(size_t)m]]) m++;
        sched.level_mid[l] = m;
    }
    simple_unit_values_ = out.unit_values;
    any_fresh_ = out.any_fresh;
    return true;
}

void Engine::flush_units() {
    const long nu = (long)staged_units_.size();
    if (nu == 0) { staged_.clear(); return; }
    need_device("update");
    const DevParams &P = params();
    const int base = tracker_.base;
    // the last unit always leaves its implicit-feedback registers in the device state slot, the way the
    // reference leaves them in the trainer's members
    staged_units_.back().flags |= UNIT_SAVE;
    Schedule sched;
    std::vector<DevUnit> du;
    schedule_units(base, sched, du);
    tracker_.base = base + (int)sched.num_levels();
    const long n = staged_.num_row();
    UnitDev &d = w_unitdev_;
    upload_units(d, sched, du);
    const DevCSR D = d.csr();
    for (size_t l = 0; l < sched.num_levels(); l++) {
        launch_svdpp_wave(P, D, d.units.p, d.fbidx.p, d.fbval.p, d.order.p, svdpp_xunits_ ? d.xunits.p : nullptr, sched.level_ptr[l], sched.level_mid[l], stream_);
        launch_svdpp(P, D, d.units.p, d.fbidx.p, d.fbval.p, d.order.p, sched.level_mid[l], sched.level_ptr[l + 1], sample_counter_, stream_);
        n_launches_++;
    }
    HIPCHECK(hipGetLastError());
    n_batches_ += (int64_t)sched.num_levels();
    n_instances_ += n;
    sample_counter_ += (unsigned)n;
    n_flushes_++;
    if (unit_open_) unit_open_on_device_ = true;
    staged_.clear();
    staged_units_.clear();
    staged_fb_index_.clear();
    staged_fb_value_.clear();
}
// Column-shaped data sets are scheduled on the GPU: the raw columns go up in file order, svdf_k_sched.hip builds the
// conflict-free levels (same order[] / level_ptr[] as the host scheduler), and the level-sorted copies are gathered in HBM.
void Engine::schedule_columns_on_device(Dataset *ds, long n, int K, const int *res_col, const unsigned *off, const unsigned *limit,
                                        const char *const *msg, int sort_col, unsigned sort_max, const std::vector<UCol> &ucols,
                                        const std::vector<FCol> &fcols) {
    std::vector<std::unique_ptr<DevBuf<unsigned>>> rawu;
    std::vector<std::unique_ptr<DevBuf<float>>> rawf;
    for (const UCol &c : ucols) { rawu.emplace_back(new DevBuf<unsigned>()); rawu.back()->upload(c.src, (size_t)n, stream_); }
    for (const FCol &c : fcols) { rawf.emplace_back(new DevBuf<float>()); rawf.back()->upload(c.src, (size_t)n, stream_); }
    const unsigned *res[SVDF_SCHED_MAX_SLOTS];
    for (int s = 0; s < K; s++) res[s] = rawu[(size_t)res_col[s]]->p;
    std::vector<DUCol> du;
    std::vector<DFCol> df;
    for (size_t c = 0; c < ucols.size(); c++) du.push_back(DUCol{rawu[c]->p, ucols[c].dst});
    for (size_t c = 0; c < fcols.size(); c++) df.push_back(DFCol{rawf[c]->p, fcols[c].dst});
    schedule_device_columns(ds, n, K, res, off, limit, msg, sort_col >= 0 ? rawu[(size_t)sort_col]->p : nullptr, sort_max, du, df);
}
void Engine::schedule_device_columns(Dataset *ds, long n, int K, const unsigned *const *res_col, const unsigned *off, const unsigned *limit,
                                     const char *const *msg, const unsigned *sort_key, unsigned sort_max, const std::vector<DUCol> &ucols,
                                     const std::vector<DFCol> &fcols) {
    SchedColumns in;
    memset(&in, 0, sizeof(in));
    in.K = K; in.n = n;
    unsigned nres = 0;
    for (int s = 0; s < K; s++) {
        in.col[s] = res_col[s]; in.off[s] = off[s]; in.limit[s] = limit[s]; in.limit_msg[s] = msg[s];
        nres = std::max(nres, off[s] + limit[s]);
    }
    in.num_res = nres;
    in.sort_key = sort_key;
    in.sort_key_max = sort_max;
    ds->order_dev.reserve((size_t)std::max<long>(n, 1));
    try {
        device_schedule(in, ds->order_dev.p, ds->sched.level_ptr, &ds->sched.max_level_size, stream_);
    } catch (const std::runtime_error &ex) {
        fail(ex.what());
    }
    ds->sched.order.clear();
    for (const DUCol &c : ucols) { c.dst->reserve((size_t)n); device_gather_u32(c.src, ds->order_dev.p, c.dst->p, n, stream_); }
    for (const DFCol &c : fcols) { c.dst->reserve((size_t)n); device_gather_f32(c.src, ds->order_dev.p, c.dst->p, n, stream_); }
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream_));
}
const int *Engine::host_order(Dataset *ds) {
    if ((long)ds->sched.order.size() != ds->num_row && ds->order_dev.p) {
        ds->sched.order.resize((size_t)ds->num_row);
        if (ds->num_row > 0) {
            HIPCHECK(hipMemcpyAsync(ds->sched.order.data(), ds->order_dev.p, (size_t)ds->num_row * sizeof(int), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
        }
    }
    return ds->sched.order.data();
}
// the two row-pointer widths in use: staged windows (int, the reference's SVDFeatureCSR) and resident data sets (int64_t)
template bool Engine::fused_shape_ok<int>(long, const int *, const unsigned *, FusedHost &);
template bool Engine::fused_shape_ok<int64_t>(long, const int64_t *, const unsigned *, FusedHost &);
template void Engine::fill_fused<int>(long, const float *, const int *, const unsigned *, const float *, const int *, FusedHost &);
template void Engine::fill_fused<int64_t>(long, const float *, const int64_t *, const unsigned *, const float *, const int *, FusedHost &);

}  // namespace svdf
