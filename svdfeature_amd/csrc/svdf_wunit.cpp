// svdf_wunit.cpp -- host side of the window-minibatch step for user units (svdf_k_wunit.hip): builds the
// window data sets of user-group (SVD++) blocks and of rows with global features -- units, segments, regrouped rows, contribution
// slots in file order -- and the one-GPU window sequence behind `amd:step = minibatch`.
//
// What the step replaces in the reference: the shared-state part of SVDPPFeature::update (apex_svd_base.h:568-582: prepare_ufeedback
// :523-538, update_ufeedback :539-554) and of update_no_decay / regularize on item rows and global biases (:383-427, :188-210) is
// applied at the window's end instead of instance by instance; the private part (the user's row, bias and feedback state) is exact.
#include <algorithm>
#include <memory>
#include <cmath>
#include <numeric>
#include <string>
#include <thread>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {


namespace {
struct HostSeg {
    unsigned user = 0;
    int64_t fb_begin = 0, fb_count = 0;   // into the caller's feedback arrays
    size_t row_first = 0, row_count = 0;  // into seg_rows (source row ids)
    bool has_user = false;
};
}  // namespace

// what one window's host build produces (wunit_build_host) and wunit_adopt uploads: plain arrays, so that the windows of a sequence can be
// built by several host threads at once (they share nothing but the caller's read-only columns)
struct WUnitHost {
    std::vector<WinUnit> units;
    std::vector<WinSeg> wsegs;
    std::vector<float> w_label, w_uval;
    std::vector<int> rptr, tptr, gptr;
    std::vector<WinEnt> ent, fbent;
    std::vector<WinFbRec> fbrec;   // deferred feedback scatter (empty: contribution rows)
    std::vector<WinTouched> touched;   // one-GPU windows (inplace builds): the targets that keep slots
    bool has_touched = false;
    long nrow = 0, nent = 0, nfbe = 0, item_entries = 0, global_entries = 0;
    int fixed_ng = -2;
    bool unit_uval = true, feedback = false;
};

WUnitSchedule Engine::wunit_view(const Dataset *ds) const {
    WUnitSchedule S;
    S.units = ds->wu_units.p; S.nunits = ds->num_units; S.segs = ds->wu_segs.p;
    S.label = ds->label.p; S.uval = ds->unit_values ? nullptr : ds->uval.p;
    S.rptr = ds->wu_estride > 0 ? nullptr : ds->wu_rptr.p; S.estride = ds->wu_estride;
    S.ent = ds->wu_ent.p; S.fbent = ds->wu_fbent.p;
    S.contrib = d_contrib_.p; S.cbias = d_cbias_.p; S.gcontrib = d_gcontrib_.p;
    S.tptr = ds->wu_tptr.p; S.gptr = ds->wu_gptr.p;
    S.nfb_rows = user_group() ? (long)num_fb_rows() : 0; S.nitem_rows = mp_.num_item; S.nglobal = mp_.num_global;
    S.contrib_bf16 = contrib_bf16_ ? 1 : 0;
    S.fbrec = ds->wu_defer_fb ? ds->wu_fbrec.p : nullptr; S.dvec = d_dvec_.p; S.dbias = d_dbias_.p;
    S.user_bias = mp_.no_user_bias ? 0 : 1;
    S.touched = ds->wu_ntouched >= 0 ? ds->wu_touched.p : nullptr; S.ntouched = std::max<long>(ds->wu_ntouched, 0);
    return S;
}

bool Engine::wunit_config_ok() const {
    return trainer_ready_ && mtype_.extend_type == 0 && !relaxed() && !lazy_decay() && mp_.common_latent_space == 0 && feat_user_.num_row() == 0 &&
           feat_item_.num_row() == 0 && g_stride_ == 1 && mp_.num_factor <= 256 && (!user_group() || mp_.common_feedback_space == 0);
}
void Engine::wunit_check_config(const char *what) const {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    check(mtype_.extend_type == 0, "window data sets: the base solvers only (extend_type 0)");
    check(!relaxed() && !lazy_decay() && mp_.common_latent_space == 0 && feat_user_.num_row() == 0 && feat_item_.num_row() == 0 && g_stride_ == 1,
          "window data sets: no side tables, relaxed ids, lazy decay or shared latent space");
    check(mp_.num_factor <= 256, "window data sets: num_factor <= 256");
    check(!user_group() || mp_.common_feedback_space == 0, "window data sets: user-group trainers need a feedback space of their own (common_feedback_space = 0)");
    (void)what;
}

// Common builder.  segs: in FILE order; seg_rows: source row ids (rows of a segment in file order).  by_row_order: random-order
// windows, where the file order of the contributions is the order of the source rows (segments of different users interleave);
// otherwise (user-group passes) a segment's rows are consecutive in the file and its feedback scatter follows them.
void Engine::wunit_build(Dataset *ds, const void *segs_v, size_t nseg, const std::vector<int64_t> &seg_rows, bool by_row_order, long num_src_row,
                         const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value,
                         const unsigned *fb_index, const float *fb_value) {
    WUnitHost H;
    wunit_build_host(H, wunit_inplace_build_, segs_v, nseg, seg_rows, by_row_order, num_src_row, row_label, row_ptr, feat_index, feat_value, fb_index, fb_value);
    wunit_adopt(ds, H);
}
void Engine::wunit_build_host(WUnitHost &H, bool inplace, const void *segs_v, size_t nseg, const std::vector<int64_t> &seg_rows, bool by_row_order,
                              long num_src_row, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value,
                              const unsigned *fb_index, const float *fb_value) const {
    const HostSeg *segs = static_cast<const HostSeg *>(segs_v);
    const long NU = mp_.num_user, NI = mp_.num_item, NG = mp_.num_global, NF = user_group() ? (long)num_fb_rows() : 0;
    const bool feedback = user_group();
    H.feedback = feedback;
    // ---- units: the segments of one user, in file order; launch order by cost (rows + feedback entries), descending
    std::vector<int> unit_of_user((size_t)NU, -1);
    std::vector<unsigned> unit_user;
    std::vector<long> unit_cost;
    std::vector<int> seg_unit(nseg, -1);
    std::vector<int> unit_nseg;
    for (size_t s = 0; s < nseg; s++) {
        if (!segs[s].has_user) continue;   // a span without rows: its feedback delta is zero (tmp - old = 0), nothing to do
        const unsigned u = segs[s].user;
        if (unit_of_user[u] < 0) { unit_of_user[u] = (int)unit_user.size(); unit_user.push_back(u); unit_cost.push_back(0); unit_nseg.push_back(0); }
        const int un = unit_of_user[u];
        seg_unit[s] = un;
        unit_cost[(size_t)un] += (long)segs[s].row_count + (long)segs[s].fb_count;
        unit_nseg[(size_t)un]++;
    }
    const size_t nunit = unit_user.size();
    std::vector<int> launch(nunit);
    std::iota(launch.begin(), launch.end(), 0);
    std::stable_sort(launch.begin(), launch.end(), [&](int a, int b) { return unit_cost[(size_t)a] > unit_cost[(size_t)b]; });
    std::vector<int> pos_of_unit(nunit);
    for (size_t j = 0; j < nunit; j++) pos_of_unit[(size_t)launch[j]] = (int)j;
    std::vector<WinUnit> &units = H.units;
    units.assign(nunit, WinUnit{});
    {
        long acc = 0;
        for (size_t j = 0; j < nunit; j++) {
            const int un = launch[j];
            units[j] = WinUnit{unit_user[(size_t)un], (int)acc, 0, 0, WinSeg{0, 0, 0, 0}};
            acc += unit_nseg[(size_t)un];
        }
    }
    // segments in launch order of their unit, file order inside a unit; rows regrouped accordingly
    size_t nseg_used = 0;
    for (size_t s = 0; s < nseg; s++) nseg_used += seg_unit[s] >= 0;
    std::vector<WinSeg> &wsegs = H.wsegs;
    wsegs.assign(nseg_used, WinSeg{});
    std::vector<int> seg_new(nseg, -1);
    for (size_t s = 0; s < nseg; s++) {
        if (seg_unit[s] < 0) continue;
        WinUnit &U = units[(size_t)pos_of_unit[(size_t)seg_unit[s]]];
        seg_new[s] = U.seg_begin + U.seg_count;
        U.seg_count++;
        U.rows += (int)segs[s].row_count;
    }
    long nrow = 0, nfbe = 0;
    for (size_t s = 0; s < nseg; s++) if (seg_unit[s] >= 0) { nrow += (long)segs[s].row_count; nfbe += (long)segs[s].fb_count; }
    check(nrow < (1L << 30) && nfbe < (1L << 30), "window data sets: at most 2^30-1 rows / feedback entries per window");
    // row and feedback ranges of the new segments: walk the segments in NEW order
    std::vector<size_t> seg_by_new(nseg_used);
    for (size_t s = 0; s < nseg; s++) if (seg_new[s] >= 0) seg_by_new[(size_t)seg_new[s]] = s;
    std::vector<long> newrow_of_src((size_t)num_src_row, -1);
    {
        long racc = 0, facc = 0;
        for (size_t q = 0; q < nseg_used; q++) {
            const HostSeg &h = segs[seg_by_new[q]];
            wsegs[q] = WinSeg{(int)facc, (int)h.fb_count, (int)racc, (int)h.row_count};
            for (size_t j = 0; j < h.row_count; j++) newrow_of_src[(size_t)seg_rows[h.row_first + j]] = racc + (long)j;
            racc += (long)h.row_count; facc += h.fb_count;
        }
    }
    // ---- regrouped rows: label, user value, entries = [global entries | item entries]
    std::vector<float> &w_label = H.w_label, &w_uval = H.w_uval;
    w_label.assign((size_t)nrow, 0.0f); w_uval.assign((size_t)nrow, 0.0f);
    std::vector<int> &rptr = H.rptr;
    rptr.assign((size_t)2 * nrow + 1, 0);
    bool unit_uval = true;
    long nent = 0;
    int fixed_ng = -2;   // -2 unknown, -1 not fixed
    std::vector<long> src_of_new((size_t)nrow, -1);
    for (long r = 0; r < num_src_row; r++) if (newrow_of_src[(size_t)r] >= 0) src_of_new[(size_t)newrow_of_src[(size_t)r]] = r;
    for (long nr = 0; nr < nrow; nr++) {
        const long r = src_of_new[(size_t)nr];
        const int64_t p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
        const int ng = (int)(p1 - p0), ni = (int)(p3 - p2);
        rptr[(size_t)2 * nr] = (int)nent; rptr[(size_t)2 * nr + 1] = (int)(nent + ng);
        nent += ng + ni;
        check(nent < (1L << 30), "window data sets: at most 2^30-1 feature entries per window");
        if (fixed_ng == -2) fixed_ng = (ni == 1) ? ng : -1;
        else if (fixed_ng >= 0 && !(ni == 1 && ng == fixed_ng)) fixed_ng = -1;
        w_label[(size_t)nr] = row_label[r];
        w_uval[(size_t)nr] = feat_value[p1];
        if (feat_value[p1] != 1.0f) unit_uval = false;
    }
    rptr[(size_t)2 * nrow] = (int)nent;
    std::vector<WinEnt> &ent = H.ent;
    ent.assign((size_t)nent, WinEnt{0u, 0.0f, 0, 0});
    // ---- slots: counts per target, then file-order assignment
    std::vector<int> &tptr = H.tptr, &gptr = H.gptr;
    tptr.assign((size_t)(NF + NI) + 1, 0); gptr.assign((size_t)NG + 1, 0);
    std::vector<unsigned> seen;   // duplicate check inside a row / a list
    for (long nr = 0; nr < nrow; nr++) {
        const long r = src_of_new[(size_t)nr];
        const int64_t p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
        int e = rptr[(size_t)2 * nr];
        seen.clear();
        for (int64_t j = p0; j < p1; j++, e++) {
            if (feat_index[j] >= (unsigned)NG) fail("global feature index exceed setting");
            ent[(size_t)e].idx = feat_index[j]; ent[(size_t)e].val = feat_value[j];
            for (unsigned x : seen) if (x == feat_index[j]) fail("window data sets: a global id listed twice in one row");
            seen.push_back(feat_index[j]);
            gptr[(size_t)feat_index[j] + 1]++;
        }
        seen.clear();
        for (int64_t j = p2; j < p3; j++, e++) {
            if (feat_index[j] >= (unsigned)NI) fail("item feature index exceed bound");
            ent[(size_t)e].idx = feat_index[j]; ent[(size_t)e].val = feat_value[j];
            for (unsigned x : seen) if (x == feat_index[j]) fail("window data sets: an item id listed twice in one row");
            seen.push_back(feat_index[j]);
            tptr[(size_t)(NF + feat_index[j]) + 1]++;
        }
    }
    std::vector<WinEnt> &fbent = H.fbent;
    fbent.assign((size_t)nfbe, WinEnt{0u, 0.0f, 0, 0});
    for (size_t q = 0; q < nseg_used; q++) {
        const HostSeg &h = segs[seg_by_new[q]];
        std::vector<unsigned> ids(fb_index + h.fb_begin, fb_index + h.fb_begin + h.fb_count);
        std::sort(ids.begin(), ids.end());
        for (size_t j = 1; j < ids.size(); j++) if (ids[j] == ids[j - 1]) fail("window data sets: a feedback id listed twice in one block");
        for (int64_t j = 0; j < h.fb_count; j++) {
            const unsigned f = fb_index[h.fb_begin + j];
            if (f >= (unsigned)NF) fail("ufeedback id exceed bound");
            fbent[(size_t)wsegs[q].fb_begin + (size_t)j].idx = f; fbent[(size_t)wsegs[q].fb_begin + (size_t)j].val = fb_value[h.fb_begin + j];
            tptr[(size_t)f + 1]++;
        }
    }
    // one-GPU window sequences: a row that meets exactly ONE contribution in this window gets no slot (slot -1): the unit applies it in place
    // with the sum kernel's operations (apply_single, svdf_device.h) -- nobody else reads or writes that row inside the window
    // deferred feedback scatter: every feedback contribution keeps a slot (a record, not a row) -- the sum kernel forms (w + d val) - w itself
    const bool defer_fb = feedback && wunit_defer_fb_ != 0;
    std::vector<unsigned char> single;
    if (inplace) {
        single.assign((size_t)(NF + NI), 0);
        for (size_t t = defer_fb ? (size_t)NF : 0; t < (size_t)(NF + NI); t++) if (tptr[t + 1] == 1) { single[t] = 1; tptr[t + 1] = 0; }
    }
    auto take_slot = [&](std::vector<int> &cur, size_t t) { return (!single.empty() && single[t]) ? -1 : cur[t]++; };
    for (size_t t = 0; t < (size_t)(NF + NI); t++) tptr[t + 1] += tptr[t];
    for (size_t g = 0; g < (size_t)NG; g++) gptr[g + 1] += gptr[g];
    std::vector<int> tcur(tptr.begin(), tptr.end() - 1), gcur(gptr.begin(), gptr.end() - 1);
    if (defer_fb) H.fbrec.assign((size_t)tptr[(size_t)NF], WinFbRec{0, 0.0f});
    auto row_slots = [&](long nr) {
        const long r = src_of_new[(size_t)nr];
        const int ng = (int)(row_ptr[3 * r + 1] - row_ptr[3 * r]);
        const int e0 = rptr[(size_t)2 * nr], e1 = e0 + ng, e2 = rptr[(size_t)2 * nr + 2];
        for (int e = e0; e < e1; e++) ent[(size_t)e].slot = gcur[ent[(size_t)e].idx]++;
        for (int e = e1; e < e2; e++) ent[(size_t)e].slot = take_slot(tcur, (size_t)NF + ent[(size_t)e].idx);
    };
    if (by_row_order) {
        for (long r = 0; r < num_src_row; r++) if (newrow_of_src[(size_t)r] >= 0) row_slots(newrow_of_src[(size_t)r]);
    } else {
        for (size_t s = 0; s < nseg; s++) {   // file order of the segments: rows, then the feedback scatter of the segment's end
            if (seg_new[s] < 0) continue;
            const WinSeg &w = wsegs[(size_t)seg_new[s]];
            for (int j = 0; j < w.row_count; j++) row_slots((long)w.row_begin + j);
            for (int j = 0; j < w.fb_count; j++) {
                WinEnt &f = fbent[(size_t)w.fb_begin + (size_t)j];
                f.slot = take_slot(tcur, (size_t)f.idx);
                if (defer_fb) H.fbrec[(size_t)f.slot] = WinFbRec{seg_new[s], f.val};
            }
        }
    }
    if (inplace) {   // the in-place sums visit only the targets that have slots (a window touches a fraction of the rows; singles keep none)
        H.has_touched = true;
        for (size_t t = 0; t < (size_t)(NF + NI); t++) if (tptr[t + 1] > tptr[t]) H.touched.push_back(WinTouched{(int)t, tptr[t], tptr[t + 1]});
    }
    for (size_t j = 0; j < nunit; j++) units[j].first = wsegs[(size_t)units[j].seg_begin];   // the first segment travels with the unit record
    H.nrow = nrow; H.nent = nent; H.nfbe = nfbe; H.fixed_ng = fixed_ng; H.unit_uval = unit_uval;
    for (long nr = 0; nr < nrow; nr++) { const int g = rptr[(size_t)2 * nr + 1] - rptr[(size_t)2 * nr]; H.global_entries += g; H.item_entries += rptr[(size_t)2 * nr + 2] - rptr[(size_t)2 * nr] - g; }
    (void)NU;
}
// the uploads and the data set's fields: the calling thread, in window order
void Engine::wunit_adopt(Dataset *ds, const WUnitHost &H) {
    const std::vector<WinUnit> &units = H.units;
    const std::vector<WinSeg> &wsegs = H.wsegs;
    const std::vector<float> &w_label = H.w_label, &w_uval = H.w_uval;
    const std::vector<int> &rptr = H.rptr, &tptr = H.tptr, &gptr = H.gptr;
    const std::vector<WinEnt> &ent = H.ent, &fbent = H.fbent;
    const long nrow = H.nrow, nent = H.nent, nfbe = H.nfbe;
    const size_t nunit = units.size(), nseg_used = wsegs.size();
    const int fixed_ng = H.fixed_ng;
    const bool unit_uval = H.unit_uval;
    if (window_trained_ == ds) window_trained_ = nullptr;
    ds->kind = 7;
    ds->sched_signature = schedule_signature();
    ds->wu_feedback = H.feedback;
    // rptr as the kernel reads it: rptr[2r], rptr[2r + 1], rptr[2r + 2] -- the odd entries are the global / item boundary of row r
    ds->num_row = nrow;
    ds->num_units = (long)nunit;
    ds->win_slots = tptr.back();
    ds->wu_gslots = gptr.back();
    ds->wu_estride = fixed_ng >= 0 ? fixed_ng + 1 : 0;
    ds->unit_values = unit_uval;
    ds->wu_units.upload(units.data(), nunit, stream_);
    ds->wu_segs.upload(wsegs.data(), nseg_used, stream_);
    ds->label.upload(w_label.data(), (size_t)nrow, stream_);
    if (!unit_uval) ds->uval.upload(w_uval.data(), (size_t)nrow, stream_);
    if (ds->wu_estride == 0) ds->wu_rptr.upload(rptr.data(), (size_t)2 * nrow + 1, stream_);
    ds->wu_ent.upload(ent.data(), (size_t)nent, stream_);
    ds->wu_fbent.upload(fbent.data(), (size_t)nfbe, stream_);
    ds->wu_fbrec.upload(H.fbrec.data(), H.fbrec.size(), stream_);
    ds->wu_ntouched = H.has_touched ? (long)H.touched.size() : -1;
    if (H.has_touched) ds->wu_touched.upload(H.touched.data(), H.touched.size(), stream_);
    ds->wu_nseg = (long)nseg_used;
    ds->wu_defer_fb = !H.fbrec.empty();
    ds->wu_tptr.upload(tptr.data(), tptr.size(), stream_);
    ds->wu_gptr.upload(gptr.data(), gptr.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));   // the host columns go out of scope
    ds->sched.level_ptr = {0, nrow};
    ds->sched.max_level_size = nrow;
    // what the reference's step moves -- per row 8k (nu + ni) + 8 (nu_b + ni) + 8 ng + 16 + 8 (ng + nu + ni), per feedback entry 12k + 20
    const long k = mp_.num_factor, nub = mp_.no_user_bias ? 0 : 1;
    const long item_entries = H.item_entries, global_entries = H.global_entries;
    ds->algorithmic_bytes = nrow * (8 * k + 8 * nub + 16 + 8) + item_entries * (8 * k + 8 + 8) + global_entries * 16 + nfbe * (12 * k + 20);
}

// ---- one exchange window of rows of a random-order trainer: any number of global and item entries, exactly one user entry
void Engine::wunit_fill_from_csr(Dataset *ds, long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) {
    WUnitHost H;
    wunit_host_from_csr(H, wunit_inplace_build_, n, row_label, row_ptr, feat_index, feat_value);
    wunit_adopt(ds, H);
}
void Engine::wunit_host_from_csr(WUnitHost &H, bool inplace, long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                 const float *feat_value) const {
    const long NU = mp_.num_user;
    std::vector<int> cnt((size_t)NU, 0);
    for (long r = 0; r < n; r++) {
        check(row_ptr[3 * r + 2] - row_ptr[3 * r + 1] == 1, "window data sets: every row needs exactly one user entry");
        const unsigned u = feat_index[row_ptr[3 * r + 1]];
        if (u >= (unsigned)NU) fail("user feature index exceed bound");
        cnt[u]++;
    }
    // one segment per active user, in order of first occurrence; its rows in file order
    std::vector<int> seg_of_user((size_t)NU, -1);
    std::vector<HostSeg> segs;
    for (long r = 0; r < n; r++) {
        const unsigned u = feat_index[row_ptr[3 * r + 1]];
        if (seg_of_user[u] < 0) { seg_of_user[u] = (int)segs.size(); HostSeg h; h.user = u; h.has_user = true; h.row_count = (size_t)cnt[u]; segs.push_back(h); }
    }
    { size_t acc = 0; for (auto &h : segs) { h.row_first = acc; acc += h.row_count; h.row_count = 0; } }
    std::vector<int64_t> seg_rows((size_t)n);
    for (long r = 0; r < n; r++) {
        HostSeg &h = segs[(size_t)seg_of_user[feat_index[row_ptr[3 * r + 1]]]];
        seg_rows[h.row_first + h.row_count++] = r;
    }
    wunit_build_host(H, inplace, segs.data(), segs.size(), seg_rows, true, n, row_label, row_ptr, feat_index, feat_value, nullptr, nullptr);
}

// ---- one exchange window of a user-group pass: blocks [b0, b1), every START closed by its END inside the window
void Engine::wunit_fill_from_blocks(Dataset *ds, long b0, long b1, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                    const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                    const float *feat_value) {
    WUnitHost H;
    wunit_host_from_blocks(H, wunit_inplace_build_, b0, b1, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
    wunit_adopt(ds, H);
}
void Engine::wunit_host_from_blocks(WUnitHost &H, bool inplace, long b0, long b1, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index,
                                    const float *fb_value, const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr,
                                    const unsigned *feat_index, const float *feat_value) const {
    const long NU = mp_.num_user;
    std::vector<HostSeg> segs;
    std::vector<int64_t> seg_rows;
    bool open = false;
    for (long b = b0; b < b1; b++) {
        const int tag = extend_tag[b];
        check(tag == TAG_DEFAULT || tag == TAG_START || tag == TAG_MIDDLE || tag == TAG_END, "dataset_from_blocks: unknown extend_tag");
        if (tag == TAG_DEFAULT || tag == TAG_START) {
            check(!open, "window data sets: a START block inside an open START..END span");
            HostSeg h;
            h.fb_begin = fb_ptr[b]; h.fb_count = fb_ptr[b + 1] - fb_ptr[b]; h.row_first = seg_rows.size();
            segs.push_back(h);
            open = true;
        } else {
            check(open, "start tag,end tag error in implicit feedback");
        }
        HostSeg &h = segs.back();
        for (int64_t r = block_row_ptr[b]; r < block_row_ptr[b + 1]; r++) {
            check(row_ptr[3 * r + 2] - row_ptr[3 * r + 1] == 1, "window data sets: every row needs exactly one user entry");
            const unsigned u = feat_index[row_ptr[3 * r + 1]];
            if (u >= (unsigned)NU) fail("user feature index exceed bound");
            if (!h.has_user) { h.user = u; h.has_user = true; }
            check(h.user == u, "window data sets: the rows of one block (or START..END span) must belong to one user");
            seg_rows.push_back(r);
            h.row_count++;
        }
        if (tag == TAG_END) {
            const int64_t nf = fb_ptr[b + 1] - fb_ptr[b];
            bool same = nf == h.fb_count;
            for (int64_t j = 0; same && j < nf; j++) same = fb_index[fb_ptr[b] + j] == fb_index[h.fb_begin + j] && fb_value[fb_ptr[b] + j] == fb_value[h.fb_begin + j];
            check(same, "svdfeature_amd: START and END blocks of one user must carry the same feedback list");
        }
        if (tag == TAG_DEFAULT || tag == TAG_END) open = false;
    }
    check(!open, "window data sets: a window must not end inside a START..END span");
    const long r_lo = block_row_ptr[b0], r_hi = block_row_ptr[b1];
    // source row ids relative to the window's first row
    for (auto &r : seg_rows) r -= r_lo;
    std::vector<int64_t> ptr((size_t)3 * (r_hi - r_lo) + 1);
    for (size_t j = 0; j < ptr.size(); j++) ptr[j] = row_ptr[3 * r_lo + (long)j];
    wunit_build_host(H, inplace, segs.data(), segs.size(), seg_rows, false, r_hi - r_lo, row_label + r_lo, ptr.data(), feat_index, feat_value, fb_index, fb_value);
}

Dataset *Engine::dataset_window_from_csr(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) {
    wunit_check_config("dataset_window_from_csr");
    need_device("dataset");
    check(!user_group(), "svdf_dataset_window_from_csr: random-order (format_type 0) trainers; user-group data goes through svdf_dataset_window_from_blocks");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; an amd:gpus handle builds them itself from svdf_dataset_from_csr");
    validate_csr_pointers(n, row_ptr);
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    wunit_fill_from_csr(ds.get(), n, row_label, row_ptr, feat_index, feat_value);
    return ds.release();
}
Dataset *Engine::dataset_window_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                            const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                            const float *feat_value) {
    wunit_check_config("dataset_window_from_blocks");
    need_device("dataset");
    check(user_group(), "svdf_dataset_window_from_blocks: user-group (format_type 1) trainers");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; an amd:gpus handle builds them itself from svdf_dataset_from_blocks");
    validate_block_pointers(num_block, fb_ptr, block_row_ptr);
    validate_csr_pointers((long)(block_row_ptr[num_block] - block_row_ptr[0]), row_ptr + 3 * block_row_ptr[0]);
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    wunit_fill_from_blocks(ds.get(), 0, num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
    return ds.release();
}

// first half of the step on a kind-7 data set (train_dataset): the users' walks; the contributions stay in the trainer's scratch
void Engine::wunit_train(Dataset *ds) {
    d_contrib_.reserve((size_t)std::max<long>(ds->win_slots, 1) * (size_t)pitch_);
    d_cbias_.reserve((size_t)std::max<long>(ds->win_slots, 1));
    d_gcontrib_.reserve((size_t)std::max<long>(ds->wu_gslots, 1));
    if (ds->wu_defer_fb) { d_dvec_.reserve((size_t)std::max<long>(ds->wu_nseg, 1) * (size_t)pitch_); d_dbias_.reserve((size_t)std::max<long>(ds->wu_nseg, 1)); }
    launch_wunit_walk(params(), wunit_view(ds), ds->wu_feedback, wunit_fast_, stream_);
    window_trained_ = ds;
}
// second half: dst == nullptr adds the per-target sums to the model in place, else they go to the wire buffer
void Engine::wunit_sum(Dataset *ds, void *dst, int half) {
    launch_wunit_sum(params(), wunit_view(ds), dst, half, stream_);
}

// =============================================================================== one GPU, `amd:step = minibatch`: a sequence of windows
// OPT-IN and NOT the reference's semantics: the pass is cut into windows; inside a window the shared rows are read as of its start and
// move once, at its end (the N-rank step run by one rank -- the result does not depend on the number of ranks).
// Accuracy contract |dRMSE| <= 1e-4 against the sequential pass, like every N > 1 line; the exact level-scheduled pass stays the default.
// updates_per_target: {item rows, global biases[, feedback rows (instance-sized mass)]}.  Calibrated at the full BASELINE configs[3] sizes
// on three data seeds, 4 and 10 passes (profiles/r04_wstep_calibration.txt): item rows / global biases at 24 per window, feedback rows at 16
// keep |dRMSE| <= 6.2e-5 (SVD++: 24 -> 9.2e-5, 48 -> 1.6e-4; neighbourhood: 64 -> 2.1e-5, 256 -> 1.8e-4 after 10 passes).
long Engine::wseq_windows(long n, const std::vector<double> &updates_per_target) const {
    if (n <= 0) return 1;
    if (window_set_) return std::max<long>(1, (n + stage_window_ - 1) / stage_window_);
    double worst = 0.0;
    for (size_t j = 0; j < updates_per_target.size(); j++)
        worst = std::max(worst, updates_per_target[j] / (double)(j == 2 ? wseq_per_target_fb_ : wseq_per_target_));
    return std::max<long>(1, (long)std::ceil(worst));
}


// How many updates of its own target an entry meets per pass: sum c^2 / sum c (the MEAN over entries, what the calibrations at the uniform
// BASELINE sizes bound at 24 per window) -- and, for skewed data, the MAX: a row that collects far more updates in one window than the
// mean (a Zipf-popular item: 800 where the mean is 24) has all of them computed against its window-start value and overshoots; the pass
// diverges (NaN on Zipf(0.7) items, round 5).  max c is therefore bounded at `per_max` per window, expressed here on the mean's scale.
static double mean_updates_met(const std::vector<long> &cnt, double per_mean_over_per_max) {
    double s1 = 0.0, s2 = 0.0;
    long mx = 0;
    for (long c : cnt) { s1 += (double)c; s2 += (double)c * (double)c; mx = std::max(mx, c); }
    return std::max(s1 > 0.0 ? s2 / s1 : 0.0, (double)mx * per_mean_over_per_max);
}

// The windows of a sequence share nothing but the caller's read-only columns: their host builds run on several threads (a quarter of the
// host's hardware threads, at most 32 and at most `wseq_build_threads`), a batch of windows at a time; uploads stay with the calling thread,
// in window order.  A failure inside a worker is reported by the calling thread (the error text is thread-local).
template <typename BuildFn, typename AdoptFn>
static void wseq_build_windows(long W, int max_threads, BuildFn build, AdoptFn adopt_window) {
    const long hw = (long)std::thread::hardware_concurrency();
    const long T = std::max<long>(1, std::min<long>(std::min<long>(W, max_threads), std::max<long>(1, std::min<long>(32, hw / 4))));
    for (long w0 = 0; w0 < W; w0 += T) {
        const long nb = std::min<long>(T, W - w0);
        std::vector<WUnitHost> H((size_t)nb);
        std::vector<std::string> err((size_t)nb);
        std::vector<char> failed((size_t)nb, 0);
        auto work = [&](long j) {
            try { build(w0 + j, H[(size_t)j]); }
            catch (const std::exception &e) { failed[(size_t)j] = 1; err[(size_t)j] = e.what(); }
        };
        if (nb == 1) work(0);
        else {
            std::vector<std::thread> th;
            for (long j = 1; j < nb; j++) th.emplace_back(work, j);
            work(0);
            for (auto &t : th) t.join();
        }
        for (long j = 0; j < nb; j++) if (failed[(size_t)j]) fail(err[(size_t)j]);
        for (long j = 0; j < nb; j++) { adopt_window(w0 + j, H[(size_t)j]); H[(size_t)j] = WUnitHost(); }
    }
}

void validate_csr_pointers(long num_row, const int64_t *row_ptr) {
    check(num_row >= 0, "dataset: negative row count");
    if (num_row == 0) return;
    check(row_ptr[0] >= 0, "CSR row_ptr must not be negative");
    for (long r = 0; r < num_row; r++) {
        const int64_t *p = &row_ptr[(size_t)3 * r];
        check(p[0] <= p[1] && p[1] <= p[2] && p[2] <= p[3], "CSR row_ptr must be non-decreasing");
    }
    check(row_ptr[(size_t)3 * num_row] - row_ptr[0] < (int64_t)2147483647, "dataset: more than 2^31-1 feature entries");
}
void validate_block_pointers(long num_block, const int64_t *fb_ptr, const int64_t *block_row_ptr) {
    check(num_block >= 0, "dataset_from_blocks: negative block count");
    check(fb_ptr[0] >= 0 && block_row_ptr[0] >= 0, "dataset_from_blocks: block_row_ptr / fb_ptr must not be negative");
    for (long b = 0; b < num_block; b++)
        check(block_row_ptr[b] <= block_row_ptr[b + 1] && fb_ptr[b] <= fb_ptr[b + 1], "dataset_from_blocks: block_row_ptr / fb_ptr must be non-decreasing");
    check(fb_ptr[num_block] - fb_ptr[0] < (int64_t)2147483647 && block_row_ptr[num_block] - block_row_ptr[0] < (int64_t)2147483647,
          "dataset_from_blocks: more than 2^31-1 rows / feedback entries");
}
static bool no_id_twice(const unsigned *a, int64_t n, std::vector<unsigned> &tmp) {
    if (n < 2) return true;
    if (n <= 8) { for (int64_t i = 0; i < n; i++) for (int64_t j = i + 1; j < n; j++) if (a[i] == a[j]) return false; return true; }
    tmp.assign(a, a + n);
    std::sort(tmp.begin(), tmp.end());
    return std::adjacent_find(tmp.begin(), tmp.end()) == tmp.end();
}
bool wunit_rows_ok(long r0, long r1, const int64_t *row_ptr, const unsigned *feat_index) {
    std::vector<unsigned> tmp;
    for (long r = r0; r < r1; r++) {
        const int64_t *p = &row_ptr[(size_t)3 * r];
        if (p[2] != p[1] + 1) return false;
        if (!no_id_twice(feat_index + p[0], p[1] - p[0], tmp) || !no_id_twice(feat_index + p[2], p[3] - p[2], tmp)) return false;
    }
    return true;
}
bool wunit_blocks_ok(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const int64_t *block_row_ptr,
                     const int64_t *row_ptr, const unsigned *feat_index) {
    std::vector<unsigned> tmp;
    bool open = false, have_user = false;
    unsigned user = 0;
    for (long b = 0; b < num_block; b++) {
        const int tag = extend_tag[b];
        if (tag == TAG_DEFAULT || tag == TAG_START) { if (open) return false; have_user = false; }
        else if (tag == TAG_MIDDLE || tag == TAG_END) { if (!open) return false; }
        else return false;
        if (!no_id_twice(fb_index + fb_ptr[b], fb_ptr[b + 1] - fb_ptr[b], tmp)) return false;
        if (!wunit_rows_ok((long)block_row_ptr[b], (long)block_row_ptr[b + 1], row_ptr, feat_index)) return false;
        for (int64_t r = block_row_ptr[b]; r < block_row_ptr[b + 1]; r++) {
            const unsigned u = feat_index[(size_t)row_ptr[(size_t)3 * r + 1]];
            if (have_user && u != user) return false;
            user = u; have_user = true;
        }
        open = (tag == TAG_START || tag == TAG_MIDDLE);
    }
    return !open;
}

Dataset *Engine::wseq_from_csr(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) {
    wunit_check_config("dataset_from_csr");
    validate_csr_pointers(n, row_ptr);
    std::vector<long> ci((size_t)mp_.num_item, 0), cg((size_t)mp_.num_global, 0);
    for (long r = 0; r < n; r++) {
        for (int64_t j = row_ptr[3 * r]; j < row_ptr[3 * r + 1]; j++) { if (feat_index[j] >= (unsigned)mp_.num_global) fail("global feature index exceed setting"); cg[feat_index[j]]++; }
        for (int64_t j = row_ptr[3 * r + 2]; j < row_ptr[3 * r + 3]; j++) { if (feat_index[j] >= (unsigned)mp_.num_item) fail("item feature index exceed bound"); ci[feat_index[j]]++; }
    }
    const long W = wseq_windows(n, {mean_updates_met(ci, wseq_max_ratio()), mean_updates_met(cg, wseq_max_ratio())});
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 8; ds->num_row = n;
    const bool inplace = wunit_inplace_ != 0;   // a window is summed in place right after its walk (wseq_train): single contributions need no slot
    wseq_build_windows(W, wseq_build_threads_,
        [&](long w, WUnitHost &H) {
            const long b0 = n * w / W, b1 = n * (w + 1) / W;
            wunit_host_from_csr(H, inplace, b1 - b0, row_label + b0, row_ptr + 3 * b0, feat_index, feat_value);
        },
        [&](long, const WUnitHost &H) {
            std::unique_ptr<Dataset> c(new Dataset());
            adopt(c.get());
            wunit_adopt(c.get(), H);
            ds->algorithmic_bytes += c->algorithmic_bytes; ds->num_units += c->num_units;
            ds->wchild.push_back(c.release());
        });
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = W > 0 ? (n + W - 1) / W : n;
    return ds.release();
}

Dataset *Engine::wseq_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                  const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                  const float *feat_value) {
    wunit_check_config("dataset_from_blocks");
    validate_block_pointers(num_block, fb_ptr, block_row_ptr);
    validate_csr_pointers((long)(block_row_ptr[num_block] - block_row_ptr[0]), row_ptr + 3 * block_row_ptr[0]);
    const long n = (long)(block_row_ptr[num_block] - block_row_ptr[0]);
    std::vector<long> ci((size_t)mp_.num_item, 0), cg((size_t)mp_.num_global, 0);
    for (long r = block_row_ptr[0]; r < block_row_ptr[num_block]; r++) {
        for (int64_t j = row_ptr[3 * r]; j < row_ptr[3 * r + 1]; j++) { if (feat_index[j] >= (unsigned)mp_.num_global) fail("global feature index exceed setting"); cg[feat_index[j]]++; }
        for (int64_t j = row_ptr[3 * r + 2]; j < row_ptr[3 * r + 3]; j++) { if (feat_index[j] >= (unsigned)mp_.num_item) fail("item feature index exceed bound"); ci[feat_index[j]]++; }
    }
    // a feedback row moves by whole-block steps: a block of n rows pushes about n |value| instance-sized updates into every row of its
    // list at once (update_ufeedback, apex_svd_base.h:539-554) -- the same measure as svdf_multi.cpp's window heuristic
    std::vector<double> mass((size_t)std::max(num_fb_rows(), 1), 0.0);
    {
        long open_rows = 0;
        for (long b = 0; b < num_block; b++) {
            const int tag = extend_tag[b];
            if (tag == TAG_DEFAULT || tag == TAG_START) open_rows = 0;
            open_rows += (long)(block_row_ptr[b + 1] - block_row_ptr[b]);
            if (tag == TAG_DEFAULT || tag == TAG_END)
                for (int64_t j = fb_ptr[b]; j < fb_ptr[b + 1]; j++) {
                    if (fb_index[j] >= (unsigned)num_fb_rows()) fail("ufeedback id exceed bound");
                    mass[fb_index[j]] += (double)open_rows * std::fabs((double)fb_value[j]);
                }
        }
    }
    double m1 = 0.0, m2 = 0.0;
    for (double m : mass) { m1 += m; m2 += m * m; }
    const long W0 = std::min<long>(std::max<long>(num_block, 1), wseq_windows(n, {mean_updates_met(ci, wseq_max_ratio()), mean_updates_met(cg, wseq_max_ratio()), m1 > 0.0 ? m2 / m1 : 0.0}));
    // cuts in blocks (the rule of multi_gpu.block_window_bounds and svdf_multi.cpp): even block positions moved forward to the next
    // position where no START..END span is open
    std::vector<long> cut{0};
    for (long w = 1; w < W0; w++) {
        long pos = std::max<long>(num_block * w / W0, cut.back());
        while (pos < num_block && pos > 0 && (extend_tag[pos - 1] == TAG_START || extend_tag[pos - 1] == TAG_MIDDLE)) pos++;
        cut.push_back(pos);
    }
    cut.push_back(num_block);
    if (num_block > 0 && (extend_tag[num_block - 1] == TAG_START || extend_tag[num_block - 1] == TAG_MIDDLE)) fail("dataset_from_blocks: the last user's END block is missing");
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 8; ds->num_row = n;
    const bool inplace = wunit_inplace_ != 0;
    wseq_build_windows((long)cut.size() - 1, wseq_build_threads_,
        [&](long w, WUnitHost &H) {
            wunit_host_from_blocks(H, inplace, cut[(size_t)w], cut[(size_t)w + 1], extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr,
                                   feat_index, feat_value);
        },
        [&](long, const WUnitHost &H) {
            std::unique_ptr<Dataset> c(new Dataset());
            adopt(c.get());
            wunit_adopt(c.get(), H);
            ds->algorithmic_bytes += c->algorithmic_bytes; ds->num_units += c->num_units;
            ds->wchild.push_back(c.release());
        });
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = n;
    return ds.release();
}

// Ordered sub-steps (svdf_k_window.hip: k_window_apply): plain ratings of the configurations the window kernels walk with unit values and fp32
// contribution rows, on the one-GPU sequence (in-place sums).
bool Engine::wseq_hot_ok() const {
    return wseq_hot_sub_ > 0 && !contrib_bf16_ && !user_group() && basic_fast_path_allowed() && gpus_ == 1 && !multi_ && !is_peer_;
}
// The window count with the hot lane.  What the round-5 rule bounded through ONE number per row -- updates met per window -- are two different things:
//   * how many changes of a row are computed against one value of it (overshoot: diverges on Zipf items): with sub-steps at most hot_sub for every row,
//     so the MEAN over entries  sum_i min(c_i / W, hot_sub) c_i / n  is kept at window_per_target (24, the calibration of the uniform sizes);
//   * how stale the USERS' view of a hot row gets (they read it as of the window start): at most window_hot_max updates (2 048; CPU simulation on a
//     10 M-rating Zipf(0.7) stream, tools/substep_sim.py: 500 per window +1.5e-5, 1 000 +2.8e-5, 2 000 +6.0e-5, 4 000 +9.1e-5 against the 1e-4 contract;
//     on the MI355X at the configs[1] size, 3 data seeds: 1 024 -> max 4.2e-5, 2 048 -> 6.6e-5, 3 072 -> 7.2e-5, profiles/r06_hot_lane_calibration.txt).
long Engine::wseq_windows_hot(long n, const std::vector<long> &item_count) const {
    if (n <= 0) return 1;
    if (window_set_) return std::max<long>(1, (n + stage_window_ - 1) / stage_window_);
    long mx = 0;
    for (long c : item_count) mx = std::max(mx, c);
    auto met = [&](long W) {
        double s = 0.0;
        for (long c : item_count) s += std::min((double)c / (double)W, (double)wseq_hot_sub_) * (double)c;
        return s / (double)n;
    };
    long lo = std::max<long>(1, (mx + wseq_hot_max_ - 1) / wseq_hot_max_);
    if (met(lo) <= (double)wseq_per_target_) return lo;
    long hi = lo;
    while (met(hi) > (double)wseq_per_target_ && hi < n) hi *= 2;
    while (lo + 1 < hi) { const long mid = (lo + hi) / 2; if (met(mid) <= (double)wseq_per_target_) hi = mid; else lo = mid; }
    return hi;
}

Dataset *Engine::wseq_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    std::vector<long> ci((size_t)mp_.num_item, 0);
    for (long r = 0; r < n; r++) { if (item[r] >= (unsigned)mp_.num_item) fail("item feature index exceed bound"); ci[item[r]]++; }
    const bool hot_lane = wseq_hot_ok();
    const long W = hot_lane ? wseq_windows_hot(n, ci) : wseq_windows(n, {mean_updates_met(ci, wseq_max_ratio())});
    // which windows hold a hot item (more than window_hot_sub slots of one item): one scan with per-item stamps
    std::vector<char> whot((size_t)W, 0);
    if (hot_lane) {
        std::vector<int> stamp((size_t)mp_.num_item, -1), cnt((size_t)mp_.num_item, 0);
        for (long w = 0; w < W; w++) {
            const long b0 = n * w / W, b1 = n * (w + 1) / W;
            for (long r = b0; r < b1; r++) {
                const unsigned it = item[r];
                if (stamp[it] != (int)w) { stamp[it] = (int)w; cnt[it] = 0; }
                if (++cnt[it] > wseq_hot_sub_) { whot[(size_t)w] = 1; break; }
            }
        }
    }
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 8; ds->num_row = n;
    // the columns go to HBM in ONE copy each (large pageable copies run at the PCIe rate, 56 GB/s; window-sized ones at a fifth of it), the
    // windows are regrouped from slices of them (svdf_k_wbuild.hip)
    DevBuf<unsigned> d_user, d_item;
    DevBuf<float> d_label;
    const bool resident = device_window_ready() && n > 0;
    if (resident) { need_device("dataset"); d_user.upload(user, (size_t)n, stream_); d_item.upload(item, (size_t)n, stream_); d_label.upload(label, (size_t)n, stream_); }
    for (long w = 0; w < W; w++) {
        const long b0 = n * w / W, b1 = n * (w + 1) / W;
        std::unique_ptr<Dataset> c(new Dataset());
        adopt(c.get());
        if (resident && b1 > b0) { window_build_header(c.get(), b1 - b0, false); window_build_resident(c.get(), b1 - b0, d_user.p + b0, d_item.p + b0, d_label.p + b0, nullptr); }
        else window_build(c.get(), b1 - b0, user + b0, item + b0, label + b0);
        c->win_hot = whot[(size_t)w] != 0;
        ds->algorithmic_bytes += c->algorithmic_bytes; ds->num_units += c->num_units;
        ds->wchild.push_back(c.release());
    }
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = W > 0 ? (n + W - 1) / W : n;
    return ds.release();
}

// rank pairs (BASELINE configs[4]): two signed item entries per instance, two contribution slots per pair.  The window rule counts both
// entries; amd:window (pairs per window) overrides -- the demo-rate calibration allows ~320 updates per item per window
// (profiles/r04_pairs_windows_demo_rate.txt), an order of magnitude more than the rating default kept here.
Dataset *Engine::wseq_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    std::vector<long> ci((size_t)mp_.num_item, 0);
    for (long r = 0; r < n; r++) {
        if (pos[r] >= (unsigned)mp_.num_item || neg[r] >= (unsigned)mp_.num_item) fail("item feature index exceed bound");
        ci[pos[r]]++; ci[neg[r]]++;
    }
    const long W = wseq_windows(n, {mean_updates_met(ci, wseq_max_ratio())});
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 8; ds->num_row = n;
    DevBuf<unsigned> d_user, d_pos, d_neg;
    const bool resident = device_window_ready() && n > 0;
    if (resident) { need_device("dataset"); d_user.upload(user, (size_t)n, stream_); d_pos.upload(pos, (size_t)n, stream_); d_neg.upload(neg, (size_t)n, stream_); }
    for (long w = 0; w < W; w++) {
        const long b0 = n * w / W, b1 = n * (w + 1) / W;
        std::unique_ptr<Dataset> c(new Dataset());
        adopt(c.get());
        if (resident && b1 > b0) { window_build_header(c.get(), b1 - b0, true); window_build_resident(c.get(), b1 - b0, d_user.p + b0, d_pos.p + b0, nullptr, d_neg.p + b0); }
        else window_build(c.get(), b1 - b0, user + b0, pos + b0, nullptr, neg + b0);
        ds->algorithmic_bytes += c->algorithmic_bytes; ds->num_units += c->num_units;
        ds->wchild.push_back(c.release());
    }
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = W > 0 ? (n + W - 1) / W : n;
    return ds.release();
}

// one pass over a window sequence: per window the users' walks, then the per-target sums added in place (two launches per window)
void Engine::wseq_train(Dataset *ds) {
    const DevParams &P = params();
    bool any_hot = false;
    long max_slots = 1;
    for (Dataset *c : ds->wchild) if (c->kind == 5 && c->win_hot) { any_hot = true; max_slots = std::max(max_slots, c->win_slots); }
    const bool hot_lane = any_hot && wseq_hot_ok();
    if (any_hot) check(hot_lane, "train_dataset: the window sequence was built with ordered sub-steps for hot items (window_hot_sub); the configuration changed since");
    if (hot_lane) d_clabel_.reserve((size_t)max_slots);
    for (Dataset *c : ds->wchild) {
        if (c->kind == 5) {
            d_contrib_.reserve((size_t)std::max<long>(c->win_slots, 1) * (size_t)pitch_);
            d_cbias_.reserve((size_t)std::max<long>(c->win_slots, 1));
            WindowSchedule S = window_view(c);
            if (hot_lane && c->win_hot) { S.hot_sub = wseq_hot_sub_; S.clabel = d_clabel_.p; }
            launch_window_users(P, S, window_slots_, window_groups_, stream_);
            if (S.hot_sub > 0) launch_window_apply(P, S, mp_.num_item, dW_.p + (size_t)item_off_ * pitch_, dbias_.p + item_off_, stream_);
            else launch_window_items_local(S, pitch_, mp_.num_factor, 0, mp_.num_item, dW_.p + (size_t)item_off_ * pitch_, dbias_.p + item_off_, stream_, c->win_slots);
        } else {
            wunit_train(c);
            wunit_sum(c, nullptr, 0);
        }
        n_launches_ += 2;
        window_trained_ = nullptr;
    }
}

}  // namespace svdf
