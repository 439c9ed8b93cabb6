// svdf_pivot.cpp -- exact passes over data with HOT rows (knob "pivot_exec"): ratings whose items (or users) are
// Zipf-popular have as many conflict-free levels as the hottest row has ratings -- every update of that row reads what the previous one
// wrote -- and a level costs a kernel boundary plus its rows' round trip however few instances it holds.  Here up to pivot_run (256) consecutive
// ratings of a hot row become ONE unit: a wave keeps the hot ("pivot") row in registers and walks the unit's ratings in file order, the
// partner rows through memory -- the user-unit walker the engine already has for SVD++ users (k_svdpp_wave, svdf_k_wave.hip), fed with
// units that carry no feedback list.  When the hot side is the ITEMS the walker runs on TRANSPOSED parameters (user / item offsets, decays
// and bias decays swapped): update_inner (apex_svd_base.h:456-462) on a (user:1, item:1) instance is symmetric in the two rows -- the
// products of the dot commute, the double bias sum has two terms, each row's step is (row + s * other) * its own decay -- so the walker
// computes bit for bit what the contract kernel computes (tests/test_gpu_pivot.py: equal to the level-by-level pass and to the oracle).
// Units and the remaining ("cold") ratings are levelled TOGETHER on the host in one scan in file order:
//   cold rating:  level = 1 + max(last[user], last[item])
//   hot rating:   joins its row's open unit U when the unit has room and last[partner] < level(U) (the partner's previous toucher runs in
//                 an earlier level, and nobody can slip in between: a later toucher of the partner sees last = level(U)); otherwise the unit
//                 is closed and a new one opens at 1 + max(last[pivot], last[partner]).
// Rows that share a level share no parameter row; rows that share a parameter row keep their file order across levels: the result is the
// sequential one.  A level is two launches (the cold ratings through the contract kernel, the units one wave each).
#include <algorithm>
#include <cstring>
#include <memory>
#include <thread>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {


bool Engine::pivot_config_ok() const {
    return pivot_exec_ != 0 && !host_only_ && !user_group() && basic_fast_path_allowed() && mtype_.extend_type == 0 && tp_.reg_method == 0 &&
           mp_.user_nonnegative == 0 && mp_.item_nonnegative == 0 && mp_.no_user_bias == 0 && u_param_.bound.empty() && i_param_.bound.empty() &&
           mp_.num_factor <= 256;
}

// nullptr: no hot row (or the configuration is outside the symmetric form): the caller builds the plain level schedule
Dataset *Engine::pivot_dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    if (!pivot_config_ok() || n < (long)pivot_min_ || n >= 0x7FFFFFF0L) return nullptr;
    const long NU = mp_.num_user, NI = mp_.num_item;
    // ratings per row, counted on several host threads (this runs for every data set of plain ratings: 100 M ratings in ~0.1 s)
    std::vector<int> cu((size_t)NU, 0), ci((size_t)NI, 0);
    {
        const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        const long T = n < (1L << 22) ? 1 : (long)hw;
        std::vector<std::vector<int>> lu((size_t)T), li((size_t)T);
        std::vector<int> bad((size_t)T, 0);
        auto work = [&](long t) {
            std::vector<int> &a = lu[(size_t)t], &b = li[(size_t)t];
            a.assign((size_t)NU, 0); b.assign((size_t)NI, 0);
            const long lo = n * t / T, hi = n * (t + 1) / T;
            for (long r = lo; r < hi; r++) {
                if (user[r] >= (unsigned)NU) { bad[(size_t)t] |= 1; continue; }
                if (item[r] >= (unsigned)NI) { bad[(size_t)t] |= 2; continue; }
                a[user[r]]++; b[item[r]]++;
            }
        };
        std::vector<std::thread> th;
        for (long t = 1; t < T; t++) th.emplace_back(work, t);
        work(0);
        for (auto &x : th) x.join();
        for (long t = 0; t < T; t++) {
            if (bad[(size_t)t] & 1) fail("user feature index exceed bound");
            if (bad[(size_t)t] & 2) fail("item feature index exceed bound");
            for (long r = 0; r < NU; r++) cu[(size_t)r] += lu[(size_t)t][(size_t)r];
            for (long r = 0; r < NI; r++) ci[(size_t)r] += li[(size_t)t][(size_t)r];
        }
    }
    const int mxu = NU ? *std::max_element(cu.begin(), cu.end()) : 0, mxi = NI ? *std::max_element(ci.begin(), ci.end()) : 0;
    if (std::max(mxu, mxi) < pivot_min_) return nullptr;
    // hot means hot AMONG its kind: a row with many times the ratings of an average (touched) row.  A large uniform data set, where every
    // item has thousands of ratings, has no hot row -- its schedule is wide, and short runs (svdf_runs.cpp) are what pays there
    {
        long tu = 0, ti = 0;
        for (int c : cu) tu += c > 0;
        for (int c : ci) ti += c > 0;
        const double mean_u = (double)n / (double)std::max<long>(tu, 1), mean_i = (double)n / (double)std::max<long>(ti, 1);
        const bool skew = (double)mxu >= 16.0 * mean_u || (double)mxi >= 16.0 * mean_i;   // ... or one that holds a sizeable share of ALL ratings (a small catalogue)
        if (!skew && (long)std::max(mxu, mxi) * 64 < n) return nullptr;
    }
    if (getenv("SVDF_PIVOT_TRACE")) fprintf(stderr, "[pivot] max ratings per user %d, per item %d\n", mxu, mxi);
    const bool item_pivot = mxi >= mxu;   // (the side whose hottest row is hotter)
    const std::vector<int> &cp = item_pivot ? ci : cu;
    const unsigned *pcol = item_pivot ? item : user, *qcol = item_pivot ? user : item;
    const long NP = item_pivot ? NI : NU;
    // ---- one scan in file order: levels of the cold ratings, units of the hot ones
    // ratings per unit at most: a unit is one level, and a level lasts as long as its longest unit.  pivot_run (256) while cold ratings share the
    // levels, pivot_run_long beyond the horizon no cold rating can reach (a cold row has fewer than pivot_min ratings, and levels through such rows
    // stay below ~1.7 x their largest count).  Measured on Zipf(0.7) items at the configs[1] size (profiles/r05_zipf_pivot.txt): 64 / 64 398 ms per
    // pass, 128 / 128 348, 192 / 192 332, 256 / 256 327 (pivot_min 2048), 128 / 1024 422: a longer cap in the tail is SLOWER, both default to 256.
    const int cold_horizon = 2 * pivot_min_;
    struct Open { int level = 0, len = 0, unit = -1, cap = 0; };
    std::vector<int> hot_slot((size_t)NP, -1);
    int nhot = 0;
    for (long r = 0; r < NP; r++) if (cp[(size_t)r] >= pivot_min_) hot_slot[(size_t)r] = nhot++;
    std::vector<Open> open((size_t)nhot);
    std::vector<int> lastp((size_t)NP, 0), lastq((size_t)(item_pivot ? NU : NI), 0);
    std::vector<int> plainp((size_t)NP, 0), plainq((size_t)(item_pivot ? NU : NI), 0);   // the plain level schedule of the same stream, for the comparison below
    std::vector<int> lvl((size_t)n), unit_of((size_t)n, -1);
    std::vector<int> unit_level, unit_len;
    int max_level = 0, plain_levels = 0;
    for (long t = 0; t < n; t++) {
        const unsigned pv = pcol[t], qv = qcol[t];
        { const int pl = 1 + std::max(plainp[pv], plainq[qv]); plainp[pv] = pl; plainq[qv] = pl; plain_levels = std::max(plain_levels, pl); }
        const int hs = hot_slot[pv];
        if (hs < 0) {
            const int l = 1 + std::max(lastp[pv], lastq[qv]);
            lastp[pv] = l; lastq[qv] = l; lvl[(size_t)t] = l;
            max_level = std::max(max_level, l);
            continue;
        }
        Open &U = open[(size_t)hs];
        if (U.unit >= 0 && U.len < U.cap && lastq[qv] < U.level) {
            lastq[qv] = U.level;
            unit_of[(size_t)t] = U.unit; unit_len[(size_t)U.unit]++; U.len++;
            lvl[(size_t)t] = U.level;
            continue;
        }
        const int l = 1 + std::max(lastp[pv], lastq[qv]);
        lastp[pv] = l; lastq[qv] = l;
        U.level = l; U.len = 1; U.unit = (int)unit_level.size();
        U.cap = l > cold_horizon ? pivot_run_long_ : pivot_run_;
        unit_level.push_back(l); unit_len.push_back(1);
        unit_of[(size_t)t] = U.unit; lvl[(size_t)t] = l;
        max_level = std::max(max_level, l);
    }
    const long nunit = (long)unit_level.size();
    if (getenv("SVDF_PIVOT_TRACE")) fprintf(stderr, "[pivot] %ld units (%.2f ratings each on average), %d levels against %d plain levels\n", nunit, nunit ? (double)(n) / (double)nunit : 0.0, max_level, plain_levels);
    {   // Units pay only where a hot row's chain is what makes the schedule deep.  A level of this data set lasts as long as its longest unit
        // (~5 us + 0.3 us per rating), a plain level ~3 us: when hot and clustered rows interleave (ratings sorted by user over Zipf items: the
        // partner rows are busy, units are cut after a few ratings) the unit form is SLOWER than plain levels -- measured 33 s against ~15 s per
        // pass of 20 M such ratings -- and the plain schedule is built instead.
        std::vector<int> longest((size_t)max_level + 1, 0);
        for (long j = 0; j < nunit; j++) longest[(size_t)unit_level[(size_t)j]] = std::max(longest[(size_t)unit_level[(size_t)j]], unit_len[(size_t)j]);
        double t_units = 0.0;
        for (int l = 1; l <= max_level; l++) t_units += longest[(size_t)l] > 0 ? 5.0 + 0.3 * longest[(size_t)l] : 3.0;
        const double t_plain = 3.0 * (double)plain_levels;
        if (!(t_units < 0.6 * t_plain)) return nullptr;
    }
    // ---- cold ratings level-sorted (stable), units level-sorted (stable), a unit's rows contiguous in file order
    const int L = max_level;
    std::vector<long> cptr((size_t)L + 2, 0), uptr((size_t)L + 2, 0);
    long ncold = 0;
    for (long t = 0; t < n; t++) if (unit_of[(size_t)t] < 0) { cptr[(size_t)lvl[(size_t)t] + 1]++; ncold++; }
    for (long j = 0; j < nunit; j++) uptr[(size_t)unit_level[(size_t)j] + 1]++;
    for (int l = 1; l <= L + 1; l++) { cptr[(size_t)l] += cptr[(size_t)l - 1]; uptr[(size_t)l] += uptr[(size_t)l - 1]; }
    // (the three columns hold ALL n ratings -- the cold ones level-sorted in front, the units' rows behind them in launch order -- so that the
    // evaluator scores the data set like a plain one; the kernels of a pass read the front part and the units' own CSR)
    std::vector<unsigned> c_user((size_t)n), c_item((size_t)n);
    std::vector<float> c_label((size_t)n);
    std::vector<int> c_pos((size_t)n);   // file position of every column slot: predict_dataset reports in file order (scatter, as kinds 0 / 2 do)
    {
        std::vector<long> cur(cptr.begin(), cptr.end());
        for (long t = 0; t < n; t++) if (unit_of[(size_t)t] < 0) {
            const long s = cur[(size_t)lvl[(size_t)t]]++;
            c_user[(size_t)s] = user[t]; c_item[(size_t)s] = item[t]; c_label[(size_t)s] = label[t]; c_pos[(size_t)s] = (int)t;
        }
    }
    std::vector<long> unit_pos((size_t)std::max<long>(nunit, 1));   // launch position of unit j
    {
        std::vector<long> cur(uptr.begin(), uptr.end());
        for (long j = 0; j < nunit; j++) unit_pos[(size_t)j] = cur[(size_t)unit_level[(size_t)j]]++;
    }
    const long nhotrows = n - ncold;
    std::vector<long> row_begin((size_t)nunit + 1, 0);   // by launch position
    for (long j = 0; j < nunit; j++) row_begin[(size_t)unit_pos[(size_t)j] + 1] = unit_len[(size_t)j];
    for (long s = 0; s < nunit; s++) row_begin[(size_t)s + 1] += row_begin[(size_t)s];
    std::vector<float> h_label((size_t)std::max<long>(nhotrows, 1));
    std::vector<unsigned> h_index((size_t)std::max<long>(2 * nhotrows, 2));
    std::vector<int> h_ptr((size_t)3 * nhotrows + 1);
    std::vector<DevUnitX> xu((size_t)std::max<long>(nunit, 1));
    {
        std::vector<long> fill((size_t)std::max<long>(nunit, 1), 0);
        for (long t = 0; t < n; t++) {
            const int j = unit_of[(size_t)t];
            if (j < 0) continue;
            const long s = unit_pos[(size_t)j];
            const long r = row_begin[(size_t)s] + fill[(size_t)j]++;
            h_label[(size_t)r] = label[t];
            c_user[(size_t)(ncold + r)] = user[t]; c_item[(size_t)(ncold + r)] = item[t]; c_label[(size_t)(ncold + r)] = label[t]; c_pos[(size_t)(ncold + r)] = (int)t;
            h_index[(size_t)2 * r] = pcol[t];       // the walker's "user" entry: the pivot row
            h_index[(size_t)2 * r + 1] = qcol[t];   // its "item" entry: the partner row
            xu[(size_t)s].user = pcol[t];
        }
        for (long r = 0; r < nhotrows; r++) { h_ptr[(size_t)3 * r] = (int)(2 * r); h_ptr[(size_t)3 * r + 1] = (int)(2 * r); h_ptr[(size_t)3 * r + 2] = (int)(2 * r + 1); }
        h_ptr[(size_t)3 * nhotrows] = (int)(2 * nhotrows);
        for (long s = 0; s < nunit; s++) {
            DevUnit u{0, 0, (int)row_begin[(size_t)s], (int)row_begin[(size_t)s + 1], UNIT_START | UNIT_END | UNIT_SIMPLE};
            xu[(size_t)s].u = u; xu[(size_t)s].e0 = 2 * (int)row_begin[(size_t)s]; xu[(size_t)s].pad = 0;
        }
    }
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    ds->kind = 9; ds->num_row = n; ds->num_units = nunit;
    ds->pv_item_pivot = item_pivot; ds->pv_cold = ncold; ds->pv_hot_rows = nhot;
    ds->user.upload(c_user.data(), c_user.size(), stream_);
    ds->item.upload(c_item.data(), c_item.size(), stream_);
    ds->label.upload(c_label.data(), c_label.size(), stream_);
    ds->order_dev.upload(c_pos.data(), c_pos.size(), stream_);
    ds->unit_values = true;
    UnitDev &d = ds->unitdev;
    d.label.upload(h_label.data(), h_label.size(), stream_);
    d.index.upload(h_index.data(), h_index.size(), stream_);
    d.ptr.upload(h_ptr.data(), h_ptr.size(), stream_);
    {   // feature values (all 1): read by the walker's forms for widths that are not a multiple of 64
        std::vector<float> ones(h_index.size(), 1.0f);
        d.value.upload(ones.data(), ones.size(), stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
    }
    d.xunits.upload(xu.data(), xu.size(), stream_);
    d.unit_values = true; d.has_fresh = false;
    // level l (1-based) -> sched.level_ptr[l - 1 .. l] for the cold ratings, pv_unit_ptr likewise for the units
    ds->sched.level_ptr.assign(cptr.begin() + 1, cptr.end());
    ds->pv_unit_ptr.assign(uptr.begin() + 1, uptr.end());
    ds->sched.max_level_size = 0;
    for (int l = 0; l < L; l++) ds->sched.max_level_size = std::max(ds->sched.max_level_size, ds->sched.level_ptr[(size_t)l + 1] - ds->sched.level_ptr[(size_t)l]);
    const long nb = 2;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 2 + 8 * nb + 16 + 8 * 2);
    HIPCHECK(hipStreamSynchronize(stream_));
    return ds.release();
}

void Engine::pivot_train(Dataset *ds) {
    const DevParams &P = params();
    DevParams Pt = P;             // what the walker sees: its "user" is the pivot row
    Pt.svdpp_helpers = 1;         // no feedback lists: nothing for helper waves to do
    if (ds->pv_item_pivot) {
        std::swap(Pt.user_off, Pt.item_off);
        std::swap(Pt.num_user, Pt.num_item);
        std::swap(Pt.wd_user, Pt.wd_item);
        std::swap(Pt.wd_user_bias, Pt.wd_item_bias);
        std::swap(Pt.u_rng, Pt.i_rng);
    }
    const UnitDev &d = ds->unitdev;
    const DevCSR D{d.label.p, d.ptr.p, d.index.p, d.value.p, 1, nullptr};
    BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, nullptr, nullptr};
    const std::vector<long> &cp = ds->sched.level_ptr, &up = ds->pv_unit_ptr;
    const size_t L = cp.size() - 1;
    int64_t launches = 0;
    for (size_t l = 0; l < L; l++) {
        if (cp[l + 1] > cp[l]) { launch_basicmf(P, S, cp[l], cp[l + 1], groups_per_wave_, block_threads_, stream_); launches++; }
        if (up[l + 1] > up[l]) { launch_svdpp_wave(Pt, D, nullptr, nullptr, nullptr, nullptr, d.xunits.p, up[l], up[l + 1], stream_); launches++; }
    }
    HIPCHECK(hipGetLastError());
    n_launches_ += launches;
    n_batches_ += (int64_t)L;
    n_pivot_passes_++;
}

}  // namespace svdf
