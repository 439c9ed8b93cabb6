// svdf_dataset.cpp -- part of the host engine (class Engine, svdf_engine.h): HBM-resident data sets (svdf_dataset_from_*), scoring and evaluation over them
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

Dataset *Engine::dataset_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                     const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                     const float *feat_value) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(user_group(), "svdfeature_amd: block datasets are for user-group (format_type 1) trainers");
    if (multi_ && !in_multi_scope()) {
        check(mp_.common_feedback_space == 0, "svdfeature_amd: amd:gpus > 1 needs a feedback space of its own (common_feedback_space = 0)");
        return multi_dataset_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
    }
    flush();
    check(!unit_open_, "dataset_from_blocks: a START block is pending in the trainer");
    if (single_minibatch()) return wseq_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
    if (auto_step_active()) {
        auto_building_ = true;
        struct Done { bool &f; ~Done() { f = false; } } done{auto_building_};
        const bool ok = wunit_config_ok() && wunit_blocks_ok(num_block, extend_tag, fb_ptr, fb_index, block_row_ptr, row_ptr, feat_index);
        auto windows = [&]() { return wseq_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value); };
        // a large pass is judged on a prefix of whole users (like the rating / pair streams above): SVD++ blocks keep ~6 users per level whatever
        // their number, so the first 2 M rows tell -- and 100 M rows are not staged and level-scheduled (2.5 s) only to be dropped for the windows
        const int64_t nrows = num_block > 0 ? block_row_ptr[num_block] - block_row_ptr[0] : 0;
        if (ok && nrows > AUTO_PROBE_MIN) {
            long bp = 0;
            while (bp < num_block && (block_row_ptr[bp] - block_row_ptr[0] < AUTO_PROBE_ROWS || bp == 0 ||
                                      !(extend_tag[bp - 1] == TAG_DEFAULT || extend_tag[bp - 1] == TAG_END))) bp++;
            if (bp > 0 && bp < num_block &&
                auto_probe_deep(dataset_from_blocks(bp, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value), (long)nrows))
                return auto_step(nullptr, true, windows);
        }
        Dataset *exact = dataset_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
        return auto_step(exact, ok, windows);
    }
    if (imfb()) {   // multi-level units: every span of the pass must be closed inside it
        check(imfb_depth_ == 0, "dataset_from_blocks: a START block is pending in the trainer");
        const long saved_window = stage_window_;
        stage_window_ = (long)1 << 60;
        struct Restore { Engine *e; long w; ~Restore() { e->stage_window_ = w; } } restore{this, saved_window};
        std::vector<int> p32;
        try {
            for (long b = 0; b < num_block; b++) {
                const int64_t r0 = block_row_ptr[b], r1 = block_row_ptr[b + 1];
                const int64_t e0 = row_ptr[3 * r0];
                p32.resize((size_t)(3 * (r1 - r0) + 1));
                for (int64_t j = 0; j <= 3 * (r1 - r0); j++) p32[(size_t)j] = (int)(row_ptr[3 * r0 + j] - e0);
                update_block((int)(fb_ptr[b + 1] - fb_ptr[b]), extend_tag[b], fb_index + fb_ptr[b], fb_value + fb_ptr[b], (int)(r1 - r0),
                             row_label + r0, p32.data(), feat_index + e0, feat_value + e0);
            }
            check(imfb_depth_ == 0, "dataset_from_blocks: the last user's END block is missing");
        } catch (...) { drop_staged_units(); imfb_depth_ = 0; iunit_open_ = false; throw; }
        std::unique_ptr<Dataset> ds(new Dataset());
        adopt(ds.get()); ds->kind = 4; ds->num_row = staged_.num_row();
        LevelTracker saved;
        std::swap(saved, tracker_);
        schedule_iunits(0, ds->sched);
        std::swap(saved, tracker_);
        upload_iunits(ds->unitdev, ds->sched);
        const long nb = mp_.no_user_bias ? 1 : 2;
        ds->algorithmic_bytes = ds->num_row * (8L * mp_.num_factor * 2 + 8 * nb + 16 + 16) + (long)staged_fb_index_.size() * (12L * mp_.num_factor + 20);
        ds->num_units = (long)staged_iunits_.size();
        drop_staged_units();
        return ds.release();
    }
    if (rows_without_feedback_ && fb_ptr[num_block] == fb_ptr[0]) {
        // No block carries implicit feedback (the shape of demo/pairwiseRank): tmp_ufeedback and its bias stay +0 and
        // norm_ufeedback is 0 through every update_svdpp (apex_svd_base.h:512-520, 524-527), update_ufeedback returns at
        // once (:539), so update(block) is exactly update_inner(row) for its rows (:557-561) -- the users need not be
        // walked as sequential units and the rows are scheduled one by one like a random-order pass.
        bool open = false;
        for (long b = 0; b < num_block; b++) {
            const int tag = extend_tag[b];
            check(tag == TAG_DEFAULT || tag == TAG_START || tag == TAG_MIDDLE || tag == TAG_END, "dataset_from_blocks: unknown extend_tag");
            open = !(tag == TAG_DEFAULT || tag == TAG_END);
        }
        if (open) fail("dataset_from_blocks: the last user's END block is missing");
        const int64_t r0 = block_row_ptr[0], r1 = block_row_ptr[num_block];
        rows_as_instances_ = true;
        struct Reset { bool &f; ~Reset() { f = false; } } reset{rows_as_instances_};
        return dataset_from_csr((long)(r1 - r0), row_label + r0, row_ptr + 3 * r0, feat_index, feat_value);
    }
    const long saved_window = stage_window_;
    stage_window_ = (long)1 << 60;
    std::vector<int> ptr32;
    if (num_block > 0) {   // the staged copies grow once, not by doubling (100 M rows: 4 GB of host vectors)
        const int64_t nr = block_row_ptr[num_block] - block_row_ptr[0], ne = row_ptr[3 * block_row_ptr[num_block]] - row_ptr[3 * block_row_ptr[0]];
        const int64_t nf = fb_ptr[num_block] - fb_ptr[0];
        if (nr > 0 && nr < 2147483647L && ne >= 0 && ne < 2147483647L && nf >= 0 && nf < 2147483647L) {
            staged_.row_label.reserve((size_t)nr); staged_.row_ptr.reserve((size_t)(3 * nr + 1));
            staged_.feat_index.reserve((size_t)ne); staged_.feat_value.reserve((size_t)ne);
            staged_fb_index_.reserve((size_t)nf); staged_fb_value_.reserve((size_t)nf);
            staged_units_.reserve((size_t)num_block);
        }
    }
    for (long b = 0; b < num_block; b++) {
        const int64_t r0 = block_row_ptr[b], r1 = block_row_ptr[b + 1];
        const int64_t e0 = row_ptr[3 * r0];
        ptr32.resize((size_t)(3 * (r1 - r0) + 1));
        for (int64_t j = 0; j <= 3 * (r1 - r0); j++) ptr32[(size_t)j] = (int)(row_ptr[3 * r0 + j] - e0);
        update_block((int)(fb_ptr[b + 1] - fb_ptr[b]), extend_tag[b], fb_index + fb_ptr[b], fb_value + fb_ptr[b], (int)(r1 - r0),
                     row_label + r0, ptr32.data(), feat_index + e0, feat_value + e0);
    }
    stage_window_ = saved_window;
    auto drop = [&]() { staged_.clear(); staged_units_.clear(); staged_fb_index_.clear(); staged_fb_value_.clear(); unit_open_ = false; unit_open_on_device_ = false; };
    if (unit_open_) { drop(); fail("dataset_from_blocks: the last user's END block is missing"); }
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 3; ds->num_row = staged_.num_row();
    if (!staged_units_.empty()) staged_units_.back().flags |= UNIT_SAVE;
    std::vector<DevUnit> du;
    const auto sched_t0 = std::chrono::steady_clock::now();
    const bool on_device = schedule_units_on_device(ds->unitdev, ds->sched, du);
    if (!on_device) {
        LevelTracker saved;
        std::swap(saved, tracker_);   // a dataset pass is preceded by a flush: schedule against an empty tracker
        schedule_units(0, ds->sched, du);
        std::swap(saved, tracker_);
    }
    unit_sched_us_ = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - sched_t0).count();
    unit_sched_on_device_ = on_device;
    upload_units(ds->unitdev, ds->sched, du, on_device);
    long nfb = (long)staged_fb_index_.size();
    const long nb = mp_.no_user_bias ? 1 : 2;
    ds->algorithmic_bytes = ds->num_row * (8L * mp_.num_factor * 2 + 8 * nb + 16 + 16) + nfb * (12L * mp_.num_factor + 20);
    ds->num_units = (long)du.size();
    for (auto &x : du) ds->num_simple_units += (x.flags & UNIT_SIMPLE) ? 1 : 0;
    drop();
    return ds.release();
}

Dataset *Engine::dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    if (multi_ && !in_multi_scope()) return multi_dataset_from_triples(n, user, item, label);
    if (single_minibatch() && !user_group() && basic_fast_path_allowed()) return wseq_from_triples(n, user, item, label);
    if (auto_step_active()) {
        auto_building_ = true;
        struct Done { bool &f; ~Done() { f = false; } } done{auto_building_};
        const bool wok = wunit_config_ok() && !user_group() && basic_fast_path_allowed();
        if (wok && n > AUTO_PROBE_MIN && auto_probe_deep(dataset_from_triples(AUTO_PROBE_ROWS, user, item, label), n))
            return auto_step(nullptr, true, [&]() { return wseq_from_triples(n, user, item, label); });
        Dataset *exact = dataset_from_triples(n, user, item, label);
        return auto_step(exact, wok, [&]() { return wseq_from_triples(n, user, item, label); });
    }
    if (!basic_fast_path_allowed()) {
        // fall back to the general representation (side tables / shared latent space / user-group trainer)
        std::vector<int64_t> ptr((size_t)3 * n + 1);
        std::vector<unsigned> idx((size_t)2 * n);
        std::vector<float> val((size_t)2 * n, 1.0f);
        for (long r = 0; r < n; r++) {
            ptr[(size_t)3 * r] = 2 * r; ptr[(size_t)3 * r + 1] = 2 * r; ptr[(size_t)3 * r + 2] = 2 * r + 1;
            idx[(size_t)2 * r] = user[r]; idx[(size_t)2 * r + 1] = item[r];
        }
        ptr[(size_t)3 * n] = 2 * n;
        return dataset_from_csr(n, label, ptr.data(), idx.data(), val.data());
    }
    if (Dataset *pv = pivot_dataset_from_triples(n, user, item, label)) return pv;
    if (Dataset *rn = runs_dataset_from_triples(n, user, item, label)) return rn;   // the contract configuration: runs of an item's consecutive ratings (svdf_runs.cpp)   // hot rows: runs of their ratings as walker units (svdf_pivot.cpp)
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->num_row = n; ds->kind = 0;
    const long nb_ = mp_.no_user_bias ? 1 : 2;
    if (device_sched_ && n > 0) {   // bounds are checked by the device pass (same messages)
        const int res_col[2] = {0, 1};
        const unsigned off[2] = {0u, (unsigned)mp_.num_user}, limit[2] = {(unsigned)mp_.num_user, (unsigned)mp_.num_item};
        const char *msg[2] = {"user feature index exceed bound", "item feature index exceed bound"};
        const int sort_col = sort_batches_ == 1 ? 1 : (sort_batches_ == 2 ? 0 : -1);
        schedule_columns_on_device(ds.get(), n, 2, res_col, off, limit, msg, sort_col, sort_col >= 0 ? limit[sort_col] : 0u,
                                   {UCol{user, &ds->user}, UCol{item, &ds->item}}, {FCol{label, &ds->label}});
        ds->unit_values = true;
        ds->algorithmic_bytes = n * (8L * mp_.num_factor * 2 + 8 * nb_ + 16 + 8 * 2);
        return ds.release();
    }
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)mp_.num_user) fail("user feature index exceed bound");
        if (item[r] >= (unsigned)mp_.num_item) fail("item feature index exceed bound");
    }
    // levels relative to an empty tracker: a dataset pass is always preceded by a flush and all launches
    // are stream ordered, so it only has to be conflict-free within itself
    std::vector<int> lastu((size_t)mp_.num_user, 0), lasti((size_t)mp_.num_item, 0), levels((size_t)n);
    for (long r = 0; r < n; r++) {
        const int l = std::max(lastu[user[r]], lasti[item[r]]) + 1;
        lastu[user[r]] = l; lasti[item[r]] = l;
        levels[(size_t)r] = l;
    }
    build_schedule(levels, 0, ds->sched);
    { std::vector<int>().swap(levels); }
    if (sort_batches_ == 1) sort_batches(ds->sched, item);
    else if (sort_batches_ == 2) sort_batches(ds->sched, user);
    std::vector<unsigned> tmp((size_t)n);
    const int *order = ds->sched.order.data();
    parallel_gather(tmp.data(), user, order, n, 1, 0);
    ds->user.upload(tmp.data(), (size_t)n, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    parallel_gather(tmp.data(), item, order, n, 1, 0);
    ds->item.upload(tmp.data(), (size_t)n, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    parallel_gather(reinterpret_cast<float *>(tmp.data()), label, order, n, 1, 0);
    ds->label.upload(reinterpret_cast<float *>(tmp.data()), (size_t)n, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    ds->unit_values = true;
    // 8k*(rows) + 8*(biases) + 16 + 8*nnz per instance
    const long nb = mp_.no_user_bias ? 1 : 2;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 2 + 8 * nb + 16 + 8 * 2);
    return ds.release();
}

// Rank pairs (user, positive item, negative item): the instance PairwiseRankGenerator emits for two rows that carry one
// item entry of value 1 each (apex_svd_data.cpp:828-860 merges the two item lists by index with the negative's sign flipped,
// label 1, :905-911): no global entry, user:1, {min(pos,neg): +-1, max(pos,neg): -+1}.  Few-row fused kernel, 3 rows per pair.
Dataset *Engine::dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    if (multi_ && !in_multi_scope()) return multi_dataset_from_pairs(n, user, pos, neg);
    if (single_minibatch() && !user_group() && basic_fast_path_allowed()) return wseq_from_pairs(n, user, pos, neg);
    if (auto_step_active()) {
        auto_building_ = true;
        struct Done { bool &f; ~Done() { f = false; } } done{auto_building_};
        const bool wok = wunit_config_ok() && !user_group() && basic_fast_path_allowed();
        if (wok && n > AUTO_PROBE_MIN && auto_probe_deep(dataset_from_pairs(AUTO_PROBE_ROWS, user, pos, neg), n))
            return auto_step(nullptr, true, [&]() { return wseq_from_pairs(n, user, pos, neg); });
        Dataset *exact = dataset_from_pairs(n, user, pos, neg);
        return auto_step(exact, wok, [&]() { return wseq_from_pairs(n, user, pos, neg); });
    }
    if (Dataset *pu = punit_dataset_from_pairs(n, user, pos, neg)) return pu;   // a user-grouped stream (the generator's own order): user-run units (svdf_punit.cpp)
    if (device_sched_ && n > 0 && fused_allowed() && !user_group() && !relaxed()) {
        // everything on the device: the three columns go up as they are, the schedule columns (lower / higher item id, signs)
        // are formed there, ids are checked by the scheduling pass, pos == neg by the preparation kernel
        std::unique_ptr<Dataset> ds(new Dataset());
        adopt(ds.get()); ds->num_row = n; ds->kind = 2;
        DevBuf<unsigned> ru, rp, rq, lo_, hi_, flag;
        DevBuf<float> vlo, vhi, one;
        ru.upload(user, (size_t)n, stream_); rp.upload(pos, (size_t)n, stream_); rq.upload(neg, (size_t)n, stream_);
        lo_.reserve((size_t)n); hi_.reserve((size_t)n); vlo.reserve((size_t)n); vhi.reserve((size_t)n); one.reserve((size_t)n); flag.reserve(1);
        HIPCHECK(hipMemsetAsync(flag.p, 0, sizeof(unsigned), stream_));
        launch_pairs_prepare(n, rp.p, rq.p, lo_.p, hi_.p, vlo.p, vhi.p, one.p, flag.p, stream_);
        unsigned bad = 0;
        HIPCHECK(hipMemcpyAsync(&bad, flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
        if (bad) fail("dataset_from_pairs: positive and negative item of a pair must differ");
        rp.release(); rq.release();
        const unsigned *res[3] = {ru.p, lo_.p, hi_.p};
        const unsigned off[3] = {0u, (unsigned)mp_.num_user, (unsigned)mp_.num_user};
        const unsigned limit[3] = {(unsigned)mp_.num_user, (unsigned)mp_.num_item, (unsigned)mp_.num_item};
        const char *msg[3] = {"user feature index exceed bound", "item feature index exceed bound", "item feature index exceed bound"};
        const unsigned *key = sort_batches_ == 1 ? lo_.p : (sort_batches_ == 2 ? ru.p : nullptr);
        FusedDev &f = ds->fused;
        f.max_nu = 1; f.max_ni = 2; f.has_g = false; f.inline_g = false;
        schedule_device_columns(ds.get(), n, 3, res, off, limit, msg, key, sort_batches_ == 1 ? limit[1] : limit[0],
                                {DUCol{ru.p, &f.uidx[0]}, DUCol{lo_.p, &f.iidx[0]}, DUCol{hi_.p, &f.iidx[1]}},
                                {DFCol{one.p, &f.label}, DFCol{one.p, &f.uval[0]}, DFCol{vlo.p, &f.ival[0]}, DFCol{vhi.p, &f.ival[1]}});
        const long nb2 = (mp_.no_user_bias ? 0 : 1) + 2;
        ds->algorithmic_bytes = n * (8L * mp_.num_factor * 3 + 8 * nb2 + 16 + 8 * 3);
        return ds.release();
    }
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)mp_.num_user) fail("user feature index exceed bound");
        if (pos[r] >= (unsigned)mp_.num_item || neg[r] >= (unsigned)mp_.num_item) fail("item feature index exceed bound");
        if (pos[r] == neg[r]) fail("dataset_from_pairs: positive and negative item of a pair must differ");
    }
    if (!fused_allowed() || user_group() || relaxed()) {   // general representation (side tables, lazy decay, wide rows ...)
        std::vector<int64_t> ptr((size_t)3 * n + 1);
        std::vector<unsigned> idx((size_t)3 * n);
        std::vector<float> val((size_t)3 * n), lab((size_t)n, 1.0f);
        for (long r = 0; r < n; r++) {
            ptr[(size_t)3 * r] = 3 * r; ptr[(size_t)3 * r + 1] = 3 * r; ptr[(size_t)3 * r + 2] = 3 * r + 1;
            const bool pf = pos[r] < neg[r];
            idx[(size_t)3 * r] = user[r]; val[(size_t)3 * r] = 1.0f;
            idx[(size_t)3 * r + 1] = pf ? pos[r] : neg[r]; val[(size_t)3 * r + 1] = pf ? 1.0f : -1.0f;
            idx[(size_t)3 * r + 2] = pf ? neg[r] : pos[r]; val[(size_t)3 * r + 2] = pf ? -1.0f : 1.0f;
        }
        ptr[(size_t)3 * n] = 3 * n;
        return dataset_from_csr(n, lab.data(), ptr.data(), idx.data(), val.data());
    }
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->num_row = n; ds->kind = 2;
    {
        std::vector<int> lastu((size_t)mp_.num_user, 0), lasti((size_t)mp_.num_item, 0), levels((size_t)n);
        for (long r = 0; r < n; r++) {
            const int l = std::max(lastu[user[r]], std::max(lasti[pos[r]], lasti[neg[r]])) + 1;
            lastu[user[r]] = l; lasti[pos[r]] = l; lasti[neg[r]] = l;
            levels[(size_t)r] = l;
        }
        build_schedule(levels, 0, ds->sched);
    }
    std::vector<unsigned> lo((size_t)n);
    for (long r = 0; r < n; r++) lo[(size_t)r] = std::min(pos[r], neg[r]);
    if (sort_batches_ == 1) sort_batches(ds->sched, lo.data());
    else if (sort_batches_ == 2) sort_batches(ds->sched, user);
    const int *order = ds->sched.order.data();
    FusedHost fh;
    fh.max_nu = 1; fh.max_ni = 2; fh.has_g = false; fh.inline_g = false;
    fh.label.assign((size_t)n, 1.0f);
    fh.uidx[0].resize((size_t)n); fh.uval[0].assign((size_t)n, 1.0f);
    for (int a = 0; a < 2; a++) { fh.iidx[a].resize((size_t)n); fh.ival[a].resize((size_t)n); }
    for (long s = 0; s < n; s++) {
        const long r = order[s];
        const bool pf = pos[r] < neg[r];
        fh.uidx[0][(size_t)s] = user[r];
        fh.iidx[0][(size_t)s] = pf ? pos[r] : neg[r]; fh.ival[0][(size_t)s] = pf ? 1.0f : -1.0f;
        fh.iidx[1][(size_t)s] = pf ? neg[r] : pos[r]; fh.ival[1][(size_t)s] = pf ? -1.0f : 1.0f;
    }
    ds->fused.upload(fh, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    const long nb = (mp_.no_user_bias ? 0 : 1) + 2;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 3 + 8 * nb + 16 + 8 * 3);   // 3128 B/pair at k=128 without user bias
    return ds.release();
}

// Few-row instances with global features (<= 2 user ids, <= 2 item ids, <= 4 distinct global ids each: the neighbourhood
// / time-bias shape) scheduled on the device like the triples: the host only spreads the rows into columns (one linear
// pass), the level assignment (svdf_k_sched.hip, one resource slot per id) and the gathers into level order run in HBM.
// Returns nullptr when the rows do not fit the shape (the host scheduler takes them).
Dataset *Engine::dataset_fewrow_on_device(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) {
    std::atomic<int> amu(0), ami(0), amg(0), fits(1), has_item(1), has_user(1);
    std::atomic<long> a_nnz(0), a_rows(0), a_bias(0), a_g(0);
    parallel_rows(n, [&](long lo, long hi) {
        int mu = 0, mi = 0, mg = 0;
        bool ok = true, ki = true, ku = true;
        long nnz = 0, rows = 0, nbias = 0, ngt = 0;
        for (long r = lo; r < hi && ok; r++) {
            const int64_t *p = row_ptr + 3 * r;
            const int ng = (int)(p[1] - p[0]), nu = (int)(p[2] - p[1]), ni = (int)(p[3] - p[2]);
            if (nu > 2 || ni > 2 || ng > 4) { ok = false; break; }
            if (nu == 2 && feat_index[p[1]] == feat_index[p[1] + 1]) ok = false;
            if (ni == 2 && feat_index[p[2]] == feat_index[p[2] + 1]) ok = false;
            for (int x = 0; x < ng; x++)
                for (int y = x + 1; y < ng; y++) if (feat_index[p[0] + x] == feat_index[p[0] + y]) ok = false;
            mu = std::max(mu, nu); mi = std::max(mi, ni); mg = std::max(mg, ng);
            ki = ki && ni > 0; ku = ku && nu > 0;
            nnz += ng + nu + ni; ngt += ng; rows += nu + ni; nbias += (mp_.no_user_bias ? 0 : nu) + ni;
        }
        if (!ok) fits = 0;
        if (!ki) has_item = 0;
        if (!ku) has_user = 0;
        int v;
        v = amu.load(); while (mu > v && !amu.compare_exchange_weak(v, mu)) {}
        v = ami.load(); while (mi > v && !ami.compare_exchange_weak(v, mi)) {}
        v = amg.load(); while (mg > v && !amg.compare_exchange_weak(v, mg)) {}
        a_nnz += nnz; a_rows += rows; a_bias += nbias; a_g += ngt;
    });
    int mu = amu.load(), mi = ami.load();
    const int mg = amg.load();
    if (!fits.load() || mg == 0 || mu + mi + mg > SVDF_SCHED_MAX_SLOTS) return nullptr;
    if ((sort_batches_ == 1 && !has_item.load()) || (sort_batches_ == 2 && !has_user.load())) return nullptr;
    mu = std::max(mu, 1); mi = std::max(mi, 1);
    const long nnz = a_nnz.load(), nrows_touched = a_rows.load(), nbias = a_bias.load(), ng_total = a_g.load();
    std::vector<unsigned> cu[2], ci[2], cg[4];
    std::vector<float> vu[2], vi[2], vg[4];
    for (int a = 0; a < mu; a++) { cu[a].resize((size_t)n); vu[a].resize((size_t)n); }
    for (int a = 0; a < mi; a++) { ci[a].resize((size_t)n); vi[a].resize((size_t)n); }
    for (int j = 0; j < 4; j++) { cg[j].resize((size_t)n); vg[j].resize((size_t)n); }
    parallel_rows(n, [&](long lo, long hi) {
        for (long r = lo; r < hi; r++) {
            const int64_t *p = row_ptr + 3 * r;
            const int ng = (int)(p[1] - p[0]), nu = (int)(p[2] - p[1]), ni = (int)(p[3] - p[2]);
            for (int j = 0; j < 4; j++) {
                cg[j][(size_t)r] = j < ng ? feat_index[p[0] + j] : (unsigned)SLOT_ABSENT; vg[j][(size_t)r] = j < ng ? feat_value[p[0] + j] : 0.0f;
            }
            for (int j = 0; j < mu; j++) {
                cu[j][(size_t)r] = j < nu ? feat_index[p[1] + j] : (unsigned)SLOT_ABSENT; vu[j][(size_t)r] = j < nu ? feat_value[p[1] + j] : 0.0f;
            }
            for (int j = 0; j < mi; j++) {
                ci[j][(size_t)r] = j < ni ? feat_index[p[2] + j] : (unsigned)SLOT_ABSENT; vi[j][(size_t)r] = j < ni ? feat_value[p[2] + j] : 0.0f;
            }
        }
    });
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->num_row = n; ds->kind = 2;
    FusedDev &f = ds->fused;
    f.max_nu = mu; f.max_ni = mi; f.has_g = true; f.inline_g = true;
    f.dense_slots = mu == 1 && mi == 1 && has_item.load() && has_user.load();
    const int zero = 0;
    f.gptr.upload(&zero, 1, stream_);   // non-null marks "has global features"; the ids themselves sit in the inline slots
    std::vector<UCol> uc;
    std::vector<FCol> fc;
    int res_col[SVDF_SCHED_MAX_SLOTS];
    unsigned off[SVDF_SCHED_MAX_SLOTS], limit[SVDF_SCHED_MAX_SLOTS];
    const char *msg[SVDF_SCHED_MAX_SLOTS];
    int K = 0, sort_col = -1;
    for (int a = 0; a < mu; a++) {
        if (a == 0 && sort_batches_ == 2) sort_col = (int)uc.size();
        res_col[K] = (int)uc.size(); off[K] = 0u; limit[K] = (unsigned)mp_.num_user; msg[K] = "user feature index exceed bound"; K++;
        uc.push_back(UCol{cu[a].data(), &f.uidx[a]}); fc.push_back(FCol{vu[a].data(), &f.uval[a]});
    }
    for (int a = 0; a < mi; a++) {
        if (a == 0 && sort_batches_ == 1) sort_col = (int)uc.size();
        res_col[K] = (int)uc.size(); off[K] = (unsigned)mp_.num_user; limit[K] = (unsigned)mp_.num_item; msg[K] = "item feature index exceed bound"; K++;
        uc.push_back(UCol{ci[a].data(), &f.iidx[a]}); fc.push_back(FCol{vi[a].data(), &f.ival[a]});
    }
    for (int j = 0; j < 4; j++) {
        if (j < mg) {
            res_col[K] = (int)uc.size(); off[K] = (unsigned)(mp_.num_user + mp_.num_item); limit[K] = (unsigned)mp_.num_global;
            msg[K] = "global feature index exceed bound"; K++;
        }
        uc.push_back(UCol{cg[j].data(), &f.gsi[j]}); fc.push_back(FCol{vg[j].data(), &f.gsv[j]});
    }
    fc.push_back(FCol{row_label, &f.label});
    schedule_columns_on_device(ds.get(), n, K, res_col, off, limit, msg, sort_col,
                               sort_col < 0 ? 0u : (sort_batches_ == 1 ? (unsigned)mp_.num_item : (unsigned)mp_.num_user), uc, fc);
    ds->algorithmic_bytes = 8L * mp_.num_factor * nrows_touched + 8 * nbias + 8 * ng_total + 16 * n + 8 * nnz;
    return ds.release();
}

Dataset *Engine::dataset_from_csr(long num_row, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(!user_group() || rows_as_instances_, "svdfeature_amd: resident datasets are for random-order (format_type 0) trainers");
    if (multi_ && !in_multi_scope()) return multi_dataset_from_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    if (single_minibatch() && !user_group()) return wseq_from_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    if (auto_step_active() && !user_group()) {
        auto_building_ = true;
        struct Done { bool &f; ~Done() { f = false; } } done{auto_building_};
        if (num_row > AUTO_PROBE_MIN) {
            validate_csr_pointers(num_row, row_ptr);
            if (wunit_config_ok() && wunit_rows_ok(0, num_row, row_ptr, feat_index) &&
                auto_probe_deep(dataset_from_csr(AUTO_PROBE_ROWS, row_label, row_ptr, feat_index, feat_value), num_row))
                return auto_step(nullptr, true, [&]() { return wseq_from_csr(num_row, row_label, row_ptr, feat_index, feat_value); });
        }
        Dataset *exact = dataset_from_csr(num_row, row_label, row_ptr, feat_index, feat_value);   // validates the pointers
        const bool ok = wunit_config_ok() && wunit_rows_ok(0, num_row, row_ptr, feat_index);
        return auto_step(exact, ok, [&]() { return wseq_from_csr(num_row, row_label, row_ptr, feat_index, feat_value); });
    }
    const long n = num_row;
    const int64_t p00 = row_ptr[0];
    check(row_ptr[3 * n] - p00 < (int64_t)2147483647, "dataset: more than 2^31-1 feature entries");
    bool basic = basic_fast_path_allowed();
    bool unit = true;
    {   // row checks on several host threads; a failing row is reported by a serial pass (first error in file order)
        std::atomic<int> bad(0), not_basic(0);
        auto check_rows = [&](long lo, long hi, bool &is_basic) {
            for (long r = lo; r < hi; r++) {
                const int64_t *p = row_ptr + 3 * r;
                check(p[0] <= p[1] && p[1] <= p[2] && p[2] <= p[3], "CSR row_ptr must be non-decreasing");
                check_row((int)(p[1] - p[0]), (int)(p[2] - p[1]), (int)(p[3] - p[2]), feat_index + p[0]);
                if (is_basic) is_basic = (p[1] == p[0]) && (p[2] == p[1] + 1) && (p[3] == p[2] + 1);
            }
        };
        const unsigned lim_g = (unsigned)mp_.num_global, lim_u = (unsigned)mp_.num_user, lim_i = (unsigned)mp_.num_item;
        parallel_rows(n, [&](long lo, long hi) {   // the same conditions as a predicate (no message, no exit from a thread)
            bool b = true, ok = true;
            for (long r = lo; r < hi; r++) {
                const int64_t *p = row_ptr + 3 * r;
                ok = ok && p[0] <= p[1] && p[1] <= p[2] && p[2] <= p[3];
                if (!ok) break;
                for (int64_t j = p[0]; j < p[1]; j++) ok = ok && feat_index[j] < lim_g;
                for (int64_t j = p[1]; j < p[2]; j++) ok = ok && feat_index[j] < lim_u;
                for (int64_t j = p[2]; j < p[3]; j++) ok = ok && feat_index[j] < lim_i;
                b = b && (p[1] == p[0]) && (p[2] == p[1] + 1) && (p[3] == p[2] + 1);
            }
            if (!ok) bad = 1;
            if (!b) not_basic = 1;
        });
        if (bad.load()) { bool b = true; check_rows(0, n, b); }
        if (not_basic.load()) basic = false;
    }
    if (basic) {
        for (long r = 0; r < n && unit; r++) unit = feat_value[row_ptr[3 * r]] == 1.0f && feat_value[row_ptr[3 * r] + 1] == 1.0f;
        if (unit) {
            std::vector<unsigned> u((size_t)n), it((size_t)n);
            for (long r = 0; r < n; r++) { u[(size_t)r] = feat_index[row_ptr[3 * r]]; it[(size_t)r] = feat_index[row_ptr[3 * r] + 1]; }
            return dataset_from_triples(n, u.data(), it.data(), row_label);
        }
    }
    if (!basic && device_sched_ && n > 0 && fused_allowed() && !user_group() && !relaxed())
        if (Dataset *d = dataset_fewrow_on_device(n, row_label, row_ptr, feat_index, feat_value)) return d;
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->num_row = n;
    std::vector<int> levels((size_t)n);
    LevelTracker saved;
    std::swap(saved, tracker_);   // schedule against an empty tracker (see dataset_from_triples)
    tracker_.resize(num_resources() + 1);
    long nnz = 0, nrows_touched = 0, nbias = 0, ng_total = 0;
    for (long r = 0; r < n; r++) {
        const int64_t *p = row_ptr + 3 * r;
        const unsigned *ig = feat_index + p[0], *iu = feat_index + p[1], *ii = feat_index + p[2];
        const int ng = (int)(p[1] - p[0]), nu = (int)(p[2] - p[1]), ni = (int)(p[3] - p[2]);
        const int lvl = level_of_row(ig, ng, iu, nu, ii, ni, 0) + 1;
        touch_row(ig, ng, iu, nu, ii, ni, lvl);
        levels[(size_t)r] = lvl;
        long nc_u = 0, nc_i = 0;
        for (int j = 0; j < nu; j++) if (iu[j] < feat_user_.num_row()) nc_u += feat_user_.row_ptr[iu[j] + 1] - feat_user_.row_ptr[iu[j]];
        for (int j = 0; j < ni; j++) if (ii[j] < feat_item_.num_row()) nc_i += feat_item_.row_ptr[ii[j] + 1] - feat_item_.row_ptr[ii[j]];
        nnz += ng + nu + ni; ng_total += ng;
        nrows_touched += nu + ni + nc_u + nc_i;
        nbias += (mp_.no_user_bias ? 0 : nu + nc_u) + ni + nc_i;
    }
    std::swap(saved, tracker_);
    build_schedule(levels, 0, ds->sched);
    ds->algorithmic_bytes = 8L * mp_.num_factor * nrows_touched + 8 * nbias + 8 * ng_total + 16 * n + 8 * nnz;
    if (basic) {   // basic structure with non-unit feature values
        ds->kind = 0; ds->unit_values = false;
        const int *order = ds->sched.order.data();
        std::vector<unsigned> tu((size_t)n), ti((size_t)n);
        std::vector<float> tl((size_t)n), tva((size_t)n), tvb((size_t)n);
        for (long s = 0; s < n; s++) {
            const int64_t p = row_ptr[3 * (long)order[s]];
            tu[(size_t)s] = feat_index[p]; ti[(size_t)s] = feat_index[p + 1];
            tva[(size_t)s] = feat_value[p]; tvb[(size_t)s] = feat_value[p + 1];
            tl[(size_t)s] = row_label[order[s]];
        }
        ds->user.upload(tu.data(), (size_t)n, stream_); ds->item.upload(ti.data(), (size_t)n, stream_);
        ds->label.upload(tl.data(), (size_t)n, stream_);
        ds->uval.upload(tva.data(), (size_t)n, stream_); ds->ival.upload(tvb.data(), (size_t)n, stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
        return ds.release();
    }
    {
        FusedHost fh;
        if (fused_allowed() && fused_shape_ok(n, row_ptr, feat_index, fh)) {
            ds->kind = 2;
            if (sort_batches_ != 0) {   // batch-internal order is free: walk the item (or user) table in id order
                std::vector<unsigned> key((size_t)n, 0u);
                for (long r = 0; r < n; r++) {
                    const int64_t *p = row_ptr + 3 * r;
                    if (relax_user_from_ != 0xFFFFFFFFu) {   // relaxed shared user feature: runs of the same shared id
                        key[(size_t)r] = (p[2] - p[1] == 2 && feat_index[p[1] + 1] >= relax_user_from_) ? feat_index[p[1] + 1] : 0xFFFFFFFFu;
                    } else if (sort_batches_ == 1 && p[3] > p[2]) key[(size_t)r] = feat_index[p[2]];
                    else if (sort_batches_ == 2 && p[2] > p[1]) key[(size_t)r] = feat_index[p[1]];
                }
                sort_batches(ds->sched, key.data());
            }
            fill_fused(n, row_label, row_ptr, feat_index, feat_value, ds->sched.order.data(), fh);
            ds->fused.upload(fh, stream_);
            HIPCHECK(hipStreamSynchronize(stream_));
            return ds.release();
        }
    }
    check(!relaxed(), "svdfeature_amd: relaxed shared ids need few-row instances (at most 2 user and 2 item ids, no side tables)");
    ds->kind = 1;
    std::vector<int> ptr32((size_t)3 * n + 1);
    for (long j = 0; j <= 3 * n; j++) ptr32[(size_t)j] = (int)(row_ptr[j] - p00);
    ds->row_label.upload(row_label, (size_t)n, stream_);
    ds->row_ptr.upload(ptr32.data(), ptr32.size(), stream_);
    ds->feat_index.upload(feat_index + p00, (size_t)ptr32.back(), stream_);
    ds->feat_value.upload(feat_value + p00, (size_t)ptr32.back(), stream_);
    ds->order.upload(ds->sched.order.data(), (size_t)n, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    return ds.release();
}

// `amd:step = auto`: keep the exact level schedule or rebuild the data set as a window sequence.  What is compared (both from the
// schedule the engine has just built, nothing is run):
//   dag_ms    = levels x the latency of ONE unit launched alone -- a pass of exact sequential semantics cannot be faster (bench.py's
//               dag_bound measures the same figure): 4.5 us for an instance (kernel boundary + record -> rows -> dot chain -> stores,
//               DESIGN.md section 5), 5 us + 0.42 us per row for a user unit of the SVD++ kernels (46.6 us at 100 rows);
//   stream_ms = the pass's algorithmic bytes at the rate random 256 / 512-byte row read-modify-writes reach on this
//               chip (0.57 x 8 TB/s, the contract line).
// dag_ms <= 2 x stream_ms: the levels are wide enough to stream, the exact pass stays (bit parity with the reference).  Otherwise the
// data's dependency depth binds and the window step (user side exact, shared rows once per window; |dRMSE| <= 1e-4) is taken.  The
// decision is printed once per data set and kept in counters 16 .. 20.
static void auto_measures(const Dataset *ex, long &levels, double &unit_us, double &dag_ms, double &stream_ms, int pivot_run = 256) {
    levels = (long)ex->sched.num_levels();
    const long units = ex->kind == 3 || ex->kind == 4 ? std::max<long>(ex->num_units, 1) : std::max<long>(ex->num_row, 1);
    unit_us = (ex->kind == 3 || ex->kind == 4) ? 5.0 + 0.42 * (double)ex->num_row / (double)units : (ex->kind == 9 ? 5.0 + 0.3 * pivot_run : (ex->kind == 10 ? 6.0 : (ex->kind == 11 ? 3.0 + 0.7 * (double)ex->num_row / (double)std::max<long>(ex->num_units, 1) : 4.5)));   // kind 9: a level lasts as long as its longest run of a hot row's ratings
    dag_ms = (double)levels * unit_us * 1e-3;
    stream_ms = (double)ex->algorithmic_bytes / (0.57 * 8.0e12) * 1e3;
}
// Deep streams are expensive to level-schedule in full (a user-grouped rank pass of 200 M pairs has 56 M levels: 85 s on the device
// scheduler): the first AUTO_PROBE_ROWS rows are scheduled first, and when THEY are deep by a wide margin (dag bound > 16 x the stream
// model; both grow linearly with the row count on such streams) the window step is chosen without ever building the full schedule.
bool Engine::auto_probe_deep(Dataset *probe, long n_full) {
    std::unique_ptr<Dataset> p(probe);
    long levels; double unit_us, dag_ms, stream_ms;
    auto_measures(p.get(), levels, unit_us, dag_ms, stream_ms);
    if (dag_ms <= 16.0 * stream_ms) return false;
    const double scale = (double)n_full / (double)std::max<long>(p->num_row, 1);
    auto_probe_ = AutoDecision();
    auto_probe_.levels = (long)((double)levels * scale);
    auto_probe_.dag_ms = dag_ms * scale;
    auto_probe_.stream_ms = stream_ms * scale;
    auto_probe_.decided = -1;   // "from a prefix"
    return true;
}
Dataset *Engine::auto_step(Dataset *exact, bool window_ok, const std::function<Dataset *()> &build_window) {
    std::unique_ptr<Dataset> ex(exact);
    AutoDecision D;
    double unit_us = 4.5;
    const bool from_probe = !ex;
    if (from_probe) { D = auto_probe_; D.decided = 0; }
    else auto_measures(ex.get(), D.levels, unit_us, D.dag_ms, D.stream_ms, pivot_run_);
    const bool deep = D.dag_ms > 2.0 * D.stream_ms;
    const char *why;
    if (!deep) { D.decided = 1; why = "exact levels kept (wide enough to stream)"; }
    else if (!window_ok) { D.decided = 3; why = "exact levels kept: the dependency depth binds, but the window step does not cover this configuration / these rows"; }
    else { D.decided = 2; why = "window step chosen: the dependency depth of exact sequential semantics binds"; }
    Dataset *out = ex.get();
    if (D.decided == 2) {
        ex.reset();   // its HBM goes back before the windows are built
        out = build_window();
        D.windows = (long)out->wchild.size();
    } else {
        ex.release();
    }
    auto_last_ = D;
    if (!getenv("SVDF_QUIET"))
        fprintf(stderr, "[svdfeature_amd] amd:step = auto: %ld rows, %s%ld conflict-free levels: dag bound %.2f ms, stream model %.2f ms -> %s%s\n",
                (long)out->num_row, from_probe ? "extrapolated from the first 2 M rows: ~" : "", D.levels, D.dag_ms, D.stream_ms, why,
                D.decided == 2 ? (" (" + std::to_string(D.windows) + " windows)").c_str() : "");
    return out;
}

// The path a resident data set takes through the engine, in one place: the schedule form (from ds->kind) and the kernel family the
// launchers will pick for it under the current configuration (the same predicates the launch sites consult).
std::string Engine::path_for(const Dataset *ds) const {
    const DevParams &P = const_cast<Engine *>(this)->params();
    const long L = (long)ds->sched.num_levels();
    char buf[512];
    auto levels = [&](const char *what) { snprintf(buf, sizeof(buf), "exact, %ld conflict-free levels: %s", L, what); return std::string(buf); };
    switch (ds->kind) {
    case 0: return levels(chain_width_ > 0 ? "contract kernel k_basicmf / k_basicmf_slots, one (user, item) instance per lane group; narrow levels chained (k_basicmf_slots_chain)"
                                           : "contract kernel k_basicmf / k_basicmf_slots, one (user, item) instance per lane group");
    case 1: return levels("general sparse kernel k_general (any number of ids per row, side tables, lazy decay)");
    case 2: {
        const FusedSchedule S = ds->fused.view();
        if (fewrow_gslots_ && fewrow_fast_ && fewrow_gslots_applies(P, S, ds->fused.max_nu, ds->fused.max_ni, ds->fused.dense_slots))
            return levels("few-row kernel with global ids k_fewrow_gslots (neighbourhood shape)");
        return levels(fewrow_fast_ && fewrow_fast_applies(P, S) ? "few-row kernel k_fewrow_slots (<= 2 user + 2 item rows per instance: rank pairs, side ids); narrow levels chained"
                                                                 : "fused few-row kernel k_fused");
    }
    case 3: snprintf(buf, sizeof(buf), "exact, %ld conflict-free levels of user units: k_svdpp_wave (one wave per user, %ld of %ld units) + k_svdpp (general units)",
                     L, ds->num_simple_units, ds->num_units); return buf;
    case 4: return levels("multi-level implicit feedback units k_imfb");
    case 5: return "window data set (one window of the window-minibatch step, user side exact): k_window_users / k_window_items";
    case 6: return "amd:gpus handle: per-rank windows (svdf_multi.cpp)";
    case 7: return "window data set of user units: k_wunit_* (svdf_k_wunit.hip)";
    case 8: snprintf(buf, sizeof(buf), "window sequence (amd:step = minibatch / auto): %zu windows, each trained and applied in place; NOT the reference's sequential semantics (|dRMSE| <= 1e-4 contract)",
                     ds->wchild.size()); return buf;
    case 9: snprintf(buf, sizeof(buf), "exact, %ld levels: hot rows walked as units (k_svdpp_wave on %s parameters, %ld units) + cold ratings through the contract kernel",
                     L, ds->pv_item_pivot ? "transposed" : "plain", ds->num_units); return buf;
    case 10: snprintf(buf, sizeof(buf), "exact, %ld levels of runs: k_basicmf_runs_soa (up to %d consecutive ratings of one item per lane group, the item's row in registers)", L, ds->rn_len); return buf;
    case 11: snprintf(buf, sizeof(buf), "exact, %ld levels of user-run units: k_pair_units (%ld units of up to %d consecutive pairs of one user, the user's row in registers)",
                      L, ds->num_units, pair_unit_cap_); return buf;
    default: return "unknown";
    }
}
void Engine::note_dataset(Dataset *ds) {
    if (!ds || host_only_ || is_peer_) return;
    const bool verbose = getenv("SVDF_VERBOSE") != nullptr, quiet = getenv("SVDF_QUIET") != nullptr;
    if (verbose && !quiet) fprintf(stderr, "[svdfeature_amd] data set of %ld rows -> %s\n", (long)ds->num_row, path_for(ds).c_str());
    // the guard of the DEFAULT step: only level-scheduled (exact) data sets of a one-GPU handle; `amd:step` set = the caller has chosen
    if (step_auto_set_ || step_minibatch_set_ || multi_ || gpus_ != 1) return;
    if (!(ds->kind == 0 || ds->kind == 1 || ds->kind == 2 || ds->kind == 3 || ds->kind == 4 || ds->kind == 9 || ds->kind == 10 || ds->kind == 11) || ds->num_row <= 0) return;
    AutoDecision D;
    double unit_us = 4.5;
    auto_measures(ds, D.levels, unit_us, D.dag_ms, D.stream_ms, pivot_run_);
    guard_last_ = D;
    if (D.dag_ms <= 10.0 * D.stream_ms || D.dag_ms < 50.0) return;   // (passes below 50 ms are not worth a line)
    n_guard_warnings_++;
    if (quiet) return;
    const bool window_ok = wunit_config_ok() && (ds->kind == 3 || ds->kind == 4 || basic_fast_path_allowed());
    fprintf(stderr, "[svdfeature_amd] default (exact) step: %ld rows in %ld conflict-free levels -- the data's dependency depth binds: about %.0f ms per pass "
                    "(%.2f M rows/s; levels x %.1f us) against %.1f ms if the rows streamed.  The exact pass keeps the reference's sequential result bit for bit; "
                    "%s\n",
            (long)ds->num_row, D.levels, D.dag_ms, (double)ds->num_row / D.dag_ms * 1e-3, unit_us, D.stream_ms,
            window_ok ? "`amd:step = auto` would train this data set with the window step (user side exact, shared rows once per window; |dRMSE| <= 1e-4 contract) near the streaming rate"
                      : "the window step of `amd:step = auto` does not cover this configuration");
}

Dataset::~Dataset() {
    for (auto &per_rank : mchild) for (Dataset *c : per_rank) delete c;
    for (Dataset *c : wchild) delete c;
    if (graph_exec) (void)hipGraphExecDestroy(graph_exec);
    if (owner) owner->disown(this);
}
// A dataset outliving its trainer must not reach into it: the trainer forgets its datasets when it goes (their device
// buffers stay valid and are freed by the dataset itself).
void Engine::adopt(Dataset *ds) { ds->owner = this; ds->sched_signature = schedule_signature(); datasets_.push_back(ds); }
// Everything a dataset's conflict schedule and kernel routing were computed under: a dataset built under one setting must
// not be launched under another (e.g. scheduled with relaxed globals, then run with plain read-modify-writes).
uint64_t Engine::schedule_signature() const {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(relax_global_); mix(relax_feedback_); mix(relax_user_from_); mix(relax_item_from_);
    mix(feat_user_.num_row()); mix(feat_user_.index.size()); mix(feat_item_.num_row()); mix(feat_item_.index.size());
    mix(user_group()); mix(lazy_decay()); mix((uint64_t)mp_.num_factor); mix(use_fused_); mix(use_simple_units_);
    mix((uint64_t)mtype_.extend_type);
    return h;
}
void Engine::disown(Dataset *ds) {
    if (window_trained_ == ds) window_trained_ = nullptr;
    for (size_t i = 0; i < datasets_.size(); i++)
        if (datasets_[i] == ds) { datasets_[i] = datasets_.back(); datasets_.pop_back(); break; }
}

// =============================================================================== window-minibatch data sets (N > 1 ranks)
// One exchange window of a rank's shard, grouped by user (svdf_k_window.hip).  svdf_train_dataset on it is
// the first half of the window step (user side exact, item side read-only); window_delta_pack sums the item-side contributions
// into the wire buffer; after the all-reduce window_delta_apply adds the sum on every rank.  Replaces what one instance
void Engine::predict_dataset(Dataset *ds, float *out) {
    check(ds && ds->owner == this, "predict_dataset: dataset belongs to another trainer");
    check(ds->kind != 7 && ds->kind != 8, "predict_dataset: window data sets are training sets (their rows are regrouped by user); score rows with svdf_predict_csr_batch / svdf_predict_block or a level-scheduled data set of the same rows");
    check(ds->kind != 5 && ds->kind != 6, "predict_dataset: window / multi-GPU data sets are training sets (their rows are regrouped: there is no file order to report predictions in); svdf_eval_dataset gives their squared error, svdf_predict_csr_batch scores rows (routed to the owner of each user)");
    check(ds->sched_signature == schedule_signature(),
          "predict_dataset: the dataset was scheduled under another configuration; build it again");
    flush();
    const DevParams &P = params();
    const long n = ds->num_row;
    if (n == 0) return;
    w_out_.reserve((size_t)n);
    if (ds->kind == 4) {
        const UnitDev &d = ds->unitdev;
        launch_imfb(P, d.csr(), d.units.p, d.blks.p, d.fbidx.p, d.fbval.p, nullptr, 0, ds->num_units, sample_counter_, w_out_.p, stream_);
        HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else if (ds->kind == 3) {
        const UnitDev &d = ds->unitdev;
        const DevCSR D = d.csr();
        launch_svdpp_predict(P, D, d.units.p, d.fbidx.p, d.fbval.p, ds->num_units, w_out_.p, stream_);
        HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else if (ds->kind == 11) {   // user-run units of rank pairs: columns in file order
        punit_predict(ds, w_out_.p);
        HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else if (ds->kind == 10) {   // the columns of a runs data set are in file order: no permutation to undo
        BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, nullptr, nullptr};
        launch_predict_basic(P, S, n, w_out_.p, stream_);
        HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else if (ds->kind == 0 || ds->kind == 2 || ds->kind == 9) {   // (kind 9: cold ratings level-sorted, then the units' rows; order_dev = file positions, svdf_pivot.cpp)
        if (ds->kind != 2) {
            BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
            launch_predict_basic(P, S, n, w_out_.p, stream_);
        } else {
            launch_predict_fused(P, ds->fused.view(), ds->fused.max_nu, ds->fused.max_ni, n, w_out_.p, stream_);
        }
        // back into the caller's instance order on the device (a host scatter of 1e8 predictions costs more than the scoring),
        // then one copy out; a host-built schedule's order goes to HBM once
        if (!ds->order_dev.p) { check(ds->sched.order.size() == (size_t)n, "predict_dataset: the data set keeps no file order"); ds->order_dev.upload(ds->sched.order.data(), (size_t)n, stream_); }
        w_pred_.reserve((size_t)n);
        device_scatter_f32(w_out_.p, ds->order_dev.p, w_pred_.p, n, stream_);
        HIPCHECK(hipMemcpyAsync(out, w_pred_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else {
        DevCSR D{ds->row_label.p, ds->row_ptr.p, ds->feat_index.p, ds->feat_value.p};
        launch_predict(P, D, n, w_out_.p, stream_);
        HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    }
    n_launches_++;
}

// RMSEEvaluator (svd_feature_infer.cpp:38-56) over a resident data set without bringing the predictions back: the squared
// errors are summed in fp64 per workgroup on the device (fixed tree), the few hundred partial sums in long double on the host
// like the reference's accumulator.  The reference adds one instance at a time in long double; the tree differs from that by
// rounding only (relative 1e-13 at 1e8 instances), stated in the test.
void Engine::eval_dataset(Dataset *ds, float scale, double *sum_sq, int64_t *count) {
    check(ds && ds->owner == this, "eval_dataset: dataset belongs to another trainer");
    if (ds->kind == 6 && multi_ && !in_multi_scope()) {   // an amd:gpus handle: every (rank, window) piece is scored where it lives
        flush();
        MultiScope local;
        long double acc = 0.0L;
        int64_t cnt = 0;
        for (int d = 0; d < gpus_; d++) {
            Engine *e = rank_engine(d);
            HIPCHECK(hipSetDevice(e->device_));
            for (Dataset *c : ds->mchild[(size_t)d]) {
                double s = 0.0; int64_t m = 0;
                e->eval_dataset(c, scale, &s, &m);
                acc += (long double)s; cnt += m;
            }
        }
        HIPCHECK(hipSetDevice(device_));
        *sum_sq = (double)acc; *count = cnt;
        return;
    }
    check(ds->kind != 6, "eval_dataset: not a data set of this handle");
    check(ds->kind != 7 && ds->kind != 8, "eval_dataset: user-unit window data sets are training sets; evaluate a level-scheduled data set of the same rows");
    check(ds->kind != 5 || ds->fused.max_ni == 1, "eval_dataset: rank-pair window data sets have no label to compare a score with");
    check(ds->sched_signature == schedule_signature(), "eval_dataset: the dataset was scheduled under another configuration; build it again");
    flush();
    const DevParams &P = params();
    const long n = ds->num_row;
    *sum_sq = 0.0; *count = n;
    if (n == 0) return;
    w_out_.reserve((size_t)n);
    const float *labels = nullptr;
    if (ds->kind == 4) {
        const UnitDev &d = ds->unitdev;
        launch_imfb(P, d.csr(), d.units.p, d.blks.p, d.fbidx.p, d.fbval.p, nullptr, 0, ds->num_units, sample_counter_, w_out_.p, stream_);
        labels = d.label.p;
    } else if (ds->kind == 3) {
        const UnitDev &d = ds->unitdev;
        launch_svdpp_predict(P, d.csr(), d.units.p, d.fbidx.p, d.fbval.p, ds->num_units, w_out_.p, stream_);
        labels = d.label.p;
    } else if (ds->kind == 5) {   // a window data set: instances grouped by user; the user column is written out for the scoring kernel
        w_pred_.reserve((size_t)n);
        unsigned *ucol = reinterpret_cast<unsigned *>(w_pred_.p);
        launch_window_user_column(ds->win_urec.p, (int)ds->num_units, ucol, stream_);
        BasicSchedule S{ucol, ds->item.p, ds->label.p, nullptr, nullptr};
        launch_predict_basic(P, S, n, w_out_.p, stream_);
        labels = ds->label.p;
    } else if (ds->kind == 0 || ds->kind == 9 || ds->kind == 10) {   // (kind 9: the columns hold the cold ratings, then the units' rows: svdf_pivot.cpp; kind 10: file order)
        BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
        launch_predict_basic(P, S, n, w_out_.p, stream_);
        labels = ds->label.p;   // same (level) order as the predictions
    } else if (ds->kind == 11) {   // user-run units of rank pairs: every label is 1 (apex_svd_data.cpp:905-911)
        punit_predict(ds, w_out_.p);
        if (!ds->pu_one.p) { ds->pu_one.reserve((size_t)n); launch_runs_fill_u32(reinterpret_cast<unsigned *>(ds->pu_one.p), n, 0x3f800000u, stream_); }
        labels = ds->pu_one.p;
    } else if (ds->kind == 2) {
        launch_predict_fused(P, ds->fused.view(), ds->fused.max_nu, ds->fused.max_ni, n, w_out_.p, stream_);
        labels = ds->fused.label.p;
    } else {
        DevCSR D{ds->row_label.p, ds->row_ptr.p, ds->feat_index.p, ds->feat_value.p};
        launch_predict(P, D, n, w_out_.p, stream_);
        labels = ds->row_label.p;
    }
    const int g = sqerr_partials_grid(n);
    if (d_partials_.cap < (size_t)g) { if (d_partials_.p) (void)hipFree(d_partials_.p); d_partials_.p = nullptr; HIPCHECK(hipMalloc((void **)&d_partials_.p, (size_t)g * sizeof(double))); d_partials_.cap = (size_t)g; }
    launch_sqerr_partials(w_out_.p, labels, n, scale, d_partials_.p, stream_);
    std::vector<double> part((size_t)g);
    HIPCHECK(hipMemcpyAsync(part.data(), d_partials_.p, (size_t)g * sizeof(double), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    long double acc = 0.0L;
    for (double x : part) acc += (long double)x;
    *sum_sq = (double)acc;
    n_launches_ += 2;
}

}  // namespace svdf
