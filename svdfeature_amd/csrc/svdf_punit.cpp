// svdf_punit.cpp -- user-run units of rank pairs (svdf_k_wave.hip: k_pair_units; knob "pair_units"; round 6).
// PairwiseRankGenerator (apex_svd_data.cpp:946-965) emits a user's pairs back to back: 200 consecutive instances that all read and write the user's row.
// Level by level that order is one kernel boundary per pair of a user (56 M levels per 200 M pairs, 3.5 pairs each): 1.6 M pairs/s, 0.45 x the CPU path.
// Here up to `pair_unit_cap` CONSECUTIVE pairs of one user whose item ids are pairwise distinct become one unit: a wave keeps the user's row in registers
// and walks them in file order (update_inner, apex_svd_base.h:456-462, pair after pair: same bits as the level-by-level pass and the oracle,
// tests/test_gpu_punit.py).  Units are levelled like instances: level(U) = 1 + max(last[user], last[every item of U]); units of a level share no row, rows
// shared across levels keep their file order.  What this can reach is bounded by the DATA: the pair-level conflict DAG of that order has a critical path of
// 0.28 n pairs, 0.43 us per dependent step in registers against 0.27 us per pair on a CPU core -- <= 2.2 x the CPU path for any
// exact executor; units of 16 reach ~1.2 x.  Chosen only for streams that ARE user-grouped (more than half of the pairs follow a pair of the same user).
#include <algorithm>
#include <cstring>
#include <memory>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {

bool Engine::punit_config_ok() const {
    return pair_units_ != 0 && !host_only_ && fused_allowed() && !user_group() && !relaxed() && mtype_.extend_type == 0 && tp_.reg_method <= 3 &&
           mp_.num_factor <= 256;
}

// The unit schedule of n pairs given as (user, lower item, higher item) in file order: units in launch order + level boundaries over units.
// false: not a user-grouped stream (fewer than half of the pairs follow a pair of the same user).  Ids must have been checked.
bool Engine::punit_build(long n, const unsigned *user, const unsigned *lo, const unsigned *hi, std::vector<PairUnit> &sorted, std::vector<long> &level_ptr) const {
    long same = 0;
    for (long t = 1; t < n; t++) same += user[t] == user[t - 1];
    if (2 * same < n) return false;
    const long NU = mp_.num_user, NI = mp_.num_item;
    // ---- one scan in file order: units (consecutive pairs of one user, distinct items, at most cap) and their levels
    const int cap = std::max(1, pair_unit_cap_);
    std::vector<int> lastu((size_t)NU, 0), lasti((size_t)NI, 0);
    std::vector<long> in_unit((size_t)NI, -1);        // the unit (by index) an item already belongs to
    std::vector<PairUnit> units;
    std::vector<int> ulevel;
    units.reserve((size_t)(n / cap + 16));
    int max_level = 0;
    long s = 0;
    while (s < n) {
        const unsigned uu = user[s];
        const long uid = (long)units.size();
        long e = s;
        int lmax = lastu[uu];
        while (e < n && e - s < cap && user[e] == uu && in_unit[lo[e]] != uid && in_unit[hi[e]] != uid) {
            in_unit[lo[e]] = uid; in_unit[hi[e]] = uid;
            lmax = std::max(lmax, std::max(lasti[lo[e]], lasti[hi[e]]));
            e++;
        }
        const int l = lmax + 1;
        lastu[uu] = l;
        for (long t = s; t < e; t++) { lasti[lo[t]] = l; lasti[hi[t]] = l; }
        units.push_back(PairUnit{uu, (int)s, (int)(e - s), 0});
        ulevel.push_back(l);
        max_level = std::max(max_level, l);
        s = e;
    }
    const long nunit = (long)units.size();
    // ---- units level-sorted (stable)
    std::vector<long> lptr((size_t)max_level + 2, 0);
    for (long j = 0; j < nunit; j++) lptr[(size_t)ulevel[(size_t)j] + 1]++;
    for (int l = 1; l <= max_level + 1; l++) lptr[(size_t)l] += lptr[(size_t)l - 1];
    sorted.resize((size_t)nunit);
    {
        std::vector<long> cur(lptr.begin(), lptr.end());
        for (long j = 0; j < nunit; j++) sorted[(size_t)cur[(size_t)ulevel[(size_t)j]]++] = units[(size_t)j];
    }
    level_ptr.assign(lptr.begin() + 1, lptr.end());
    return true;
}

// nullptr: not a user-grouped stream (or a configuration outside the walker): the caller builds the plain level schedule
Dataset *Engine::punit_dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    if (!punit_config_ok() || n < 2 || n >= 0x7FFFFFF0L) return nullptr;
    {
        long same = 0;
        for (long t = 1; t < n; t++) same += user[t] == user[t - 1];
        if (2 * same < n) return nullptr;   // (before the id checks: a random-order stream leaves them, and their messages, to the plain builder)
    }
    const long NU = mp_.num_user, NI = mp_.num_item;
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)NU) fail("user feature index exceed bound");
        if (pos[r] >= (unsigned)NI || neg[r] >= (unsigned)NI) fail("item feature index exceed bound");
        if (pos[r] == neg[r]) fail("dataset_from_pairs: positive and negative item of a pair must differ");
    }
    // ---- columns in file order: lower / higher item id, sign of the lower entry
    std::vector<unsigned> lo((size_t)n), hi((size_t)n);
    std::vector<float> vlo((size_t)n);
    for (long t = 0; t < n; t++) {
        const bool pf = pos[t] < neg[t];
        lo[(size_t)t] = pf ? pos[t] : neg[t]; hi[(size_t)t] = pf ? neg[t] : pos[t]; vlo[(size_t)t] = pf ? 1.0f : -1.0f;
    }
    std::vector<PairUnit> sorted;
    std::vector<long> lptr;
    if (!punit_build(n, user, lo.data(), hi.data(), sorted, lptr)) return nullptr;
    const long nunit = (long)sorted.size();
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    ds->kind = 11; ds->num_row = n; ds->num_units = nunit;
    ds->pu_units.upload(sorted.data(), sorted.size(), stream_);
    ds->pu_lo.upload(lo.data(), lo.size(), stream_);
    ds->pu_hi.upload(hi.data(), hi.size(), stream_);
    ds->pu_vlo.upload(vlo.data(), vlo.size(), stream_);
    ds->sched.level_ptr = lptr;
    ds->sched.max_level_size = 0;
    for (size_t l = 0; l + 1 < lptr.size(); l++) ds->sched.max_level_size = std::max(ds->sched.max_level_size, lptr[l + 1] - lptr[l]);
    const long nb = (mp_.no_user_bias ? 0 : 1) + 2;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 3 + 8 * nb + 16 + 8 * 3);   // what the reference's step moves per pair
    HIPCHECK(hipStreamSynchronize(stream_));
    return ds.release();
}

// A staged window (svdf_update_csr calls: the reference CLI's per-instance route, svd_feature.cpp:220-248 with input_type = 2) whose rows are all rank pairs
// in the generator's shape -- no global entry, one user entry of value 1, two item entries of values (v, -v), |v| = 1, label 1 -- and user-grouped: walked
// as user-run units like a resident pair data set.  false: not that shape (the caller schedules the window level by level).
bool Engine::punit_flush(HostCSR &src) {
    const long n = src.num_row();
    if (!punit_config_ok() || n < 64) return false;
    const DevParams &P = params();
    if (!pair_units_applies(P)) return false;
    std::vector<unsigned> user((size_t)n), lo((size_t)n), hi((size_t)n);
    std::vector<float> vlo((size_t)n);
    const long NU = mp_.num_user, NI = mp_.num_item;
    for (long r = 0; r < n; r++) {
        const int *p = &src.row_ptr[(size_t)3 * r];
        if (p[1] != p[0] || p[2] != p[1] + 1 || p[3] != p[2] + 2 || src.row_label[(size_t)r] != 1.0f) return false;
        const unsigned *ix = &src.feat_index[(size_t)p[1]];
        const float *vx = &src.feat_value[(size_t)p[1]];
        if (vx[0] != 1.0f || !(vx[1] == 1.0f || vx[1] == -1.0f) || vx[2] != -vx[1] || ix[1] >= ix[2]) return false;
        if (ix[0] >= (unsigned)NU || ix[1] >= (unsigned)NI || ix[2] >= (unsigned)NI) return false;   // (the plain path raises the reference's message)
        user[(size_t)r] = ix[0]; lo[(size_t)r] = ix[1]; hi[(size_t)r] = ix[2]; vlo[(size_t)r] = vx[1];
    }
    std::vector<PairUnit> sorted;
    std::vector<long> lptr;
    if (!punit_build(n, user.data(), lo.data(), hi.data(), sorted, lptr)) return false;
    need_device("update");
    w_pu_units_.upload(sorted.data(), sorted.size(), stream_);
    w_user_.upload(lo.data(), (size_t)n, stream_);
    w_item_.upload(hi.data(), (size_t)n, stream_);
    w_label_.upload(vlo.data(), (size_t)n, stream_);
    HIPCHECK(hipStreamSynchronize(stream_));   // (the host vectors go out of scope)
    const PairUnitSchedule S{w_pu_units_.p, w_user_.p, w_item_.p, w_label_.p};
    for (size_t l = 0; l + 1 < lptr.size(); l++) launch_pair_units(P, S, lptr[l], lptr[l + 1], nullptr, stream_);
    HIPCHECK(hipGetLastError());
    const int64_t L = (int64_t)lptr.size() - 1;
    n_launches_ += L;
    n_batches_ += L;
    n_instances_ += n;
    sample_counter_ += (unsigned)n;
    n_flushes_++;
    n_punit_passes_++;
    src.clear();
    return true;
}

PairUnitSchedule Engine::punit_view(const Dataset *ds) const { return PairUnitSchedule{ds->pu_units.p, ds->pu_lo.p, ds->pu_hi.p, ds->pu_vlo.p}; }

void Engine::punit_train(Dataset *ds) {
    const DevParams &P = params();
    check(pair_units_applies(P) && punit_config_ok(), "train_dataset: the data set was built as user-run units of rank pairs (svdf_punit.cpp); the configuration changed since");
    const PairUnitSchedule S = punit_view(ds);
    const std::vector<long> &lp = ds->sched.level_ptr;
    const size_t NL = lp.size() - 1;
    // one launch per level, a 64-thread workgroup per unit: every unit gets a CU of its own.  (Walking runs of narrow levels inside ONE launch -- one
    // workgroup, a barrier per level, the scheme of k_fewrow_slots_chain -- was built and measured SLOWER here: 2.62 against 3.39 M pairs/s at cap 16; the
    // units of a level then share one CU's memory pipeline, and a level is 11 us of walk against 3 us of boundary.)
    for (size_t l = 0; l < NL; l++) launch_pair_units(P, S, lp[l], lp[l + 1], nullptr, stream_);
    HIPCHECK(hipGetLastError());
    const int64_t L = (int64_t)NL;
    n_launches_ += L;
    n_batches_ += L;
    n_punit_passes_++;
}

// scores of every pair in file order (the columns are in file order: nothing to un-permute)
void Engine::punit_predict(Dataset *ds, float *d_out) {
    const DevParams &P = params();
    launch_pair_units(P, punit_view(ds), 0, ds->num_units, d_out, stream_);
}

}  // namespace svdf
