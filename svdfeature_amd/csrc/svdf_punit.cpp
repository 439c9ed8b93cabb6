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

// nullptr: not a user-grouped stream (or a configuration outside the walker): the caller builds the plain level schedule
Dataset *Engine::punit_dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    if (!punit_config_ok() || n < 2 || n >= 0x7FFFFFF0L) return nullptr;
    long same = 0;
    for (long t = 1; t < n; t++) same += user[t] == user[t - 1];
    if (2 * same < n) return nullptr;
    const long NU = mp_.num_user, NI = mp_.num_item;
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)NU) fail("user feature index exceed bound");
        if (pos[r] >= (unsigned)NI || neg[r] >= (unsigned)NI) fail("item feature index exceed bound");
        if (pos[r] == neg[r]) fail("dataset_from_pairs: positive and negative item of a pair must differ");
    }
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
        while (e < n && e - s < cap && user[e] == uu && in_unit[pos[e]] != uid && in_unit[neg[e]] != uid) {
            in_unit[pos[e]] = uid; in_unit[neg[e]] = uid;
            lmax = std::max(lmax, std::max(lasti[pos[e]], lasti[neg[e]]));
            e++;
        }
        const int l = lmax + 1;
        lastu[uu] = l;
        for (long t = s; t < e; t++) { lasti[pos[t]] = l; lasti[neg[t]] = l; }
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
    std::vector<PairUnit> sorted((size_t)nunit);
    {
        std::vector<long> cur(lptr.begin(), lptr.end());
        for (long j = 0; j < nunit; j++) sorted[(size_t)cur[(size_t)ulevel[(size_t)j]]++] = units[(size_t)j];
    }
    // ---- columns in file order: lower / higher item id, sign of the lower entry
    std::vector<unsigned> lo((size_t)n), hi((size_t)n);
    std::vector<float> vlo((size_t)n);
    for (long t = 0; t < n; t++) {
        const bool pf = pos[t] < neg[t];
        lo[(size_t)t] = pf ? pos[t] : neg[t]; hi[(size_t)t] = pf ? neg[t] : pos[t]; vlo[(size_t)t] = pf ? 1.0f : -1.0f;
    }
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    ds->kind = 11; ds->num_row = n; ds->num_units = nunit;
    ds->pu_units.upload(sorted.data(), sorted.size(), stream_);
    ds->pu_lo.upload(lo.data(), lo.size(), stream_);
    ds->pu_hi.upload(hi.data(), hi.size(), stream_);
    ds->pu_vlo.upload(vlo.data(), vlo.size(), stream_);
    ds->sched.level_ptr.assign(lptr.begin() + 1, lptr.end());
    ds->sched.max_level_size = 0;
    for (int l = 0; l < max_level; l++) ds->sched.max_level_size = std::max(ds->sched.max_level_size, ds->sched.level_ptr[(size_t)l + 1] - ds->sched.level_ptr[(size_t)l]);
    const long nb = (mp_.no_user_bias ? 0 : 1) + 2;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 3 + 8 * nb + 16 + 8 * 3);   // what the reference's step moves per pair
    HIPCHECK(hipStreamSynchronize(stream_));
    return ds.release();
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
