// svdf_stream.cpp -- host side of the in-launch DAG executor (svdf_k_stream.hip; knob "stream_exec"): builds a data
// set's tile plan from its level schedule (tiles, the predecessor tile of every row an instance touches) and issues a pass as ONE launch.
// What it replaces in the reference: nothing changes in SVDFeature::update_inner's arithmetic or order (apex_svd_base.h:456-462) -- only
// HOW the conflict-free order of DESIGN.md section 2 is enforced on the device (exact predecessors instead of level boundaries).
#include <algorithm>
#include <cstring>
#include <memory>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {

// cols[s]: the row ids of slot s in SCHEDULE order (device); rows of different slots are different parameter rows unless `same_space`
// marks slots whose ids live in one id space (the two item entries of a rank pair): those are chained together.
void Engine::stream_build(Dataset *ds, int TS, const unsigned *const *cols, int nslots, const int *space_of_slot) {
    const long n = ds->num_row;
    const Schedule &sc = ds->sched;
    const long L = (long)sc.num_levels();
    check(n < (long)0xFFFFFFF0u, "stream_exec: at most 2^32 - 16 instances per data set");
    std::vector<unsigned> tile_base((size_t)L + 1), lp((size_t)L + 1);
    unsigned long long nt = 0;
    for (long l = 0; l < L; l++) {
        tile_base[(size_t)l] = (unsigned)nt;
        lp[(size_t)l] = (unsigned)sc.level_ptr[(size_t)l];
        nt += (unsigned long long)((sc.level_ptr[(size_t)l + 1] - sc.level_ptr[(size_t)l] + TS - 1) / TS);
        check(nt < 0xFFFFFFF0ull, "stream_exec: too many tiles");
    }
    tile_base[(size_t)L] = (unsigned)nt;
    lp[(size_t)L] = (unsigned)sc.level_ptr[(size_t)L];
    ds->st_ntiles = (unsigned)nt;
    DevBuf<unsigned> d_base, d_lp, tile_of_pos, keys_a, keys_b, vals_a, vals_b;
    d_base.upload(tile_base.data(), tile_base.size(), stream_);
    d_lp.upload(lp.data(), lp.size(), stream_);
    ds->st_tile_hdr.reserve((size_t)nt);
    tile_of_pos.reserve((size_t)n);
    launch_stream_tiles(d_base.p, d_lp.p, L, TS, (unsigned)nt, ds->st_tile_hdr.p, tile_of_pos.p, stream_);
    HIPCHECK(hipGetLastError());
    keys_a.reserve((size_t)n); keys_b.reserve((size_t)n); vals_a.reserve((size_t)n); vals_b.reserve((size_t)n);
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    // slots of one id space are chained TOGETHER (an item row touched as the lower id of one pair and the higher id of another is one
    // row): their (id, position) pairs are sorted as one array of m * n entries.  Distinct spaces get their own sort.
    for (int s = 0; s < nslots; s++) {
        bool first_of_space = true;
        int members = 0;
        for (int j = 0; j < nslots; j++) if (space_of_slot[j] == space_of_slot[s]) { if (j < s) first_of_space = false; members++; }
        ds->st_pred[s].reserve((size_t)n);
        if (!first_of_space) continue;
        if (members == 1) {
            HIPCHECK(hipMemcpyAsync(keys_a.p, cols[s], (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, stream_));
            launch_stream_iota(vals_a.p, n, stream_);
            device_sort_pairs_u32(keys_a.p, keys_b.p, vals_a.p, vals_b.p, n, &tmp, &tmp_bytes, stream_);
            launch_stream_preds(keys_b.p, vals_b.p, n, 0xFFFFFFFFu, tile_of_pos.p, ds->st_pred[s].p, 1, 0, stream_);
        } else {
            // entry e = member * n + position; sorted by id, ties by entry: a stable sort of the concatenation would put member 0's
            // entries of an id before member 1's whatever their positions, so the sort key is (id, position) in two stable passes
            const long mn = (long)members * n;
            check(mn < (long)0xFFFFFFF0u, "stream_exec: too many entries in one id space");
            DevBuf<unsigned> ka, kb, va, vb;
            ka.reserve((size_t)mn); kb.reserve((size_t)mn); va.reserve((size_t)mn); vb.reserve((size_t)mn);
            // pass 1 is free: entries are generated position-major (entry e = position * members + member), i.e. already sorted by position
            std::vector<const unsigned *> mem_cols;
            std::vector<int> mem_slots;
            for (int j = 0; j < nslots; j++) if (space_of_slot[j] == space_of_slot[s]) { mem_cols.push_back(cols[j]); mem_slots.push_back(j); }
            launch_stream_interleave(mem_cols.data(), members, n, ka.p, va.p, stream_);   // ka[e] = id, va[e] = e
            device_sort_pairs_u32(ka.p, kb.p, va.p, vb.p, mn, &tmp, &tmp_bytes, stream_);
            for (int mi = 0; mi < members; mi++)
                launch_stream_preds(kb.p, vb.p, mn, 0xFFFFFFFFu, tile_of_pos.p, ds->st_pred[mem_slots[(size_t)mi]].p, members, mi, stream_);
        }
        HIPCHECK(hipGetLastError());
    }
    ds->st_done.reserve((size_t)std::max<unsigned long long>(nt, 1));
    HIPCHECK(hipMemsetAsync(ds->st_done.p, 0, (size_t)std::max<unsigned long long>(nt, 1) * sizeof(unsigned), stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    if (tmp) (void)hipFree(tmp);
    ds->st_pass = 0;
    ds->st_nslots = nslots;
    ds->st_built = true;
}

StreamPlan Engine::stream_view(const Dataset *ds) {
    if (!stream_err_) {   // host-mapped: a timed-out wait is seen without a copy
        HIPCHECK(hipHostMalloc((void **)&stream_err_, 64, hipHostMallocMapped));
        *stream_err_ = 0u;
        HIPCHECK(hipHostGetDevicePointer((void **)&stream_err_dev_, stream_err_, 0));
    }
    StreamPlan T;
    memset(&T, 0, sizeof(T));
    T.tile_hdr = ds->st_tile_hdr.p;
    for (int s = 0; s < 3; s++) T.pred[s] = s < ds->st_nslots ? ds->st_pred[s].p : nullptr;
    T.done = ds->st_done.p;
    T.err = stream_err_dev_;
    T.ntiles = ds->st_ntiles;
    T.spin_limit = (unsigned)stream_spin_limit_;
    T.debug_mode = stream_debug_mode_;
    const unsigned long long wb = (unsigned long long)n_uiset_ * (unsigned long long)pitch_ * 4ull;
    T.w_bytes = (unsigned)std::min<unsigned long long>(wb, 0xFFFFFFFFull);
    return T;
}

// true when this data set's pass can run as one in-launch DAG (the plan is built on first use)
bool Engine::stream_applies(Dataset *ds) {
    if (!stream_exec_ || host_only_ || lazy_decay() || relaxed()) return false;
    if ((unsigned long long)n_uiset_ * (unsigned long long)pitch_ * 4ull > 0xFFFFFFFFull) return false;   // 32-bit buffer offsets
    const DevParams &P = params();
    if (ds->kind == 0) {
        BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
        if (!stream_basic_applies(P, S) || ds->num_row == 0) return false;
        if (!ds->st_built) {
            const unsigned *cols[2] = {ds->user.p, ds->item.p};
            const int space[2] = {0, 1};
            stream_build(ds, stream_basic_tile_size(P), cols, 2, space);
        }
        return true;
    }
    return false;
}

void Engine::stream_train(Dataset *ds) {
    const DevParams &P = params();
    StreamPlan T = stream_view(ds);
    ds->st_pass++;
    if (ds->st_pass == 0u) {   // the stamp wrapped: 0 is "never"
        HIPCHECK(hipMemsetAsync(ds->st_done.p, 0, (size_t)std::max<unsigned>(ds->st_ntiles, 1u) * sizeof(unsigned), stream_));
        ds->st_pass = 1u;
    }
    if (stream_num_cu_ == 0) {
        hipDeviceProp_t prop;
        HIPCHECK(hipGetDeviceProperties(&prop, device_));
        stream_num_cu_ = prop.multiProcessorCount;
    }
    // static tile assignment: every wave must be resident (svdf_k_stream.hip)
    int waves = stream_waves_ > 0 ? stream_waves_ : stream_num_cu_ * 8;
    waves = std::min(waves, stream_basic_max_waves(stream_num_cu_));
    waves = (int)std::min<long>(waves, (long)std::max<unsigned>(ds->st_ntiles, 1u));
    if (ds->kind == 0) {
        BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, nullptr, nullptr};
        launch_basicmf_stream(P, S, T, ds->st_pass, waves, stream_);
    }
    HIPCHECK(hipGetLastError());
    n_stream_passes_++;
}

void Engine::stream_fail_if_dead(const char *where) {
    if (stream_err_ && *stream_err_ != 0u) {
        *stream_err_ = 0u;
        fail(std::string(where) + ": a wait of the in-launch DAG executor (knob stream_exec) hit its spin limit -- the pass was abandoned and the "
             "model holds a partial pass; this is a bug of the tile plan or a hung GPU, not a property of the data");
    }
}

}  // namespace svdf
