// svdf_runs.cpp -- host side of the runs schedule (svdf_k_runs.hip; knob "runs_exec"): plain (user, item, rating)
// triples of the contract configuration become runs of up to R consecutive ratings of one item, formed and level-scheduled in HBM.
// Nothing of SVDFeature::update_inner (apex_svd_base.h:456-462) changes: same instances, every row's touches in file order, the item's row
// kept in registers across a run instead of being written and read back between two of its ratings.
#include <algorithm>
#include <cstring>
#include <memory>

#include "svdf_engine.h"
#include "svdf_kernels.h"
#include "svdf_internal.h"

namespace svdf {

bool Engine::runs_config_ok() const {
    return runs_exec_ != 0 && !host_only_ && device_sched_ && !user_group() && basic_fast_path_allowed() && mtype_.extend_type == 0 && mtype_.active_type == ACT_LINEAR &&
           tp_.reg_method == 0 && mp_.user_nonnegative == 0 && mp_.no_user_bias == 0 && u_param_.bound.empty() && i_param_.bound.empty() && (mp_.num_factor == 64 || mp_.num_factor == 128) &&
           basic_i8_ != 0 && store_mode_ == 0;
}

// nullptr: the configuration has no runs kernel: the caller builds the plain level schedule
Dataset *Engine::runs_dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    if (!runs_config_ok() || n < runs_min_rows_ || n >= 0x7FFFFFF0L) return nullptr;
    const int RF = std::max(2, std::min(7, runs_len_));           // ratings per run at most (the level scheduler takes 8 row slots per unit: 1 item + 7 users)
    const int R = RF <= 2 ? 2 : (RF <= 4 ? 4 : 7);               // the kernel's width: columns beyond a run's length hold SLOT_ABSENT
    const unsigned NU = (unsigned)mp_.num_user, NI = (unsigned)mp_.num_item;
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    ds->kind = 10; ds->num_row = n; ds->rn_len = R;
    // the raw columns stay in file order: the evaluator and predict_dataset score them as they are
    ds->user.upload(user, (size_t)n, stream_);
    ds->item.upload(item, (size_t)n, stream_);
    ds->label.upload(label, (size_t)n, stream_);
    ds->unit_values = true;
    DevBuf<unsigned> flag, ka, kb, va, vb, prev, head, head_of, unit_at, c_item, c_user;
    DevBuf<unsigned char> idx;
    DevBuf<float> c_label;
    flag.reserve(4);
    HIPCHECK(hipMemsetAsync(flag.p, 0, 4 * sizeof(unsigned), stream_));
    launch_runs_check(ds->user.p, ds->item.p, n, NU, NI, flag.p, stream_);
    unsigned hflag = 0;
    HIPCHECK(hipMemcpyAsync(&hflag, flag.p, sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    if (hflag & 1u) fail("user feature index exceed bound");
    if (hflag & 2u) fail("item feature index exceed bound");
    ka.reserve((size_t)n); kb.reserve((size_t)n); va.reserve((size_t)n); vb.reserve((size_t)n);
    prev.reserve((size_t)n); head.reserve((size_t)n); head_of.reserve((size_t)n); unit_at.reserve((size_t)n); idx.reserve((size_t)n);
    void *tmp = nullptr;
    size_t tmp_bytes = 0;
    struct FreeTmp { void *&p; ~FreeTmp() { if (p) (void)hipFree(p); } } free_tmp{tmp};
    // 1. the previous rating of every rating's user
    HIPCHECK(hipMemcpyAsync(ka.p, ds->user.p, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, stream_));
    launch_runs_iota(va.p, n, stream_);
    device_sort_pairs_u32(ka.p, kb.p, va.p, vb.p, n, &tmp, &tmp_bytes, stream_);
    launch_runs_prev(kb.p, vb.p, n, prev.p, stream_);
    // 2. the item-major list, 3. runs
    HIPCHECK(hipMemcpyAsync(ka.p, ds->item.p, (size_t)n * sizeof(unsigned), hipMemcpyDeviceToDevice, stream_));
    launch_runs_iota(va.p, n, stream_);
    device_sort_pairs_u32(ka.p, kb.p, va.p, vb.p, n, &tmp, &tmp_bytes, stream_);
    launch_runs_form(kb.p, vb.p, n, NI, prev.p, RF, head.p, head_of.p, idx.p, stream_);
    // 4. runs numbered by the file position of their head
    const long nunit = device_exclusive_scan_u32(head.p, unit_at.p, n, &tmp, &tmp_bytes, stream_);
    HIPCHECK(hipGetLastError());
    // 5. the runs' columns
    c_item.reserve((size_t)nunit); c_user.reserve((size_t)R * (size_t)nunit); c_label.reserve((size_t)R * (size_t)nunit);
    launch_runs_fill_u32(c_user.p, (long)R * nunit, (unsigned)SLOT_ABSENT, stream_);
    HIPCHECK(hipMemsetAsync(c_label.p, 0, (size_t)R * (size_t)nunit * sizeof(float), stream_));
    launch_runs_fill(ds->user.p, ds->item.p, ds->label.p, n, unit_at.p, head_of.p, idx.p, nunit, c_item.p, c_user.p, c_label.p, stream_);
    HIPCHECK(hipGetLastError());
    HIPCHECK(hipStreamSynchronize(stream_));
    ka.release(); kb.release(); va.release(); vb.release(); prev.release(); head.release(); head_of.release(); unit_at.release(); idx.release();
    // 6. the level schedule of the runs: slot 0 the item row, slots 1 .. R the user rows (absent ones skipped by the scheduler)
    const unsigned *res[SVDF_SCHED_MAX_SLOTS];
    unsigned off[SVDF_SCHED_MAX_SLOTS], limit[SVDF_SCHED_MAX_SLOTS];
    const char *msg[SVDF_SCHED_MAX_SLOTS];
    res[0] = c_item.p; off[0] = NU; limit[0] = NI; msg[0] = "item feature index exceed bound";
    std::vector<DUCol> du{DUCol{c_item.p, &ds->rn_item}};
    std::vector<DFCol> df;
    for (int j = 0; j < R; j++) {
        res[1 + j] = c_user.p + (size_t)j * (size_t)nunit; off[1 + j] = 0u; limit[1 + j] = NU; msg[1 + j] = "user feature index exceed bound";
        du.push_back(DUCol{c_user.p + (size_t)j * (size_t)nunit, &ds->rn_user[j]});
        df.push_back(DFCol{c_label.p + (size_t)j * (size_t)nunit, &ds->rn_label[j]});
    }
    schedule_device_columns(ds.get(), nunit, 1 + R, res, off, limit, msg, sort_batches_ == 1 ? c_item.p : nullptr, NI, du, df);
    ds->num_units = nunit;
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * 2 + 8 * 2 + 16 + 8 * 2);
    return ds.release();
}

void Engine::runs_train(Dataset *ds) {
    const DevParams &P = params();
    RunSchedule S;
    memset(&S, 0, sizeof(S));
    S.item = ds->rn_item.p;
    for (int j = 0; j < ds->rn_len; j++) { S.user[j] = ds->rn_user[j].p; S.label[j] = ds->rn_label[j].p; }
    const Schedule &sc = ds->sched;
    for (size_t l = 0; l < sc.num_levels(); l++) launch_basicmf_runs_soa(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], ds->rn_len, runs_sets_, runs_block_, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_ += (int64_t)sc.num_levels();
    n_batches_ += (int64_t)sc.num_levels();
    n_runs_passes_++;
}

}  // namespace svdf
