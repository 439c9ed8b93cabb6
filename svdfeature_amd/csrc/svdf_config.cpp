// svdf_config.cpp -- part of the host engine (class Engine, svdf_engine.h): config keys (SVDTrainParam / SVDModelParam / ParameterSet / side tables, extension keys), counters, tuning knobs
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

// =============================================================================== small parsers
void ParamSet::set_param(const char *name, const char *val) {  // apex_svd_base.h:48-68
    if (!strncmp(name, prefix_a.c_str(), prefix_a.size())) name += prefix_a.size();
    else if (!strncmp(name, prefix_b.c_str(), prefix_b.size())) name += prefix_b.size();
    else return;
    if (!strcmp("bound", name)) {
        unsigned bd = (unsigned)atoi(val);
        check(bd > 0, "can't give 0 as bound");
        check(bound.empty() || bound.back() < bd, "bound must be given in order");
        check(bound.size() + 1 == wd.size(), "must specifiy wd in each range");
        bound.push_back(bd - 1);
    }
    if (!strcmp("wd", name)) {
        check(wd.size() == bound.size(), "setting must be exactly");
        wd.push_back((float)atof(val));
    }
}
void SideTable::load(const char *fname) {  // apex-utils/apex_utils.h:172-195
    row_ptr.assign(1, 0);
    index.clear();
    value.clear();
    FILE *fi = fopen(fname, "r");
    if (!fi) fail(std::string("can not open file \"") + fname + "\"");
    int n;
    while (fscanf(fi, "%d", &n) == 1) {
        row_ptr.push_back(row_ptr.back() + (unsigned)n);
        for (int i = 0; i < n; i++) {
            unsigned idx;
            float v;
            if (fscanf(fi, "%u:%f", &idx, &v) != 2) { fclose(fi); fail("load sparse feature"); }
            index.push_back(idx);
            value.push_back(v);
        }
    }
    fclose(fi);
}
// ---- SVDTrainParam / SVDModelParam config keys (apex_svd_model.h:350-368, 456-476).  One row per key: the fields it sets
// (a key like wd_uiset sets two) and how the text is read (the reference uses atof / atoi; unsigned fields take atoi's value).
// Unknown keys fall through without a word, as in the reference (svd_feature.cpp:145-150).
namespace {
template <typename P>
struct KeyRow {
    const char *key;
    float P::*f[2];
    int P::*i[2];
    unsigned P::*u;
};
template <typename P>
void apply_key(const KeyRow<P> *rows, size_t n, P &p, const char *name, const char *val) {
    for (size_t r = 0; r < n; r++) {
        if (strcmp(rows[r].key, name)) continue;
        for (int j = 0; j < 2; j++) {
            if (rows[r].f[j]) p.*(rows[r].f[j]) = (float)atof(val);
            if (rows[r].i[j]) p.*(rows[r].i[j]) = atoi(val);
        }
        if (rows[r].u) p.*(rows[r].u) = (unsigned)atoi(val);
    }
}
#define KF(P, key, a) {key, {&P::a, nullptr}, {nullptr, nullptr}, nullptr}
#define KF2(P, key, a, b) {key, {&P::a, &P::b}, {nullptr, nullptr}, nullptr}
#define KI(P, key, a) {key, {nullptr, nullptr}, {&P::a, nullptr}, nullptr}
#define KI2(P, key, a, b) {key, {nullptr, nullptr}, {&P::a, &P::b}, nullptr}
#define KU(P, key, a) {key, {nullptr, nullptr}, {nullptr, nullptr}, &P::a}
const KeyRow<TrainParam> kTrainKeys[] = {
    KF(TrainParam, "learning_rate", learning_rate),
    KF(TrainParam, "wd_user", wd_user), KF(TrainParam, "wd_item", wd_item), KF2(TrainParam, "wd_uiset", wd_user, wd_item),
    KF(TrainParam, "wd_user_bias", wd_user_bias), KF(TrainParam, "wd_item_bias", wd_item_bias), KF2(TrainParam, "wd_uiset_bias", wd_user_bias, wd_item_bias),
    KF(TrainParam, "wd_global", wd_global),
    KI(TrainParam, "reg_method", reg_method), KI(TrainParam, "reg_global", reg_global), KU(TrainParam, "num_regfree_global", num_regfree_global),
    KI(TrainParam, "decay_learning_rate", decay_learning_rate), KF(TrainParam, "min_learning_rate", min_learning_rate), KF(TrainParam, "decay_rate", decay_rate),
    KF(TrainParam, "scale_lr_ufeedback", scale_lr_ufeedback), KF(TrainParam, "wd_ufeedback", wd_ufeedback), KF(TrainParam, "wd_ufeedback_bias", wd_ufeedback_bias),
};
const KeyRow<ModelParam> kModelKeys[] = {
    KI(ModelParam, "num_user", num_user), KI(ModelParam, "num_item", num_item), KI2(ModelParam, "num_uiset", num_user, num_item),
    KI(ModelParam, "num_global", num_global), KI(ModelParam, "num_factor", num_factor), KI(ModelParam, "num_ufeedback", num_ufeedback),
    KF(ModelParam, "u_init_sigma", u_init_sigma), KF(ModelParam, "i_init_sigma", i_init_sigma), KF2(ModelParam, "ui_init_sigma", u_init_sigma, i_init_sigma),
    KF(ModelParam, "ufeedback_init_sigma", ufeedback_init_sigma), KF(ModelParam, "base_score", base_score), KI(ModelParam, "no_user_bias", no_user_bias),
    KI(ModelParam, "num_randinit_ufactor", num_randinit_ufactor), KI(ModelParam, "num_randinit_ifactor", num_randinit_ifactor),
    KI2(ModelParam, "num_randinit_uifactor", num_randinit_ifactor, num_randinit_ufactor),
    KI(ModelParam, "common_latent_space", common_latent_space), KI(ModelParam, "common_feedback_space", common_feedback_space),
    KI(ModelParam, "user_nonnegative", user_nonnegative), KI(ModelParam, "item_nonnegative", item_nonnegative),
};
#undef KF
#undef KF2
#undef KI
#undef KI2
#undef KU
}  // namespace
void config_set_train_param(TrainParam &p, const char *name, const char *val) { apply_key(kTrainKeys, sizeof(kTrainKeys) / sizeof(kTrainKeys[0]), p, name, val); }
void config_set_model_param(ModelParam &p, const char *name, const char *val) { apply_key(kModelKeys, sizeof(kModelKeys) / sizeof(kModelKeys[0]), p, name, val); }

void Engine::set_param(const char *name, const char *val) {  // apex_svd_base.h:126-136
    if (save_async_.active) save_model_end();   // (the asynchronous writer reads mp_ live)
    if (trainer_ready_ && !host_only_) flush();   // staged instances were issued under the old parameters
    // N GPUs behind one handle (svdf_multi.cpp): extension keys, ignored by the reference like any unknown key
    if (!strcmp(name, "amd:gpus")) { check(!multi_ && !space_allocated_, "amd:gpus must be set before the model is created"); gpus_ = std::max(1, atoi(val)); }
    else if (!is_peer_) param_log_.emplace_back(name, val);
    if (!strcmp(name, "amd:delta_half")) delta_half_ = atoi(val) != 0;
    if (!strcmp(name, "amd:exchange")) {
        check(!strcmp(val, "p2p") || !strcmp(val, "rccl"), "amd:exchange must be p2p or rccl");
        check(!multi_, "amd:exchange must be set before the model is created");
        multi_exchange_mode_ = !strcmp(val, "rccl") ? 1 : 0;
    }
    if (!strcmp(name, "amd:step")) {
        check(!strcmp(val, "minibatch") || !strcmp(val, "levels") || !strcmp(val, "auto"), "amd:step must be minibatch, levels or auto");
        check(!multi_, "amd:step must be set before the model is created");
        multi_step_levels_ = !strcmp(val, "levels");
        step_minibatch_set_ = !strcmp(val, "minibatch");   // one GPU: opt-in window-minibatch SGD (svdf_wunit.cpp: resident data sets become window sequences)
        step_auto_set_ = !strcmp(val, "auto");             // one GPU: opt-in, decided per resident data set from its level schedule (svdf_dataset.cpp: auto_step)
    }
    if (!strcmp(name, "amd:contrib")) {   // window-minibatch step: storage format of the contribution rows (sums are fp32 either way)
        check(!strcmp(val, "fp32") || !strcmp(val, "bf16"), "amd:contrib must be fp32 or bf16");
        contrib_bf16_ = !strcmp(val, "bf16");
    }
    if (!strcmp(name, "amd:window")) { stage_window_ = std::max<long>(1, atol(val)); window_set_ = true; }
    // the data-driven window rule of the window / N-rank steps (svdf_wunit.cpp, svdf_multi.cpp): updates a shared row may meet per window, mean and most
    if (!strcmp(name, "amd:window_per_target")) { check(atoi(val) >= 1, "amd:window_per_target must be positive"); wseq_per_target_ = atoi(val); }
    if (!strcmp(name, "amd:window_per_target_max")) { check(atoi(val) >= 1, "amd:window_per_target_max must be positive"); wseq_per_target_max_ = atoi(val); }
    if (multi_) for (int d = 1; d < gpus_; d++) rank_engine(d)->set_param(name, val);
    if (!strcmp(name, "feature_user")) name_feat_user_ = val;
    if (!strcmp(name, "feature_item")) name_feat_item_ = val;
    // extension keys (ignored by the reference like any unknown key): relaxed handling of shared ids
    if (imfb() && !strcmp(name, "ufeedback_disable_level")) {   // apex_multi_imfb.h:58-67
        const int level = atoi(val);
        check(level >= 0, "ufeedback_disable_level must not be negative");
        if (level < 32) imfb_disable_ |= 1u << level;          // levels beyond IMFB_DEPTH can never open here
    }
    if (bilinear()) {   // apex_svd_bilinear.h:187-193
        if (!strcmp(name, "reg_bi_feedback")) reg_bi_feedback_ = atoi(val);
        if (!bi_allocated_) {
            if (!strcmp(name, "num_bi_feedback")) bi_param_.num_bi_feedback = atoi(val);
            if (!strcmp(name, "start_ufeedback")) bi_param_.start_ufeedback = atoi(val);
        }
    }
    if (!strcmp(name, "amd:relax_global")) relax_global_ = atoi(val) != 0;
    if (!strcmp(name, "amd:relax_feedback")) relax_feedback_ = atoi(val) != 0;
    if (!strcmp(name, "amd:relax_user_from")) relax_user_from_ = (unsigned)strtoul(val, nullptr, 10);
    if (!strcmp(name, "amd:relax_item_from")) relax_item_from_ = (unsigned)strtoul(val, nullptr, 10);
    pair_sampler_.set_param(name, val);   // the reference hands every config pair to the data iterator too (svd_feature.cpp:128-143)
    config_set_train_param(tp_, name, val);
    u_param_.set_param(name, val);
    i_param_.set_param(name, val);
    g_param_.set_param(name, val);
    if (!space_allocated_) config_set_model_param(mp_, name, val);
    params_dirty_ = true;
}
int64_t Engine::counter(int what) const {
    switch (what) {
    case 0: return n_instances_;
    case 1: return n_launches_;
    case 2: return n_batches_;
    case 3: return n_flushes_;
    case 4: return n_kind_[0];
    case 5: return n_kind_[1];
    case 6: return n_kind_[2];
    case 7: return n_device_rank_passes_;
    case 8: return multi_counter(0);    // item-delta exchanges of an amd:gpus > 1 handle
    case 9: return multi_counter(1);    // 1 when they run through RCCL
    case 10: return multi_counter(2);   // 1 when every rank has a device of its own
    case 11: return multi_counter(3);   // exchange windows trained with the window-minibatch step
    case 12: return multi_counter(4);   // exchange path: 0 p2p, 1 rccl
    case 13: return n_init_reports_;    // init_model on the device: values the host libm decided (near a float rounding boundary)
    case 14: return n_init_draws_;      // init_model on the device: rand() draws consumed
    case 15: return n_chained_levels_;  // conflict-free levels executed inside chained launches (k_fewrow_slots_chain)
    // amd:step = auto, the decision taken for the data set built last (svdf_dataset.cpp: auto_step)
    case 16: return auto_last_.decided;                           // 0 none yet, 1 exact levels kept, 2 window step chosen, 3 exact kept because the window step does not cover the configuration / the data
    case 17: return auto_last_.levels;
    case 18: return (int64_t)(auto_last_.dag_ms * 1000.0);        // levels x unit latency, microseconds
    case 19: return (int64_t)(auto_last_.stream_ms * 1000.0);     // algorithmic bytes at the measured random-row rate, microseconds
    case 20: return auto_last_.windows;
    case 23: return n_runs_passes_;     // passes over data sets scheduled as runs of an item's consecutive ratings (svdf_runs.cpp)
    case 22: return n_pivot_passes_;    // passes over data sets with hot rows walked as units (svdf_pivot.cpp)
    case 24: return unit_sched_us_;     // microseconds the last user-unit data set's schedule took (levels, order, fast-path flags)
    case 25: return unit_sched_on_device_ ? 1 : 0;   // ... built by svdf_k_sched.hip's device_schedule_units (1) or the host scan (0)
    case 26: return n_guard_warnings_;                            // data sets whose DEFAULT (exact) step was predicted > 10 x slower than the streaming model (note_dataset)
    case 27: return (int64_t)(guard_last_.dag_ms * 1000.0);       // ... the last noted data set's dag bound / stream model, microseconds
    case 28: return (int64_t)(guard_last_.stream_ms * 1000.0);
    case 29: return n_punit_passes_;   // passes over rank pairs walked as user-run units (svdf_punit.cpp)
    case 21: return 0;   // (was: passes of the in-launch DAG executor, removed in round 6 -- DESIGN_APPENDIX.md section K)
    default: return -1;
    }
}
int Engine::set_knob(const char *name, long value) {
    launch_version_++;   // any knob may change what a captured pass would launch
    // tuning knobs reach every rank of an amd:gpus handle (they never change a result; the exchange window is the handle's own)
    if (multi_ && !is_peer_ && strcmp(name, "stage_window") != 0 && strcmp(name, "async_flush") != 0)
        for (int d = 1; d < gpus_; d++) (void)rank_engine(d)->set_knob(name, value);
    if (!strcmp(name, "use_graph")) { use_graph_ = value != 0; return 0; }
    if (!strcmp(name, "stage_window")) { check(value >= 1, "stage_window must be >= 1"); stage_window_ = value; window_set_ = true; return 0; }
    if (!strcmp(name, "groups_per_wave")) {
        check(value >= 0 && value <= 8 && value != 7, "groups_per_wave must be 0 (auto), 1 ... 6 or 8");
        groups_per_wave_ = (int)value;
        return 0;
    }
    if (!strcmp(name, "xcd_remap")) { xcd_remap_ = value != 0; params_dirty_ = true; return 0; }
    if (!strcmp(name, "fewrow_i16")) { check(value >= 0 && value <= 1, "fewrow_i16 must be 0 or 1"); fewrow_i16_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "svdpp_helpers")) { check(value == 1 || value == 4 || value == 8 || value == 16, "svdpp_helpers must be 1, 4, 8 or 16"); svdpp_helpers_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "basic_i8")) { check(value >= 0 && value <= 1, "basic_i8 must be 0 or 1"); basic_i8_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "load_mode")) { check(value >= 0 && value <= 2, "load_mode must be 0, 1 or 2 (auto)"); load_mode_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "store_mode")) { check(value >= 0 && value <= 2, "store_mode must be 0, 1 or 2"); store_mode_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "sort_batches")) { check(value >= 0 && value <= 2, "sort_batches must be 0, 1 (by item) or 2 (by user)"); sort_batches_ = (int)value; return 0; }
    if (!strcmp(name, "async_flush")) { flush(); async_flush_ = value != 0; return 0; }
    if (!strcmp(name, "use_fused")) { use_fused_ = value != 0; return 0; }
    if (!strcmp(name, "device_schedule")) { device_sched_ = value != 0; return 0; }
    if (!strcmp(name, "device_rank")) { device_rank_ = value != 0; return 0; }
    if (!strcmp(name, "device_init")) { device_init_ = value != 0; return 0; }
    if (!strcmp(name, "device_window")) { device_window_ = value != 0; return 0; }
    if (!strcmp(name, "device_load")) { device_load_ = value != 0; return 0; }
    if (!strcmp(name, "runs_exec")) { check(value == 0 || value == 1, "runs_exec must be 0 or 1"); runs_exec_ = (int)value; return 0; }
    if (!strcmp(name, "runs_len")) { check(value >= 2 && value <= 7, "runs_len must be in 2 .. 7"); runs_len_ = (int)value; return 0; }
    if (!strcmp(name, "runs_sets")) { check(value >= 1 && value <= 2, "runs_sets must be 1 or 2"); runs_sets_ = (int)value; return 0; }
    if (!strcmp(name, "runs_block")) { check(value == 64 || value == 128 || value == 256, "runs_block must be 64, 128 or 256"); runs_block_ = (int)value; return 0; }
    if (!strcmp(name, "runs_min_rows")) { check(value >= 0, "runs_min_rows must not be negative"); runs_min_rows_ = value; return 0; }
    if (!strcmp(name, "pair_units")) { check(value == 0 || value == 1, "pair_units must be 0 or 1"); pair_units_ = (int)value; return 0; }
    if (!strcmp(name, "pair_unit_cap")) { check(value >= 1 && value <= 4096, "pair_unit_cap must be in 1 .. 4096"); pair_unit_cap_ = (int)value; return 0; }
    if (!strcmp(name, "pivot_exec")) { check(value == 0 || value == 1, "pivot_exec must be 0 or 1"); pivot_exec_ = (int)value; return 0; }
    if (!strcmp(name, "pivot_run")) { check(value >= 1 && value <= 65536, "pivot_run must be in 1 .. 65536"); pivot_run_ = (int)value; return 0; }
    if (!strcmp(name, "pivot_min")) { check(value >= 2, "pivot_min must be at least 2"); pivot_min_ = (int)value; return 0; }
    if (!strcmp(name, "ipc_spin_limit")) { ipc_set_spin_limit(value); return 0; }
    if (!strcmp(name, "chain_width")) { check(value >= 0, "chain_width must not be negative"); chain_width_ = value; return 0; }
    if (!strcmp(name, "device_init_margin_log2")) { check(value >= 8 && value <= 52, "device_init_margin_log2 must be in 8 .. 52"); device_init_margin_log2_ = (int)value; return 0; }
    if (!strcmp(name, "device_schedule_min")) { check(value >= 1, "device_schedule_min must be >= 1"); device_sched_min_ = value; return 0; }
    if (!strcmp(name, "use_simple_units")) { use_simple_units_ = value != 0; return 0; }
    if (!strcmp(name, "rows_without_feedback")) { rows_without_feedback_ = value != 0; return 0; }
    if (!strcmp(name, "fewrow_gslots")) { fewrow_gslots_ = value != 0; launch_version_++; return 0; }
    if (!strcmp(name, "wunit_inplace")) { wunit_inplace_ = value != 0; return 0; }
    if (!strcmp(name, "wunit_defer_fb")) { wunit_defer_fb_ = value != 0; return 0; }
    if (!strcmp(name, "wunit_fast")) { check(value >= 0 && value <= 2, "wunit_fast must be 0, 1 or 2"); wunit_fast_ = (int)value; return 0; }
    if (!strcmp(name, "window_per_target_fb")) { check(value >= 1, "window_per_target_fb must be positive"); wseq_per_target_fb_ = (int)value; return 0; }
    if (!strcmp(name, "window_hot_sub")) { check(value >= 0 && value <= 4096, "window_hot_sub must be in 0 .. 4096"); wseq_hot_sub_ = (int)value; return 0; }
    if (!strcmp(name, "window_hot_max")) { check(value >= 1, "window_hot_max must be positive"); wseq_hot_max_ = (int)value; return 0; }
    if (!strcmp(name, "window_per_target_max")) { check(value >= 1, "window_per_target_max must be positive"); wseq_per_target_max_ = (int)value; return 0; }
    if (!strcmp(name, "window_per_target")) { check(value >= 1, "window_per_target must be positive"); wseq_per_target_ = (int)value; return 0; }
    if (!strcmp(name, "window_slots")) { window_slots_ = value != 0; return 0; }
    if (!strcmp(name, "window_groups")) { check(value >= 0 && value <= 2, "window_groups must be 0 (auto), 1 or 2"); window_groups_ = (int)value; return 0; }
    if (!strcmp(name, "block_threads")) {
        check(value == 0 || value == 64 || value == 128 || value == 256, "block_threads must be 0 (auto), 64, 128 or 256");
        block_threads_ = (int)value;
        return 0;
    }
    return -1;
}

}  // namespace svdf
