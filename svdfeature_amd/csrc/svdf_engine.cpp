// svdf_engine.cpp -- part of the host engine (class Engine, svdf_engine.h): errors, device buffers, lifecycle, staging of update() calls, flush, predict, train_dataset
// Reference citations are relative to /root/reference.
#include <malloc.h>
#include <mutex>

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


// =============================================================================== errors
static int g_error_mode = 0;
static thread_local std::string g_last_error;
void set_error_mode(int m) { g_error_mode = m; }
const char *last_error() { return g_last_error.c_str(); }
void note_error(const std::string &m) { g_last_error = m; }

[[noreturn]] void fail(const std::string &msg) {
    g_last_error = msg;
    if (g_error_mode == 0) {  // apex-utils/apex_utils.h:47-50
        fprintf(stderr, "%s\n", msg.c_str());
        exit(-1);
    }
    throw Error(msg);
}
// "this thread is executing the multi-GPU code of a handle" (svdf_multi.cpp): rank 0 of an amd:gpus handle is the handle itself, so
// while a window is being trained or a resident data set is being built its own flush() / dataset_from_*() must act like a plain
// single-GPU engine's.  Per THREAD, not per handle: the background window thread may be inside multi_flush while the caller's
// thread enters flush() and has to wait for it.
static thread_local int tl_multi_depth = 0;
bool in_multi_scope() { return tl_multi_depth > 0; }
MultiScope::MultiScope() { tl_multi_depth++; }
MultiScope::~MultiScope() { tl_multi_depth--; }

// =============================================================================== lifecycle

// glibc raises its mmap threshold to the size of every mmapped block that is freed ("dynamic threshold"), after which blocks of that size
// come from the arena heaps -- and helper threads that free them make their arenas shrink (madvise / remap of the heap tail).  Each of those
// runs the GPU driver's MMU notifier for the process, and the device queues stall while it does: a ranker call of 38 tiles took 5 ms or
// 25 - 40 ms from one process to the next depending on where malloc had put the blocks (profiles/r05_ranker_malloc.txt; any of
// MALLOC_MMAP_THRESHOLD_ / MALLOC_TRIM_THRESHOLD_ / MALLOC_TOP_PAD_ in the environment: always 5 ms).  Setting the threshold to the value it
// already has switches the adjustment off and changes nothing else.  SVDF_KEEP_MALLOC_DYNAMIC=1 leaves malloc alone.
// Applied by the RANKER only (the component it was measured on; svdf_ranker_create), never by a trainer handle, and not at all when the host
// application has set any MALLOC_* tunable itself: a library does not override the allocator policy of the process it is loaded into.
void pin_malloc_threshold() {
    static std::once_flag once;
    std::call_once(once, [] {
        const char *e = getenv("SVDF_KEEP_MALLOC_DYNAMIC");
        if (e && atoi(e) != 0) return;
        for (const char *name : {"MALLOC_MMAP_THRESHOLD_", "MALLOC_TRIM_THRESHOLD_", "MALLOC_TOP_PAD_", "MALLOC_MMAP_MAX_", "MALLOC_ARENA_MAX", "GLIBC_TUNABLES"})
            if (getenv(name)) return;
        (void)mallopt(M_MMAP_THRESHOLD, 128 * 1024);
    });
}

Engine::Engine(TypeParam mtype, int device) : mtype_(mtype) {
    memset(&mp_, 0, sizeof(mp_));
    mp_.u_init_sigma = mp_.i_init_sigma = 0.01f;   // SVDModelParam() apex_svd_model.h:436-450
    mp_.base_score = 0.5f;
    memset(&tp_, 0, sizeof(tp_));
    tp_.learning_rate = 0.01f;                     // SVDTrainParam() apex_svd_model.h:334-344
    tp_.decay_rate = 1.0f;
    tp_.scale_lr_ufeedback = 1.0f;
    u_param_.prefix_a = "up:"; u_param_.prefix_b = "uip:";   // apex_svd_base.h:103
    i_param_.prefix_a = "ip:"; i_param_.prefix_b = "uip:";
    g_param_.prefix_a = "gp:"; g_param_.prefix_b = "gp:";
    memset(&dev_params_, 0, sizeof(dev_params_));
    memset(&bi_param_, 0, sizeof(bi_param_));
    // apex_svd.cpp:32-44 dispatches on extend_type: 2 = multi-level implicit feedback, 15 = bilinear, 30 / 31 = GBRT, 1 = SVD++
    if (mtype.extend_type != 0 && mtype.extend_type != 1 && mtype.extend_type != 2 && mtype.extend_type != 15)
        fail("svdfeature_amd: extend_type " + std::to_string((int)mtype.extend_type) +
             " is not supported (0 / 1 base solver, 2 multi-level implicit feedback, 15 bilinear; the GBRT solvers 30 / 31 are another algorithm family)");
    if (device == -2) {   // host-only handle: config / model file / scheduler logic, no compute
        host_only_ = true;
        return;
    }
    // The first real HIP call initialises the ROCm runtime, which disturbs libc's rand() state (measured:
    // tools/check_hip_init_rand.cpp).  The reference seeds once in main() (svd_feature.cpp:293) and then relies on the
    // rand() stream for rand_init and for pairwise sampling, so runtime start-up is run on a scratch PRNG
    // state and the caller's state is put back exactly (initstate/setstate save and restore the position).
    RandStateGuard keep_callers_rand_stream;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
        fail("svdfeature_amd: no HIP device visible -- this engine has no CPU fallback");
    if (device >= 0) HIPCHECK(hipSetDevice(device));
    HIPCHECK(hipGetDevice(&device_));
    HIPCHECK(hipStreamCreate(&stream_));
    void *warm = nullptr;
    HIPCHECK(hipMalloc(&warm, 256));
    HIPCHECK(hipMemsetAsync(warm, 0, 256, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    HIPCHECK(hipFree(warm));
}


void Engine::save_pipe_free(SavePipe &sp) {
    for (int b = 0; b < 2; b++) {
        if (sp.pin[b]) (void)hipHostFree(sp.pin[b]);
        if (sp.ev[b]) (void)hipEventDestroy(sp.ev[b]);
        sp.pin[b] = nullptr; sp.ev[b] = nullptr;
    }
}
Engine::~Engine() {
    if (save_async_.th.joinable()) save_async_.th.join();   // a writer still streaming the snapshot: let it finish, the caller owns the file
    if (multi_ && !host_only_ && stream_) { try { flush(); } catch (...) {} }
    multi_.reset();
    if (!host_only_ && device_ >= 0) (void)hipSetDevice(device_);
    rank_prefetch_drop();
    for (Dataset *ds : datasets_) ds->owner = nullptr;
    datasets_.clear();
    if (getenv("SVDF_PROFILE"))
        fprintf(stderr, "[svdfeature_amd] host time: flush (schedule+upload+launch) %.3fs  model save/load %.3fs  instances %ld flushes %ld\n",
                ns_flush_ * 1e-9, ns_model_ * 1e-9, (long)n_instances_, (long)n_flushes_);
    if (!host_only_ && stream_) {
        try { flush(); } catch (...) {}
        if (worker_.joinable()) {
            { std::lock_guard<std::mutex> g(mu_); worker_stop_ = true; }
            cv_.notify_all();
            worker_.join();
        }
        (void)hipStreamSynchronize(stream_);
        if (pred_pin_) (void)hipHostFree(pred_pin_);
        save_pipe_free(save_pipe_);
        save_pipe_free(save_async_.pipe);
        if (save_async_.ready) (void)hipEventDestroy(save_async_.ready);
        if (save_async_.st) (void)hipStreamDestroy(save_async_.st);
        if (owns_stream_) (void)hipStreamDestroy(stream_);
    }
}

void Engine::need_device(const char *what) {
    if (host_only_) fail(std::string("svdfeature_amd: ") + what + " needs a GPU (handle was created host-only)");
    HIPCHECK(hipSetDevice(device_));
}
void Engine::init_trainer() {  // apex_svd_base.h:151-173, 499-503
    check(space_allocated_, "init_trainer: call init_model or load_model first");
    if (name_feat_user_ != "NULL") feat_user_.load(name_feat_user_.c_str());
    if (name_feat_item_ != "NULL") feat_item_.load(name_feat_item_.c_str());
    // the reference indexes children unchecked; validate once here instead
    for (unsigned c : feat_user_.index) check(c < (unsigned)mp_.num_user, "feature_user: child index exceed bound");
    for (unsigned c : feat_item_.index) check(c < (unsigned)mp_.num_item, "feature_item: child index exceed bound");
    check(tp_.reg_method >= 0 && tp_.reg_method <= 5, "unknown reg_method");
    check(tp_.reg_global == 0 || tp_.reg_global == 1 || tp_.reg_global == 4 || tp_.reg_global == 5, "unknown global decay method");
    if (mtype_.extend_type != 0 && !user_group())
        fail("svdfeature_amd: extend_type 1 / 2 / 15 (implicit-feedback solvers) need the user-group format (format_type = 1)");
    if (bilinear()) {
        // reg_feedback(lr, iid) (apex_svd_bilinear.h:103-119) runs once per item entry of every update: modes 2 / 3 decay the
        // row of W_bi, which only matters for a loaded model with non-zero W_bi (training itself never makes it non-zero)
        check(reg_bi_feedback_ >= 0 && reg_bi_feedback_ <= 5, "unknown bi feedback decay method");
        bool nz = false;
        for (float v : hbi_) nz = nz || v != 0.0f;
        if (nz && (reg_bi_feedback_ == 2 || reg_bi_feedback_ == 3))
            fail("svdfeature_amd: a loaded bilinear model with non-zero W_bi and reg_bi_feedback 2 / 3 is not supported");
    }
    if (imfb()) check(mp_.common_latent_space == 0, "svdfeature_amd: extend_type 2 with common_latent_space is not supported");
    imfb_depth_ = 0; iunit_open_ = false;   // the reference's `top` starts at 0 (apex_multi_imfb.h:46-48)
    trainer_ready_ = true;
    sample_counter_ = 0;   // :157
    if (host_only_) return;
    if (mp_.num_factor > max_supported_factor())
        fail("svdfeature_amd: num_factor > 1024 is not supported by the gfx950 kernels");
    if (multi_) {
        for (int d = 1; d < gpus_; d++) { Engine *e = rank_engine(d); HIPCHECK(hipSetDevice(e->device_)); e->init_trainer(); }
        HIPCHECK(hipSetDevice(device_));
    }
    if (!device_model_) upload_model();
    tracker_.resize(num_resources() + 1);
    d_ref_ui_.release(); d_ref_global_.release();   // ref_user/ref_item/ref_global start at 0 (:159-170)
    params_dirty_ = true;
}

void Engine::set_round(int nround) {  // apex_svd_base.h:470-478
    if (multi_ && tp_.decay_learning_rate != 0) { flush(); for (int d = 1; d < gpus_; d++) rank_engine(d)->set_round(nround); }
    if (tp_.decay_learning_rate != 0) {
        check(round_counter_ <= nround, "round counter restriction");
        if (round_counter_ < nround && trainer_ready_ && !host_only_) flush();
        while (round_counter_ < nround) {
            tp_.learning_rate *= tp_.decay_rate;
            round_counter_++;
        }
        params_dirty_ = true;
    }
}
void Engine::finish_round() {
    if (trainer_ready_ && !host_only_) flush();
}

const DevParams &Engine::params() {
    if (!params_dirty_) return dev_params_;
    need_device("building kernel parameters");
    DevParams &P = dev_params_;
    memset(&P, 0, sizeof(P));
    P.W = dW_.p; P.bias = dbias_.p; P.g_bias = dg_.p; P.svdpp_state = dstate_.p;
    P.pitch = pitch_; P.k = mp_.num_factor;
    P.user_off = user_off_; P.item_off = item_off_; P.fb_off = fb_off_;
    P.num_user = mp_.num_user; P.num_item = mp_.num_item; P.num_global = mp_.num_global; P.num_ufeedback = mp_.num_ufeedback;
    P.base_score = mp_.base_score;
    P.active_type = mtype_.active_type; P.no_user_bias = mp_.no_user_bias; P.user_nonnegative = mp_.user_nonnegative;
    P.user_group = user_group() ? 1 : 0;
    P.store_mode = store_mode_;
    P.load_mode = load_mode_ == 2 ? (pitch_ >= 64 ? 1 : 0) : load_mode_;   // auto: nontemporal row gathers for rows of two cache lines or more
    P.basic_i8 = basic_i8_;
    P.small_blocks = small_blocks_;
    P.svdpp_helpers = svdpp_helpers_;
    P.fewrow_i16 = fewrow_i16_;
    P.xcd_remap = xcd_remap_;
    P.imfb_disable = imfb_disable_;
    P.imfb_deep = imfb_deep_ ? 1 : 0;
    P.fewrow_fast = fewrow_fast_ ? 1 : 0;
    P.hot_reduce = hot_reduce_; P.relax_global = relax_global_ ? 1 : 0; P.relax_feedback = relax_feedback_ ? 1 : 0;
    if (device_model_ && g_stride_ != wanted_g_stride() && mp_.num_global > 0) {   // relax_global switched after the upload: re-lay out
        std::vector<float> g((size_t)mp_.num_global);
        download_globals(g.data());
        HIPCHECK(hipStreamSynchronize(stream_));
        std::swap(g, hg_);
        upload_globals(wanted_g_stride());
        HIPCHECK(hipStreamSynchronize(stream_));
        std::swap(g, hg_);
    }
    P.g_stride = g_stride_; P.g_bias = dg_.p; P.relax_user_from = relax_user_from_; P.relax_item_from = relax_item_from_;
    P.lr = tp_.learning_rate; P.wd_user = tp_.wd_user; P.wd_item = tp_.wd_item;
    P.wd_user_bias = tp_.wd_user_bias; P.wd_item_bias = tp_.wd_item_bias; P.wd_global = tp_.wd_global;
    P.reg_method = tp_.reg_method; P.reg_global = tp_.reg_global; P.num_regfree_global = tp_.num_regfree_global;
    check(tp_.reg_method >= 0 && tp_.reg_method <= 5, "unknown reg_method");
    check(tp_.reg_global == 0 || tp_.reg_global == 1 || tp_.reg_global == 4 || tp_.reg_global == 5, "unknown global decay method");
    if (tp_.reg_method >= 4 && !d_ref_ui_.p) {        // one word per W_uiset row: with common_latent_space users and items
        d_ref_ui_.reserve((size_t)n_uiset_ + 1);        // share rows and therefore refs, like ref_item = ref_user (:167)
        HIPCHECK(hipMemsetAsync(d_ref_ui_.p, 0, ((size_t)n_uiset_ + 1) * sizeof(unsigned), stream_));
    }
    if (tp_.reg_global >= 4 && !d_ref_global_.p) {
        d_ref_global_.reserve((size_t)mp_.num_global + 1);
        HIPCHECK(hipMemsetAsync(d_ref_global_.p, 0, ((size_t)mp_.num_global + 1) * sizeof(unsigned), stream_));
    }
    P.ref_ui = d_ref_ui_.p; P.ref_global = d_ref_global_.p;
    P.scale_lr_ufeedback = tp_.scale_lr_ufeedback; P.wd_ufeedback = tp_.wd_ufeedback; P.wd_ufeedback_bias = tp_.wd_ufeedback_bias;
    auto up_ranges = [&](const ParamSet &ps, DevBuf<unsigned> &db, DevBuf<float> &dw, DevRanges &out, unsigned max_id) {
        out.n = 0; out.bound = nullptr; out.wd = nullptr;
        if (ps.bound.empty()) return;
        // ParameterSet::get_wd asserts idx < bound.size() for every id it is asked about
        if (max_id > 0) check(ps.bound.back() >= max_id - 1, "bound set err");
        db.upload(ps.bound.data(), ps.bound.size(), stream_);
        dw.upload(ps.wd.data(), ps.bound.size(), stream_);
        out.n = (int)ps.bound.size(); out.bound = db.p; out.wd = dw.p;
    };
    up_ranges(u_param_, d_ubound_, d_uwd_, P.u_rng, (unsigned)mp_.num_user);
    up_ranges(i_param_, d_ibound_, d_iwd_, P.i_rng, (unsigned)mp_.num_item);
    up_ranges(g_param_, d_gbound_, d_gwd_, P.g_rng, (unsigned)mp_.num_global);
    auto up_table = [&](const SideTable &t, DevBuf<unsigned> &dp, DevBuf<unsigned> &di, DevBuf<float> &dv, DevSideTable &out) {
        out.num_row = 0; out.row_ptr = nullptr; out.index = nullptr; out.value = nullptr;
        if (t.num_row() == 0) return;
        dp.upload(t.row_ptr.data(), t.row_ptr.size(), stream_);
        di.upload(t.index.data(), t.index.size(), stream_);
        dv.upload(t.value.data(), t.value.size(), stream_);
        out.num_row = t.num_row(); out.row_ptr = dp.p; out.index = di.p; out.value = dv.p;
    };
    up_table(feat_user_, d_fu_ptr_, d_fu_idx_, d_fu_val_, P.feat_user);
    up_table(feat_item_, d_fi_ptr_, d_fi_idx_, d_fi_val_, P.feat_item);
    HIPCHECK(hipStreamSynchronize(stream_));
    params_dirty_ = false;
    launch_version_++;
    return dev_params_;
}

// =============================================================================== staging
void Engine::check_row(int ng, int nu, int ni, const unsigned *index) {  // asserts of apex_svd_base.h:320,327,343,360
    for (int j = 0; j < ng; j++) check(index[j] < (unsigned)mp_.num_global, "global feature index exceed setting");
    for (int j = 0; j < nu; j++) check(index[ng + j] < (unsigned)mp_.num_user, "user feature index exceed bound");
    for (int j = 0; j < ni; j++) check(index[ng + nu + j] < (unsigned)mp_.num_item, "item feature index exceed bound");
}
void Engine::stage_rows(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    stage_rows_into(staged_, num_row, row_label, row_ptr, feat_index, feat_value);
}
// All rows are validated before the first one is appended: a failing row (error mode 1 throws) leaves `dst` untouched.
void Engine::stage_rows_into(HostCSR &staged_, int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index,
                             const float *feat_value) {
    for (int r = 0; r < num_row; r++) {
        const int p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
        check(p0 <= p1 && p1 <= p2 && p2 <= p3, "CSR row_ptr must be non-decreasing");
        check_row(p1 - p0, p2 - p1, p3 - p2, feat_index + p0);
    }
    if (num_row > 0)
        check((long)staged_.row_ptr.back() + (long)(row_ptr[3 * num_row] - row_ptr[0]) < 2147483647L, "svdfeature_amd: more than 2^31-1 feature entries in one window");
    for (int r = 0; r < num_row; r++) {
        const int p0 = row_ptr[3 * r], p1 = row_ptr[3 * r + 1], p2 = row_ptr[3 * r + 2], p3 = row_ptr[3 * r + 3];
        const int base = staged_.row_ptr.back() - p0;
        staged_.row_label.push_back(row_label[r]);
        staged_.row_ptr.push_back(p1 + base);
        staged_.row_ptr.push_back(p2 + base);
        staged_.row_ptr.push_back(p3 + base);
        staged_.feat_index.insert(staged_.feat_index.end(), feat_index + p0, feat_index + p3);
        staged_.feat_value.insert(staged_.feat_value.end(), feat_value + p0, feat_value + p3);
    }
}
bool Engine::basic_fast_path_allowed() const {
    return !relaxed() && !lazy_decay() && mp_.num_factor <= max_fast_path_factor() && (!user_group() || rows_as_instances_) && mp_.common_latent_space == 0 && feat_user_.num_row() == 0 && feat_item_.num_row() == 0;
}

void Engine::update_csr(float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    if (!user_group() && trainer_ready_ && !host_only_) {   // one instance per call: append in place
        check(ng >= 0 && nu >= 0 && ni >= 0, "negative feature count");
        check_row(ng, nu, ni, index);
        const int nv = ng + nu + ni;
        const int b = staged_.row_ptr.back();
        staged_.row_label.push_back(label);
        staged_.row_ptr.push_back(b + ng);
        staged_.row_ptr.push_back(b + ng + nu);
        staged_.row_ptr.push_back(b + nv);
        for (int j = 0; j < nv; j++) { staged_.feat_index.push_back(index[j]); staged_.feat_value.push_back(value[j]); }
        if (staged_.num_row() >= stage_window_) submit_window();
        return;
    }
    const int ptr[4] = {0, ng, ng + nu, ng + nu + ni};
    update_csr_batch(1, &label, ptr, index, value);
}
void Engine::update_csr_batch(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    // hot when the reference's CLI feeds one instance per call: no HIP API call and no timer in here
    check(trainer_ready_, "update: init_trainer has not been called");
    if (host_only_) need_device("update");
    if (imfb()) {   // update(Elem) of the multi-level solver: the rows see whatever levels are open (a MIDDLE block without tags)
        update_block_imfb(0, TAG_MIDDLE, nullptr, nullptr, num_row, row_label, row_ptr, feat_index, feat_value);
        return;
    }
    if (user_group()) {
        // SVDPPFeature inherits update(Elem) (apex_svd_base.h:464-466): rows run against the current
        // implicit-feedback state without prepare/scatter
        const int h = (int)staged_.num_row();
        stage_rows(num_row, row_label, row_ptr, feat_index, feat_value);
        if (!staged_units_.empty() && !(staged_units_.back().flags & (UNIT_START | UNIT_END)) && staged_units_.back().row_end == h)
            staged_units_.back().row_end = h + num_row;
        else
            staged_units_.push_back(HostUnit{0, 0, h, h + num_row, UNIT_LOAD | UNIT_SAVE});
    } else {
        stage_rows(num_row, row_label, row_ptr, feat_index, feat_value);
    }
    if (staged_.num_row() >= stage_window_) submit_window();
}

void Engine::update_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                          const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    check(trainer_ready_, "update: init_trainer has not been called");
    if (host_only_) need_device("update");
    check(user_group(), "not implemented");   // SVDFeature has no update(SVDPlusBlock) (apex_svd.h:97)
    check(!multi_ || in_multi_scope(), "svdfeature_amd: amd:gpus > 1 trains user-group data from resident data sets (svdf_dataset_from_blocks / svdf_dataset_from_buffer_file), not block by block");
    for (int j = 0; j < nfb; j++) check(ifb[j] < (unsigned)mp_.num_ufeedback, "ufeedback id exceed bound");
    if (imfb()) { update_block_imfb(nfb, tag, ifb, vfb, num_row, row_label, row_ptr, feat_index, feat_value); return; }
    const int h = (int)staged_.num_row();
    const bool starts = (tag == TAG_DEFAULT || tag == TAG_START);
    const bool ends = (tag == TAG_DEFAULT || tag == TAG_END);
    const bool continues = !starts && !staged_units_.empty() && unit_open_ && !unit_open_on_device_ && staged_units_.back().row_end == h &&
                           !(staged_units_.back().flags & UNIT_END);
    if (continues && tag == TAG_END && (staged_units_.back().flags & UNIT_START)) {
        // START and END merged in one flush: the scatter list must equal the prepare list (checked before anything is staged)
        const HostUnit &su = staged_units_.back();
        bool same = (su.fb_end - su.fb_begin) == nfb;
        for (int j = 0; same && j < nfb; j++)
            same = staged_fb_index_[(size_t)su.fb_begin + j] == ifb[j] && staged_fb_value_[(size_t)su.fb_begin + j] == vfb[j];
        check(same, "svdfeature_amd: START and END blocks of one user must carry the same feedback list");
    }
    stage_rows(num_row, row_label, row_ptr, feat_index, feat_value);
    auto push_fb = [&](int &b, int &e) {
        b = (int)staged_fb_index_.size();
        staged_fb_index_.insert(staged_fb_index_.end(), ifb, ifb + nfb);
        staged_fb_value_.insert(staged_fb_value_.end(), vfb, vfb + nfb);
        e = (int)staged_fb_index_.size();
    };
    HostUnit *u = nullptr;
    // a MIDDLE/END block continues the unit staged just before it, if that unit is still open here
    if (continues) {
        u = &staged_units_.back();
        u->row_end = h + num_row;
    } else {
        HostUnit nu{0, 0, h, h + num_row, 0};
        if (starts) { push_fb(nu.fb_begin, nu.fb_end); nu.flags |= UNIT_START; }
        else { nu.flags |= UNIT_LOAD; unit_open_on_device_ = false; }   // the open user continues in this window
        staged_units_.push_back(nu);
        u = &staged_units_.back();
    }
    if (ends) {
        // update_ufeedback scatters through the END block's own feedback list (apex_svd_base.h:579-581).
        // It is kept next to the prepare list: [fb_begin,fb_end) prepare, the scatter list is appended and
        // recorded by re-pointing fb_* when the unit did not start here.
        if (!(u->flags & UNIT_START)) push_fb(u->fb_begin, u->fb_end);
        u->flags |= UNIT_END;
        unit_open_ = false;
        unit_open_on_device_ = false;
    } else {
        unit_open_ = true;
    }
    if (staged_.num_row() >= stage_window_) flush();
}

// ---- extend_type 2: SVDPPMultiIMFB::update (apex_multi_imfb.h:173-192) staged as blocks; a unit is a run of blocks from an
// empty stack (or the start of the window) until the stack is empty again (or the window ends)
void Engine::update_block_imfb(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                               const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    check(tag == TAG_DEFAULT || tag == TAG_START || tag == TAG_MIDDLE || tag == TAG_END, "unknown extend_tag");
    const bool starts = (tag == TAG_DEFAULT || tag == TAG_START), ends = (tag == TAG_DEFAULT || tag == TAG_END);
    if (starts) {
        check(imfb_depth_ < IMFB_DEPTH_MAX, "svdfeature_amd: more than 16 nested implicit-feedback levels (the reference's stack is an unbounded std::vector, apex_multi_imfb.h:41-58; this engine holds 16)");
        if (imfb_depth_ >= IMFB_DEPTH && !imfb_deep_) { imfb_deep_ = true; params_dirty_ = true; }
    }
    if (ends && !starts) check(imfb_depth_ > 0, "start tag,end tag error in implicit feedback");   // :183
    const int h = (int)staged_.num_row();
    stage_rows(num_row, row_label, row_ptr, feat_index, feat_value);
    DevBlk k;
    memset(&k, 0, sizeof(k));
    k.row_begin = h; k.row_end = h + num_row; k.tag = tag;
    if (starts || ends) {
        const int b = (int)staged_fb_index_.size();
        staged_fb_index_.insert(staged_fb_index_.end(), ifb, ifb + nfb);
        staged_fb_value_.insert(staged_fb_value_.end(), vfb, vfb + nfb);
        if (starts) { k.fb_begin = b; k.fb_end = b + nfb; }
        if (ends) { k.sc_begin = b; k.sc_end = b + nfb; }
    }
    const int bi = (int)staged_blks_.size();
    staged_blks_.push_back(k);
    if (!iunit_open_) {
        staged_iunits_.push_back(DevUnit{bi, bi + 1, h, h + num_row, imfb_depth_ > 0 ? UNIT_LOAD : 0});   // open levels live in the device state slot
        iunit_open_ = true;
    } else {
        staged_iunits_.back().fb_end = bi + 1;
        staged_iunits_.back().row_end = h + num_row;
    }
    if (starts) imfb_depth_++;
    if (ends) imfb_depth_--;
    if (imfb_depth_ == 0) iunit_open_ = false;
    if (staged_.num_row() >= stage_window_) flush();
}
void Engine::schedule_iunits(int base, Schedule &sched) {
    const long nu = (long)staged_iunits_.size();
    tracker_.resize(num_resources() + 1);
    const size_t state_res = num_resources();
    std::vector<int> levels((size_t)nu);
    int *last = tracker_.last.data();
    const unsigned *idx = staged_.feat_index.data();
    check(!relaxed(), "svdfeature_amd: the relaxed modes are not available for extend_type 2");
    for (long t = 0; t < nu; t++) {
        const DevUnit &u = staged_iunits_[(size_t)t];
        int lvl = base;
        for (int r = u.row_begin; r < u.row_end; r++) {
            const int *p = &staged_.row_ptr[(size_t)3 * r];
            lvl = level_of_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], lvl);
        }
        for (int b = u.fb_begin; b < u.fb_end; b++) {
            const DevBlk &k = staged_blks_[(size_t)b];
            for (int j = k.fb_begin; j < k.fb_end; j++) lvl = std::max(lvl, last[fb_off_ + staged_fb_index_[(size_t)j]]);
            for (int j = k.sc_begin; j < k.sc_end; j++) lvl = std::max(lvl, last[fb_off_ + staged_fb_index_[(size_t)j]]);
        }
        if (u.flags & (UNIT_LOAD | UNIT_SAVE)) lvl = std::max(lvl, last[state_res]);
        lvl += 1;
        for (int r = u.row_begin; r < u.row_end; r++) {
            const int *p = &staged_.row_ptr[(size_t)3 * r];
            touch_row(idx + p[0], p[1] - p[0], idx + p[1], p[2] - p[1], idx + p[2], p[3] - p[2], lvl);
        }
        for (int b = u.fb_begin; b < u.fb_end; b++) {
            const DevBlk &k = staged_blks_[(size_t)b];
            for (int j = k.fb_begin; j < k.fb_end; j++) last[fb_off_ + staged_fb_index_[(size_t)j]] = lvl;
            for (int j = k.sc_begin; j < k.sc_end; j++) last[fb_off_ + staged_fb_index_[(size_t)j]] = lvl;
        }
        if (u.flags & (UNIT_LOAD | UNIT_SAVE)) last[state_res] = lvl;
        levels[(size_t)t] = lvl;
    }
    build_schedule(levels, base, sched);
}
void Engine::upload_iunits(UnitDev &d, const Schedule &sched) {
    d.label.upload(staged_.row_label.data(), staged_.row_label.size(), stream_);
    d.ptr.upload(staged_.row_ptr.data(), staged_.row_ptr.size(), stream_);
    d.index.upload(staged_.feat_index.data(), staged_.feat_index.size(), stream_);
    d.value.upload(staged_.feat_value.data(), staged_.feat_value.size(), stream_);
    d.fbidx.upload(staged_fb_index_.data(), staged_fb_index_.size(), stream_);
    d.fbval.upload(staged_fb_value_.data(), staged_fb_value_.size(), stream_);
    d.units.upload(staged_iunits_.data(), staged_iunits_.size(), stream_);
    d.blks.upload(staged_blks_.data(), staged_blks_.size(), stream_);
    d.order.upload(sched.order.data(), sched.order.size(), stream_);
    d.unit_values = false; d.has_fresh = false;
    HIPCHECK(hipStreamSynchronize(stream_));
}
void Engine::drop_staged_units() {
    staged_.clear(); staged_units_.clear(); staged_fb_index_.clear(); staged_fb_value_.clear();
    staged_blks_.clear(); staged_iunits_.clear();
}
void Engine::flush_iunits() {
    if (staged_iunits_.empty()) { drop_staged_units(); return; }
    need_device("update");
    const DevParams &P = params();
    const int base = tracker_.base;
    if (iunit_open_) staged_iunits_.back().flags |= UNIT_SAVE;   // open levels wait in the device state slot for the next window
    Schedule sched;
    schedule_iunits(base, sched);
    tracker_.base = base + (int)sched.num_levels();
    const long n = staged_.num_row();
    UnitDev &d = w_unitdev_;
    upload_iunits(d, sched);
    const DevCSR D = d.csr();
    for (size_t l = 0; l < sched.num_levels(); l++) {
        launch_imfb(P, D, d.units.p, d.blks.p, d.fbidx.p, d.fbval.p, d.order.p, sched.level_ptr[l], sched.level_ptr[l + 1], sample_counter_, nullptr, stream_);
        n_launches_++;
    }
    HIPCHECK(hipGetLastError());
    n_batches_ += (int64_t)sched.num_levels();
    n_instances_ += n;
    sample_counter_ += (unsigned)n;
    n_flushes_++;
    iunit_open_ = false;   // what is still open continues as a new unit with UNIT_LOAD
    drop_staged_units();
}
// =============================================================================== flush
void Engine::flush() {
    if (host_only_ || !trainer_ready_ || in_multi_scope()) return;
    wait_worker();   // a window handed to the background thread earlier must land first
    ScopedNs timer(ns_flush_);
    if (imfb()) flush_iunits();
    else if (user_group()) flush_units();
    else if (multi_) {
        MultiScope scope;
        multi_flush(staged_);
    } else flush_csr(staged_);
}

// ---- background window flush -----------------------------------------------------------------------
// While the caller keeps staging instances of window w+1 (the reference's CLI hands them over one virtual
// call at a time), a worker thread schedules, uploads and launches window w.  Windows are processed
// strictly in order on one HIP stream; every synchronisation point goes through flush(), which first
// waits for the worker, so the observable semantics are those of the synchronous path.
void Engine::wait_worker() {
    if (!worker_.joinable()) return;
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !worker_busy_; });
    if (!worker_error_.empty()) {
        std::string m;
        m.swap(worker_error_);
        lk.unlock();
        fail(m);
    }
}
void Engine::worker_main() {
    (void)hipSetDevice(device_);
    std::unique_lock<std::mutex> lk(mu_);
    for (;;) {
        cv_.wait(lk, [&] { return worker_busy_ || worker_stop_; });
        if (worker_stop_) return;
        lk.unlock();
        try {
            ScopedNs timer(ns_flush_);
            if (multi_) { MultiScope scope; multi_flush(job_); }
            else flush_csr(job_);
        } catch (const std::exception &ex) {
            std::lock_guard<std::mutex> g(mu_);
            worker_error_ = ex.what();
        }
        job_.clear();
        lk.lock();
        worker_busy_ = false;
        cv_.notify_all();
    }
}
void Engine::submit_window() {
    if (!async_flush_ || user_group()) { flush(); return; }
    if (!worker_.joinable()) worker_ = std::thread([this] { worker_main(); });
    wait_worker();
    {
        std::lock_guard<std::mutex> g(mu_);
        std::swap(job_, staged_);   // job_ was cleared by the worker; staged_ starts empty again
        worker_busy_ = true;
    }
    cv_.notify_all();
}
// =============================================================================== predict
float Engine::predict_csr(float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    const int ptr[4] = {0, ng, ng + nu, ng + nu + ni};
    float out = 0.0f;
    predict_csr_batch(1, &label, ptr, index, value, &out);
    return out;
}
void Engine::predict_csr_batch(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index,
                               const float *feat_value, float *out) {
    check(trainer_ready_, "predict: init_trainer has not been called");
    need_device("predict");
    if (num_row <= 0) return;
    if (user_group()) {  // SVDFeature::predict(Elem) against the current implicit-feedback state
        predict_block(0, TAG_MIDDLE, nullptr, nullptr, num_row, row_label, row_ptr, feat_index, feat_value, out);
        return;
    }
    flush();
    if (multi_) { multi_predict(num_row, row_label, row_ptr, feat_index, feat_value, out); return; }
    predict_csr_batch_local(num_row, row_label, row_ptr, feat_index, feat_value, out);
}
void Engine::predict_csr_batch_local(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index,
                                     const float *feat_value, float *out) {
    need_device("predict");
    if (num_row <= 0) return;
    flush();
    const DevParams &P = params();
    {   // a few rows (the per-instance predict(Elem) of ISVDTrainer): no copies at all -- the rows are written into ONE pinned, device-mapped
        // buffer, the kernel reads them and writes the predictions through it; a call is one launch + one stream synchronisation
        const long nv = (long)row_ptr[3 * num_row] - (long)row_ptr[0];
        const size_t words = (size_t)num_row * 2 + (size_t)3 * num_row + 1 + (size_t)2 * std::max<long>(nv, 0);
        if (num_row <= 256 && nv >= 0 && words <= PRED_PIN_WORDS) {
            for (int r = 0; r < num_row; r++) {
                const int *p = row_ptr + 3 * r;
                check(p[0] <= p[1] && p[1] <= p[2] && p[2] <= p[3], "CSR row_ptr must be non-decreasing");
                check_row(p[1] - p[0], p[2] - p[1], p[3] - p[2], feat_index + p[0]);
            }
            if (!pred_pin_) HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&pred_pin_), PRED_PIN_WORDS * sizeof(unsigned), hipHostMallocMapped));
            float *l = reinterpret_cast<float *>(pred_pin_), *o = l + num_row;
            int *ptr = reinterpret_cast<int *>(o + num_row);
            unsigned *idx = reinterpret_cast<unsigned *>(ptr + 3 * num_row + 1);
            float *val = reinterpret_cast<float *>(idx + nv);
            memcpy(l, row_label, (size_t)num_row * sizeof(float));
            const int p0 = row_ptr[0];
            for (int j = 0; j <= 3 * num_row; j++) ptr[j] = row_ptr[j] - p0;
            if (nv > 0) { memcpy(idx, feat_index + p0, (size_t)nv * sizeof(unsigned)); memcpy(val, feat_value + p0, (size_t)nv * sizeof(float)); }
            DevCSR D{l, ptr, idx, val};
            launch_predict(P, D, num_row, o, stream_);
            n_launches_++;
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(stream_));
            memcpy(out, o, (size_t)num_row * sizeof(float));
            return;
        }
    }
    HostCSR tmp;   // prediction rows never enter the training stage
    tmp.row_label.reserve((size_t)num_row);
    stage_rows_into(tmp, num_row, row_label, row_ptr, feat_index, feat_value);
    w_label_.upload(tmp.row_label.data(), tmp.row_label.size(), stream_);
    w_ptr_.upload(tmp.row_ptr.data(), tmp.row_ptr.size(), stream_);
    w_index_.upload(tmp.feat_index.data(), tmp.feat_index.size(), stream_);
    w_value_.upload(tmp.feat_value.data(), tmp.feat_value.size(), stream_);
    w_out_.reserve((size_t)num_row);
    DevCSR D{w_label_.p, w_ptr_.p, w_index_.p, w_value_.p};
    launch_predict(P, D, num_row, w_out_.p, stream_);
    n_launches_++;
    HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)num_row * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
}
void Engine::predict_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                           const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out) {
    check(trainer_ready_, "predict: init_trainer has not been called");
    need_device("predict");
    check(user_group(), "not implemented");
    for (int j = 0; j < nfb; j++) check(ifb[j] < (unsigned)mp_.num_ufeedback, "ufeedback id exceed bound");
    flush();
    if (multi_ && !in_multi_scope()) {
        // the block is scored by the owner of its user (the first user entry of its first row; a block without one -- the MIDDLE / END
        // pieces of a span carry their rows' user too -- goes where the previous block went)
        for (int r = 0; r < num_row; r++) {   // the owner is read from the rows before any rank validates them
            const int *q = row_ptr + 3 * r;
            check(q[0] >= 0 && q[0] <= q[1] && q[1] <= q[2] && q[2] <= q[3], "CSR row_ptr must be non-decreasing");
        }
        if (num_row > 0 && row_ptr[2] > row_ptr[1]) multi_predict_rank_ = (int)(feat_index[row_ptr[1]] % (unsigned)gpus_);
        Engine *e = rank_engine(multi_predict_rank_);
        if (e != this) {
            HIPCHECK(hipSetDevice(e->device_));
            e->predict_block(nfb, tag, ifb, vfb, num_row, row_label, row_ptr, feat_index, feat_value, out);
            HIPCHECK(hipSetDevice(device_));
            return;
        }
    }
    const DevParams &P = params();
    if (imfb()) {   // SVDPPMultiIMFB::predict (apex_multi_imfb.h:193-207): push on DEFAULT / START, score, pop (no scatter) on DEFAULT / END
        const bool st = (tag == TAG_DEFAULT || tag == TAG_START), en = (tag == TAG_DEFAULT || tag == TAG_END);
        if (st) {
            check(imfb_depth_ < IMFB_DEPTH_MAX, "svdfeature_amd: more than 16 nested implicit-feedback levels (the reference's stack is an unbounded std::vector, apex_multi_imfb.h:41-58; this engine holds 16)");
            if (imfb_depth_ >= IMFB_DEPTH && !imfb_deep_) { imfb_deep_ = true; params_dirty_ = true; }
        }
        if (en && !st) check(imfb_depth_ > 0, "start tag,end tag error in implicit feedback");
        HostCSR rows;
        stage_rows_into(rows, num_row, row_label, row_ptr, feat_index, feat_value);
        DevBlk k;
        memset(&k, 0, sizeof(k));
        k.row_end = num_row; k.tag = tag;
        if (st) k.fb_end = nfb;
        if (en) k.sc_end = nfb;
        const int after = imfb_depth_ + (st ? 1 : 0) - (en ? 1 : 0);
        DevUnit u{0, 1, 0, num_row, (imfb_depth_ > 0 ? UNIT_LOAD : 0) | (after > 0 ? UNIT_SAVE : 0)};
        w_label_.upload(rows.row_label.data(), rows.row_label.size(), stream_);
        w_ptr_.upload(rows.row_ptr.data(), rows.row_ptr.size(), stream_);
        w_index_.upload(rows.feat_index.data(), rows.feat_index.size(), stream_);
        w_value_.upload(rows.feat_value.data(), rows.feat_value.size(), stream_);
        w_fbidx_.upload(ifb, (size_t)nfb, stream_);
        w_fbval_.upload(vfb, (size_t)nfb, stream_);
        w_units_.upload(&u, 1, stream_);
        w_unitdev_.blks.upload(&k, 1, stream_);
        w_out_.reserve((size_t)std::max(num_row, 1));
        DevCSR D{w_label_.p, w_ptr_.p, w_index_.p, w_value_.p};
        launch_imfb(P, D, w_units_.p, w_unitdev_.blks.p, w_fbidx_.p, w_fbval_.p, nullptr, 0, 1, sample_counter_, w_out_.p, stream_);
        n_launches_++;
        if (num_row > 0) HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)num_row * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
        imfb_depth_ = after;
        return;
    }
    const bool starts = (tag == TAG_DEFAULT || tag == TAG_START);
    DevUnit u{0, starts ? nfb : 0, 0, num_row, (starts ? UNIT_START : UNIT_LOAD) | UNIT_SAVE};
    {   // the usual block (a user's rows and feedback list: a few hundred entries): no copies -- everything the kernel reads, and the
        // predictions it writes, live in ONE pinned, device-mapped buffer (see predict_csr_batch_local)
        const long nv = num_row > 0 ? (long)row_ptr[3 * num_row] - (long)row_ptr[0] : 0;
        const long nf = starts ? nfb : 0;
        const size_t uw = (sizeof(DevUnit) + 3) / 4;
        const size_t words = (size_t)2 * num_row + (size_t)3 * num_row + 1 + (size_t)2 * nv + (size_t)2 * nf + uw + 8;
        if (nv >= 0 && words <= PRED_PIN_WORDS) {
            for (int r = 0; r < num_row; r++) {
                const int *p = row_ptr + 3 * r;
                check(p[0] <= p[1] && p[1] <= p[2] && p[2] <= p[3], "CSR row_ptr must be non-decreasing");
                check_row(p[1] - p[0], p[2] - p[1], p[3] - p[2], feat_index + p[0]);
            }
            if (!pred_pin_) HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&pred_pin_), PRED_PIN_WORDS * sizeof(unsigned), hipHostMallocMapped));
            DevUnit *du = reinterpret_cast<DevUnit *>(pred_pin_);
            float *l = reinterpret_cast<float *>(pred_pin_ + uw), *o = l + num_row;
            int *ptr = reinterpret_cast<int *>(o + num_row);
            unsigned *idx = reinterpret_cast<unsigned *>(ptr + 3 * num_row + 1);
            float *val = reinterpret_cast<float *>(idx + nv);
            unsigned *fi = reinterpret_cast<unsigned *>(val + nv);
            float *fv = reinterpret_cast<float *>(fi + nf);
            *du = u;
            if (num_row > 0) memcpy(l, row_label, (size_t)num_row * sizeof(float));
            const int p0 = num_row > 0 ? row_ptr[0] : 0;
            for (int j = 0; j <= 3 * num_row; j++) ptr[j] = num_row > 0 ? row_ptr[j] - p0 : 0;
            if (nv > 0) { memcpy(idx, feat_index + p0, (size_t)nv * sizeof(unsigned)); memcpy(val, feat_value + p0, (size_t)nv * sizeof(float)); }
            if (nf > 0) { memcpy(fi, ifb, (size_t)nf * sizeof(unsigned)); memcpy(fv, vfb, (size_t)nf * sizeof(float)); }
            DevCSR D{l, ptr, idx, val};
            launch_svdpp_predict(P, D, du, fi, fv, 1, o, stream_);
            n_launches_++;
            HIPCHECK(hipGetLastError());
            HIPCHECK(hipStreamSynchronize(stream_));
            if (num_row > 0) memcpy(out, o, (size_t)num_row * sizeof(float));
            return;
        }
    }
    HostCSR tmp;   // prediction rows never enter the training stage
    stage_rows_into(tmp, num_row, row_label, row_ptr, feat_index, feat_value);
    w_label_.upload(tmp.row_label.data(), tmp.row_label.size(), stream_);
    w_ptr_.upload(tmp.row_ptr.data(), tmp.row_ptr.size(), stream_);
    w_index_.upload(tmp.feat_index.data(), tmp.feat_index.size(), stream_);
    w_value_.upload(tmp.feat_value.data(), tmp.feat_value.size(), stream_);
    w_fbidx_.upload(ifb, starts ? (size_t)nfb : 0, stream_);
    w_fbval_.upload(vfb, starts ? (size_t)nfb : 0, stream_);
    w_units_.upload(&u, 1, stream_);
    w_out_.reserve((size_t)std::max(num_row, 1));
    DevCSR D{w_label_.p, w_ptr_.p, w_index_.p, w_value_.p};
    launch_svdpp_predict(P, D, w_units_.p, w_fbidx_.p, w_fbval_.p, 1, w_out_.p, stream_);
    n_launches_++;
    if (num_row > 0) HIPCHECK(hipMemcpyAsync(out, w_out_.p, (size_t)num_row * sizeof(float), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
}

// =============================================================================== datasets
void Engine::train_dataset(Dataset *ds) {
    check(ds && ds->owner == this, "train_dataset: dataset belongs to another trainer");
    if (ds->kind == 6) { check(multi_ != nullptr && !in_multi_scope(), "train_dataset: the data set belongs to an amd:gpus handle"); multi_train_dataset(ds); return; }
    check(ds->sched_signature == schedule_signature(),
          "train_dataset: the dataset was scheduled under another configuration (relaxed-id keys, side tables, lazy decay or kernel-routing knobs changed since it was built); build it again");
    flush();
    if (ds->kind == 8) {   // one GPU, amd:step = minibatch: the pass as a sequence of windows, each trained and applied in place (svdf_wunit.cpp)
        wseq_train(ds);
        HIPCHECK(hipGetLastError());
        n_batches_ += (int64_t)ds->wchild.size();
        n_instances_ += ds->num_row;
        sample_counter_ += (unsigned)ds->num_row;
        return;
    }
    if (ds->kind == 11) {   // user-run units of rank pairs (svdf_punit.cpp)
        punit_train(ds);
        n_instances_ += ds->num_row;
        sample_counter_ += (unsigned)ds->num_row;
        return;
    }
    if (ds->kind == 10) {   // runs of an item's consecutive ratings (svdf_runs.cpp)
        check(runs_config_ok(), "train_dataset: the data set was built for the contract configuration's runs kernel (svdf_runs.cpp); the configuration changed since");
        runs_train(ds);
        n_instances_ += ds->num_row;
        sample_counter_ += (unsigned)ds->num_row;
        return;
    }
    if (ds->kind == 9) {   // ratings with hot rows: cold ratings level by level, runs of a hot row's ratings as walker units (svdf_pivot.cpp)
        check(pivot_config_ok(), "train_dataset: the data set was built for the symmetric basicMF configuration (svdf_pivot.cpp); the configuration changed since");
        pivot_train(ds);
        n_instances_ += ds->num_row;
        sample_counter_ += (unsigned)ds->num_row;
        return;
    }
    if (ds->kind == 5) {   // the trainer-owned contribution scratch is sized BEFORE anything is issued (and never inside a stream capture)
        d_contrib_.reserve((size_t)ds->win_slots * (size_t)pitch_);
        d_cbias_.reserve((size_t)ds->win_slots);
        window_trained_ = ds;
    }
    const DevParams &P = params();
    const Schedule &sc = ds->sched;
    check(!lazy_decay() || ds->kind == 1 || ds->kind == 4 || (ds->kind == 3 && ds->num_simple_units == 0),
          "train_dataset: the dataset was scheduled before lazy decay (reg_method/reg_global >= 4) was selected");
    if ((ds->kind == 2 || ds->kind == 0) && chain_width_ > 0 && !ds->d_level_ptr_ok && sc.num_levels() >= 4) {   // the level boundaries for chained launches: once, outside any capture
        ds->d_level_ptr.upload(sc.level_ptr.data(), sc.level_ptr.size(), stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
        ds->d_level_ptr_ok = true;
    }
    auto issue = [&]() {
        if (ds->kind == 0) {
            BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
            // runs of NARROW levels (the tail of a pass over Zipf-popular items) through one launch per run, everything else level by level
            const size_t L = sc.num_levels();
            const long cw = std::min<long>(chain_width_, 128);   // one round of the chained workgroup: 16 waves x 8 instances
            int64_t chained = 0;
            const bool can_chain = cw > 0 && ds->d_level_ptr_ok && launch_basicmf_chain(P, S, nullptr, 0, 0, stream_);
            for (size_t l = 0; l < L;) {
                size_t e = l;
                if (can_chain) while (e < L && sc.level_ptr[e + 1] - sc.level_ptr[e] <= cw) e++;
                if (e >= l + 4) {
                    (void)launch_basicmf_chain(P, S, ds->d_level_ptr.p, (long)l, (long)e, stream_);
                    chained += (int64_t)(e - l);
                    l = e;
                    continue;
                }
                const size_t stop = std::max(e, l + 1);
                for (; l < stop; l++) launch_basicmf(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
            }
            ds->chained_levels = chained;
        } else if (ds->kind == 5) {
            // window-minibatch step, first half: the users' exact walks; the item side is only read, its would-be change goes to the
            // contribution slots that window_delta_pack sums (svdf_k_window.hip)
            launch_window_users(P, window_view(ds), window_slots_, window_groups_, stream_);
        } else if (ds->kind == 7) {   // the same step for user units (svdf_k_wunit.hip)
            wunit_train(ds);
        } else if (ds->kind == 3) {
            const UnitDev &d = ds->unitdev;
            const DevCSR D = d.csr();
            for (size_t l = 0; l < sc.num_levels(); l++) {
                launch_svdpp_wave(P, D, d.units.p, d.fbidx.p, d.fbval.p, d.order.p, svdpp_xunits_ ? d.xunits.p : nullptr, sc.level_ptr[l], sc.level_mid[l], stream_);
                launch_svdpp(P, D, d.units.p, d.fbidx.p, d.fbval.p, d.order.p, sc.level_mid[l], sc.level_ptr[l + 1], sample_counter_, stream_);
            }
        } else if (ds->kind == 4) {
            const UnitDev &d = ds->unitdev;
            const DevCSR D = d.csr();
            for (size_t l = 0; l < sc.num_levels(); l++)
                launch_imfb(P, D, d.units.p, d.blks.p, d.fbidx.p, d.fbval.p, d.order.p, sc.level_ptr[l], sc.level_ptr[l + 1], sample_counter_, nullptr, stream_);
        } else if (ds->kind == 2) {
            const FusedSchedule S = ds->fused.view();
            ds->chained_levels = 0;
            if (fewrow_gslots_ && fewrow_fast_ && fewrow_gslots_applies(P, S, ds->fused.max_nu, ds->fused.max_ni, ds->fused.dense_slots)) {
                for (size_t l = 0; l < sc.num_levels(); l++) launch_fewrow_gslots(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], stream_);
            } else {
                // runs of NARROW levels (a rank pass in file order: tens of thousands of levels of a few dozen pairs) go through one launch per
                // run -- one workgroup walks the levels with a barrier in between (k_fewrow_slots_chain) --, everything else level by level
                const size_t L = sc.num_levels();
                const long cw = chain_width_;
                int64_t chained = 0;
                const bool can_chain = cw > 0 && ds->d_level_ptr_ok && launch_fewrow_chain(P, S, ds->fused.max_nu, ds->fused.max_ni, nullptr, 0, 0, stream_);
                for (size_t l = 0; l < L;) {
                    size_t e = l;
                    if (can_chain) while (e < L && sc.level_ptr[e + 1] - sc.level_ptr[e] <= cw) e++;
                    if (e >= l + 4) {
                        (void)launch_fewrow_chain(P, S, ds->fused.max_nu, ds->fused.max_ni, ds->d_level_ptr.p, (long)l, (long)e, stream_);
                        chained += (int64_t)(e - l);
                        l = e;
                        continue;
                    }
                    const size_t stop = std::max(e, l + 1);
                    for (; l < stop; l++)
                        launch_fused(P, S, ds->fused.max_nu, ds->fused.max_ni, sc.level_ptr[l], sc.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
                }
                ds->chained_levels = chained;   // what this launch sequence chains: counted per PASS below (issue() runs once per graph capture)
            }
        } else {
            DevCSR D{ds->row_label.p, ds->row_ptr.p, ds->feat_index.p, ds->feat_value.p};
            for (size_t l = 0; l < sc.num_levels(); l++) launch_general(P, D, ds->order.p, sc.level_ptr[l], sc.level_ptr[l + 1], sample_counter_, stream_);
        }
    };
    // A pass over a resident dataset is the same launch sequence every time: capture it once into a hipGraph and
    // replay it (short batches are launch-bound on the host otherwise).  Re-captured when kernel parameters, launch
    // knobs or the stream change; the lazy decay modes pass a per-pass counter and stay on plain launches.
    if (use_graph_ && !lazy_decay() && ds->kind != 5 && ds->kind != 7 && sc.num_levels() >= (size_t)graph_min_levels_) {   // window data sets use trainer-owned scratch that may be re-sized between passes: plain launches
        if (!ds->graph_exec || ds->graph_version != launch_version_ || ds->graph_stream != stream_) {
            if (ds->graph_exec) { (void)hipGraphExecDestroy(ds->graph_exec); ds->graph_exec = nullptr; }
            hipGraph_t g = nullptr;
            HIPCHECK(hipStreamBeginCapture(stream_, hipStreamCaptureModeThreadLocal));
            issue();
            HIPCHECK(hipStreamEndCapture(stream_, &g));
            HIPCHECK(hipGraphInstantiate(&ds->graph_exec, g, nullptr, nullptr, 0));
            (void)hipGraphDestroy(g);
            ds->graph_version = launch_version_;
            ds->graph_stream = stream_;
        }
        HIPCHECK(hipGraphLaunch(ds->graph_exec, stream_));
    } else {
        issue();
    }
    HIPCHECK(hipGetLastError());
    n_chained_levels_ += ds->chained_levels;
    n_launches_ += (int64_t)sc.num_levels();
    if (ds->kind < 3) n_kind_[ds->kind] += (int64_t)sc.num_levels();
    n_batches_ += (int64_t)sc.num_levels();
    n_instances_ += ds->num_row;
    sample_counter_ += (unsigned)ds->num_row;
}

void Engine::set_stream(hipStream_t s) {
    need_device("set_stream");
    flush();
    HIPCHECK(hipStreamSynchronize(stream_));
    if (owns_stream_ && stream_) (void)hipStreamDestroy(stream_);
    stream_ = s;
    owns_stream_ = false;
}
void Engine::synchronize() {
    if (host_only_) return;
    if (multi_) { multi_synchronize(); return; }
    HIPCHECK(hipStreamSynchronize(stream_));
    ipc_fail_if_dead("svdf_synchronize");
}
}  // namespace svdf
