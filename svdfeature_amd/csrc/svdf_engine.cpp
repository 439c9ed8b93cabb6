// svdf_engine.cpp -- see svdf_engine.h.  Reference citations are relative to /root/reference.
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
static inline void check(bool ok, const char *msg) { if (!ok) fail(msg); }
// "this thread is executing the multi-GPU code of a handle" (svdf_multi.cpp): rank 0 of an amd:gpus handle is the handle itself, so
// while a window is being trained or a resident data set is being built its own flush() / dataset_from_*() must act like a plain
// single-GPU engine's.  Per THREAD, not per handle: the background window thread may be inside multi_flush while the caller's
// thread enters flush() and has to wait for it.
static thread_local int tl_multi_depth = 0;
bool in_multi_scope() { return tl_multi_depth > 0; }
MultiScope::MultiScope() { tl_multi_depth++; }
MultiScope::~MultiScope() { tl_multi_depth--; }
#define HIPCHECK(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)

// =============================================================================== DevBuf
template <typename T>
void DevBuf<T>::release() {
    if (p) (void)hipFree(p);
    p = nullptr;
    cap = 0;
}
template <typename T>
void DevBuf<T>::reserve(size_t n) {
    if (n <= cap && p) return;
    release();
    size_t want = n ? n : 1;
    HIPCHECK(hipMalloc((void **)&p, want * sizeof(T)));
    cap = want;
}
template <typename T>
void DevBuf<T>::upload(const T *src, size_t n, hipStream_t st) {
    reserve(n);
    if (n) HIPCHECK(hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
}
template struct DevBuf<float>;
template struct DevBuf<int>;
template struct DevBuf<unsigned>;
template struct DevBuf<DevUnit>;
template struct DevBuf<DevBlk>;
template struct DevBuf<signed char>;
template struct DevBuf<double>;
template struct DevBuf<long>;
template struct DevBuf<char>;
template struct DevBuf<WinUser>;
template struct DevBuf<WinUnit>;
template struct DevBuf<WinSeg>;
template struct DevBuf<WinEnt>;

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
static void tp_set_param(TrainParam &p, const char *name, const char *val) {  // apex_svd_model.h:350-368
    if (!strcmp("learning_rate", name)) p.learning_rate = (float)atof(val);
    if (!strcmp("wd_user", name)) p.wd_user = (float)atof(val);
    if (!strcmp("wd_item", name)) p.wd_item = (float)atof(val);
    if (!strcmp("wd_uiset", name)) p.wd_user = p.wd_item = (float)atof(val);
    if (!strcmp("wd_user_bias", name)) p.wd_user_bias = (float)atof(val);
    if (!strcmp("wd_item_bias", name)) p.wd_item_bias = (float)atof(val);
    if (!strcmp("wd_uiset_bias", name)) p.wd_user_bias = p.wd_item_bias = (float)atof(val);
    if (!strcmp("wd_global", name)) p.wd_global = (float)atof(val);
    if (!strcmp("reg_method", name)) p.reg_method = atoi(val);
    if (!strcmp("reg_global", name)) p.reg_global = atoi(val);
    if (!strcmp("num_regfree_global", name)) p.num_regfree_global = (unsigned)atoi(val);
    if (!strcmp("decay_learning_rate", name)) p.decay_learning_rate = atoi(val);
    if (!strcmp("min_learning_rate", name)) p.min_learning_rate = (float)atof(val);
    if (!strcmp("decay_rate", name)) p.decay_rate = (float)atof(val);
    if (!strcmp("scale_lr_ufeedback", name)) p.scale_lr_ufeedback = (float)atof(val);
    if (!strcmp("wd_ufeedback", name)) p.wd_ufeedback = (float)atof(val);
    if (!strcmp("wd_ufeedback_bias", name)) p.wd_ufeedback_bias = (float)atof(val);
}
static void mp_set_param(ModelParam &p, const char *name, const char *val) {  // apex_svd_model.h:456-476
    if (!strcmp("num_user", name)) p.num_user = atoi(val);
    if (!strcmp("num_item", name)) p.num_item = atoi(val);
    if (!strcmp("num_uiset", name)) p.num_user = p.num_item = atoi(val);
    if (!strcmp("num_global", name)) p.num_global = atoi(val);
    if (!strcmp("num_factor", name)) p.num_factor = atoi(val);
    if (!strcmp("u_init_sigma", name)) p.u_init_sigma = (float)atof(val);
    if (!strcmp("i_init_sigma", name)) p.i_init_sigma = (float)atof(val);
    if (!strcmp("ui_init_sigma", name)) p.u_init_sigma = p.i_init_sigma = (float)atof(val);
    if (!strcmp("base_score", name)) p.base_score = (float)atof(val);
    if (!strcmp("no_user_bias", name)) p.no_user_bias = atoi(val);
    if (!strcmp("num_ufeedback", name)) p.num_ufeedback = atoi(val);
    if (!strcmp("num_randinit_ufactor", name)) p.num_randinit_ufactor = atoi(val);
    if (!strcmp("num_randinit_ifactor", name)) p.num_randinit_ifactor = atoi(val);
    if (!strcmp("num_randinit_uifactor", name)) p.num_randinit_ifactor = p.num_randinit_ufactor = atoi(val);
    if (!strcmp("ufeedback_init_sigma", name)) p.ufeedback_init_sigma = (float)atof(val);
    if (!strcmp("common_latent_space", name)) p.common_latent_space = atoi(val);
    if (!strcmp("common_feedback_space", name)) p.common_feedback_space = atoi(val);
    if (!strcmp("user_nonnegative", name)) p.user_nonnegative = atoi(val);
    if (!strcmp("item_nonnegative", name)) p.item_nonnegative = atoi(val);
}

// =============================================================================== lifecycle
namespace {
struct RandStateGuard {
    char scratch[256];
    char *old;
    RandStateGuard() { old = initstate(1u, scratch, sizeof(scratch)); }
    ~RandStateGuard() { if (old) setstate(old); }
};
}  // namespace

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

namespace {
struct ScopedNs {
    int64_t &acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedNs(int64_t &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedNs() { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};
}  // namespace

Engine::~Engine() {
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
        for (int b = 0; b < 2; b++) {
            if (save_pin_[b]) (void)hipHostFree(save_pin_[b]);
            if (save_ev_[b]) (void)hipEventDestroy(save_ev_[b]);
        }
        if (owns_stream_) (void)hipStreamDestroy(stream_);
    }
}

void Engine::need_device(const char *what) {
    if (host_only_) fail(std::string("svdfeature_amd: ") + what + " needs a GPU (handle was created host-only)");
    HIPCHECK(hipSetDevice(device_));
}

void Engine::set_param(const char *name, const char *val) {  // apex_svd_base.h:126-136
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
        check(!strcmp(val, "minibatch") || !strcmp(val, "levels"), "amd:step must be minibatch or levels");
        check(!multi_, "amd:step must be set before the model is created");
        multi_step_levels_ = !strcmp(val, "levels");
        step_minibatch_set_ = !strcmp(val, "minibatch");   // one GPU: opt-in window-minibatch SGD (svdf_wunit.cpp: resident data sets become window sequences)
    }
    if (!strcmp(name, "amd:contrib")) {   // window-minibatch step: storage format of the contribution rows (sums are fp32 either way)
        check(!strcmp(val, "fp32") || !strcmp(val, "bf16"), "amd:contrib must be fp32 or bf16");
        contrib_bf16_ = !strcmp(val, "bf16");
    }
    if (!strcmp(name, "amd:window")) { stage_window_ = std::max<long>(1, atol(val)); window_set_ = true; }
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
    tp_set_param(tp_, name, val);
    u_param_.set_param(name, val);
    i_param_.set_param(name, val);
    g_param_.set_param(name, val);
    if (!space_allocated_) mp_set_param(mp_, name, val);
    params_dirty_ = true;
}

void Engine::compute_geometry() {  // SVDModel::alloc_space apex_svd_model.h:511-556
    const int ustart = (mp_.common_feedback_space == 0 && user_group()) ? mp_.num_ufeedback : 0;
    if (mp_.common_latent_space == 0) {
        n_uiset_ = (long)ustart + mp_.num_user + mp_.num_item;
        user_off_ = (unsigned)ustart;
        item_off_ = (unsigned)(ustart + mp_.num_user);
    } else {
        check(mp_.num_user == mp_.num_item, "num_user and num_item must be the same to use common latent space");
        check(mp_.common_feedback_space != 0, "common latent space must enforce common feedback space");
        n_uiset_ = mp_.num_item;
        user_off_ = item_off_ = (unsigned)ustart;
    }
    fb_off_ = mp_.common_feedback_space == 0 ? 0u : user_off_;
    pitch_ = ((mp_.num_factor + 3) / 4) * 4;   // ceil(4k/16)*16 bytes (apex_tensor_sse.h:26-27)
    space_allocated_ = true;
}
void Engine::alloc_host_model() {
    compute_geometry();
    hW_.assign((size_t)n_uiset_ * pitch_, 0.0f);
    hbias_.assign((size_t)n_uiset_, 0.0f);
    hg_.assign((size_t)mp_.num_global, 0.0f);
    host_model_valid_ = true;
}

// ---- PRNG: apex-tensor/apex_random.h:42-77 over libc rand(), so the starting point is bit-identical
static inline double next_double2() { return ((double)rand() + 1.0) / ((double)RAND_MAX + 2.0); }
static inline double sample_normal() {
    double x, y, s;
    do {
        x = 2 * next_double2() - 1.0;
        y = 2 * next_double2() - 1.0;
        s = x * x + y * y;
    } while (s >= 1.0 || s == 0.0);
    return x * sqrt(-2.0 * log(s) / s);
}
static void sample_gaussian(float *w, long rows, int cols, int pitch, float sd) {  // apex_tensor_cpu_inline_common.h:249-253
    for (long y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++) w[(size_t)y * pitch + x] = (float)sample_normal() * sd;
}
static float calc_base_score(float base_score, int type) {  // apex_svd_model.h:220-237
    switch (type) {
    case ACT_LINEAR: case ACT_HINGE_L2: case ACT_HINGE_SMOOTH: return base_score;
    case ACT_SIGMOID_L2: case ACT_SIGMOID_LIKELIHOOD: case ACT_SIGMOID_RANK: case ACT_SIGMOID_QSGRAD:
        check(base_score > 0.0f && base_score < 1.0f, "sigmoid range constrain");
        return -logf(1.0f / base_score - 1.0f);
    default: fail("unkown active type");
    }
}
void Engine::rand_init() {  // SVDModel::rand_init apex_svd_model.h:665-705
    mp_.base_score = calc_base_score(mp_.base_score, mtype_.active_type);
    const int k = mp_.num_factor;
    float *Wu = hW_.data() + (size_t)user_off_ * pitch_;
    float *Wi = hW_.data() + (size_t)item_off_ * pitch_;
    {
        long rows = mp_.num_randinit_ufactor != 0 ? mp_.num_randinit_ufactor : mp_.num_user;
        sample_gaussian(Wu, rows, k, pitch_, mp_.u_init_sigma);
        if (mp_.user_nonnegative)
            for (long y = 0; y < mp_.num_user; y++)
                for (int x = 0; x < k; x++) Wu[(size_t)y * pitch_ + x] = fabsf(Wu[(size_t)y * pitch_ + x]);
    }
    if (mp_.common_latent_space == 0) {
        long rows = mp_.num_randinit_ifactor != 0 ? mp_.num_randinit_ifactor : mp_.num_item;
        sample_gaussian(Wi, rows, k, pitch_, mp_.i_init_sigma);
        if (mp_.item_nonnegative)
            for (long y = 0; y < rows; y++)
                for (int x = 0; x < k; x++) Wi[(size_t)y * pitch_ + x] = fabsf(Wi[(size_t)y * pitch_ + x]);
    }
    if (user_group())  // draws are consumed even when sigma == 0 (apex_svd_model.h:702-704)
        sample_gaussian(hW_.data() + (size_t)fb_off_ * pitch_, num_fb_rows(), k, pitch_, mp_.ufeedback_init_sigma);
}

// SVDModel::rand_init on the device (svdf_k_init.hip): the same draws, the same accepted attempts, the same float products; the few
// values whose double sits within `margin` of a float rounding boundary are recomputed here with the host libm (the one function the
// device cannot restate) and patched.  Leaves the model in HBM (no host copy) and libc's generator where the reference's calls would
// have left it.  Returns false -- nothing touched -- when the path does not apply: libc's generator not in its 31-word mode, or a
// feedback space that aliases the user rows (W_ufeedback is then written over W_user in sequence).
bool Engine::rand_init_device() {
    if (host_only_ || !device_init_) return false;
    compute_geometry();
    if (user_group() && mp_.common_feedback_space != 0) return false;
    LibcRand s0;
    if (!libc_rand_capture(s0)) return false;
    const int k = mp_.num_factor;
    InitPlan plan;
    memset(&plan, 0, sizeof(plan));
    plan.pitch = pitch_;
    plan.margin = std::ldexp(1.0, -device_init_margin_log2_);
    long total = 0;
    auto add = [&](long rows, unsigned row0, float sigma, bool absf) {
        InitSeg &g = plan.seg[plan.nseg++];
        g.begin = total; g.count = rows * (long)k; g.row0 = (long)row0; g.k = std::max(k, 1); g.sigma = sigma; g.absf = absf ? 1 : 0;
        total += g.count;
    };
    add(mp_.num_randinit_ufactor != 0 ? mp_.num_randinit_ufactor : mp_.num_user, user_off_, mp_.u_init_sigma, mp_.user_nonnegative != 0);
    if (mp_.common_latent_space == 0)
        add(mp_.num_randinit_ifactor != 0 ? mp_.num_randinit_ifactor : mp_.num_item, item_off_, mp_.i_init_sigma, mp_.item_nonnegative != 0);
    if (user_group()) add(num_fb_rows(), fb_off_, mp_.ufeedback_init_sigma, false);
    for (int g = plan.nseg; g < 3; g++) plan.seg[g].begin = total;
    plan.total = total;
    for (int g = 0; g < plan.nseg; g++)   // a view that does not fit its matrix (inconsistent shape keys): leave it to the host loop, as before
        if (plan.seg[g].count < 0 || plan.seg[g].row0 + plan.seg[g].count / std::max(k, 1) > (long)n_uiset_) return false;
    need_device("init_model");
    const float base_score = calc_base_score(mp_.base_score, mtype_.active_type);
    dW_.reserve(std::max<size_t>((size_t)n_uiset_ * pitch_, 1));
    HIPCHECK(hipMemsetAsync(dW_.p, 0, (size_t)n_uiset_ * pitch_ * sizeof(float), stream_));
    n_init_reports_ = 0;
    if (total > 0) {
        const long TILE = 1L << 25, C = 16384;   // attempts per tile (256 MB of raw draws), draws per jump-ahead chunk
        const int report_cap = 1 << 20;
        DevBuf<unsigned> raw, flag, off, tables;
        DevBuf<char> tmp;
        DevBuf<unsigned long long> state;
        DevBuf<InitReport> reports;
        state.reserve(4); reports.reserve((size_t)report_cap);
        HIPCHECK(hipMemsetAsync(state.p, 0, 4 * sizeof(unsigned long long), stream_));
        long accepted = 0, draws_total = 0;
        LibcRand cur = s0, after = s0;
        std::vector<uint32_t> htab;
        for (bool done = false; !done;) {
            const long need = total - accepted;
            const long A = std::min<long>(TILE, (long)((double)need / 0.78539816339744831) + 8 * (long)std::sqrt((double)need) + 4096);
            const long D = 2 * A, nchunks = (D + C - 1) / C;
            libc_rand_chunk_states(cur, nchunks, C, htab);
            tables.upload(htab.data(), htab.size(), stream_);
            raw.reserve((size_t)D); flag.reserve((size_t)A); off.reserve((size_t)A);
            const size_t tb = init_scan_tmp_bytes(A);
            tmp.reserve(std::max<size_t>(tb, 1));
            launch_init_expand(tables.p, nchunks, C, D, raw.p, stream_);
            HIPCHECK(hipMemsetAsync(state.p, 0, 2 * sizeof(unsigned long long), stream_));
            launch_init_tile(raw.p, A, accepted, plan, dW_.p, flag.p, off.p, tmp.p, tb, state.p, reports.p, report_cap, stream_);
            unsigned long long hs[4];
            HIPCHECK(hipMemcpyAsync(hs, state.p, sizeof(hs), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
            long used = D;   // draws of this tile that the reference's loop would have taken
            if (hs[1] != 0) { used = 2 * (long)hs[1]; done = true; }
            else { accepted += (long)hs[0]; check((long)hs[0] > 0, "init_model: the device sampler made no progress"); }
            LibcRand nxt = cur;
            if (used >= 31) {
                HIPCHECK(hipMemcpyAsync(nxt.x, raw.p + (used - 31), 31 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
                HIPCHECK(hipStreamSynchronize(stream_));
            } else {
                uint32_t head[31];
                HIPCHECK(hipMemcpyAsync(head, raw.p, (size_t)used * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
                HIPCHECK(hipStreamSynchronize(stream_));
                for (long j = 0; j < 31 - used; j++) nxt.x[j] = cur.x[used + j];
                for (long j = 0; j < used; j++) nxt.x[31 - used + j] = head[j];
            }
            cur = nxt; after = nxt;
            draws_total += used;
            n_init_reports_ = (int64_t)hs[2];
        }
        // values near a float rounding boundary: the host libm decides (apex_random.h:67-77 as written)
        if (n_init_reports_ > report_cap) {   // a margin wider than the float spacing reports everything: the host loop does the whole job
            n_init_reports_ = 0; n_init_draws_ = 0;   // (libc's generator and the parameters have not been touched)
            return false;
        }
        if (n_init_reports_ > 0) {
            std::vector<InitReport> rep((size_t)n_init_reports_);
            HIPCHECK(hipMemcpyAsync(rep.data(), reports.p, rep.size() * sizeof(InitReport), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
            std::vector<long> pidx(rep.size());
            std::vector<float> pval(rep.size());
            for (size_t q = 0; q < rep.size(); q++) {
                const double x = 2 * (((double)(int)(rep[q].r1 >> 1) + 1.0) / ((double)RAND_MAX + 2.0)) - 1.0;
                const double y = 2 * (((double)(int)(rep[q].r2 >> 1) + 1.0) / ((double)RAND_MAX + 2.0)) - 1.0;
                const double sq = x * x + y * y;
                const double v = x * sqrt(-2.0 * log(sq) / sq);
                int g = 0;
                while (g + 1 < plan.nseg && rep[q].j >= plan.seg[g + 1].begin) g++;
                const InitSeg &sg = plan.seg[g];
                const long jj = rep[q].j - sg.begin, row = jj / sg.k, col = jj - row * sg.k;
                float w = (float)v * sg.sigma;
                if (sg.absf) w = fabsf(w);
                pidx[q] = (sg.row0 + row) * (long)pitch_ + col;
                pval[q] = w;
            }
            DevBuf<long> didx;
            DevBuf<float> dval;
            didx.upload(pidx.data(), pidx.size(), stream_);
            dval.upload(pval.data(), pval.size(), stream_);
            launch_init_patch((long)pidx.size(), didx.p, dval.p, dW_.p, stream_);
            HIPCHECK(hipStreamSynchronize(stream_));
        }
        libc_rand_restore(after);   // libc's generator moves on by exactly the draws of the reference's loop
        n_init_draws_ = draws_total;
    }
    mp_.base_score = base_score;
    // the rest of a fresh model: biases and global biases 0 (apex_svd_model.h:666-667), the kernels' state words 0
    dbias_.reserve(std::max<size_t>((size_t)n_uiset_, 1));
    HIPCHECK(hipMemsetAsync(dbias_.p, 0, std::max<size_t>((size_t)n_uiset_, 1) * sizeof(float), stream_));
    g_stride_ = wanted_g_stride();
    dg_.reserve(std::max<size_t>((size_t)mp_.num_global * (size_t)g_stride_, 1));
    HIPCHECK(hipMemsetAsync(dg_.p, 0, std::max<size_t>((size_t)mp_.num_global * (size_t)g_stride_, 1) * sizeof(float), stream_));
    std::vector<float> zero(imfb() ? (size_t)4 + IMFB_DEPTH_MAX * ((size_t)2 * pitch_ + 4) : (size_t)2 * pitch_ + 4, 0.0f);
    dstate_.upload(zero.data(), zero.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    device_model_ = true;
    params_dirty_ = true;
    hW_.clear(); hW_.shrink_to_fit(); hbias_.clear(); hbias_.shrink_to_fit(); hg_.clear();
    host_model_valid_ = false;
    return true;
}

void Engine::init_model() {  // apex_svd_base.h:146-149
    if (rand_init_device()) {
        if (gpus_ > 1) download_model();   // the other ranks of an amd:gpus handle start from a host copy of rank 0's model
    } else {
        alloc_host_model();
        rand_init();
    }
    if (bilinear()) {   // BModel::alloc_space (apex_svd_bilinear.h:49-54, :202-205): W_bi[num_item][num_bi_feedback] = 0
        check(bi_param_.num_bi_feedback >= 0, "num_bi_feedback must not be negative");
        hbi_.assign((size_t)mp_.num_item * (size_t)bi_param_.num_bi_feedback, 0.0f);
        bi_allocated_ = true;
    }
    multi_setup();
    multi_copy_model_to_peers();
    if (device_model_ && host_model_valid_) {   // a host-built model (or the host copy an amd:gpus handle starts its ranks from) replaces what the device held
        if (multi_) for (int d = 1; d < gpus_; d++) { Engine *e = rank_engine(d); if (e->device_model_) { HIPCHECK(hipSetDevice(e->device_)); e->upload_model(); } }
        HIPCHECK(hipSetDevice(device_));
        upload_model();
    }
}

// ---- model file: apex_svd_model.h:570-660; tensors: int header x_max[,y_max] + unpadded rows
static void save_1d(FILE *fo, const float *v, int n) {
    fwrite(&n, sizeof(int), 1, fo);
    fwrite(v, sizeof(float), (size_t)n, fo);
}
static void save_2d(FILE *fo, const float *w, int rows, int cols, int pitch) {
    int hdr[2] = {cols, rows};
    fwrite(hdr, sizeof(int), 2, fo);
    if (cols == pitch) { fwrite(w, sizeof(float), (size_t)rows * cols, fo); return; }   // k % 4 == 0: rows are contiguous
    for (int y = 0; y < rows; y++) fwrite(w + (size_t)y * pitch, sizeof(float), (size_t)cols, fo);
}
static void load_1d(FILE *fi, float *v, int n) {
    int x;
    check(fread(&x, sizeof(int), 1, fi) > 0, "tensor::load_from_file");
    check(x == n, "tensor::load_from_file: shape does not match the model header");
    if (n > 0) check(fread(v, sizeof(float), (size_t)n, fi) > 0, "tensor::load_from_file");
}
static void load_2d(FILE *fi, float *w, int rows, int cols, int pitch) {
    int hdr[2];
    check(fread(hdr, sizeof(int), 2, fi) > 0, "tensor::load_from_file");
    check(hdr[0] == cols && hdr[1] == rows, "tensor::load_from_file: shape does not match the model header");
    if (cols == pitch && rows > 0 && cols > 0) {
        check(fread(w, sizeof(float), (size_t)rows * cols, fi) == (size_t)rows * cols, "tensor::load_from_file");
        return;
    }
    for (int y = 0; y < rows; y++)
        if (cols > 0) check(fread(w + (size_t)y * pitch, sizeof(float), (size_t)cols, fi) > 0, "tensor::load_from_file");
}
void Engine::write_model(FILE *fo) {
    const int k = mp_.num_factor;
    fwrite(&mp_, sizeof(ModelParam), 1, fo);
    if (mp_.common_latent_space == 0) {
        save_1d(fo, hbias_.data() + user_off_, mp_.num_user);
        save_2d(fo, hW_.data() + (size_t)user_off_ * pitch_, mp_.num_user, k, pitch_);
        save_1d(fo, hbias_.data() + item_off_, mp_.num_item);
        save_2d(fo, hW_.data() + (size_t)item_off_ * pitch_, mp_.num_item, k, pitch_);
    } else {
        save_1d(fo, hbias_.data(), (int)n_uiset_);
        save_2d(fo, hW_.data(), (int)n_uiset_, k, pitch_);
    }
    save_1d(fo, hg_.data(), mp_.num_global);
    if (user_group() && mp_.common_feedback_space == 0) {
        save_1d(fo, hbias_.data(), mp_.num_ufeedback);
        save_2d(fo, hW_.data(), mp_.num_ufeedback, k, pitch_);
    }
}
// The same file straight from the device model (no 282 MB host mirror for a 1 M x 64 user table): the tables travel in chunks through two
// pinned buffers, chunk c+1 is copied out while chunk c goes to the file.  Rows are compacted by the copy itself (2-D copy: k floats
// of every pitch_-float row), so the file bytes are write_model's.
void Engine::dev_to_file(FILE *fo, const float *dsrc, long rows, long cols, long pitch) {
    if (rows <= 0 || cols <= 0) return;
    const size_t cap = (size_t)8 << 20;   // floats per buffer (32 MB)
    if (!save_pin_[0]) {
        for (int b = 0; b < 2; b++) {
            HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&save_pin_[b]), cap * sizeof(float), hipHostMallocDefault));
            HIPCHECK(hipEventCreateWithFlags(&save_ev_[b], hipEventDisableTiming));
        }
    }
    check((size_t)cols <= cap, "save_model: a row wider than the staging buffer");
    const long per = std::max<long>(1, (long)(cap / (size_t)cols));
    long prev_rows = 0;
    int c = 0;
    for (long r0 = 0; r0 < rows || prev_rows > 0; r0 += per, c++) {
        const long nr = r0 < rows ? std::min(per, rows - r0) : 0;
        if (nr > 0) {
            float *dst = save_pin_[c & 1];
            if (cols == pitch) HIPCHECK(hipMemcpyAsync(dst, dsrc + (size_t)r0 * pitch, (size_t)nr * cols * sizeof(float), hipMemcpyDeviceToHost, stream_));
            else HIPCHECK(hipMemcpy2DAsync(dst, (size_t)cols * sizeof(float), dsrc + (size_t)r0 * pitch, (size_t)pitch * sizeof(float), (size_t)cols * sizeof(float),
                                           (size_t)nr, hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipEventRecord(save_ev_[c & 1], stream_));
        }
        if (prev_rows > 0) {
            HIPCHECK(hipEventSynchronize(save_ev_[(c - 1) & 1]));
            fwrite(save_pin_[(c - 1) & 1], sizeof(float), (size_t)prev_rows * cols, fo);
        }
        prev_rows = nr;
    }
}
void Engine::write_model_from_device(FILE *fo) {
    const int k = mp_.num_factor;
    auto d1 = [&](const float *d, int n) { fwrite(&n, sizeof(int), 1, fo); dev_to_file(fo, d, n, 1, 1); };
    auto d2 = [&](const float *d, int rows) { int hdr[2] = {k, rows}; fwrite(hdr, sizeof(int), 2, fo); dev_to_file(fo, d, rows, k, pitch_); };
    fwrite(&mp_, sizeof(ModelParam), 1, fo);
    if (mp_.common_latent_space == 0) {
        d1(dbias_.p + user_off_, mp_.num_user);
        d2(dW_.p + (size_t)user_off_ * pitch_, mp_.num_user);
        d1(dbias_.p + item_off_, mp_.num_item);
        d2(dW_.p + (size_t)item_off_ * pitch_, mp_.num_item);
    } else {
        d1(dbias_.p, (int)n_uiset_);
        d2(dW_.p, (int)n_uiset_);
    }
    {   // globals: a few words (strided on the device in the relaxed mode): through the host vector
        hg_.resize((size_t)mp_.num_global);
        if (!hg_.empty()) { download_globals(hg_.data()); HIPCHECK(hipStreamSynchronize(stream_)); }
        save_1d(fo, hg_.data(), mp_.num_global);
    }
    if (user_group() && mp_.common_feedback_space == 0) {
        d1(dbias_.p, mp_.num_ufeedback);
        d2(dW_.p, mp_.num_ufeedback);
    }
}
// the mirror of dev_to_file: a tensor's rows from the file into HBM through the two pinned buffers, fread of chunk c + 1 beside the copy of chunk c
void Engine::file_to_dev(FILE *fi, float *ddst, long rows, long cols, long pitch) {
    if (rows <= 0 || cols <= 0) return;
    const size_t cap = (size_t)8 << 20;   // floats per buffer (32 MB)
    if (!save_pin_[0]) {
        for (int b = 0; b < 2; b++) {
            HIPCHECK(hipHostMalloc(reinterpret_cast<void **>(&save_pin_[b]), cap * sizeof(float), hipHostMallocDefault));
            HIPCHECK(hipEventCreateWithFlags(&save_ev_[b], hipEventDisableTiming));
        }
    }
    check((size_t)cols <= cap, "load_model: a row wider than the staging buffer");
    const long per = std::max<long>(1, (long)(cap / (size_t)cols));
    int c = 0;
    for (long r0 = 0; r0 < rows; r0 += per, c++) {
        const long nr = std::min(per, rows - r0);
        float *buf = save_pin_[c & 1];
        if (c >= 2) HIPCHECK(hipEventSynchronize(save_ev_[c & 1]));   // the copy that read this buffer two chunks ago
        check(fread(buf, sizeof(float), (size_t)nr * cols, fi) == (size_t)nr * cols, "tensor::load_from_file");
        if (cols == pitch) HIPCHECK(hipMemcpyAsync(ddst + (size_t)r0 * pitch, buf, (size_t)nr * cols * sizeof(float), hipMemcpyHostToDevice, stream_));
        else HIPCHECK(hipMemcpy2DAsync(ddst + (size_t)r0 * pitch, (size_t)pitch * sizeof(float), buf, (size_t)cols * sizeof(float), (size_t)cols * sizeof(float),
                                       (size_t)nr, hipMemcpyHostToDevice, stream_));
        HIPCHECK(hipEventRecord(save_ev_[c & 1], stream_));
    }
    HIPCHECK(hipStreamSynchronize(stream_));   // the buffers are free for the next tensor
}
// SVDModel::load_from_file (apex_svd_model.h:570-585) straight into HBM: same order, same shape checks as read_model, no host copy of the matrices
void Engine::read_model_to_device(FILE *fi) {
    if (fread(&mp_, sizeof(ModelParam), 1, fi) == 0) fail("error loading CF SVD model");
    compute_geometry();
    need_device("loading the model");
    const int k = mp_.num_factor;
    dW_.reserve(std::max<size_t>((size_t)n_uiset_ * pitch_, 1));
    dbias_.reserve(std::max<size_t>((size_t)n_uiset_, 1));
    if (pitch_ != k) HIPCHECK(hipMemsetAsync(dW_.p, 0, (size_t)n_uiset_ * pitch_ * sizeof(float), stream_));   // the pad floats of every row stay 0
    auto d1 = [&](float *d, int n) {
        int x;
        check(fread(&x, sizeof(int), 1, fi) > 0, "tensor::load_from_file");
        check(x == n, "tensor::load_from_file: shape does not match the model header");
        file_to_dev(fi, d, n, 1, 1);
    };
    auto d2 = [&](float *d, int rows) {
        int hdr[2];
        check(fread(hdr, sizeof(int), 2, fi) > 0, "tensor::load_from_file");
        check(hdr[0] == k && hdr[1] == rows, "tensor::load_from_file: shape does not match the model header");
        file_to_dev(fi, d, rows, k, pitch_);
    };
    if (mp_.common_latent_space == 0) {
        d1(dbias_.p + user_off_, mp_.num_user);
        d2(dW_.p + (size_t)user_off_ * pitch_, mp_.num_user);
        d1(dbias_.p + item_off_, mp_.num_item);
        d2(dW_.p + (size_t)item_off_ * pitch_, mp_.num_item);
    } else {
        d1(dbias_.p, (int)n_uiset_);
        d2(dW_.p, (int)n_uiset_);
    }
    hg_.assign((size_t)mp_.num_global, 0.0f);
    load_1d(fi, hg_.data(), mp_.num_global);
    upload_globals(wanted_g_stride());
    if (user_group() && mp_.common_feedback_space == 0) {
        d1(dbias_.p, mp_.num_ufeedback);
        d2(dW_.p, mp_.num_ufeedback);
    }
    std::vector<float> zero(imfb() ? (size_t)4 + IMFB_DEPTH_MAX * ((size_t)2 * pitch_ + 4) : (size_t)2 * pitch_ + 4, 0.0f);
    dstate_.upload(zero.data(), zero.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    device_model_ = true;
    hW_.clear(); hW_.shrink_to_fit(); hbias_.clear(); hbias_.shrink_to_fit(); hg_.clear();
    host_model_valid_ = false;
}
void Engine::read_model(FILE *fi) {
    if (fread(&mp_, sizeof(ModelParam), 1, fi) == 0) fail("error loading CF SVD model");
    alloc_host_model();
    const int k = mp_.num_factor;
    if (mp_.common_latent_space == 0) {
        load_1d(fi, hbias_.data() + user_off_, mp_.num_user);
        load_2d(fi, hW_.data() + (size_t)user_off_ * pitch_, mp_.num_user, k, pitch_);
        load_1d(fi, hbias_.data() + item_off_, mp_.num_item);
        load_2d(fi, hW_.data() + (size_t)item_off_ * pitch_, mp_.num_item, k, pitch_);
    } else {
        load_1d(fi, hbias_.data(), (int)n_uiset_);
        load_2d(fi, hW_.data(), (int)n_uiset_, k, pitch_);
    }
    load_1d(fi, hg_.data(), mp_.num_global);
    if (user_group() && mp_.common_feedback_space == 0) {
        load_1d(fi, hbias_.data(), mp_.num_ufeedback);
        load_2d(fi, hW_.data(), mp_.num_ufeedback, k, pitch_);
    }
}
void Engine::load_model(FILE *fi) {  // apex_svd_base.h:138-140
    if (trainer_ready_ && !host_only_) flush();
    if (!host_only_ && gpus_ <= 1 && device_load_) {   // the matrices stream file -> pinned chunks -> HBM
        try { read_model_to_device(fi); }
        catch (...) {   // a truncated / mismatching file leaves no half-loaded model behind: the handle has no model until the next init / load
            (void)hipStreamSynchronize(stream_);
            device_model_ = false; host_model_valid_ = false; space_allocated_ = false; trainer_ready_ = false;
            throw;
        }
    } else read_model(fi);
    if (bilinear()) {   // BModel::load_from_file (apex_svd_bilinear.h:64-68, :194-197)
        check(fread(&bi_param_, sizeof(BiParam), 1, fi) > 0, "load from file");
        check(bi_param_.num_bi_feedback >= 0, "num_bi_feedback must not be negative");
        hbi_.assign((size_t)mp_.num_item * (size_t)bi_param_.num_bi_feedback, 0.0f);
        bi_allocated_ = true;
        load_2d(fi, hbi_.data(), mp_.num_item, bi_param_.num_bi_feedback, bi_param_.num_bi_feedback);
    }
    params_dirty_ = true;
    multi_setup();
    multi_copy_model_to_peers();
    if (multi_) for (int d = 1; d < gpus_; d++) { Engine *e = rank_engine(d); e->params_dirty_ = true; if (e->device_model_) { HIPCHECK(hipSetDevice(e->device_)); e->upload_model(); } }
    if (multi_ && !host_only_) HIPCHECK(hipSetDevice(device_));
    if (device_model_ && host_model_valid_) upload_model();
}
void Engine::save_model(FILE *fo) {  // apex_svd_base.h:142-144
    ScopedNs timer(ns_model_);
    check(space_allocated_, "save_model: model is not initialised");
    if (device_model_) {
        flush();
        if (multi_) multi_gather_user_rows();
        need_device("saving the model");
        write_model_from_device(fo);
    } else {
        check(host_model_valid_, "save_model: no model");
        write_model(fo);
    }
    if (bilinear()) {   // BModel::save_to_file (apex_svd_bilinear.h:60-63, :198-201).  W_bi is inert: SVDPPFeature::update binds its OWN
        // non-virtual prepare_ufeedback (apex_svd_base.h:523,571), so the derived one that would fill up_index never runs and
        // get_bias_plugin / update_bias_plugin (:133-162) loop over nothing -- training is SVDPPFeature's, W_bi rides along
        check(bi_allocated_, "save_model: bilinear model is not initialised");
        fwrite(&bi_param_, sizeof(BiParam), 1, fo);
        save_2d(fo, hbi_.data(), mp_.num_item, bi_param_.num_bi_feedback, bi_param_.num_bi_feedback);
    }
    if (device_model_) { hW_.clear(); hW_.shrink_to_fit(); hbias_.clear(); hg_.clear(); host_model_valid_ = false; }
}

// g_bias on the device: hg_ scattered with `stride` floats between entries (padding zero)
void Engine::upload_globals(int stride) {
    g_stride_ = stride;
    const size_t n = hg_.size();
    dg_.reserve(std::max<size_t>(n * (size_t)stride, 1));
    if (n == 0) return;
    if (stride == 1) { dg_.upload(hg_.data(), n, stream_); return; }
    HIPCHECK(hipMemsetAsync(dg_.p, 0, n * (size_t)stride * sizeof(float), stream_));
    HIPCHECK(hipMemcpy2DAsync(dg_.p, (size_t)stride * sizeof(float), hg_.data(), sizeof(float), sizeof(float), n, hipMemcpyHostToDevice, stream_));
}
void Engine::download_globals(float *dst) {
    const size_t n = (size_t)mp_.num_global;
    if (n == 0) return;
    if (g_stride_ == 1) HIPCHECK(hipMemcpyAsync(dst, dg_.p, n * sizeof(float), hipMemcpyDeviceToHost, stream_));
    else HIPCHECK(hipMemcpy2DAsync(dst, sizeof(float), dg_.p, (size_t)g_stride_ * sizeof(float), sizeof(float), n, hipMemcpyDeviceToHost, stream_));
}

void Engine::upload_model() {
    need_device("uploading the model");
    check(host_model_valid_, "upload_model: no host model");
    dW_.upload(hW_.data(), hW_.size(), stream_);
    dbias_.upload(hbias_.data(), hbias_.size(), stream_);
    upload_globals(wanted_g_stride());
    std::vector<float> zero(imfb() ? (size_t)4 + IMFB_DEPTH_MAX * ((size_t)2 * pitch_ + 4) : (size_t)2 * pitch_ + 4, 0.0f);
    dstate_.upload(zero.data(), zero.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
    device_model_ = true;
    params_dirty_ = true;
    hW_.clear(); hW_.shrink_to_fit(); hbias_.clear(); hbias_.shrink_to_fit(); hg_.clear();
    host_model_valid_ = false;
}
void Engine::download_model() {
    need_device("downloading the model");
    hW_.resize((size_t)n_uiset_ * pitch_);
    hbias_.resize((size_t)n_uiset_);
    hg_.resize((size_t)mp_.num_global);
    if (!hW_.empty()) HIPCHECK(hipMemcpyAsync(hW_.data(), dW_.p, hW_.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (!hbias_.empty()) HIPCHECK(hipMemcpyAsync(hbias_.data(), dbias_.p, hbias_.size() * sizeof(float), hipMemcpyDeviceToHost, stream_));
    if (!hg_.empty()) download_globals(hg_.data());
    HIPCHECK(hipStreamSynchronize(stream_));
    host_model_valid_ = true;
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
static void sort_batches(Schedule &sched, const unsigned *key) {
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

// fn(a, b) over [0, n) in contiguous chunks on up to 16 host threads (fn must not throw)
template <typename F>
static void parallel_rows(long n, F fn) {
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < (1 << 18) || hw == 1) { fn(0L, n); return; }
    std::vector<std::thread> th;
    const long chunk = (n + hw - 1) / hw;
    for (unsigned t = 0; t < hw; t++) {
        const long lo = t * chunk, hi = std::min(n, lo + chunk);
        if (lo >= hi) break;
        th.emplace_back([=]() { fn(lo, hi); });
    }
    for (auto &x : th) x.join();
}

template <typename T>
static void parallel_gather(T *dst, const T *src, const int *order, long n, long stride, long offset) {
    // dst[s] = src[order[s]*stride + offset]
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    if (n < (1 << 20) || hw == 1) {
        for (long s = 0; s < n; s++) dst[s] = src[(long)order[s] * stride + offset];
        return;
    }
    std::vector<std::thread> th;
    const long chunk = (n + hw - 1) / hw;
    for (unsigned t = 0; t < hw; t++) {
        const long a = t * chunk, b = std::min(n, a + chunk);
        if (a >= b) break;
        th.emplace_back([=]() { for (long s = a; s < b; s++) dst[s] = src[(long)order[s] * stride + offset]; });
    }
    for (auto &x : th) x.join();
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

void Engine::flush_csr(HostCSR &src) {
    const long n = src.num_row();
    if (n == 0) return;
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
void Engine::upload_units(UnitDev &d, const Schedule &sched, const std::vector<DevUnit> &du) {
    d.label.upload(staged_.row_label.data(), staged_.row_label.size(), stream_);
    d.ptr.upload(staged_.row_ptr.data(), staged_.row_ptr.size(), stream_);
    d.index.upload(staged_.feat_index.data(), staged_.feat_index.size(), stream_);
    d.value.upload(staged_.feat_value.data(), staged_.feat_value.size(), stream_);
    d.fbidx.upload(staged_fb_index_.data(), staged_fb_index_.size(), stream_);
    d.fbval.upload(staged_fb_value_.data(), staged_fb_value_.size(), stream_);
    d.units.upload(du.data(), du.size(), stream_);
    d.order.upload(sched.order.data(), sched.order.size(), stream_);
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
    if (any_fresh_) d.fresh.upload(staged_fresh_.data(), staged_fresh_.size(), stream_);
    HIPCHECK(hipStreamSynchronize(stream_));
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
    LevelTracker saved;
    std::swap(saved, tracker_);   // a dataset pass is preceded by a flush: schedule against an empty tracker
    std::vector<DevUnit> du;
    schedule_units(0, ds->sched, du);
    std::swap(saved, tracker_);
    upload_units(ds->unitdev, ds->sched, du);
    long nfb = (long)staged_fb_index_.size();
    const long nb = mp_.no_user_bias ? 1 : 2;
    ds->algorithmic_bytes = ds->num_row * (8L * mp_.num_factor * 2 + 8 * nb + 16 + 16) + nfb * (12L * mp_.num_factor + 20);
    ds->num_units = (long)du.size();
    for (auto &x : du) ds->num_simple_units += (x.flags & UNIT_SIMPLE) ? 1 : 0;
    drop();
    return ds.release();
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

Dataset *Engine::dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    if (multi_ && !in_multi_scope()) return multi_dataset_from_triples(n, user, item, label);
    if (single_minibatch() && !user_group() && basic_fast_path_allowed()) return wseq_from_triples(n, user, item, label);
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
// contributes in /root/reference/solvers/base-solver/apex_svd_base.h:383-427 being applied at once by "applied at the window's end".
WindowSchedule Engine::window_view(const Dataset *ds) const {
    const bool pairs = ds->win_item1.p != nullptr && ds->fused.max_ni == 2;
    return WindowSchedule{ds->win_urec.p, ds->num_units, ds->item.p, pairs ? nullptr : ds->label.p, ds->win_slot.p, ds->unit_values ? nullptr : ds->uval.p,
                          (ds->unit_values && !pairs) ? nullptr : ds->ival.p, ds->win_iptr.p, d_contrib_.p, d_cbias_.p,
                          pairs ? ds->win_item1.p : nullptr, pairs ? ds->win_slot1.p : nullptr, pairs ? ds->win_ival1.p : nullptr, contrib_bf16_ ? 1 : 0};
}
Dataset *Engine::dataset_window_from_triples(long n, const unsigned *user, const unsigned *item, const float *label) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; an amd:gpus handle builds them itself from svdf_dataset_from_triples");
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    window_build(ds.get(), n, user, item, label);
    return ds.release();
}
// rank pairs (user, positive item, negative item) of one exchange window: the instance PairwiseRankGenerator emits for two plain rows
// (apex_svd_data.cpp:828-860, :905-911: label 1, user:1, the two items in index order with the negative's sign flipped), BASELINE
// configs[4].  Two contribution slots per pair.
Dataset *Engine::dataset_window_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg) {
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    check(!multi_ || in_multi_scope(), "window data sets are per rank; shard rank pairs through svdfeature_amd.multi_gpu");
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get());
    window_build(ds.get(), n, user, pos, nullptr, neg);
    return ds.release();
}
// (re)fills ds in place: the staged path of an amd:gpus handle rebuilds one window data set per rank every window.
// neg != nullptr: rank pairs, `item` holds the positive items and the labels are 1.
void Engine::window_build_header(Dataset *ds, long n, bool pairs) {
    check(!user_group() && mtype_.extend_type == 0, "window data sets: random-order trainers only");
    check(basic_fast_path_allowed(), "window data sets: no side tables, relaxed ids, lazy decay or shared latent space; num_factor <= 256");
    check(n >= 0 && n < (1L << 30), "window data sets: at most 2^30-1 instances per window");
    if (window_trained_ == ds) window_trained_ = nullptr;
    ds->num_row = n; ds->kind = 5;
    ds->win_slots = pairs ? 2 * n : n;
    ds->fused.max_ni = pairs ? 2 : 1;
}
void Engine::window_build(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg) {
    check(!user_group() && mtype_.extend_type == 0, "window data sets: random-order trainers only");
    check(basic_fast_path_allowed(), "window data sets: no side tables, relaxed ids, lazy decay or shared latent space; num_factor <= 256");
    check(n >= 0 && n < (1L << 30), "window data sets: at most 2^30-1 instances per window");
    const long NU = mp_.num_user, NI = mp_.num_item;
    const bool pairs = neg != nullptr;
    window_build_header(ds, n, pairs);
    if (window_build_device(ds, n, user, item, label, neg)) return;
    std::vector<int> ucnt((size_t)NU, 0), iptr((size_t)NI + 1, 0);
    for (long r = 0; r < n; r++) {
        if (user[r] >= (unsigned)NU) fail("user feature index exceed bound");
        if (item[r] >= (unsigned)NI) fail("item feature index exceed bound");
        ucnt[user[r]]++;
        iptr[(size_t)item[r] + 1]++;
        if (pairs) {
            if (neg[r] >= (unsigned)NI) fail("item feature index exceed bound");
            if (neg[r] == item[r]) fail("rank pair: positive and negative item must differ");
            iptr[(size_t)neg[r] + 1]++;
        }
    }
    for (long i = 0; i < NI; i++) iptr[(size_t)i + 1] += iptr[(size_t)i];
    // users in launch order: by instance count, descending (the lane groups of a wave then run the same number of iterations),
    // ties by user id; a user's instances are contiguous in that order
    int maxc = 0;
    long nact = 0;
    for (long u = 0; u < NU; u++) { maxc = std::max(maxc, ucnt[(size_t)u]); nact += ucnt[(size_t)u] > 0; }
    std::vector<long> start((size_t)maxc + 2, 0);
    for (long u = 0; u < NU; u++) if (ucnt[(size_t)u] > 0) start[(size_t)ucnt[(size_t)u]]++;
    { long acc = 0; for (int c = maxc; c >= 1; c--) { const long m = start[(size_t)c]; start[(size_t)c] = acc; acc += m; } }
    std::vector<WinUser> urec((size_t)nact);
    for (long u = 0; u < NU; u++) {
        const int c = ucnt[(size_t)u];
        if (c > 0) urec[(size_t)start[(size_t)c]++] = WinUser{(unsigned)u, 0, c, 0};
    }
    std::vector<int> ubegin((size_t)NU, 0);
    { long acc = 0; for (long j = 0; j < nact; j++) { urec[(size_t)j].begin = (int)acc; ubegin[urec[(size_t)j].user] = (int)acc; acc += urec[(size_t)j].count; } }
    std::vector<unsigned> w_item((size_t)n), w_item1(pairs ? (size_t)n : 0);
    std::vector<float> w_label(pairs ? 0 : (size_t)n), w_v0(pairs ? (size_t)n : 0), w_v1(pairs ? (size_t)n : 0);
    std::vector<int> w_slot((size_t)n), w_slot1(pairs ? (size_t)n : 0), icur(iptr.begin(), iptr.end() - 1);
    for (long r = 0; r < n; r++) {   // file order: a user's instances and an item's slots both keep it
        const int at = ubegin[user[r]]++;
        if (!pairs) {
            w_item[(size_t)at] = item[r];
            w_label[(size_t)at] = label[r];
            w_slot[(size_t)at] = icur[item[r]]++;
        } else {   // entry 0 = the lower item id (the merged row is index sorted), the negative's sign flipped
            const bool pf = item[r] < neg[r];
            const unsigned lo = pf ? item[r] : neg[r], hi = pf ? neg[r] : item[r];
            w_item[(size_t)at] = lo; w_item1[(size_t)at] = hi;
            w_v0[(size_t)at] = pf ? 1.0f : -1.0f; w_v1[(size_t)at] = pf ? -1.0f : 1.0f;
            w_slot[(size_t)at] = icur[lo]++; w_slot1[(size_t)at] = icur[hi]++;
        }
    }
    ds->win_urec.upload(urec.data(), (size_t)nact, stream_);
    ds->item.upload(w_item.data(), (size_t)n, stream_);
    ds->win_slot.upload(w_slot.data(), (size_t)n, stream_);
    ds->win_iptr.upload(iptr.data(), (size_t)NI + 1, stream_);
    if (!pairs) {
        ds->label.upload(w_label.data(), (size_t)n, stream_);
        ds->win_item1.release();
    } else {
        ds->win_item1.upload(w_item1.data(), (size_t)n, stream_);
        ds->win_slot1.upload(w_slot1.data(), (size_t)n, stream_);
        ds->ival.upload(w_v0.data(), (size_t)n, stream_);
        ds->win_ival1.upload(w_v1.data(), (size_t)n, stream_);
    }
    HIPCHECK(hipStreamSynchronize(stream_));   // the host columns go out of scope
    ds->sched_signature = schedule_signature();   // a data set refilled in place (the staged path of an amd:gpus handle) is valid under the CURRENT configuration
    ds->win_item_lo = NI; ds->win_item_hi = -1;    // the item ids the window touches: window_delta_apply_local checks them against the active block
    for (long i = 0; i < NI; i++) if (iptr[(size_t)i + 1] > iptr[(size_t)i]) { if (ds->win_item_lo == NI) ds->win_item_lo = i; ds->win_item_hi = i; }
    ds->unit_values = true;
    ds->num_units = nact;
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = n;
    const long nrow_touched = pairs ? 3 : 2, nb = (mp_.no_user_bias ? 0 : 1) + (pairs ? 2 : 1);
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * nrow_touched + 8 * nb + 16 + 8 * nrow_touched);   // SURVEY 8(d4), what the reference's step moves per instance
}
// The same arrays from the device (svdf_k_wbuild.hip): the window's columns go up as they are, three stable sorts and two scans regroup them
// in HBM.  false = not taken (host-only handle, knob device_window = 0, an empty window): the host builder above runs.
bool Engine::window_build_device(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg) {
    if (host_only_ || !device_window_ || n <= 0) return false;
    need_device("dataset");
    const bool pairs = neg != nullptr;
    wb_user_.upload(user, (size_t)n, stream_);
    wb_item_.upload(item, (size_t)n, stream_);
    if (pairs) wb_neg_.upload(neg, (size_t)n, stream_); else wb_label_.upload(label, (size_t)n, stream_);
    window_build_resident(ds, n, wb_user_.p, wb_item_.p, pairs ? nullptr : wb_label_.p, pairs ? wb_neg_.p : nullptr);
    return true;
}
// the columns already in HBM (a whole data set handed over in one copy, its windows built from slices: wseq_from_triples / _pairs).
// pairs: d_label == nullptr, d_neg != nullptr.  The caller has set the window's header fields (window_build).
void Engine::window_build_resident(Dataset *ds, long n, const unsigned *d_user, const unsigned *d_item, const float *d_label, const unsigned *d_neg) {
    const long NU = mp_.num_user, NI = mp_.num_item;
    const bool pairs = d_neg != nullptr;
    const long E = pairs ? 2 * n : n;
    wb_k0_.reserve((size_t)E); wb_k1_.reserve((size_t)E); wb_v0_.reserve((size_t)E); wb_v1_.reserve((size_t)E);
    wb_inst_.reserve((size_t)n); wb_slot_e_.reserve((size_t)E); wb_head_.reserve((size_t)n); wb_mark_.reserve((size_t)n);
    wb_run_user_.reserve((size_t)n); wb_run_start_.reserve((size_t)n); wb_run_begin_.reserve((size_t)n);
    wb_state_.reserve(8);
    const size_t tb = wbuild_tmp_bytes(E);
    wb_tmp_.reserve(std::max<size_t>(tb, 1));
    ds->win_urec.reserve((size_t)std::min<long>(n, std::max<long>(NU, 1)));
    ds->item.reserve((size_t)n); ds->win_slot.reserve((size_t)n); ds->win_iptr.reserve((size_t)NI + 1);
    if (!pairs) { ds->label.reserve((size_t)n); ds->win_item1.release(); }
    else { ds->win_item1.reserve((size_t)n); ds->win_slot1.reserve((size_t)n); ds->ival.reserve((size_t)n); ds->win_ival1.reserve((size_t)n); }
    WBuildIn in{n, pairs ? 1 : 0, d_user, d_item, d_neg, d_label, NU, NI};
    WBuildBuffers B{wb_k0_.p, wb_k1_.p, wb_v0_.p, wb_v1_.p, wb_inst_.p, wb_slot_e_.p, wb_head_.p, wb_mark_.p, wb_run_user_.p, wb_run_start_.p, wb_run_begin_.p,
                    wb_tmp_.p, tb, wb_state_.p};
    WBuildOut out{ds->win_urec.p, ds->item.p, pairs ? ds->win_item1.p : nullptr, pairs ? nullptr : ds->label.p, pairs ? ds->ival.p : nullptr,
                  pairs ? ds->win_ival1.p : nullptr, ds->win_slot.p, pairs ? ds->win_slot1.p : nullptr, ds->win_iptr.p};
    long nact = 0, lo = 0, hi = -1;
    try {
        device_window_build(in, B, out, &nact, &lo, &hi, stream_);
    } catch (const std::runtime_error &e) {
        fail(e.what());
    }
    ds->sched_signature = schedule_signature();
    ds->win_item_lo = lo; ds->win_item_hi = hi;
    ds->unit_values = true;
    ds->num_units = nact;
    ds->sched.level_ptr = {0, n};
    ds->sched.max_level_size = n;
    const long nrow_touched = pairs ? 3 : 2, nb = (mp_.no_user_bias ? 0 : 1) + (pairs ? 2 : 1);
    ds->algorithmic_bytes = n * (8L * mp_.num_factor * nrow_touched + 8 * nb + 16 + 8 * nrow_touched);   // SURVEY 8(d4), as in the host builder
}
void Engine::window_delta_pack(Dataset *ds, void *device_dst, int half, int64_t *count) {
    check(trainer_ready_, "window_delta: init_trainer has not been called");
    check(ds && ds->owner == this && (ds->kind == 5 || ds->kind == 7), "window_delta_pack: not a window data set of this trainer");
    if (ds->kind == 7) {   // user units: the whole replicated side in one piece, [feedback rows | item rows | their biases | global biases]
        check(delta_nparts_ == 1, "window_delta_pack: user-unit window data sets exchange the replicated side in one piece");
        const DeltaRanges R7 = delta_ranges();
        const long T = (user_group() ? (long)num_fb_rows() : 0) + (long)mp_.num_item;
        check(R7.off[R7.n] == T * (pitch_ + 1) + (long)mp_.num_global, "window_delta_pack: unexpected layout of the replicated ranges");
        if (count) *count = R7.off[R7.n];
        if (!device_dst) return;
        need_device("window_delta");
        check(window_trained_ == ds, "window_delta_pack: train this window data set first (svdf_train_dataset)");
        wunit_sum(ds, device_dst, half);
        HIPCHECK(hipGetLastError());
        n_launches_++;
        return;
    }
    check(!relaxed() && g_stride_ == 1 && user_off_ == 0, "window_delta_pack: random-order trainers without relaxed ids only");
    const long ni = mp_.num_item;
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    const long nglobal = (delta_nparts_ == 1 || delta_part_ == 0) ? (long)mp_.num_global : 0;
    const DeltaRanges R = delta_ranges();
    check(R.off[R.n] == (hi - lo) * (pitch_ + 1) + nglobal, "window_delta_pack: unexpected layout of the replicated ranges");
    if (count) *count = R.off[R.n];
    if (!device_dst) return;
    need_device("window_delta");
    check(window_trained_ == ds, "window_delta_pack: train this window data set first (svdf_train_dataset)");
    launch_window_items(window_view(ds), pitch_, mp_.num_factor, lo, hi, nglobal, device_dst, half, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
// Stratified schedule: the rank owns the active item block (svdf_item_delta_select) exclusively while it trains
// a stratum, so the window's per-item sums go straight into the model -- no wire buffer, no sum over ranks.
void Engine::window_delta_apply_local(Dataset *ds) {
    check(ds && ds->owner == this && (ds->kind == 5 || ds->kind == 7), "window_delta_apply_local: not a window data set of this trainer");
    need_device("window_delta");
    check(window_trained_ == ds, "window_delta_apply_local: train this window data set first (svdf_train_dataset)");
    if (ds->kind == 7) {   // user units: every per-target sum of the window, added in place (one rank holds the whole replicated side)
        check(delta_nparts_ == 1, "window_delta_apply_local: user-unit window data sets apply the replicated side in one piece");
        wunit_sum(ds, nullptr, 0);
        HIPCHECK(hipGetLastError());
        n_launches_++;
        window_trained_ = nullptr;   // the sums are in the model: applying them twice would be a silent error
        return;
    }
    check(!relaxed() && g_stride_ == 1 && user_off_ == 0, "window_delta_apply_local: random-order trainers without relaxed ids only");
    const long ni = mp_.num_item;
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    check(ds->win_item_hi < 0 || (ds->win_item_lo >= lo && ds->win_item_hi < hi),
          "window_delta_apply_local: the window holds instances of items outside the active item block (svdf_item_delta_select): their updates would be lost");
    launch_window_items_local(window_view(ds), pitch_, mp_.num_factor, lo, hi, dW_.p + (size_t)item_off_ * pitch_, dbias_.p + item_off_, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
void Engine::item_block_copy(float *device_buf, int set, int64_t *count) {
    check(trainer_ready_, "item_block: init_trainer has not been called");
    const DeltaRanges R = delta_ranges();
    if (count) *count = R.off[R.n];
    if (!device_buf) return;
    need_device("item_block");
    flush();
    launch_ranges_copy(R, device_buf, set, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}
// One stratum step of the stratified schedule in ONE call: every window data set trained and summed in place into the active
// item block, then (device_out != nullptr) the block copied out for its hand-over -- the sequence multi_gpu.StratifiedTrainer issued as
// 4 + 4 W calls.  The host thread has ~40 us per step at N = 8; the calls were a quarter of that.
void Engine::stratum_step(Dataset *const *ds, int n, int block, int nblocks, float *device_out) {
    check(n >= 0 && (n == 0 || ds != nullptr), "stratum_step: bad window list");
    for (int w = 0; w < n; w++) {
        item_delta_select(0, 1);
        train_dataset(ds[w]);
        item_delta_select(block, nblocks);
        window_delta_apply_local(ds[w]);
    }
    if (device_out) {
        item_delta_select(block, nblocks);
        item_block_copy(device_out, 0, nullptr);
    }
    item_delta_select(0, 1);
}
void Engine::item_block_set_at(int block, int nblocks, const float *device_src) {
    item_delta_select(block, nblocks);
    item_block_copy(const_cast<float *>(device_src), 1, nullptr);
    item_delta_select(0, 1);
}
void Engine::window_delta_apply(const void *device_src, int half) {
    need_device("window_delta");
    flush();
    launch_delta_addto(delta_ranges(), device_src, half, stream_);
    HIPCHECK(hipGetLastError());
    n_launches_++;
}

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
    if (ds->kind == 5) {   // the trainer-owned contribution scratch is sized BEFORE anything is issued (and never inside a stream capture)
        d_contrib_.reserve((size_t)ds->win_slots * (size_t)pitch_);
        d_cbias_.reserve((size_t)ds->win_slots);
        window_trained_ = ds;
    }
    const DevParams &P = params();
    const Schedule &sc = ds->sched;
    check(!lazy_decay() || ds->kind == 1 || ds->kind == 4 || (ds->kind == 3 && ds->num_simple_units == 0),
          "train_dataset: the dataset was scheduled before lazy decay (reg_method/reg_global >= 4) was selected");
    if (ds->kind == 2 && chain_width_ > 0 && !ds->d_level_ptr_ok && sc.num_levels() >= 4) {   // the level boundaries for chained launches: once, outside any capture
        ds->d_level_ptr.upload(sc.level_ptr.data(), sc.level_ptr.size(), stream_);
        HIPCHECK(hipStreamSynchronize(stream_));
        ds->d_level_ptr_ok = true;
    }
    auto issue = [&]() {
        if (ds->kind == 0) {
            BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
            for (size_t l = 0; l < sc.num_levels(); l++) launch_basicmf(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
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
            if (fewrow_gslots_ && fewrow_fast_ && fewrow_gslots_applies(P, S, ds->fused.max_nu, ds->fused.max_ni, ds->fused.dense_slots)) {
                for (size_t l = 0; l < sc.num_levels(); l++) launch_fewrow_gslots(P, S, sc.level_ptr[l], sc.level_ptr[l + 1], stream_);
            } else {
                // runs of NARROW levels (a rank pass in file order: tens of thousands of levels of a few dozen pairs) go through one launch per
                // run -- one workgroup walks the levels with a barrier in between (k_fewrow_slots_chain) --, everything else level by level
                const size_t L = sc.num_levels();
                const long cw = chain_width_;
                const bool can_chain = cw > 0 && ds->d_level_ptr_ok && launch_fewrow_chain(P, S, ds->fused.max_nu, ds->fused.max_ni, nullptr, 0, 0, stream_);
                for (size_t l = 0; l < L;) {
                    size_t e = l;
                    if (can_chain) while (e < L && sc.level_ptr[e + 1] - sc.level_ptr[e] <= cw) e++;
                    if (e >= l + 4) {
                        (void)launch_fewrow_chain(P, S, ds->fused.max_nu, ds->fused.max_ni, ds->d_level_ptr.p, (long)l, (long)e, stream_);
                        n_chained_levels_ += (int64_t)(e - l);
                        l = e;
                        continue;
                    }
                    const size_t stop = std::max(e, l + 1);
                    for (; l < stop; l++)
                        launch_fused(P, S, ds->fused.max_nu, ds->fused.max_ni, sc.level_ptr[l], sc.level_ptr[l + 1], groups_per_wave_, block_threads_, stream_);
                }
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
    n_launches_ += (int64_t)sc.num_levels();
    if (ds->kind < 3) n_kind_[ds->kind] += (int64_t)sc.num_levels();
    n_batches_ += (int64_t)sc.num_levels();
    n_instances_ += ds->num_row;
    sample_counter_ += (unsigned)ds->num_row;
}

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
    } else if (ds->kind == 0 || ds->kind == 2) {
        if (ds->kind == 0) {
            BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
            launch_predict_basic(P, S, n, w_out_.p, stream_);
        } else {
            launch_predict_fused(P, ds->fused.view(), ds->fused.max_nu, ds->fused.max_ni, n, w_out_.p, stream_);
        }
        // back into the caller's instance order on the device (a host scatter of 1e8 predictions costs more than the scoring),
        // then one copy out; a host-built schedule's order goes to HBM once
        if (!ds->order_dev.p) ds->order_dev.upload(ds->sched.order.data(), (size_t)n, stream_);
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
    } else if (ds->kind == 0) {
        BasicSchedule S{ds->user.p, ds->item.p, ds->label.p, ds->unit_values ? nullptr : ds->uval.p, ds->unit_values ? nullptr : ds->ival.p};
        launch_predict_basic(P, S, n, w_out_.p, stream_);
        labels = ds->label.p;   // same (level) order as the predictions
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

// =============================================================================== item-side delta (multi-GPU)
std::vector<Engine::Range> Engine::shared_ranges() {
    check(mp_.common_latent_space == 0, "svdfeature_amd: user sharding needs separate user and item spaces");
    std::vector<Range> r;
    if (user_off_ > 0) r.push_back({dW_.p, (long)user_off_ * pitch_});
    r.push_back({dW_.p + (size_t)item_off_ * pitch_, (long)(n_uiset_ - item_off_) * pitch_});
    if (user_off_ > 0) r.push_back({dbias_.p, (long)user_off_});
    r.push_back({dbias_.p + item_off_, (long)(n_uiset_ - item_off_)});
    if (mp_.num_global > 0) r.push_back({dg_.p, (long)mp_.num_global * g_stride_});   // (padding floats stay 0: zero deltas)
    return r;
}
void Engine::item_delta_begin() {
    check(trainer_ready_, "item_delta: init_trainer has not been called");
    need_device("item_delta");
    flush();
    item_delta_begin_local();
}
void Engine::item_delta_begin_local() {
    need_device("item_delta");
    auto rg = shared_ranges();
    long total = 0;
    for (auto &x : rg) total += x.n;
    d_snap_.reserve((size_t)total);
    d_delta_.reserve((size_t)total);
    long off = 0;
    for (auto &x : rg) {
        HIPCHECK(hipMemcpyAsync(d_snap_.p + off, x.base, (size_t)x.n * sizeof(float), hipMemcpyDeviceToDevice, stream_));
        off += x.n;
    }
}
void *Engine::item_delta_buffer(int64_t *count) {
    need_device("item_delta");
    flush();
    auto rg = shared_ranges();
    long off = 0;
    for (auto &x : rg) {
        launch_delta_sub(x.base, d_snap_.p + off, d_delta_.p + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
    if (count) *count = off;
    return d_delta_.p;
}
void Engine::item_delta_apply() {
    need_device("item_delta");
    auto rg = shared_ranges();
    long off = 0;
    for (auto &x : rg) {
        launch_delta_add(x.base, d_snap_.p + off, d_delta_.p + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
}

void Engine::item_delta_into(float *device_dst, int64_t *count) {
    need_device("item_delta");
    flush();
    long off = 0;
    for (auto &x : shared_ranges()) {
        launch_delta_sub(x.base, d_snap_.p + off, device_dst + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
    if (count) *count = off;
}
void Engine::item_delta_apply_from(const float *device_src) {
    need_device("item_delta");
    long off = 0;
    for (auto &x : shared_ranges()) {
        launch_delta_add(x.base, d_snap_.p + off, device_src + off, x.n, stream_);
        off += x.n;
    }
    HIPCHECK(hipGetLastError());
}
// Replicated ranges of the ACTIVE exchange partition (svdf_item_delta_select): the item rows are cut into nparts id ranges so
// that a window's exchange can be split into pieces that overlap with training on the other pieces' instances; everything that
// is not indexed by item id (feedback rows, global biases) travels with partition 0.  snap_off addresses the full snapshot.
DeltaRanges Engine::delta_ranges() {
    DeltaRanges R;
    memset(&R, 0, sizeof(R));
    auto rg = shared_ranges();
    check(rg.size() <= SVDF_MAX_DELTA_RANGES, "item_delta: too many replicated ranges");
    // shared_ranges(): [W_fb] W_item [bias_fb] bias_item [g_bias]; item ranges are the ones starting at item_off_
    const long ni = (long)(n_uiset_ - item_off_);
    const long lo = ni * delta_part_ / delta_nparts_, hi = ni * (delta_part_ + 1) / delta_nparts_;
    long off = 0, snap = 0;
    int n = 0;
    for (size_t q = 0; q < rg.size(); q++) {
        const bool is_w_item = rg[q].base == dW_.p + (size_t)item_off_ * pitch_;
        const bool is_b_item = rg[q].base == dbias_.p + item_off_;
        if (delta_nparts_ > 1 && (is_w_item || is_b_item)) {
            const long unit = is_w_item ? pitch_ : 1;
            R.base[n] = rg[q].base + lo * unit; R.off[n] = off; R.snap_off[n] = snap + lo * unit;
            off += (hi - lo) * unit; n++;
        } else if (delta_nparts_ == 1 || delta_part_ == 0) {
            R.base[n] = rg[q].base; R.off[n] = off; R.snap_off[n] = snap;
            off += rg[q].n; n++;
        }
        snap += rg[q].n;
    }
    R.n = n;
    for (int q = n; q <= SVDF_MAX_DELTA_RANGES; q++) R.off[q] = off;
    return R;
}
void Engine::item_delta_select(int part, int nparts) {
    check(nparts >= 1 && part >= 0 && part < nparts, "item_delta_select: bad partition");
    delta_part_ = part; delta_nparts_ = nparts;
}
void Engine::item_delta_pack(void *device_dst, int half, int64_t *count) {
    check(trainer_ready_, "item_delta: init_trainer has not been called");
    if (device_dst) need_device("item_delta");
    const DeltaRanges R = delta_ranges();
    if (count) *count = R.off[R.n];
    if (!device_dst) return;   // size query
    check(d_snap_.p != nullptr && (long)d_snap_.cap >= R.off[R.n], "item_delta: call item_delta_begin first");
    flush();
    launch_delta_pack(R, d_snap_.p, device_dst, half, stream_);
    HIPCHECK(hipGetLastError());
}
void Engine::item_delta_unpack(const void *device_src, int half, int refresh_snapshot) {
    need_device("item_delta");
    const DeltaRanges R = delta_ranges();
    check(d_snap_.p != nullptr && (long)d_snap_.cap >= R.off[R.n], "item_delta: call item_delta_begin first");
    launch_delta_unpack(R, d_snap_.p, device_src, half, refresh_snapshot, stream_);
    HIPCHECK(hipGetLastError());
}
void Engine::set_stream(hipStream_t s) {
    need_device("set_stream");
    flush();
    HIPCHECK(hipStreamSynchronize(stream_));
    if (owns_stream_ && stream_) (void)hipStreamDestroy(stream_);
    stream_ = s;
    owns_stream_ = false;
}
void Engine::item_delta_copy(float *device_dst, const float *device_src) {
    need_device("item_delta");
    long total = 0;
    for (auto &x : shared_ranges()) total += x.n;
    check(d_delta_.p != nullptr && (long)d_delta_.cap >= total, "item_delta: call item_delta_begin / item_delta_buffer first");
    if (device_dst) HIPCHECK(hipMemcpyAsync(device_dst, d_delta_.p, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    if (device_src) HIPCHECK(hipMemcpyAsync(d_delta_.p, device_src, (size_t)total * sizeof(float), hipMemcpyDeviceToDevice, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
}

// =============================================================================== introspection
void Engine::view_shape(int which, int *rows, int *cols) {
    *rows = -1; *cols = 0;
    if (!space_allocated_) return;
    switch (which) {
    case 0: *rows = mp_.num_user; *cols = 1; break;
    case 1: *rows = mp_.num_user; *cols = mp_.num_factor; break;
    case 2: *rows = mp_.num_item; *cols = 1; break;
    case 3: *rows = mp_.num_item; *cols = mp_.num_factor; break;
    case 4: *rows = mp_.num_global; *cols = 1; break;
    case 5: if (user_group()) { *rows = num_fb_rows(); *cols = 1; } break;
    case 6: if (user_group()) { *rows = num_fb_rows(); *cols = mp_.num_factor; } break;
    default: break;
    }
}
int64_t Engine::get_view(int which, float *out, int64_t capacity) {
    int rows, cols;
    view_shape(which, &rows, &cols);
    if (rows < 0) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (n > capacity) return -1;
    if (n == 0) return 0;
    const bool matrix = (which == 1 || which == 3 || which == 6);
    const unsigned off = (which <= 1) ? user_off_ : (which <= 3) ? item_off_ : fb_off_;
    if (device_model_ && multi_ && which <= 1) {   // user rows live on their owners
        flush();
        multi_gather_user_rows();
        download_model();
        if (!matrix) memcpy(out, hbias_.data() + off, (size_t)n * sizeof(float));
        else for (int y = 0; y < rows; y++) memcpy(out + (size_t)y * cols, hW_.data() + ((size_t)off + y) * pitch_, (size_t)cols * sizeof(float));
        hW_.clear(); hW_.shrink_to_fit(); hbias_.clear(); hg_.clear(); host_model_valid_ = false;
        return n;
    }
    if (device_model_) {
        flush();
        if (which == 4) download_globals(out);
        else if (!matrix) HIPCHECK(hipMemcpyAsync(out, dbias_.p + off, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, stream_));
        else HIPCHECK(hipMemcpy2DAsync(out, (size_t)cols * sizeof(float), dW_.p + (size_t)off * pitch_, (size_t)pitch_ * sizeof(float),
                                       (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else {
        check(host_model_valid_, "get_view: no model");
        if (which == 4) memcpy(out, hg_.data(), (size_t)n * sizeof(float));
        else if (!matrix) memcpy(out, hbias_.data() + off, (size_t)n * sizeof(float));
        else for (int y = 0; y < rows; y++) memcpy(out + (size_t)y * cols, hW_.data() + ((size_t)off + y) * pitch_, (size_t)cols * sizeof(float));
    }
    return n;
}
// overwrite a parameter view from rows*cols unpadded floats (multi-GPU: gathering the owners' user rows before a save)
int64_t Engine::set_view(int which, const float *in, int64_t count) {
    int rows, cols;
    view_shape(which, &rows, &cols);
    if (rows < 0) return -1;
    const int64_t n = (int64_t)rows * cols;
    if (n != count) return -1;
    if (n == 0) return 0;
    check(!multi_, "set_view: not available with amd:gpus > 1");
    const bool matrix = (which == 1 || which == 3 || which == 6);
    const unsigned off = (which <= 1) ? user_off_ : (which <= 3) ? item_off_ : fb_off_;
    if (device_model_) {
        flush();
        if (which == 4) {
            std::vector<float> keep;
            keep.assign(in, in + n);
            std::swap(keep, hg_);
            upload_globals(g_stride_);
            HIPCHECK(hipStreamSynchronize(stream_));
            std::swap(keep, hg_);
            return n;
        }
        if (!matrix) HIPCHECK(hipMemcpyAsync(dbias_.p + off, in, (size_t)n * sizeof(float), hipMemcpyHostToDevice, stream_));
        else HIPCHECK(hipMemcpy2DAsync(dW_.p + (size_t)off * pitch_, (size_t)pitch_ * sizeof(float), in, (size_t)cols * sizeof(float),
                                       (size_t)cols * sizeof(float), (size_t)rows, hipMemcpyHostToDevice, stream_));
        HIPCHECK(hipStreamSynchronize(stream_));
    } else {
        check(host_model_valid_, "set_view: no model");
        if (which == 4) memcpy(hg_.data(), in, (size_t)n * sizeof(float));
        else if (!matrix) memcpy(hbias_.data() + off, in, (size_t)n * sizeof(float));
        else for (int y = 0; y < rows; y++) memcpy(hW_.data() + ((size_t)off + y) * pitch_, in + (size_t)y * cols, (size_t)cols * sizeof(float));
    }
    return n;
}
void Engine::synchronize() {
    if (host_only_) return;
    if (multi_) { multi_synchronize(); return; }
    HIPCHECK(hipStreamSynchronize(stream_));
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
    default: return -1;
    }
}
int Engine::set_knob(const char *name, long value) {
    launch_version_++;   // any knob may change what a captured pass would launch
    // tuning knobs reach every rank of an amd:gpus handle (they never change a result; the exchange window is the handle's own)
    if (multi_ && !is_peer_ && strcmp(name, "stage_window") != 0 && strcmp(name, "async_flush") != 0)
        for (int d = 1; d < gpus_; d++) (void)rank_engine(d)->set_knob(name, value);
    if (!strcmp(name, "use_graph")) { use_graph_ = value != 0; return 0; }
    if (!strcmp(name, "graph_min_levels")) { check(value >= 1, "graph_min_levels must be >= 1"); graph_min_levels_ = (int)value; return 0; }
    if (!strcmp(name, "stage_window")) { check(value >= 1, "stage_window must be >= 1"); stage_window_ = value; window_set_ = true; return 0; }
    if (!strcmp(name, "groups_per_wave")) {
        check(value >= 0 && value <= 8 && value != 7, "groups_per_wave must be 0 (auto), 1 ... 6 or 8");
        groups_per_wave_ = (int)value;
        return 0;
    }
    if (!strcmp(name, "xcd_remap")) { xcd_remap_ = value != 0; params_dirty_ = true; return 0; }
    if (!strcmp(name, "hot_reduce")) { hot_reduce_ = value != 0; params_dirty_ = true; return 0; }
    if (!strcmp(name, "fewrow_i16")) { check(value >= 0 && value <= 1, "fewrow_i16 must be 0 or 1"); fewrow_i16_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "small_blocks")) { check(value == 0 || value == 1, "small_blocks must be 0 or 1"); small_blocks_ = (int)value; params_dirty_ = true; return 0; }
    if (!strcmp(name, "svdpp_xunits")) { check(value == 0 || value == 1, "svdpp_xunits must be 0 or 1"); svdpp_xunits_ = (int)value; return 0; }
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
    if (!strcmp(name, "chain_width")) { check(value >= 0, "chain_width must not be negative"); chain_width_ = value; return 0; }
    if (!strcmp(name, "wseq_build_threads")) { check(value >= 1 && value <= 256, "wseq_build_threads must be in 1 .. 256"); wseq_build_threads_ = (int)value; return 0; }
    if (!strcmp(name, "device_init_margin_log2")) { check(value >= 8 && value <= 52, "device_init_margin_log2 must be in 8 .. 52"); device_init_margin_log2_ = (int)value; return 0; }
    if (!strcmp(name, "fewrow_fast")) { fewrow_fast_ = value != 0; params_dirty_ = true; return 0; }
    if (!strcmp(name, "device_schedule_min")) { check(value >= 1, "device_schedule_min must be >= 1"); device_sched_min_ = value; return 0; }
    if (!strcmp(name, "use_simple_units")) { use_simple_units_ = value != 0; return 0; }
    if (!strcmp(name, "rows_without_feedback")) { rows_without_feedback_ = value != 0; return 0; }
    if (!strcmp(name, "fewrow_gslots")) { fewrow_gslots_ = value != 0; launch_version_++; return 0; }
    if (!strcmp(name, "wunit_inplace")) { wunit_inplace_ = value != 0; return 0; }
    if (!strcmp(name, "wunit_fast")) { check(value >= 0 && value <= 2, "wunit_fast must be 0, 1 or 2"); wunit_fast_ = (int)value; return 0; }
    if (!strcmp(name, "window_per_target_fb")) { check(value >= 1, "window_per_target_fb must be positive"); wseq_per_target_fb_ = (int)value; return 0; }
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
