// svdf_internal.h -- helpers shared by the translation units of the host engine (svdf_engine / _config / _model / _sched /
// _dataset / _window .cpp).  Not part of the boundary; include/svdfeature_amd.h is.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstdlib>
#include <string>
#include <thread>
#include <vector>

#include "svdf_engine.h"

namespace svdf {

static inline void check(bool ok, const char *msg) { if (!ok) fail(msg); }
#define HIPCHECK(call)                                                                           \
    do {                                                                                         \
        hipError_t e_ = (call);                                                                  \
        if (e_ != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)

// The first real HIP call initialises the ROCm runtime, which disturbs libc's rand() state (tools/check_hip_init_rand.cpp);
// runtime start-up runs on a scratch PRNG state and the caller's state is put back exactly.
struct RandStateGuard {
    char scratch[256];
    char *old;
    RandStateGuard() { old = initstate(1u, scratch, sizeof(scratch)); }
    ~RandStateGuard() { if (old) setstate(old); }
};
struct ScopedNs {   // host-side time accounting (SVDF_PROFILE=1)
    int64_t &acc;
    std::chrono::steady_clock::time_point t0;
    explicit ScopedNs(int64_t &a) : acc(a), t0(std::chrono::steady_clock::now()) {}
    ~ScopedNs() { acc += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count(); }
};

// config keys of SVDTrainParam / SVDModelParam (svdf_config.cpp: one table row per key)
void config_set_train_param(TrainParam &p, const char *name, const char *val);
void config_set_model_param(ModelParam &p, const char *name, const char *val);

// Pointer arrays handed over by a caller are checked before anything indexes through them (the messages of dataset_from_csr /
// dataset_from_blocks): counts not negative, pointers starting at >= 0 and non-decreasing, fewer than 2^31 entries.
void validate_csr_pointers(long num_row, const int64_t *row_ptr);
void validate_block_pointers(long num_block, const int64_t *fb_ptr, const int64_t *block_row_ptr);
// the shapes the user-unit window step takes (svdf_wunit.cpp's builders fail on anything else): one user entry per row, one user per
// block / START..END span, no id twice in a row's global or item entries, no feedback id twice in a block.  Pointers must be valid.
bool wunit_rows_ok(long r0, long r1, const int64_t *row_ptr, const unsigned *feat_index);
bool wunit_blocks_ok(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const int64_t *block_row_ptr,
                     const int64_t *row_ptr, const unsigned *feat_index);

// Instances of one batch commute: sorting a batch by a key changes no bit of the result (svdf_sched.cpp)
void sort_batches(Schedule &sched, const unsigned *key);

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

}  // namespace svdf
