// svdf_k_wbuild.hip -- window data sets of the window-minibatch step built ON THE DEVICE.
//
// A window of plain ratings or rank pairs (kind 5, Engine::window_build) is two regroupings of the window's instances:
//   * by USER, a user's instances contiguous and in file order (the lane group of k_window_users walks them with the user's row in
//     registers -- the exact sequential part of the step: update_inner, apex_svd_base.h:456-462, on the user's current row), users in
//     launch order: by instance count descending, ties by id;
//   * by ITEM entry: every item entry gets the slot its contribution is stored in, item by item and in FILE order inside an item
//     (k_window_items adds them in that order -- what the reference's instance-by-instance updates of the row become, :383-427).
// The host does this with counting sorts over cache-missing scatters: 16 ns per instance, 1.6 s for the 100 M ratings of BASELINE
// configs[2] -- a hundred passes of the step it prepares.  Here it is three stable radix sorts (rocPRIM, a library primitive like the
// sorts of svdf_k_sched.hip), two scans and a handful of gather kernels per window; the arrays are identical to the host builder's.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <stdexcept>
#include <string>

#include "svdf_kernels.h"

namespace svdf {

namespace {

enum { WB_ERR = 0, WB_NACT = 1, WB_LO = 2, WB_HI = 3, WB_WORDS = 8 };
enum { WB_ERR_USER = 1, WB_ERR_ITEM = 2, WB_ERR_SAME = 4 };

inline int wb_bits(unsigned long long v) {
    int b = 1;
    while (b < 64 && (v >> b) != 0ull) b++;
    return b;
}
inline unsigned wb_grid(long n) {
    long g = (n + 255) / 256;
    return (unsigned)std::max<long>(1, std::min<long>(g, 1L << 20));
}

// item entries: ratings one per instance; pairs two, entry 0 = the lower id (the merged row is index sorted, apex_svd_data.cpp:828-860)
__global__ __launch_bounds__(256) void k_wb_item_keys(long n, int pairs, const unsigned *item, const unsigned *neg, unsigned NI, unsigned *keys, unsigned *vals,
                                                      unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long E = pairs ? 2 * n : n;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < E; e += stride) {
        unsigned id;
        if (!pairs) {
            id = item[e];
        } else {
            const unsigned a = item[e >> 1], b = neg[e >> 1];
            if (b >= NI) atomicOr(&state[WB_ERR], (unsigned)WB_ERR_ITEM);
            if (a == b) atomicOr(&state[WB_ERR], (unsigned)WB_ERR_SAME);
            id = (e & 1) ? (a > b ? a : b) : (a < b ? a : b);
        }
        if (id >= NI) { atomicOr(&state[WB_ERR], (unsigned)WB_ERR_ITEM); id = NI ? NI - 1 : 0; }
        keys[e] = id;
        vals[e] = (unsigned)e;
    }
}
// iptr[i] = number of entries with an item id below i (lower bound in the sorted keys)
__global__ __launch_bounds__(256) void k_wb_iptr(const unsigned *sorted, long E, long NI, int *iptr, unsigned *state) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > NI) return;
    long lo = 0, hi = E;
    while (lo < hi) {
        const long mid = (lo + hi) >> 1;
        if ((long)sorted[mid] < i) lo = mid + 1; else hi = mid;
    }
    iptr[i] = (int)lo;
    if (i == 0 && E > 0) { state[WB_LO] = sorted[0]; state[WB_HI] = sorted[E - 1]; }
}
__global__ __launch_bounds__(256) void k_wb_slots(const unsigned *sorted, const unsigned *ent, long E, const int *iptr, int *slot_e) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < E; p += stride) slot_e[ent[p]] = (int)p;   // slots are global over the window: item by item
    (void)sorted; (void)iptr;
}
__global__ __launch_bounds__(256) void k_wb_user_keys(long n, const unsigned *user, unsigned NU, unsigned *keys, unsigned *vals, unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long r = (long)blockIdx.x * blockDim.x + threadIdx.x; r < n; r += stride) {
        unsigned u = user[r];
        if (u >= NU) { atomicOr(&state[WB_ERR], (unsigned)WB_ERR_USER); u = NU ? NU - 1 : 0; }
        keys[r] = u;
        vals[r] = (unsigned)r;
    }
}
__global__ __launch_bounds__(256) void k_wb_heads(const unsigned *ku, long n, int *head) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) head[p] = (p == 0 || ku[p] != ku[p - 1]) ? 1 : 0;
}
// mark[p] = 1-based run index of sorted position p (inclusive scan of the heads)
__global__ __launch_bounds__(256) void k_wb_runs(const unsigned *ku, const int *mark, long n, int *run_user, int *run_start, unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        if (p == 0 || ku[p] != ku[p - 1]) { const int j = mark[p] - 1; run_user[j] = (int)ku[p]; run_start[j] = (int)p; }
        if (p == n - 1) state[WB_NACT] = (unsigned)mark[p];
    }
}
// launch order: by count descending (key = ~count ascending), ties by user id (the runs are in id order and the sort is stable); runs past
// nact pad the array with count 0
__global__ __launch_bounds__(256) void k_wb_run_keys(const int *run_start, long n, const unsigned *state, unsigned *keys, unsigned *vals) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long nact = (long)state[WB_NACT];
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) {
        unsigned cnt = 0;
        if (j < nact) cnt = (unsigned)((j + 1 < nact ? (long)run_start[j + 1] : n) - (long)run_start[j]);
        keys[j] = ~cnt;
        vals[j] = (unsigned)j;
    }
}
__global__ __launch_bounds__(256) void k_wb_counts(const unsigned *keys_sorted, long n, unsigned *cnt) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) cnt[j] = ~keys_sorted[j];
}
__global__ __launch_bounds__(256) void k_wb_urec(const unsigned *keys_sorted, const unsigned *run_sorted, const unsigned *begin, const int *run_user, long n,
                                                 const unsigned *state, WinUser *urec, int *run_begin) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long nact = (long)state[WB_NACT];
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nact && q < n; q += stride) {
        const unsigned j = run_sorted[q];
        urec[q] = WinUser{(unsigned)run_user[j], (int)begin[q], (int)(~keys_sorted[q]), 0};
        run_begin[j] = (int)begin[q];
    }
}
// the regrouped columns: sorted position p (user-major, file order inside a user) -> its place in launch order
__global__ __launch_bounds__(256) void k_wb_place(long n, int pairs, const unsigned *inst, const int *mark, const int *run_start, const int *run_begin,
                                                  const unsigned *item, const unsigned *neg, const float *label, const int *slot_e, unsigned *w_item,
                                                  unsigned *w_item1, float *w_label, float *w_v0, float *w_v1, int *w_slot, int *w_slot1, const int *iptr) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const int j = mark[p] - 1;
        const long at = (long)run_begin[j] + (p - (long)run_start[j]);
        const long r = (long)inst[p];
        if (!pairs) {
            w_item[at] = item[r];
            w_label[at] = label[r];
            w_slot[at] = slot_e[r];
        } else {
            const unsigned a = item[r], b = neg[r];
            const bool pf = a < b;
            w_item[at] = pf ? a : b; w_item1[at] = pf ? b : a;
            w_v0[at] = pf ? 1.0f : -1.0f; w_v1[at] = pf ? -1.0f : 1.0f;
            w_slot[at] = slot_e[2 * r]; w_slot1[at] = slot_e[2 * r + 1];
        }
    }
    (void)iptr;
}

#define WCHK(call)                                                                                                    \
    do {                                                                                                              \
        hipError_t e_ = (call);                                                                                       \
        if (e_ != hipSuccess) throw std::runtime_error(std::string("device window build: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)

}  // namespace

size_t wbuild_tmp_bytes(long m) {
    size_t a = 0, b = 0, c = 0;
    WCHK(rocprim::radix_sort_pairs(nullptr, a, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr, (unsigned *)nullptr, (size_t)m, 0u, 32u, hipStream_t(0)));
    WCHK(rocprim::inclusive_scan(nullptr, b, (int *)nullptr, (int *)nullptr, (size_t)m, rocprim::plus<int>(), hipStream_t(0)));
    WCHK(rocprim::exclusive_scan(nullptr, c, (unsigned *)nullptr, (unsigned *)nullptr, 0u, (size_t)m, rocprim::plus<unsigned>(), hipStream_t(0)));
    return std::max(a, std::max(b, c));
}

// See svdf_kernels.h.  Everything is enqueued on st; the call returns after ONE small read-back (error flags, active users, item range).
void device_window_build(const WBuildIn &in, const WBuildBuffers &B, const WBuildOut &out, long *nact, long *item_lo, long *item_hi, hipStream_t st) {
    const long n = in.n, E = in.pairs ? 2 * n : n;
    if (n <= 0) throw std::runtime_error("device window build: empty window");
    size_t tb = B.tmp_bytes;
    WCHK(hipMemsetAsync(B.state, 0, WB_WORDS * sizeof(unsigned), st));
    // ---- item side: slots in (item, file order), iptr
    hipLaunchKernelGGL(k_wb_item_keys, dim3(wb_grid(E)), dim3(256), 0, st, n, in.pairs, in.item, in.neg, (unsigned)in.num_item, B.k0, B.v0, B.state);
    WCHK(rocprim::radix_sort_pairs(B.tmp, tb, B.k0, B.k1, B.v0, B.v1, (size_t)E, 0u, (unsigned)wb_bits((unsigned long long)std::max<long>(in.num_item, 1)), st));
    hipLaunchKernelGGL(k_wb_iptr, dim3((unsigned)((in.num_item + 1 + 255) / 256)), dim3(256), 0, st, B.k1, E, in.num_item, out.iptr, B.state);
    hipLaunchKernelGGL(k_wb_slots, dim3(wb_grid(E)), dim3(256), 0, st, B.k1, B.v1, E, out.iptr, B.slot_e);
    // ---- user side: runs of the user-sorted instances, launch order, places
    hipLaunchKernelGGL(k_wb_user_keys, dim3(wb_grid(n)), dim3(256), 0, st, n, in.user, (unsigned)in.num_user, B.k0, B.v0, B.state);
    tb = B.tmp_bytes;
    WCHK(rocprim::radix_sort_pairs(B.tmp, tb, B.k0, B.k1, B.v0, B.inst, (size_t)n, 0u, (unsigned)wb_bits((unsigned long long)std::max<long>(in.num_user, 1)), st));
    hipLaunchKernelGGL(k_wb_heads, dim3(wb_grid(n)), dim3(256), 0, st, B.k1, n, B.head);
    tb = B.tmp_bytes;
    WCHK(rocprim::inclusive_scan(B.tmp, tb, B.head, B.mark, (size_t)n, rocprim::plus<int>(), st));
    hipLaunchKernelGGL(k_wb_runs, dim3(wb_grid(n)), dim3(256), 0, st, B.k1, B.mark, n, B.run_user, B.run_start, B.state);
    hipLaunchKernelGGL(k_wb_run_keys, dim3(wb_grid(n)), dim3(256), 0, st, B.run_start, n, B.state, B.k0, B.v0);
    tb = B.tmp_bytes;
    WCHK(rocprim::radix_sort_pairs(B.tmp, tb, B.k0, B.k1, B.v0, B.v1, (size_t)n, 0u, 32u, st));
    hipLaunchKernelGGL(k_wb_counts, dim3(wb_grid(n)), dim3(256), 0, st, B.k1, n, B.k0);
    tb = B.tmp_bytes;
    WCHK(rocprim::exclusive_scan(B.tmp, tb, B.k0, B.v0, 0u, (size_t)n, rocprim::plus<unsigned>(), st));
    hipLaunchKernelGGL(k_wb_urec, dim3(wb_grid(n)), dim3(256), 0, st, B.k1, B.v1, B.v0, B.run_user, n, B.state, out.urec, B.run_begin);
    hipLaunchKernelGGL(k_wb_place, dim3(wb_grid(n)), dim3(256), 0, st, n, in.pairs, B.inst, B.mark, B.run_start, B.run_begin, in.item, in.neg, in.label, B.slot_e,
                       out.item, out.item1, out.label, out.v0, out.v1, out.slot, out.slot1, out.iptr);
    unsigned hs[WB_WORDS];
    WCHK(hipMemcpyAsync(hs, B.state, sizeof(hs), hipMemcpyDeviceToHost, st));
    WCHK(hipStreamSynchronize(st));
    WCHK(hipGetLastError());
    if (hs[WB_ERR] & WB_ERR_USER) throw std::runtime_error("user feature index exceed bound");
    if (hs[WB_ERR] & WB_ERR_ITEM) throw std::runtime_error("item feature index exceed bound");
    if (hs[WB_ERR] & WB_ERR_SAME) throw std::runtime_error("rank pair: positive and negative item must differ");
    *nact = (long)hs[WB_NACT];
    *item_lo = (long)hs[WB_LO];
    *item_hi = (long)hs[WB_HI];
}

}  // namespace svdf
