// svdf_k_sched.hip -- conflict-free level scheduling ON THE DEVICE.
//
// The host scheduler (svdf_engine.cpp: level_of_row / touch_row / build_schedule) walks the instance stream once and keeps
// `last[row]`; that is a sequential longest-path computation: 1.1 s for the 100 M ratings of BASELINE configs[1].  Here the
// same levels come out of a frontier peel over the per-row successor links:
//   1. one (row, entry) pair per touched parameter row, entry = unit*K + slot, radix-sorted by row (stable: units of one row
//      stay in file order)                                                                  -- rocPRIM radix_sort_pairs
//   2. neighbours in the sorted array give every entry its successor (the next unit touching the same row) and tell which
//      units head their rows; a unit is READY when it heads all of its rows
//   3. level l = the ready units; retiring a unit promotes its successors (one atomic counter per unit), the units that
//      become ready form level l+1.  One short launch per level, frontier sizes stay on the device (last-block-done hand-over),
//      the host only looks every few hundred levels whether everything is scheduled
//   4. units sorted by (level, batch key, file position) with one more radix sort: the order inside a level is free
//      (its units commute), the key is the item / user id so that neighbouring lane groups walk the factor table in order.
// level(u) = 1 + max(level of the predecessors) is unique, and ties are broken exactly like the host's stable sorts, so the
// resulting order[] and level_ptr[] are IDENTICAL to the host scheduler's (tests/test_gpu_sched.py compares them).
//
// The sort is a library primitive (rocPRIM), like a plain GEMM would be rocBLAS's; everything else is hand-written.
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdexcept>
#include <string>
#include <vector>

#include "svdf_kernels.h"

namespace svdf {

namespace {

struct Cols {
    const unsigned *col[SVDF_SCHED_MAX_SLOTS];
    unsigned off[SVDF_SCHED_MAX_SLOTS];     // resource id = off[slot] + id
    unsigned limit[SVDF_SCHED_MAX_SLOTS];   // ids must be < limit (else the error flag is raised)
    int K;
};

// state words in device memory
enum { ST_CURSOR = 0, ST_BLOCKS_DONE = 1, ST_ERROR = 2, ST_NLEVELS = 3, ST_NEXT_LEVEL = 4, ST_WORDS = 8 };

// remaining[u] starts as the number of rows unit u touches; every row it heads (links kernel) and every retired predecessor
// (peel) takes one off; the unit is ready when it reaches 0 -- one atomic per dependency, no second array to look up
__global__ __launch_bounds__(256) void k_sched_keys(const Cols C, long n, unsigned absent_key, unsigned *keys, unsigned *vals, int *remaining,
                                                    unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) {
        int nd = 0;
        for (int s = 0; s < C.K; s++) {
            const unsigned id = C.col[s][u];
            unsigned key = absent_key;
            if (id != SLOT_ABSENT) {
                if (id >= C.limit[s]) atomicOr(&state[ST_ERROR], 1u << s);
                else { key = C.off[s] + id; nd++; }
            }
            keys[u * C.K + s] = key;
            vals[u * C.K + s] = (unsigned)(u * C.K + s);
        }
        remaining[u] = nd;
    }
}

// sorted (row, entry): successor link of every entry, and one credit for the unit that heads each row
__global__ __launch_bounds__(256) void k_sched_links(const unsigned *keys, const unsigned *vals, long m, unsigned absent_key, int K, int *succ,
                                                     int *remaining) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const unsigned key = keys[j];
        if (key == absent_key) continue;
        const unsigned e = vals[j];
        int nxt = -1;
        if (j + 1 < m && keys[j + 1] == key) nxt = (int)(vals[j + 1] / (unsigned)K);
        succ[e] = nxt;
        if (j == 0 || keys[j - 1] != key) atomicSub(&remaining[e / (unsigned)K], 1);
    }
}

// Appends are staged per WORKGROUP in LDS and flushed with ONE atomic on the shared cursor per round: thousands of same-address
// device-scope atomics per level (one per wave, plus one per workgroup for the hand-over) were what a level cost before --
// 23 us measured with 512 x 256 threads; the peel therefore runs few, large workgroups (PEEL_BLOCKS x 1024).
#define PEEL_THREADS 1024
#define PEEL_STAGE (PEEL_THREADS * SVDF_SCHED_MAX_SLOTS)
struct Stage {
    int items[PEEL_STAGE];
    int count;
    unsigned base;
    bool last;
};
__device__ __forceinline__ void stage_flush(Stage &sh, int *order, unsigned *cursor) {
    __syncthreads();
    if (threadIdx.x == 0 && sh.count > 0) sh.base = atomicAdd(cursor, (unsigned)sh.count);
    __syncthreads();
    for (int j = threadIdx.x; j < sh.count; j += blockDim.x) order[sh.base + (unsigned)j] = sh.items[j];
    __syncthreads();
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
}
// last workgroup of a launch publishes where the next level ends (= the cursor now that every append of this launch is done)
__device__ __forceinline__ void publish_level_end(Stage &sh, unsigned *state, unsigned *level_end, int l) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __threadfence();
        sh.last = atomicAdd(&state[ST_BLOCKS_DONE], 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (sh.last && threadIdx.x == 0) {
        __threadfence();
        const unsigned cur = atomicAdd(&state[ST_CURSOR], 0u);
        level_end[l + 1] = cur;
        if (cur > level_end[l]) state[ST_NLEVELS] = (unsigned)(l + 1);   // level l+1 (1-based) is not empty
        state[ST_BLOCKS_DONE] = 0;
        __threadfence();
    }
}

// level 1: the units that head all of their rows.  level_end[0] = 0, level_end[1] = number of such units
__global__ __launch_bounds__(PEEL_THREADS) void k_sched_seed(long n, const int *remaining, int *order, int *level, unsigned *state,
                                                             unsigned *level_end) {
    __shared__ Stage sh;
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    const long rounds = (n + stride - 1) / stride;
    for (long it = 0; it < rounds; it++) {
        const long u = it * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        if (u < n && remaining[u] == 0) {
            level[u] = 1;
            sh.items[atomicAdd(&sh.count, 1)] = (int)u;
        }
        stage_flush(sh, order, &state[ST_CURSOR]);
    }
    publish_level_end(sh, state, level_end, 0);
}

// level l+1 from level l (l >= 1): order[level_end[l-1] .. level_end[l]) are the units of level l
template <int K>
__global__ __launch_bounds__(PEEL_THREADS) void k_sched_peel(int l, const int *succ, int *remaining, int *order, int *level,
                                                             unsigned *state, unsigned *level_end) {
    __shared__ Stage sh;
    if (threadIdx.x == 0) sh.count = 0;
    __syncthreads();
    const unsigned begin = level_end[l - 1], end = level_end[l];
    const unsigned total = end - begin;
    const unsigned stride = gridDim.x * blockDim.x;
    const unsigned rounds = (total + stride - 1) / stride;
    for (unsigned it = 0; it < rounds; it++) {
        const unsigned i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        if (i < total) {
            const int u = order[begin + i];
            int v[K];
            bool ready[K];
#pragma unroll
            for (int s = 0; s < K; s++) v[s] = succ[(long)u * K + s];
#pragma unroll
            for (int s = 0; s < K; s++) ready[s] = v[s] >= 0 && atomicSub(&remaining[v[s]], 1) == 1;   // K atomics in flight
#pragma unroll
            for (int s = 0; s < K; s++)
                if (ready[s]) { level[v[s]] = l + 1; sh.items[atomicAdd(&sh.count, 1)] = v[s]; }
        }
        stage_flush(sh, order, &state[ST_CURSOR]);
    }
    publish_level_end(sh, state, level_end, l);
}

// Deep, narrow dependency graphs (a rank pass in file order: 58 K levels of ~50 units) are bound by the launch-to-launch
// latency of the one-level kernel (~6 us).  While a frontier fits into LDS ONE workgroup peels level after level inside a
// single launch: the frontier never leaves LDS, the cursor is a register, a level costs two barriers and one round of
// atomics on the successors' counters.  The kernel returns as soon as a frontier outgrows CHAIN_CAP (the wide kernel
// continues from the global order array, which is always kept complete) or after max_levels.
#define CHAIN_CAP 4096
template <int K>
__global__ __launch_bounds__(PEEL_THREADS) void k_sched_peel_chain(int l, int max_levels, const int *succ, int *remaining, int *order,
                                                                   int *level, unsigned *state, unsigned *level_end) {
    __shared__ int fr[2][CHAIN_CAP];
    __shared__ int nnext;
    __shared__ unsigned cursor;
    unsigned begin = level_end[l - 1], end = level_end[l];
    int ncur = (int)(end - begin);
    if (ncur > CHAIN_CAP || ncur == 0) {   // too wide for one workgroup (or nothing left): leave it to the caller
        if (threadIdx.x == 0) state[ST_NEXT_LEVEL] = (unsigned)l;
        return;
    }
    for (int i = threadIdx.x; i < ncur; i += blockDim.x) fr[0][i] = order[begin + (unsigned)i];
    if (threadIdx.x == 0) { nnext = 0; cursor = end; }
    __syncthreads();
    int cur = 0, done = 0;
    while (done < max_levels && ncur > 0 && ncur <= CHAIN_CAP) {
        const unsigned base = cursor;
        for (int i = threadIdx.x; i < ncur; i += blockDim.x) {
            const int u = fr[cur][i];
            int v[K];
            bool ready[K];
#pragma unroll
            for (int s = 0; s < K; s++) v[s] = succ[(long)u * K + s];
#pragma unroll
            for (int s = 0; s < K; s++) ready[s] = v[s] >= 0 && atomicSub(&remaining[v[s]], 1) == 1;
#pragma unroll
            for (int s = 0; s < K; s++)
                if (ready[s]) {
                    level[v[s]] = l + 1;
                    const int pos = atomicAdd(&nnext, 1);
                    order[base + (unsigned)pos] = v[s];            // the global order stays complete
                    if (pos < CHAIN_CAP) fr[cur ^ 1][pos] = v[s];
                }
        }
        __syncthreads();
        const int made = nnext;
        __syncthreads();
        if (threadIdx.x == 0) {
            cursor = base + (unsigned)made;
            level_end[l + 1] = cursor;
            nnext = 0;
        }
        __syncthreads();
        l++; done++;
        cur ^= 1;
        ncur = made;
    }
    if (threadIdx.x == 0) {
        state[ST_CURSOR] = cursor;
        state[ST_NEXT_LEVEL] = (unsigned)l;
        // level l's frontier is order[level_end[l-1], level_end[l]); levels up to l-1 have been retired, level l exists iff ncur > 0
        if (ncur > 0) state[ST_NLEVELS] = (unsigned)l; else state[ST_NLEVELS] = (unsigned)(l - 1);
    }
}

template <typename KeyT>
__global__ __launch_bounds__(256) void k_sched_final_keys(long n, const int *level, const unsigned *sort_key, int key_bits, KeyT *keys, unsigned *vals) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < n; u += stride) {
        KeyT k = (KeyT)(unsigned)level[u];
        if (key_bits > 0) k = (k << key_bits) | (KeyT)sort_key[u];
        keys[u] = k;
        vals[u] = (unsigned)u;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void k_gather(const T *src, const int *order, T *dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) dst[s] = src[order[s]];
}

inline int bits_for(unsigned long long v) {   // bits needed to hold values 0..v
    int b = 1;
    while (b < 64 && (v >> b) != 0ull) b++;
    return b;
}
inline int grid_for_n(long n) {
    long g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}
#define SCHK(call)                                                                                                       \
    do {                                                                                                                 \
        hipError_t e_ = (call);                                                                                          \
        if (e_ != hipSuccess) throw std::runtime_error(std::string("device scheduler: ") + hipGetErrorString(e_) + " at " #call); \
    } while (0)

struct Scratch {   // freed when the call returns: a schedule is built once per data set
    std::vector<void *> ptrs;
    ~Scratch() { for (void *p : ptrs) (void)hipFree(p); }
    template <typename T> T *get(size_t n) {
        void *p = nullptr;
        SCHK(hipMalloc(&p, (n ? n : 1) * sizeof(T)));
        ptrs.push_back(p);
        return (T *)p;
    }
};

}  // namespace

void device_gather_u32(const unsigned *src, const int *order, unsigned *dst, long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather<unsigned>, dim3(grid_for_n(n)), dim3(256), 0, st, src, order, dst, n);
}
void device_gather_f32(const float *src, const int *order, float *dst, long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_gather<float>, dim3(grid_for_n(n)), dim3(256), 0, st, src, order, dst, n);
}

// dst[order[s]] = src[s]: predictions of a scheduled data set back into the caller's instance order
__global__ __launch_bounds__(256) void k_scatter_f32(const float *src, const int *order, float *dst, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long s = (long)blockIdx.x * blockDim.x + threadIdx.x; s < n; s += stride) dst[order[s]] = src[s];
}
void device_scatter_f32(const float *src, const int *order, float *dst, long n, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_scatter_f32, dim3(grid_for_n(n)), dim3(256), 0, st, src, order, dst, n);
}

void device_sort_pairs_u32(unsigned *keys_in, unsigned *keys_out, unsigned *vals_in, unsigned *vals_out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st) {
    if (n <= 0) return;
    size_t need = 0;
    SCHK(rocprim::radix_sort_pairs(nullptr, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 32u, st));
    if (need > *tmp_bytes) {
        if (*tmp) (void)hipFree(*tmp);
        *tmp = nullptr;
        SCHK(hipMalloc(tmp, need));
        *tmp_bytes = need;
    }
    SCHK(rocprim::radix_sort_pairs(*tmp, need, keys_in, keys_out, vals_in, vals_out, (size_t)n, 0u, 32u, st));
}

// exclusive prefix sum of n flags (0 / 1 words); returns the total.  *tmp grows as needed (the caller frees it).
long device_exclusive_scan_u32(const unsigned *in, unsigned *out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st) {
    if (n <= 0) return 0;
    size_t need = 0;
    SCHK(rocprim::exclusive_scan(nullptr, need, in, out, 0u, (size_t)n, rocprim::plus<unsigned>(), st));
    if (need > *tmp_bytes) {
        if (*tmp) (void)hipFree(*tmp);
        *tmp = nullptr;
        SCHK(hipMalloc(tmp, need));
        *tmp_bytes = need;
    }
    SCHK(rocprim::exclusive_scan(*tmp, need, in, out, 0u, (size_t)n, rocprim::plus<unsigned>(), st));
    unsigned last_in = 0, last_out = 0;
    SCHK(hipMemcpyAsync(&last_in, in + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    SCHK(hipMemcpyAsync(&last_out, out + (n - 1), sizeof(unsigned), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    return (long)last_in + (long)last_out;
}

// See svdf_kernels.h.  Returns the number of levels; throws std::runtime_error with the reference's bound messages.
long device_schedule(const SchedColumns &in, int *order_out, std::vector<long> &level_ptr, long *max_level_size, hipStream_t st) {
    const long n = in.n;
    const int K = in.K;
    level_ptr.assign(1, 0);
    if (max_level_size) *max_level_size = 0;
    if (n == 0) return 0;
    if (K < 1 || K > SVDF_SCHED_MAX_SLOTS) throw std::runtime_error("device scheduler: bad slot count");
    if ((unsigned long long)n * (unsigned long long)K >= 0xFFFFFFFFull || n >= 0x7FFFFFFFL)
        throw std::runtime_error("device scheduler: more than 2^32 row entries in one data set");
    const long m = n * K;
    Cols C;
    memset(&C, 0, sizeof(C));
    C.K = K;
    for (int s = 0; s < K; s++) { C.col[s] = in.col[s]; C.off[s] = in.off[s]; C.limit[s] = in.limit[s]; }
    const unsigned absent_key = in.num_res;   // one past the largest resource id: sorts behind every real row
    const int res_bits = bits_for(absent_key);

    Scratch S;
    unsigned *keys_a = S.get<unsigned>((size_t)m), *keys_b = S.get<unsigned>((size_t)m);
    unsigned *vals_a = S.get<unsigned>((size_t)m), *vals_b = S.get<unsigned>((size_t)m);
    int *remaining = S.get<int>((size_t)n), *level = S.get<int>((size_t)n);
    int *succ = S.get<int>((size_t)m);
    int *frontier = S.get<int>((size_t)n);
    unsigned *state = S.get<unsigned>(ST_WORDS);
    const long level_cap = n + 2;   // a chain can be as deep as the data set
    unsigned *level_end = S.get<unsigned>((size_t)level_cap);
    SCHK(hipMemsetAsync(state, 0, ST_WORDS * sizeof(unsigned), st));
    SCHK(hipMemsetAsync(succ, 0xFF, (size_t)m * sizeof(int), st));   // absent slots have no successor (-1)
    SCHK(hipMemsetAsync(level_end, 0, sizeof(unsigned), st));

    hipLaunchKernelGGL(k_sched_keys, dim3(grid_for_n(n)), dim3(256), 0, st, C, n, absent_key, keys_a, vals_a, remaining, state);
    {
        size_t tmp_bytes = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)m, 0u, (unsigned)res_bits, st));
        void *tmp = S.get<char>(tmp_bytes);
        SCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)m, 0u, (unsigned)res_bits, st));
    }
    hipLaunchKernelGGL(k_sched_links, dim3(grid_for_n(m)), dim3(256), 0, st, keys_b, vals_b, m, absent_key, K, succ, remaining);
    const int peel_grid = 64;   // few large workgroups: see Stage
    hipLaunchKernelGGL(k_sched_seed, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, n, remaining, frontier, level, state, level_end);
    auto peel = [&](int lvl) {
        switch (K) {
        case 1: hipLaunchKernelGGL(k_sched_peel<1>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 2: hipLaunchKernelGGL(k_sched_peel<2>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 3: hipLaunchKernelGGL(k_sched_peel<3>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 4: hipLaunchKernelGGL(k_sched_peel<4>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 5: hipLaunchKernelGGL(k_sched_peel<5>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 6: hipLaunchKernelGGL(k_sched_peel<6>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        case 7: hipLaunchKernelGGL(k_sched_peel<7>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        default: hipLaunchKernelGGL(k_sched_peel<8>, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, lvl, succ, remaining, frontier, level, state, level_end); break;
        }
    };

    // peel.  Narrow frontiers are chained inside one launch (k_sched_peel_chain), wide ones take one launch per level; the host
    // looks at the state between phases: how far the levels got, and whether every unit has been placed
    auto chain = [&](int lvl, int max_levels) {
        switch (K) {
        case 1: hipLaunchKernelGGL(k_sched_peel_chain<1>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 2: hipLaunchKernelGGL(k_sched_peel_chain<2>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 3: hipLaunchKernelGGL(k_sched_peel_chain<3>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 4: hipLaunchKernelGGL(k_sched_peel_chain<4>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 5: hipLaunchKernelGGL(k_sched_peel_chain<5>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 6: hipLaunchKernelGGL(k_sched_peel_chain<6>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        case 7: hipLaunchKernelGGL(k_sched_peel_chain<7>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        default: hipLaunchKernelGGL(k_sched_peel_chain<8>, dim3(1), dim3(PEEL_THREADS), 0, st, lvl, max_levels, succ, remaining, frontier, level, state, level_end); break;
        }
    };
    unsigned host_state[ST_WORDS];
    long l = 1;   // the level whose frontier is known: order[level_end[l-1], level_end[l])
    long placed_before = -1;
    for (;;) {
        chain((int)l, 1 << 20);
        SCHK(hipMemcpyAsync(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        if (host_state[ST_ERROR]) {
            for (int s = 0; s < K; s++)
                if (host_state[ST_ERROR] & (1u << s)) throw std::runtime_error(in.limit_msg[s] ? in.limit_msg[s] : "feature index exceed bound");
        }
        l = (long)host_state[ST_NEXT_LEVEL];
        if ((long)host_state[ST_CURSOR] >= n) break;
        if ((long)host_state[ST_CURSOR] == placed_before) throw std::runtime_error("device scheduler: the dependency graph did not drain");
        placed_before = (long)host_state[ST_CURSOR];
        if (l + 1 >= level_cap) throw std::runtime_error("device scheduler: the dependency graph did not drain");
        const long wide = 128;   // the frontier is wider than one workgroup's LDS: one launch per level for a while
        for (long j = 0; j < wide && l + 1 < level_cap; j++, l++) peel((int)l);
    }
    SCHK(hipMemcpyAsync(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    const long nlevels = (long)host_state[ST_NLEVELS];
    std::vector<unsigned> ends((size_t)nlevels + 1);
    SCHK(hipMemcpyAsync(ends.data(), level_end, ((size_t)nlevels + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st));

    // final order: by (level, batch key, file position)
    const int lvl_bits = bits_for((unsigned long long)nlevels);
    const int key_bits = in.sort_key ? bits_for(in.sort_key_max) : 0;
    if (lvl_bits + key_bits <= 32) {
        unsigned *k2a = keys_a, *k2b = keys_b;   // the entry arrays are free again (m >= n)
        hipLaunchKernelGGL(k_sched_final_keys<unsigned>, dim3(grid_for_n(n)), dim3(256), 0, st, n, level, in.sort_key, key_bits, k2a, vals_a);
        size_t tmp_bytes = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k2a, k2b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + key_bits), st));
        void *tmp = S.get<char>(tmp_bytes);
        SCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k2a, k2b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + key_bits), st));
    } else {
        unsigned long long *k2a = S.get<unsigned long long>((size_t)n), *k2b = S.get<unsigned long long>((size_t)n);
        hipLaunchKernelGGL(k_sched_final_keys<unsigned long long>, dim3(grid_for_n(n)), dim3(256), 0, st, n, level, in.sort_key, key_bits, k2a, vals_a);
        size_t tmp_bytes = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, k2a, k2b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + key_bits), st));
        void *tmp = S.get<char>(tmp_bytes);
        SCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, k2a, k2b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + key_bits), st));
    }
    SCHK(hipStreamSynchronize(st));
    SCHK(hipGetLastError());
    level_ptr.resize((size_t)nlevels + 1);
    long biggest = 0;
    for (long j = 0; j <= nlevels; j++) {
        level_ptr[(size_t)j] = (long)ends[(size_t)j];
        if (j > 0) biggest = std::max(biggest, level_ptr[(size_t)j] - level_ptr[(size_t)j - 1]);
    }
    if (max_level_size) *max_level_size = biggest;
    return nlevels;
}

// =============================================================================== user units (SVD++ blocks): lists of rows per unit
// Engine::schedule_units' scan (svdf_sched.cpp; the reference walks a user's blocks in file order, apex_svd_base.h:568-582) as the same
// frontier peel: a unit touches the rows of its instances (global biases, its user row, item rows), the rows of its feedback list and --
// when it loads or saves the trainer's feedback registers -- one state resource.  Entries are a list per unit (eptr), not K columns:
//   * a row met twice inside a unit is ONE dependency (the later entries are taken off the unit's count and get no successor); the same
//     neighbour test yields what the host finds with its stamp array: an item rated twice inside a simple unit (row_fresh: the wave kernel
//     reads that row at use) and a feedback id listed twice (the unit is not simple),
//   * the user entry of every row but the first is not emitted when the rows have the simple shape (it would be the same row 100 times),
//   * retiring a unit walks its list with one WAVE (lanes stride the entries); narrow frontiers are chained inside one launch as above.
namespace {

enum { UST_ANY_FRESH = 5, UST_NONUNIT = 6 };

__device__ __forceinline__ int wave_min_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ int wave_sum_i(int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__global__ __launch_bounds__(256) void k_us_count(const UnitSchedIn U, unsigned *cnt) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < U.n; u += stride) {
        const DevUnit un = U.units[u];
        cnt[u] = (unsigned)(U.row_ptr[3 * (long)un.row_end] - U.row_ptr[3 * (long)un.row_begin]) + (unsigned)(un.fb_end - un.fb_begin) +
                 ((un.flags & (UNIT_LOAD | UNIT_SAVE)) ? 1u : 0u);
    }
}

// one wave per unit: resource keys of its entries, the entry's owner, the unit's dependency count; the shape tests of the simple path
__global__ __launch_bounds__(256) void k_us_fill(const UnitSchedIn U, const unsigned *eptr, unsigned absent_key, unsigned *keys, unsigned *vals,
                                                 int *owner, int *remaining, int *first_fail, unsigned char *nonunit) {
    const int lane = threadIdx.x & 63;
    const long nwaves = (long)gridDim.x * (blockDim.x >> 6);
    for (long u = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); u < U.n; u += nwaves) {
        const DevUnit un = U.units[u];
        const unsigned e0 = eptr[u];
        const int pf = U.row_ptr[3 * (long)un.row_begin];
        const int nrow_ent = U.row_ptr[3 * (long)un.row_end] - pf;
        auto shape_of = [&](int r, unsigned uid0) {
            const int *p = U.row_ptr + 3 * (long)r;
            return p[1] == p[0] && p[2] == p[1] + 1 && p[3] == p[2] + 1 && U.index[p[1]] == uid0;
        };
        unsigned uid0 = 0;
        bool shape0 = false;
        if (un.row_end > un.row_begin) { uid0 = U.index[pf]; shape0 = shape_of(un.row_begin, uid0); }
        int ff = 0x7FFFFFFF, fn = 0x7FFFFFFF, live = 0;
        for (int r = un.row_begin + lane; r < un.row_end; r += 64) {
            const int *p = U.row_ptr + 3 * (long)r;
            const bool shape = shape_of(r, uid0);
            if (!shape) ff = min(ff, r);
            else if (U.value[p[1]] != 1.0f || U.value[p[2]] != 1.0f) fn = min(fn, r);
            for (int j = p[0]; j < p[3]; j++) {
                const unsigned id = U.index[j];
                unsigned key;
                if (j < p[1]) key = U.goff + id;
                else if (j < p[2]) key = (shape0 && shape && r != un.row_begin) ? absent_key : U.user_off + id;
                else key = U.item_off + id;
                const unsigned e = e0 + (unsigned)(j - pf);
                keys[e] = key; vals[e] = e; owner[e] = (int)u;
                live += key != absent_key;
            }
        }
        for (int j = un.fb_begin + lane; j < un.fb_end; j += 64) {
            const unsigned e = e0 + (unsigned)nrow_ent + (unsigned)(j - un.fb_begin);
            keys[e] = U.fb_off + U.fb_index[j]; vals[e] = e; owner[e] = (int)u;
            live++;
        }
        if (lane == 0 && (un.flags & (UNIT_LOAD | UNIT_SAVE))) {
            const unsigned e = e0 + (unsigned)nrow_ent + (unsigned)(un.fb_end - un.fb_begin);
            keys[e] = U.state_res; vals[e] = e; owner[e] = (int)u;
            live++;
        }
        ff = wave_min_i(ff); fn = wave_min_i(fn); live = wave_sum_i(live);
        if (lane == 0) { first_fail[u] = ff; nonunit[u] = fn < ff ? 1 : 0; remaining[u] = live; }
    }
}

// sorted (row, entry): successor unit of every entry, one credit for the unit that heads a row, repeated rows inside a unit
__global__ __launch_bounds__(256) void k_us_links(const UnitSchedIn U, const unsigned *eptr, const unsigned *keys, const unsigned *vals, long m,
                                                  unsigned absent_key, const int *owner, const int *first_fail, int *succ, int *remaining,
                                                  unsigned char *fbdup, unsigned char *fresh, unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const unsigned key = keys[j];
        if (key == absent_key) continue;
        const unsigned e = vals[j];
        const int u = owner[e];
        const bool prev_same = j > 0 && keys[j - 1] == key;
        if (prev_same && owner[vals[j - 1]] == u) {   // this row again inside the unit
            atomicSub(&remaining[u], 1);
            if (key >= U.item_off && key - U.item_off < U.num_item) {
                const DevUnit un = U.units[u];
                const int jc = U.row_ptr[3 * (long)un.row_begin] + (int)(e - eptr[u]);
                int lo = un.row_begin, hi = un.row_end - 1;   // the row holding CSR entry jc
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (U.row_ptr[3 * (long)mid] <= jc) lo = mid; else hi = mid - 1;
                }
                if (lo < first_fail[u]) { fresh[lo] = 1; state[UST_ANY_FRESH] = 1u; }
            } else if (key >= U.fb_off && key - U.fb_off < U.num_fb) {
                fbdup[u] = 1;
            }
            continue;
        }
        if (!prev_same) atomicSub(&remaining[u], 1);
        long jj = j + 1;
        while (jj < m && keys[jj] == key && owner[vals[jj]] == u) jj++;
        succ[e] = (jj < m && keys[jj] == key) ? owner[vals[jj]] : -1;
    }
}

// the fast-path flag of every unit (Engine::schedule_units: simple && use_simple_units && ...) and the key of the in-level partition
__global__ __launch_bounds__(256) void k_us_simple(const UnitSchedIn U, const int *first_fail, const unsigned char *fbdup, const unsigned char *nonunit,
                                                   unsigned char *simple_out, unsigned *sort_key, unsigned *state) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long u = (long)blockIdx.x * blockDim.x + threadIdx.x; u < U.n; u += stride) {
        const DevUnit un = U.units[u];
        const bool rows = un.row_end > un.row_begin;
        if (U.simple_ok && rows && nonunit[u]) state[UST_NONUNIT] = 1u;
        const bool fast = U.simple_ok && rows && first_fail[u] == 0x7FFFFFFF && !fbdup[u] && U.fast_ok;
        simple_out[u] = fast ? 1 : 0;
        sort_key[u] = fast ? 0u : 1u;
    }
}

__global__ __launch_bounds__(PEEL_THREADS) void k_us_seed(long n, const int *remaining, int *order, int *level, unsigned *state, unsigned *level_end) {
    const int lane = threadIdx.x & 63;
    const long stride = (long)gridDim.x * blockDim.x;
    const long rounds = (n + stride - 1) / stride;
    for (long it = 0; it < rounds; it++) {
        const long u = it * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool ready = u < n && remaining[u] == 0;
        const unsigned long long mask = __ballot(ready);
        if (mask) {
            unsigned base = 0;
            if (lane == 0) base = atomicAdd(&state[ST_CURSOR], (unsigned)__popcll(mask));
            base = __shfl(base, 0);
            if (ready) { level[u] = 1; order[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = (int)u; }
        }
    }
    __shared__ Stage sh;
    publish_level_end(sh, state, level_end, 0);
}

// level l+1 from level l: one wave per retiring unit, appends through one atomic per wave and 64 entries
__global__ __launch_bounds__(PEEL_THREADS) void k_us_peel(int l, const unsigned *eptr, const int *succ, int *remaining, int *order, int *level,
                                                          unsigned *state, unsigned *level_end) {
    const int lane = threadIdx.x & 63;
    const unsigned begin = level_end[l - 1], end = level_end[l];
    const unsigned nwaves = gridDim.x * (blockDim.x >> 6);
    for (unsigned i = begin + blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); i < end; i += nwaves) {
        const int u = order[i];
        const unsigned e0 = eptr[u], e1 = eptr[u + 1];
        for (unsigned eb = e0; eb < e1; eb += 64) {
            const unsigned e = eb + (unsigned)lane;
            const int v = e < e1 ? succ[e] : -1;
            const bool ready = v >= 0 && atomicSub(&remaining[v], 1) == 1;
            const unsigned long long mask = __ballot(ready);
            if (mask) {
                unsigned base = 0;
                if (lane == 0) base = atomicAdd(&state[ST_CURSOR], (unsigned)__popcll(mask));
                base = __shfl(base, 0);
                if (ready) { level[v] = l + 1; order[base + (unsigned)__popcll(mask & ((1ull << lane) - 1ull))] = v; }
            }
        }
    }
    __shared__ Stage sh;
    publish_level_end(sh, state, level_end, l);
}

// narrow frontiers: ONE workgroup peels level after level, the frontier in LDS (k_sched_peel_chain for lists).  A level is a chain of
// dependent memory operations -- a unit's successors, their counters, the released units' entry ranges -- and nothing else, so:
//   * the frontier holds CHUNKS of up to 128 entries (two per lane, both successor loads and both counter atomics in flight together), spread
//     over all sixteen waves: a unit of 200 entries is two work items, not four dependent rounds of one wave (4.7 -> ~2 us per level);
//   * a released unit's range is fetched while the counter that may release it is still in flight and travels with it through LDS.
#define UCHAIN_CAP 4096
#define UCHAIN_SPAN 128
__device__ __forceinline__ int wave_incl_scan_i(int v, int lane) {
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(v, o); if (lane >= o) v += t; }
    return v;
}
// A barrier that orders LDS traffic only: __syncthreads() also waits for the level's global stores (order[], level[], level_end[] -- read by
// nobody before the launch ends) to be acknowledged, one more memory round trip on the critical path of every level.
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
__global__ __launch_bounds__(PEEL_THREADS) void k_us_peel_chain(int l, int max_levels, const unsigned *eptr, const int *succ, int *remaining, int *order,
                                                                int *level, unsigned *state, unsigned *level_end) {
    __shared__ unsigned fe0[2][UCHAIN_CAP], fe1[2][UCHAIN_CAP];
    __shared__ int nchunk, nunit;
    __shared__ unsigned cursor;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nwv = blockDim.x >> 6;
    unsigned begin = level_end[l - 1], end = level_end[l];
    int units_cur = (int)(end - begin);
    if (units_cur > UCHAIN_CAP / 4 || units_cur == 0) {
        if (threadIdx.x == 0) state[ST_NEXT_LEVEL] = (unsigned)l;
        return;
    }
    if (threadIdx.x == 0) { nchunk = 0; nunit = 0; cursor = end; }
    __syncthreads();
    // chunks of the first frontier; a frontier that does not fit ends the chain (the wide kernel continues from the global order array)
    for (int i = threadIdx.x; i < units_cur; i += blockDim.x) {
        const int u = order[begin + (unsigned)i];
        const unsigned e0 = eptr[u], e1 = eptr[u + 1];
        const int nc = (int)((e1 - e0 + UCHAIN_SPAN - 1) / UCHAIN_SPAN);
        const int pos = atomicAdd(&nchunk, nc);
        for (int c = 0; c < nc; c++)
            if (pos + c < UCHAIN_CAP) { fe0[0][pos + c] = e0 + (unsigned)c * UCHAIN_SPAN; fe1[0][pos + c] = min(e1, e0 + (unsigned)(c + 1) * UCHAIN_SPAN); }
    }
    __syncthreads();
    int ncur = nchunk;
    __syncthreads();
    if (threadIdx.x == 0) nchunk = 0;
    __syncthreads();
    int cur = 0, done = 0;
    bool fits = ncur <= UCHAIN_CAP;
    while (done < max_levels && units_cur > 0 && fits) {
        const unsigned base = cursor;
        for (int w = wv; w < ncur; w += nwv) {
            const unsigned e0 = fe0[cur][w], e1 = fe1[cur][w];
            const unsigned ea = e0 + (unsigned)lane, eb = ea + 64u;
            const int va = ea < e1 ? succ[ea] : -1, vb = eb < e1 ? succ[eb] : -1;
            unsigned a0 = 0, a1 = 0, b0 = 0, b1 = 0;
            if (va >= 0) { a0 = eptr[va]; a1 = eptr[va + 1]; }
            if (vb >= 0) { b0 = eptr[vb]; b1 = eptr[vb + 1]; }
            const bool ra = va >= 0 && atomicSub(&remaining[va], 1) == 1;
            const bool rb = vb >= 0 && atomicSub(&remaining[vb], 1) == 1;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const bool ready = h ? rb : ra;
                const unsigned long long mask = __ballot(ready);
                if (!mask) continue;
                const int v = h ? vb : va;
                const unsigned r0 = h ? b0 : a0, r1 = h ? b1 : a1;
                const int nc = ready ? (int)((r1 - r0 + UCHAIN_SPAN - 1) / UCHAIN_SPAN) : 0;
                const int incl = wave_incl_scan_i(nc, lane);
                int cpos0 = 0, upos0 = 0;
                if (lane == 63) { cpos0 = atomicAdd(&nchunk, incl); upos0 = atomicAdd(&nunit, __popcll(mask)); }
                cpos0 = __shfl(cpos0, 63); upos0 = __shfl(upos0, 63);
                if (ready) {
                    level[v] = l + 1;
                    order[base + (unsigned)(upos0 + __popcll(mask & ((1ull << lane) - 1ull)))] = v;
                    const int cpos = cpos0 + incl - nc;
                    for (int c = 0; c < nc; c++)
                        if (cpos + c < UCHAIN_CAP) { fe0[cur ^ 1][cpos + c] = r0 + (unsigned)c * UCHAIN_SPAN; fe1[cur ^ 1][cpos + c] = min(r1, r0 + (unsigned)(c + 1) * UCHAIN_SPAN); }
                }
            }
        }
        lds_barrier();
        const int made_units = nunit, made_chunks = nchunk;
        lds_barrier();
        if (threadIdx.x == 0) {
            cursor = base + (unsigned)made_units;
            level_end[l + 1] = cursor;
            nunit = 0; nchunk = 0;
        }
        lds_barrier();
        l++; done++;
        cur ^= 1;
        units_cur = made_units;
        ncur = made_chunks;
        fits = made_chunks <= UCHAIN_CAP;
    }
    if (threadIdx.x == 0) {
        state[ST_CURSOR] = cursor;
        state[ST_NEXT_LEVEL] = (unsigned)l;
        // level l's frontier is order[level_end[l-1], level_end[l]); levels up to l-1 have been retired, level l exists iff it holds units
        if (units_cur > 0) state[ST_NLEVELS] = (unsigned)l; else state[ST_NLEVELS] = (unsigned)(l - 1);
    }
}

}  // namespace

long device_schedule_units(const UnitSchedIn &in, int *order_out, unsigned char *simple_out, unsigned char *fresh_out, UnitSchedOut &out, hipStream_t st) {
    const long n = in.n;
    out.level_ptr.assign(1, 0);
    out.max_level_size = 0; out.any_fresh = false; out.unit_values = true;
    if (n == 0) return 0;
    if (n >= 0x7FFFFFFFL) throw std::runtime_error("device scheduler: too many units");
    Scratch S;
    unsigned *cnt = S.get<unsigned>((size_t)n), *eptr = S.get<unsigned>((size_t)n + 1);
    hipLaunchKernelGGL(k_us_count, dim3(grid_for_n(n)), dim3(256), 0, st, in, cnt);
    void *scan_tmp = nullptr;
    size_t scan_bytes = 0;
    long m = 0;
    try { m = device_exclusive_scan_u32(cnt, eptr, n, &scan_tmp, &scan_bytes, st); }
    catch (...) { if (scan_tmp) (void)hipFree(scan_tmp); throw; }
    if (scan_tmp) (void)hipFree(scan_tmp);
    if (m >= 0xFFFFFFF0L) throw std::runtime_error("device scheduler: more than 2^32 row entries in one data set");
    {
        const unsigned mm = (unsigned)m;
        SCHK(hipMemcpyAsync(eptr + n, &mm, sizeof(unsigned), hipMemcpyHostToDevice, st));
        SCHK(hipStreamSynchronize(st));
    }
    const unsigned absent_key = in.state_res + 1;
    const int res_bits = bits_for(absent_key);
    const size_t me = (size_t)(m > 0 ? m : 1);
    unsigned *keys_a = S.get<unsigned>(std::max(me, (size_t)n)), *keys_b = S.get<unsigned>(std::max(me, (size_t)n));
    unsigned *vals_a = S.get<unsigned>(std::max(me, (size_t)n)), *vals_b = S.get<unsigned>(me);
    int *owner = S.get<int>(me), *succ = S.get<int>(me);
    int *remaining = S.get<int>((size_t)n), *level = S.get<int>((size_t)n), *first_fail = S.get<int>((size_t)n), *frontier = S.get<int>((size_t)n);
    unsigned char *nonunit = S.get<unsigned char>((size_t)n), *fbdup = S.get<unsigned char>((size_t)n);
    unsigned *sort_key = S.get<unsigned>((size_t)n);
    unsigned *state = S.get<unsigned>(ST_WORDS);
    const long level_cap = n + 2;
    unsigned *level_end = S.get<unsigned>((size_t)level_cap);
    SCHK(hipMemsetAsync(state, 0, ST_WORDS * sizeof(unsigned), st));
    SCHK(hipMemsetAsync(succ, 0xFF, me * sizeof(int), st));
    SCHK(hipMemsetAsync(fbdup, 0, (size_t)n, st));
    SCHK(hipMemsetAsync(level_end, 0, sizeof(unsigned), st));
    if (in.nrow > 0) SCHK(hipMemsetAsync(fresh_out, 0, (size_t)in.nrow, st));
    {
        long g = (n + 3) / 4;
        if (g > 8192) g = 8192;
        hipLaunchKernelGGL(k_us_fill, dim3((unsigned)g), dim3(256), 0, st, in, eptr, absent_key, keys_a, vals_a, owner, remaining, first_fail, nonunit);
    }
    if (m > 0) {
        size_t tmp_bytes = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)m, 0u, (unsigned)res_bits, st));
        void *tmp = S.get<char>(tmp_bytes);
        SCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, vals_a, vals_b, (size_t)m, 0u, (unsigned)res_bits, st));
        hipLaunchKernelGGL(k_us_links, dim3(grid_for_n(m)), dim3(256), 0, st, in, eptr, keys_b, vals_b, m, absent_key, owner, first_fail, succ, remaining,
                           fbdup, fresh_out, state);
    }
    hipLaunchKernelGGL(k_us_simple, dim3(grid_for_n(n)), dim3(256), 0, st, in, first_fail, fbdup, nonunit, simple_out, sort_key, state);
    const int peel_grid = 64;
    hipLaunchKernelGGL(k_us_seed, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, n, remaining, frontier, level, state, level_end);
    unsigned host_state[ST_WORDS];
    long l = 1;
    long placed_before = -1;
    for (;;) {
        hipLaunchKernelGGL(k_us_peel_chain, dim3(1), dim3(PEEL_THREADS), 0, st, (int)l, 1 << 20, eptr, succ, remaining, frontier, level, state, level_end);
        SCHK(hipMemcpyAsync(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost, st));
        SCHK(hipStreamSynchronize(st));
        l = (long)host_state[ST_NEXT_LEVEL];
        if ((long)host_state[ST_CURSOR] >= n) break;
        if ((long)host_state[ST_CURSOR] == placed_before) throw std::runtime_error("device scheduler: the dependency graph did not drain");
        placed_before = (long)host_state[ST_CURSOR];
        if (l + 1 >= level_cap) throw std::runtime_error("device scheduler: the dependency graph did not drain");
        const long wide = 32;
        for (long j = 0; j < wide && l + 1 < level_cap; j++, l++)
            hipLaunchKernelGGL(k_us_peel, dim3(peel_grid), dim3(PEEL_THREADS), 0, st, (int)l, eptr, succ, remaining, frontier, level, state, level_end);
    }
    SCHK(hipMemcpyAsync(host_state, state, sizeof(host_state), hipMemcpyDeviceToHost, st));
    SCHK(hipStreamSynchronize(st));
    const long nlevels = (long)host_state[ST_NLEVELS];
    out.any_fresh = host_state[UST_ANY_FRESH] != 0;
    out.unit_values = host_state[UST_NONUNIT] == 0;
    std::vector<unsigned> ends((size_t)nlevels + 1);
    SCHK(hipMemcpyAsync(ends.data(), level_end, ((size_t)nlevels + 1) * sizeof(unsigned), hipMemcpyDeviceToHost, st));
    // final order: by (level, fast-path units first, file position)
    const int lvl_bits = bits_for((unsigned long long)nlevels);
    if (lvl_bits + 1 > 32) throw std::runtime_error("device scheduler: too many levels");
    hipLaunchKernelGGL(k_sched_final_keys<unsigned>, dim3(grid_for_n(n)), dim3(256), 0, st, n, level, sort_key, 1, keys_a, vals_a);
    {
        size_t tmp_bytes = 0;
        SCHK(rocprim::radix_sort_pairs(nullptr, tmp_bytes, keys_a, keys_b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + 1), st));
        void *tmp = S.get<char>(tmp_bytes);
        SCHK(rocprim::radix_sort_pairs(tmp, tmp_bytes, keys_a, keys_b, vals_a, (unsigned *)order_out, (size_t)n, 0u, (unsigned)(lvl_bits + 1), st));
    }
    SCHK(hipStreamSynchronize(st));
    SCHK(hipGetLastError());
    out.level_ptr.resize((size_t)nlevels + 1);
    for (long j = 0; j <= nlevels; j++) {
        out.level_ptr[(size_t)j] = (long)ends[(size_t)j];
        if (j > 0) out.max_level_size = std::max(out.max_level_size, out.level_ptr[(size_t)j] - out.level_ptr[(size_t)j - 1]);
    }
    return nlevels;
}

}  // namespace svdf
