// svdf_k_runs.hip -- RUNS of an item's consecutive ratings as schedule units (knob "runs_exec").
//
// The contract workload moves 2 factor rows per rating (1072 B at k = 64).  Consecutive ratings of ONE item -- consecutive in
// the item's own file order; they are ~n / count positions apart in the file -- can share the item's row: a lane group reads it once, walks up
// to R ratings with it in registers, writes it once.  A run is a valid unit of a conflict-free schedule when nothing else touches its rows
// "inside" it, and that is a FILE-ORDER fact: rating y (user u) may join the run headed at file position h iff u's previous rating lies
// before h.  Then executing whole runs in the order of their heads is a sequential reordering that keeps every row's touches in file order:
// u's previous rating belongs to a run with an earlier head, u's next rating z can only join a run whose head lies behind y, and the item's
// own ratings are walked in order.  So:
//   1. prev[p] = file position of the previous rating of p's user        (stable radix sort of (user, position), neighbours)
//   2. the item-major list of positions                                   (stable radix sort of (item, position))
//   3. one thread per item walks its list and cuts it into runs           (k_runs_form: head flags, index in run)
//   4. runs numbered by the file position of their head                   (exclusive scan of the head flags)
//   5. the runs' columns (item, R users, R labels; absent users marked)   (k_runs_fill)
//   6. the EXISTING device level scheduler over runs as units with 1 + R row slots (svdf_k_sched.hip), columns gathered into level order
// and a level is one launch of k_basicmf_runs_soa: per rating the arithmetic of k_basicmf_slots in file order inside the run -- the same bits
// as one instance per lane group (tests/test_gpu_runs.py: == the level-by-level pass == the oracle).
// Uniform contract stream, runs of at most 4: 1 874 levels -> ~680, 3.0 - 3.5 ratings per run, the item row's bytes divided by the run length.
#include "svdf_device.h"

namespace svdf {

// ids of both columns against their limits (the reference's messages are raised by the host from the flag word: bit 0 user, bit 1 item)
__global__ __launch_bounds__(256) void k_runs_check(const unsigned *user, const unsigned *item, long n, unsigned nu, unsigned ni, unsigned *flag) {
    const long stride = (long)gridDim.x * blockDim.x;
    unsigned bad = 0u;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        if (user[p] >= nu) bad |= 1u;
        if (item[p] >= ni) bad |= 2u;
    }
    if (bad) atomicOr(flag, bad);
}
// keys / pos: positions stably sorted by user id: prev[pos[j]] = pos[j - 1] when the users agree, else 0xFFFFFFFF
__global__ __launch_bounds__(256) void k_runs_prev(const unsigned *keys, const unsigned *pos, long n, unsigned *prev) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride)
        prev[pos[j]] = (j > 0 && keys[j - 1] == keys[j]) ? pos[j - 1] : 0xFFFFFFFFu;
}
// one thread per item: its ratings are ikeys == item in ipos (ascending file positions).  head[p] = 1 at the first rating of a run,
// head_of[p] = the head's position, idx[p] = the rating's index inside its run
__global__ __launch_bounds__(256) void k_runs_form(const unsigned *ikeys, const unsigned *ipos, long n, unsigned num_item, const unsigned *prev, int R,
                                                   unsigned *head, unsigned *head_of, unsigned char *idx) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < (long)num_item; it += stride) {
        long lo = 0, hi = n;   // first j with ikeys[j] >= it
        while (lo < hi) { const long mid = (lo + hi) >> 1; if (ikeys[mid] < (unsigned)it) lo = mid + 1; else hi = mid; }
        unsigned h = 0u;
        int len = 0;
        for (long j = lo; j < n && ikeys[j] == (unsigned)it; j++) {
            const unsigned p = ipos[j];
            const unsigned pv = prev[p];
            // (the user's previous rating must lie before the head; a rating of the same user INSIDE the run -- the item rated twice in a row --
            // has prev >= h and starts a new run)
            if (len > 0 && len < R && (pv == 0xFFFFFFFFu || pv < h)) {
                head[p] = 0u; head_of[p] = h; idx[p] = (unsigned char)len;
                len++;
            } else {
                head[p] = 1u; head_of[p] = p; idx[p] = 0;
                h = p; len = 1;
            }
        }
    }
}
// unit_at[p] = exclusive scan of head[] (valid at heads).  Every rating writes itself into slot idx[p] of its run's columns.
__global__ __launch_bounds__(256) void k_runs_fill(const unsigned *user, const unsigned *item, const float *label, long n, const unsigned *unit_at,
                                                   const unsigned *head_of, const unsigned char *idx, long nunit, unsigned *c_item, unsigned *c_user, float *c_label) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
        const unsigned un = unit_at[head_of[p]];
        const int j = idx[p];
        c_user[(size_t)j * (size_t)nunit + un] = user[p];
        c_label[(size_t)j * (size_t)nunit + un] = label[p];
        if (j == 0) c_item[un] = item[p];
    }
}
__global__ __launch_bounds__(256) void k_runs_fill_u32(unsigned *v, long n, unsigned x) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) v[j] = x;
}
static inline int runs_grid(long n) { long g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }
void launch_runs_check(const unsigned *user, const unsigned *item, long n, unsigned nu, unsigned ni, unsigned *flag, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_runs_check, dim3(runs_grid(n)), dim3(256), 0, st, user, item, n, nu, ni, flag);
}
void launch_runs_prev(const unsigned *keys, const unsigned *pos, long n, unsigned *prev, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_runs_prev, dim3(runs_grid(n)), dim3(256), 0, st, keys, pos, n, prev);
}
void launch_runs_form(const unsigned *ikeys, const unsigned *ipos, long n, unsigned num_item, const unsigned *prev, int R, unsigned *head, unsigned *head_of,
                      unsigned char *idx, hipStream_t st) {
    if (n > 0 && num_item > 0) hipLaunchKernelGGL(k_runs_form, dim3(runs_grid((long)num_item)), dim3(256), 0, st, ikeys, ipos, n, num_item, prev, R, head, head_of, idx);
}
void launch_runs_fill(const unsigned *user, const unsigned *item, const float *label, long n, const unsigned *unit_at, const unsigned *head_of,
                      const unsigned char *idx, long nunit, unsigned *c_item, unsigned *c_user, float *c_label, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_runs_fill, dim3(runs_grid(n)), dim3(256), 0, st, user, item, label, n, unit_at, head_of, idx, nunit, c_item, c_user, c_label);
}
__global__ __launch_bounds__(256) void k_runs_iota(unsigned *v, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) v[j] = (unsigned)j;
}
void launch_runs_iota(unsigned *v, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_runs_iota, dim3((int)grid), dim3(256), 0, st, v, n);
}
void launch_runs_fill_u32(unsigned *v, long n, unsigned x, hipStream_t st) {
    if (n > 0) hipLaunchKernelGGL(k_runs_fill_u32, dim3(runs_grid(n)), dim3(256), 0, st, v, n, x);
}

// ------------------------------------------------------------------------------------------------- the pass
// One level of runs.  S.item[s], S.user[j][s], S.label[j][s] (j < R; S.user[j][s] == SLOT_ABSENT: the run has fewer than j + 1 ratings).
template <int LANES, int V, int R, int G>
__global__ __launch_bounds__(256) void k_basicmf_runs_soa(const DevParams P, const RunSchedule S, long begin, long end) {
    constexpr int T = 16 / LANES;
    constexpr int IPS = 64 / LANES;    // runs per wave and row set
    constexpr int K = 4 * LANES * V;
    const int lane = threadIdx.x & 63;
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int m = (lane & 15) / T;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    const int pitch = P.pitch;
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
#pragma unroll 1
    for (int g = 0; g < G; g++) {
        const long w0 = begin + (wave * G + g) * IPS;
        if (w0 >= end) return;
        const long s = w0 + gslot;
        const bool live = s < end;
        const long sc = live ? s : w0;
        const unsigned ir = P.item_off + S.item[sc];
        unsigned ur[R];
        float label[R], bu[R];
        float4 p[R][V], q[V];
        int n = 0;
#pragma unroll
        for (int j = 0; j < R; j++) {
            const unsigned u = S.user[j][sc];
            label[j] = S.label[j][sc];
            const bool on = live && u != (unsigned)SLOT_ABSENT;
            ur[j] = P.user_off + (on ? u : 0u);
            if (on) n = j + 1;
        }
#pragma unroll
        for (int v = 0; v < V; v++) q[v] = live ? load_row_nt<K / 4>(P.W, ir, pitch, m + v * LANES, K) : f4zero();
        float bi = live ? P.bias[ir] : 0.0f;
#pragma unroll
        for (int j = 0; j < R; j++) {
#pragma unroll
            for (int v = 0; v < V; v++) p[j][v] = f4zero();
            bu[j] = 0.0f;
            if (j < n) {
#pragma unroll
                for (int v = 0; v < V; v++) p[j][v] = load_row_nt<K / 4>(P.W, ur[j], pitch, m + v * LANES, K);
                bu[j] = P.bias[ur[j]];
            }
        }
        int nmax = n;   // the wave walks as many steps as its longest run (dot_slots is a wave-wide operation)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) nmax = max(nmax, __shfl_xor(nmax, o));
#pragma unroll
        for (int j = 0; j < R; j++) {
            if (j >= nmax) break;
            // the arithmetic of k_basicmf_slots (= basicmf_wave<K / 4, ., true, true, true>) on (user row j, the item row as the run has left it)
            double bs = 0.0;
            bs += (double)(1.0f * bu[j]); bs += 0.0;
            bs += 0.0;
            bs += (double)(1.0f * bi);
            double sum = (double)P.base_score + bs;
            float4 tu[V], ti[V];
#pragma unroll
            for (int v = 0; v < V; v++) { tu[v] = f4zero(); ti[v] = f4zero(); axpy4(tu[v], p[j][v], 1.0f); axpy4(ti[v], q[v], 1.0f); }
            sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
            const float pred = (float)sum;
            const float err = (label[j] - pred) * 1.0f;
            const float su = P.lr * err * 1.0f;
            const float si = P.lr * err * 1.0f;
            float nbu = bu[j] + su, nbi = bi + si;
            nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
            nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
            const bool act = j < n;
#pragma unroll
            for (int v = 0; v < V; v++) {
                float4 wu = p[j][v], wi = q[v];
                axpy4(wu, ti[v], su);
                axpy4(wi, tu[v], si);
                wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
                wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
                if (act) {
                    store_row<K / 4>(P.W, ur[j], pitch, m + v * LANES, K, wu);
                    q[v] = wi;
                }
            }
            if (act) {
                P.bias[ur[j]] = nbu;
                bi = nbi;
            }
        }
        if (live) {
#pragma unroll
            for (int v = 0; v < V; v++) store_row<K / 4>(P.W, ir, pitch, m + v * LANES, K, q[v]);
            P.bias[ir] = bi;
        }
    }
}
bool basicmf_runs_soa_applies(const DevParams &P) {
    return P.basic_i8 && P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 && P.u_rng.n == 0 && P.i_rng.n == 0 &&
           P.store_mode == 0 && (P.k == 64 || P.k == 128);
}
// k = 64: 8 lanes x 2 chunks per row, 8 runs per wave and row set; k = 128: 16 lanes x 2 chunks, 4 runs
void launch_basicmf_runs_soa(const DevParams &P, const RunSchedule &S, long begin, long end, int R, int G, int block_threads, hipStream_t st) {
    if (end <= begin) return;
    if (G < 1) G = 1;
    if (block_threads != 64 && block_threads != 128 && block_threads != 256) block_threads = 64;
    const int per_wave = P.k == 128 ? 4 : 8;
    const long per_block = (long)(block_threads / 64) * G * per_wave;
    int grid = (int)((end - begin + per_block - 1) / per_block);
    if (P.xcd_remap) grid = (grid + 7) & ~7;
#define RUNS_LAUNCH(L_, R_, G_) hipLaunchKernelGGL((k_basicmf_runs_soa<L_, 2, R_, G_>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end)
#define RUNS_BY_R(L_)                                                                                                          \
    if (R == 2) { if (G >= 2) RUNS_LAUNCH(L_, 2, 2); else RUNS_LAUNCH(L_, 2, 1); }      /* R = user / label columns of the schedule: 2, 4 or 7 */ \
    else if (R == 4) { if (G >= 2) RUNS_LAUNCH(L_, 4, 2); else RUNS_LAUNCH(L_, 4, 1); }                                       \
    else { if (G >= 2) RUNS_LAUNCH(L_, 7, 2); else RUNS_LAUNCH(L_, 7, 1); }
    if (P.k == 128) { RUNS_BY_R(16) } else { RUNS_BY_R(8) }
#undef RUNS_BY_R
#undef RUNS_LAUNCH
}

}  // namespace svdf
