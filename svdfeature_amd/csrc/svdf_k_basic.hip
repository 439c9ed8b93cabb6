// svdf_k_basic.hip -- basicMF kernels (k_basicmf, k_predict_basic) and their launchers
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_device.h"

namespace svdf {

// =====================================================================================
// Kernel 1: basicMF fused SGD step -- no global feature, one user id, one item id, no side
// tables, distinct user/item rows.  One lane group per instance, G groups per wave in flight.
// Traffic per instance (k=64): 2 x 256 B row reads + 2 x 256 B row writes + 2 bias RMW +
// 12..20 B of schedule = the 1072 B/instance algorithmic figure of SURVEY.md 8(d4).
// =====================================================================================
// One wave's G x (64/LPI) instances.  Compile-time switches strip what the contract workload never needs from the
// instruction stream (the kernel is not only HBM- but also VALU-issue-bound: a level is about one occupancy wave, so every
// SIMD runs its ~5 waves' instructions once per launch, ~40 % of the launch time):
//   FULL      num_factor == 4*LPI: no per-lane bounds test around row loads / stores, no masked chunks / tail in the dot
//   FAST      linear link, reg_method 0, user bias on, no per-range decay, no clamp, plain stores: no per-row switches
// (A separate unpredicated path for waves whose slots are all occupied was tried too: it costs 103 instead of 74 VGPRs at
// G=4 and measured slower.)
template <int LPI, int G, bool UNITVAL, bool FULL, bool FAST>
__device__ __forceinline__ void basicmf_wave(const DevParams &P, const BasicSchedule &S, long begin, long end, long w0, int L, int gslot) {
    constexpr int IPW = 64 / LPI;  // instances per wave per group slot
    const int pitch = P.pitch;
    const int k = FULL ? 4 * LPI : P.k;
    const bool use_ubias = FAST ? true : P.no_user_bias == 0;

    bool valid[G];
    unsigned ur[G], ir[G];
    float label[G], ua[G], ia[G], bu[G], bi[G];
    float4 p[G], q[G];

    // ---- stage 1: schedule records (coalesced, one address per lane group)
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long s = w0 + (long)g * IPW + gslot;
        valid[g] = s < end;
        const long sc = valid[g] ? s : begin;
        ur[g] = P.user_off + S.user[sc];
        ir[g] = P.item_off + S.item[sc];
        label[g] = S.label[sc];
        ua[g] = UNITVAL ? 1.0f : S.uval[sc];
        ia[g] = UNITVAL ? 1.0f : S.ival[sc];
    }
    // ---- stage 2: all row gathers of the wave issued back to back
#pragma unroll
    for (int g = 0; g < G; g++) {
        p[g] = f4zero(); q[g] = f4zero(); bu[g] = 0.0f; bi[g] = 0.0f;
        if (valid[g]) {
            if (P.load_mode & 1) {   // default: a level reads each row exactly once -- nontemporal hint, -2.2 ... -4.2 % per pass (tools/ab_knob.py)
                p[g] = load_row_nt<LPI>(P.W, ur[g], pitch, L, k);
                q[g] = load_row_nt<LPI>(P.W, ir[g], pitch, L, k);
            } else {
                p[g] = load_row<LPI>(P.W, ur[g], pitch, L, k);
                q[g] = load_row<LPI>(P.W, ir[g], pitch, L, k);
            }
            if (use_ubias) bu[g] = P.bias[ur[g]];
            bi[g] = P.bias[ir[g]];
        }
    }
    // row-invariant decay factors of the FAST configuration (L2: W *= 1 - lr*wd, multiply skipped when that is 1)
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
    // ---- stage 3: score, gradient, fused update + decay, scatter
#pragma unroll
    for (int g = 0; g < G; g++) {
        // calc_bias (:313-353) in double; "+ 0.0" terms are the svdpp / plugin hooks returning 0.0f
        double bs = 0.0;
        if (use_ubias) { bs += (double)(ua[g] * bu[g]); bs += 0.0; }
        bs += 0.0;
        bs += (double)(ia[g] * bi[g]);
        double sum = (double)P.base_score + bs;
        // prepare_tmp (:354-381): tmp = 0 + row*val
        float4 tu = f4zero(), ti = f4zero();
        axpy4(tu, p[g], ua[g]);
        axpy4(ti, q[g], ia[g]);
        sum += (double)group_dot<LPI>(tu, ti, L, k);
        const float pred = FAST ? (float)sum : map_active((float)sum, P.active_type);
        const float err = (FAST ? label[g] - pred : cal_grad(label[g], pred, P.active_type)) * 1.0f;
        // update_no_decay (:383-427): both rows use the pre-update snapshots
        const float su = P.lr * err * ua[g];
        const float si = P.lr * err * ia[g];
        float4 wu = p[g], wi = q[g];
        axpy4(wu, ti, su);
        axpy4(wi, tu, si);
        float nbu = bu[g] + su, nbi = bi[g] + si;
        // regularize(feature, true) (:286-311)
        if (FAST) {
            wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
            wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
        } else {
            reg_row<LPI>(P, wu, get_wd(P.u_rng, ur[g] - P.user_off, P.wd_user), false, L);
            reg_row<LPI>(P, wi, get_wd(P.i_rng, ir[g] - P.item_off, P.wd_item), true, L);
        }
        nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
        nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
        if (valid[g]) {
            if (FAST) {
                store_row<LPI>(P.W, ur[g], pitch, L, k, wu);
                store_row<LPI>(P.W, ir[g], pitch, L, k, wi);
                // every lane of the group writes the same bias word: one request, no exec-mask branch
                P.bias[ur[g]] = nbu;
                P.bias[ir[g]] = nbi;
            } else {
                store_row_policy<LPI>(P.W, ur[g], pitch, L, k, wu, P.store_mode);
                store_row_policy<LPI>(P.W, ir[g], pitch, L, k, wi, P.store_mode);
                if (L == 0) {
                    if (use_ubias) P.bias[ur[g]] = nbu;
                    P.bias[ir[g]] = nbi;
                }
            }
        }
    }
}

template <int LPI, int G, bool UNITVAL, bool FULL, bool FAST>
__global__ __launch_bounds__(256) void k_basicmf(const DevParams P, const BasicSchedule S, long begin, long end) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    // Hardware deals workgroups to the 8 XCDs round-robin (blockIdx % 8).  With xcd_remap the grid is a multiple
    // of 8 and XCD x works on the x-th contiguous eighth of the batch, so (the batch being sorted by item id)
    // neighbouring item rows / bias sectors meet in ONE XCD's L2 instead of eight.
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = begin + wave * (long)(G * IPW);
    if (w0 < end) basicmf_wave<LPI, G, UNITVAL, FULL, FAST>(P, S, begin, end, w0, L, gslot);
}

// =====================================================================================
// The contract configuration with FEWER lanes per row (basic_i8 knob; k = 64: EIGHT lanes).  The lane-group layout above spends its
// instructions per ROW SET (64 / LPI instances per wave instruction): 60 dependent DPP adds for the dot product at k = 64, ~100 more
// for addressing, the fp64 bias sums, link and decay -- the arithmetic is ~3.5 us of a 13.9 us launch that memory cannot overlap
// (anatomy).  Here a lane holds V 16-byte chunks of a row, chunk m, m + LANES, ... (V fully coalesced pieces
// of 16 LANES bytes per instance, one per load instruction), and the 16 / LANES instances of a 16-lane DPP row are interleaved
// lane by lane, so that the dot product's hand-over is ONE row_shr:T add per step for all of them (dot_slots, svdf_device.h) and
// every wave instruction serves V times the instances.  Same additions in the same order as the lane-group scan; at k = 64
// (LANES = 8, V = 2, G = 4: 32 instances per wave) the per-instance instruction count drops by 40 %.
// the schedule records of one wave's G x 64 / LANES instances starting at position w0 of the level [begin, end)
template <int G> struct BasicRecs { bool valid[G]; unsigned ur[G], ir[G]; float label[G]; };
template <int LANES, int G>
__device__ __forceinline__ BasicRecs<G> basicmf_slots_recs(const DevParams &P, const BasicSchedule &S, long begin, long end, long w0) {
    constexpr int T = 16 / LANES, IPS = 64 / LANES;
    const int lane = threadIdx.x & 63;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    BasicRecs<G> R;
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long s = w0 + (long)g * IPS + gslot;
        R.valid[g] = s < end;
        const long sc = R.valid[g] ? s : begin;
        R.ur[g] = P.user_off + S.user[sc];
        R.ir[g] = P.item_off + S.item[sc];
        R.label[g] = S.label[sc];
    }
    return R;
}
template <int LANES, int V, int G>
__device__ __forceinline__ void basicmf_slots_apply(const DevParams &P, const BasicRecs<G> &R) {
    constexpr int T = 16 / LANES;      // instances interleaved in one DPP row
    constexpr int K = 4 * LANES * V;
    const int lane = threadIdx.x & 63;
    const int m = (lane & 15) / T;
    const int pitch = P.pitch;
    const bool (&valid)[G] = R.valid;
    const unsigned (&ur)[G] = R.ur, (&ir)[G] = R.ir;
    const float (&label)[G] = R.label;
    float bu[G], bi[G];
    float4 p[G][V], q[G][V];
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int v = 0; v < V; v++) { p[g][v] = f4zero(); q[g][v] = f4zero(); }
        bu[g] = 0.0f; bi[g] = 0.0f;
        if (valid[g]) {
#pragma unroll
            for (int v = 0; v < V; v++) p[g][v] = load_row_nt<K / 4>(P.W, ur[g], pitch, m + v * LANES, K);
#pragma unroll
            for (int v = 0; v < V; v++) q[g][v] = load_row_nt<K / 4>(P.W, ir[g], pitch, m + v * LANES, K);
            bu[g] = P.bias[ur[g]];
            bi[g] = P.bias[ir[g]];
        }
    }
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
#pragma unroll
    for (int g = 0; g < G; g++) {
        // the arithmetic of basicmf_wave<K / 4, ., true, true, true>, V chunks per lane
        double bs = 0.0;
        bs += (double)(1.0f * bu[g]); bs += 0.0;
        bs += 0.0;
        bs += (double)(1.0f * bi[g]);
        double sum = (double)P.base_score + bs;
        float4 tu[V], ti[V];
#pragma unroll
        for (int v = 0; v < V; v++) { tu[v] = f4zero(); ti[v] = f4zero(); axpy4(tu[v], p[g][v], 1.0f); axpy4(ti[v], q[g][v], 1.0f); }
        sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
        const float pred = (float)sum;
        const float err = (label[g] - pred) * 1.0f;
        const float su = P.lr * err * 1.0f;
        const float si = P.lr * err * 1.0f;
        float nbu = bu[g] + su, nbi = bi[g] + si;
        nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
        nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
#pragma unroll
        for (int v = 0; v < V; v++) {
            float4 wu = p[g][v], wi = q[g][v];
            axpy4(wu, ti[v], su);
            axpy4(wi, tu[v], si);
            wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
            wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
            p[g][v] = wu; q[g][v] = wi;
        }
        if (valid[g]) {
#pragma unroll
            for (int v = 0; v < V; v++) store_row<K / 4>(P.W, ur[g], pitch, m + v * LANES, K, p[g][v]);
#pragma unroll
            for (int v = 0; v < V; v++) store_row<K / 4>(P.W, ir[g], pitch, m + v * LANES, K, q[g][v]);
            P.bias[ur[g]] = nbu;
            P.bias[ir[g]] = nbi;
        }
    }
}
template <int LANES, int V, int G>
__global__ __launch_bounds__(256) void k_basicmf_slots(const DevParams P, const BasicSchedule S, long begin, long end) {
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = begin + wave * (long)(G * (64 / LANES));
    if (w0 >= end) return;
    basicmf_slots_apply<LANES, V, G>(P, basicmf_slots_recs<LANES, G>(P, S, begin, end, w0));
}
// DEEP, NARROW dependency graphs (round 5: ratings with Zipf-popular items -- the hottest item's ratings are ONE chain, 973 K levels at the
// configs[1] size, 97 % of them narrower than 128 instances): ONE workgroup of 16 waves walks a RUN of narrow levels inside one launch,
// level l's instances, a workgroup barrier, level l + 1 -- the form of k_fewrow_slots_chain (svdf_k_fewrow.hip) for the contract kernel.  The
// waves share the CU's vector L1, so what one of them stored before the barrier is what the others load after it.  Same instances in the
// same level order, same arithmetic: the same bits as one launch per level (tests/test_gpu_chain.py).
template <int LANES, int V>
__global__ __launch_bounds__(1024) void k_basicmf_slots_chain(const DevParams P, const BasicSchedule S, const long *level_ptr, long l0, long l1) {
    // every level of the run fits ONE round of the workgroup (the caller chains levels of at most 16 x 64 / LANES instances; more row sets per
    // wave lengthen the level in proportion: 2.9 / 3.3 / 4.7 us per level at 1 / 2 / 4 sets).  Requesting level l + 1's records ahead of level
    // l's rows was built and measured: slower (3.5 against 2.8 s per pass on Zipf(0.7) items at the configs[1] size, profiles/r05_zipf_chain.txt).
    constexpr int IPW = 64 / LANES;
    const long w = threadIdx.x >> 6;
    for (long l = l0; l < l1; l++) {
        const long begin = level_ptr[l], end = level_ptr[l + 1];
        const long w0 = begin + w * IPW;
        if (w0 < end) basicmf_slots_apply<LANES, V, 1>(P, basicmf_slots_recs<LANES, 1>(P, S, begin, end, w0));
        __syncthreads();
    }
}
// a run of narrow levels [l0, l1) in one launch; false when the configuration has no chained form (the caller launches level by level)
bool launch_basicmf_chain(const DevParams &P, const BasicSchedule &S, const long *d_level_ptr, long l0, long l1, hipStream_t st) {
    const bool slots_cfg = P.basic_i8 && S.uval == nullptr && P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 &&
                           P.user_nonnegative == 0 && P.u_rng.n == 0 && P.i_rng.n == 0 && P.store_mode == 0 && P.k == 64;
    if (!slots_cfg) return false;
    if (d_level_ptr == nullptr) return true;   // query
    hipLaunchKernelGGL((k_basicmf_slots_chain<8, 2>), dim3(1), dim3(1024), 0, st, P, S, d_level_ptr, l0, l1);
    return true;
}

template <int LPI, bool UNITVAL>
__global__ __launch_bounds__(256) void k_predict_basic(const DevParams P, const BasicSchedule S, long n, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = gidx; s < n; s += stride) {
        const unsigned ur = P.user_off + S.user[s], ir = P.item_off + S.item[s];
        const float ua = UNITVAL ? 1.0f : S.uval[s], ia = UNITVAL ? 1.0f : S.ival[s];
        double bs = 0.0;
        if (P.no_user_bias == 0) bs += (double)(ua * P.bias[ur]);
        bs += (double)(ia * P.bias[ir]);
        double sum = (double)P.base_score + bs;
        float4 tu = f4zero(), ti = f4zero();
        axpy4(tu, load_row<LPI>(P.W, ur, P.pitch, L, P.k), ua);
        axpy4(ti, load_row<LPI>(P.W, ir, P.pitch, L, P.k), ia);
        sum += (double)group_dot<LPI>(tu, ti, L, P.k);
        if (L == 0) out[s] = map_active((float)sum, P.active_type);
    }
}

template <int LPI>
static void launch_basicmf_lpi(const DevParams &P, const BasicSchedule &S, long begin, long end, int G, int block_threads, hipStream_t st) {
    const long n = end - begin;
    const bool unit = S.uval == nullptr;
    auto go = [&](auto gtag) {
        constexpr int GG = decltype(gtag)::value;
        const long per_block = (long)(block_threads / 64) * GG * (64 / LPI);
        int grid = (int)((n + per_block - 1) / per_block);
        if (P.xcd_remap) grid = (grid + 7) & ~7;
        // specialised instruction stream for the configuration of the contract workload, general one otherwise
        const bool fast = P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 &&
                          P.u_rng.n == 0 && P.i_rng.n == 0 && P.store_mode == 0;
        auto slots = [&](auto lanes, auto vv) {   // LANES lanes per row, V chunks per lane: GG row sets of 64 / LANES instances per wave
            constexpr int LANES = decltype(lanes)::value, V = decltype(vv)::value;
            const long per_block8 = (long)(block_threads / 64) * GG * (64 / LANES);
            int grid8 = (int)((n + per_block8 - 1) / per_block8);
            if (P.xcd_remap) grid8 = (grid8 + 7) & ~7;
            hipLaunchKernelGGL((k_basicmf_slots<LANES, V, GG>), dim3(grid8), dim3(block_threads), 0, st, P, S, begin, end);
        };
        if (unit && fast && LPI == 16 && P.k == 64 && P.basic_i8) {
            slots(std::integral_constant<int, 8>(), std::integral_constant<int, 2>());
        } else if (unit && fast && LPI == 64 && P.k == 256 && P.basic_i8) {
            slots(std::integral_constant<int, 16>(), std::integral_constant<int, 4>());
        } else if (unit && fast && P.k == 4 * LPI) hipLaunchKernelGGL((k_basicmf<LPI, GG, true, true, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else if (unit) hipLaunchKernelGGL((k_basicmf<LPI, GG, true, false, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else hipLaunchKernelGGL((k_basicmf<LPI, GG, false, false, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    };
    switch (G) {
    case 1: go(std::integral_constant<int, 1>()); break;
    case 2: go(std::integral_constant<int, 2>()); break;
    case 3: go(std::integral_constant<int, 3>()); break;
    case 5: go(std::integral_constant<int, 5>()); break;
    case 6: go(std::integral_constant<int, 6>()); break;
    case 8: go(std::integral_constant<int, 8>()); break;
    default: go(std::integral_constant<int, 4>()); break;
    }
}

void launch_basicmf(const DevParams &P, const BasicSchedule &S, long begin, long end, int groups_per_wave, int block_threads, hipStream_t st) {
    if (end <= begin) return;
    // 0 = tuned default (tools/sweep_knobs.py on MI355X with the FULL/FAST specialisation): k=64 (16 lanes per row) wants
    // 4 row sets in flight per wave in 64-thread blocks (24.9 ms/pass; 26.3 with one row set), k=256 two row sets, every
    // other width one row set per wave in 256-thread blocks
    const int lpi_ = lanes_per_instance(P.k);
    // (tools/ab_knob.py, in-process A/B: k=64 8 lanes x 2 chunks with 4 row sets 23.6 vs 24.4 ms per 100 M; k=256 16 lanes x 4 chunks with
    // one row set 18.5 vs 20.2 ms per 25 M; k=128 16 lanes x 2 chunks 22.3 vs 22.3 ms per 50 M: no gain, the lane-group kernel stays)
    const bool slots_cfg = P.basic_i8 && S.uval == nullptr && P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 &&
                           P.user_nonnegative == 0 && P.u_rng.n == 0 && P.i_rng.n == 0;   // the configuration k_basicmf_slots is specialised for
    const bool slots256 = slots_cfg && P.k == 256;
    if (groups_per_wave <= 0) {
        groups_per_wave = lpi_ == 16 ? 4 : (lpi_ == 64 ? (slots256 ? 1 : 2) : 1);
        // k = 64 in the 8-lane layout: 8 instances per row set.  A full-size level (53 K instances) wants 4 row sets per wave (1.6 waves
        // per SIMD: 23.6 ms per pass against 26.6 with 2 and 27.5 with 1), the 17 K-instance levels of one rank's window of an 8-GPU
        // run want 1 (7.66 ms per pass against 8.09 with 2 and 9.10 with 4, tools/shard8_knobs.sh): aim at ~1 800 waves per launch
        if (slots_cfg && P.k == 64) groups_per_wave = (int)std::min<long>(4, std::max<long>(1, (end - begin + 7200) / 14400));
    }
    if (block_threads <= 0) block_threads = lpi_ == 16 ? 64 : 256;
    SVDF_DISPATCH_LPI(lanes_per_instance(P.k), launch_basicmf_lpi<LPI>(P, S, begin, end, groups_per_wave, block_threads, st));
}
void launch_predict_basic(const DevParams &P, const BasicSchedule &S, long n, float *out, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    if (S.uval == nullptr) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_predict_basic<LPI, true>), dim3(grid), dim3(256), 0, st, P, S, n, out)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_predict_basic<LPI, false>), dim3(grid), dim3(256), 0, st, P, S, n, out)); }
}

}  // namespace svdf
