// svdf_k_fewrow.hip -- the few-row fused step specialised for its usual configuration (rank pairs, BASELINE configs[4]; basicMF
// with side ids): no global feature in the data set, no relaxed ids, L2 decay (reg_method 0), no per-range decay, no clamp.
// Same arithmetic, same order, same bits as k_fused (svdf_k_fused.hip) -- which stays the general few-row kernel -- with the
// instruction stream stripped of everything this configuration never executes: the global-bias paths, the per-row decay
// switch and range lookup, the relaxed-id tests, the exec-mask branch around the bias stores.  k_fused<32,1,2> spends 448 VALU
// + 165 SALU instructions per wave on a pair (profiles/r02_pmc_pairwise.txt) and a level of 11.5 K pairs is one occupancy
// wave of the chip, so instruction issue is ~40 % of a launch: the same lever as k_basicmf's FULL / FAST specialisation.
#include "svdf_device.h"

namespace svdf {

template <int LPI, int NU, int NI, bool FULL>
__global__ __launch_bounds__(256) void k_fewrow_fast(const DevParams P, const FusedSchedule S, long begin, long end) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    long tile = blockIdx.x;   // XCD-aware tile mapping, see k_basicmf
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long s = begin + wave * (long)IPW + gslot;
    if (begin + wave * (long)IPW >= end) return;
    const bool valid = s < end;
    const long sc = valid ? s : begin;
    const int k = FULL ? 4 * LPI : P.k, pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;

    unsigned ur[NU], ir[NI];
    float ua[NU], ia[NI], bu[NU], bi[NI];
    float4 p[NU], q[NI];
    const float label = S.label[sc];
#pragma unroll
    for (int a = 0; a < NU; a++) { ur[a] = valid ? S.uidx[a][sc] : (unsigned)SLOT_ABSENT; ua[a] = S.uval[a][sc]; }
#pragma unroll
    for (int b = 0; b < NI; b++) { ir[b] = valid ? S.iidx[b][sc] : (unsigned)SLOT_ABSENT; ia[b] = S.ival[b][sc]; }
#pragma unroll
    for (int a = 0; a < NU; a++) {
        p[a] = f4zero(); bu[a] = 0.0f;
        if (ur[a] != SLOT_ABSENT) {
            p[a] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.user_off + ur[a], pitch, L, k) : load_row<LPI>(P.W, P.user_off + ur[a], pitch, L, k);
            if (use_ubias) bu[a] = P.bias[P.user_off + ur[a]];
        }
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
        q[b] = f4zero(); bi[b] = 0.0f;
        if (ir[b] != SLOT_ABSENT) {
            q[b] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.item_off + ir[b], pitch, L, k) : load_row<LPI>(P.W, P.item_off + ir[b], pitch, L, k);
            bi[b] = P.bias[P.item_off + ir[b]];
        }
    }
    // row-invariant decay factors (L2: W *= 1 - lr*wd, the multiply skipped when that rounds to one; biases are plain multiplies)
    const float dec_u = snap_to_one(1.0f - P.lr * P.wd_user), dec_i = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;
    // calc_bias (:313-353) in double, prepare_tmp (:354-381), dot, link, gradient -- as in k_fused
    double bs = 0.0;
    if (use_ubias) {
#pragma unroll
        for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) bs += (double)(ua[a] * bu[a]);
    }
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) bs += (double)(ia[b] * bi[b]);
    double sum = (double)P.base_score + bs;
    float4 tu = f4zero(), ti = f4zero();
#pragma unroll
    for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) axpy4(tu, p[a], ua[a]);
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) axpy4(ti, q[b], ia[b]);
    sum += (double)group_dot<LPI>(tu, ti, L, k);
    const float pred = map_active((float)sum, P.active_type);
    const float err = cal_grad(label, pred, P.active_type) * 1.0f;
    const float lr = P.lr;
    // update_no_decay (:383-427) + regularize (:286-311), rows written once; every lane of the group writes the same bias word
#pragma unroll
    for (int a = 0; a < NU; a++) {
        if (ur[a] == SLOT_ABSENT) continue;
        const float su = lr * err * ua[a];
        float4 w = p[a];
        axpy4(w, ti, su);
        w.x = w.x * dec_u; w.y = w.y * dec_u; w.z = w.z * dec_u; w.w = w.w * dec_u;
        store_row<LPI>(P.W, P.user_off + ur[a], pitch, L, k, w);
        if (use_ubias) P.bias[P.user_off + ur[a]] = (bu[a] + su) * dec_ub;
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
        if (ir[b] == SLOT_ABSENT) continue;
        const float si = lr * err * ia[b];
        float4 w = q[b];
        axpy4(w, tu, si);
        w.x = w.x * dec_i; w.y = w.y * dec_i; w.z = w.z * dec_i; w.w = w.w * dec_i;
        store_row<LPI>(P.W, P.item_off + ir[b], pitch, L, k, w);
        P.bias[P.item_off + ir[b]] = (bi[b] + si) * dec_ib;
    }
}

// ---- full rows with FEWER lanes per row (fewrow_i16 knob): a lane holds V chunks of a row -- chunk m, m + LANES, m + 2 LANES, ... (V
// coalesced pieces of 16 * LANES bytes per instance, one per load instruction) --, the T = 16 / LANES instances of a 16-lane DPP row are
// interleaved lane by lane (instance a on lanes T m + a), 64 / LANES instances per wave instead of 64 / (LANES V): the layout idea of
// k_basicmf_i8 (svdf_k_basic.hip), every wave instruction serves V times the instances.  The dot product is the reference's chain:
// the chunks of slot 0 in lane order (LANES - 1 row_shr:T steps), the finished sums rotate from the row's last lanes to its first
// (row_ror:T) and are folded into the addend of the next slot's first chunk -- the carry trick of group_dot<32> --, and so on
// through the slots (dot_slots, svdf_device.h).  Same additions in the same order.  k = 128: LANES = 16, V = 2 (one instance per DPP row).
// the schedule record of one instance (the level-sorted columns of FusedSchedule): what a chained launch requests one level ahead
template <int NU, int NI>
struct FewrowRec { float label; unsigned ur[NU], ir[NI]; float ua[NU], ia[NI]; };
template <int LANES, int NU, int NI>
__device__ __forceinline__ FewrowRec<NU, NI> fewrow_slots_rec(const FusedSchedule &S, long begin, long end, long wave) {
    constexpr int T = 16 / LANES, IPW = 64 / LANES;
    const int lane = threadIdx.x & 63;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    const long s = begin + wave * (long)IPW + gslot;
    const bool valid = s < end;
    const long sc = valid ? s : begin;
    FewrowRec<NU, NI> R;
    R.label = S.label[sc];
#pragma unroll
    for (int a = 0; a < NU; a++) { R.ur[a] = valid ? S.uidx[a][sc] : (unsigned)SLOT_ABSENT; R.ua[a] = S.uval[a][sc]; }
#pragma unroll
    for (int b = 0; b < NI; b++) { R.ir[b] = valid ? S.iidx[b][sc] : (unsigned)SLOT_ABSENT; R.ia[b] = S.ival[b][sc]; }
    return R;
}
// the instances [begin + wave IPW, begin + (wave + 1) IPW) of a level, by one wave, their records in R
template <int LANES, int V, int NU, int NI>
__device__ __forceinline__ void fewrow_slots_apply(const DevParams &P, const FewrowRec<NU, NI> &R) {
    constexpr int T = 16 / LANES;          // instances interleaved in one DPP row
    constexpr int K = 4 * LANES * V;
    const int lane = threadIdx.x & 63;
    const int m = (lane & 15) / T;
    const int pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;

    unsigned ur[NU], ir[NI];
    float ua[NU], ia[NI], bu[NU], bi[NI];
    float4 p[NU][V], q[NI][V];
    const float label = R.label;
#pragma unroll
    for (int a = 0; a < NU; a++) { ur[a] = R.ur[a]; ua[a] = R.ua[a]; }
#pragma unroll
    for (int b = 0; b < NI; b++) { ir[b] = R.ir[b]; ia[b] = R.ia[b]; }
#pragma unroll
    for (int a = 0; a < NU; a++) {
#pragma unroll
        for (int v = 0; v < V; v++) p[a][v] = f4zero();
        bu[a] = 0.0f;
        if (ur[a] != SLOT_ABSENT) {
#pragma unroll
            for (int v = 0; v < V; v++) p[a][v] = load_row_nt<K / 4>(P.W, P.user_off + ur[a], pitch, m + v * LANES, K);
            if (use_ubias) bu[a] = P.bias[P.user_off + ur[a]];
        }
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
#pragma unroll
        for (int v = 0; v < V; v++) q[b][v] = f4zero();
        bi[b] = 0.0f;
        if (ir[b] != SLOT_ABSENT) {
#pragma unroll
            for (int v = 0; v < V; v++) q[b][v] = load_row_nt<K / 4>(P.W, P.item_off + ir[b], pitch, m + v * LANES, K);
            bi[b] = P.bias[P.item_off + ir[b]];
        }
    }
    const float dec_u = snap_to_one(1.0f - P.lr * P.wd_user), dec_i = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;
    // the arithmetic of k_fewrow_fast<K / 4, NU, NI, true>, V chunks per lane
    double bs = 0.0;
    if (use_ubias) {
#pragma unroll
        for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) bs += (double)(ua[a] * bu[a]);
    }
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) bs += (double)(ia[b] * bi[b]);
    double sum = (double)P.base_score + bs;
    float4 tu[V], ti[V];
#pragma unroll
    for (int v = 0; v < V; v++) { tu[v] = f4zero(); ti[v] = f4zero(); }
#pragma unroll
    for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) {
#pragma unroll
        for (int v = 0; v < V; v++) axpy4(tu[v], p[a][v], ua[a]);
    }
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) {
#pragma unroll
        for (int v = 0; v < V; v++) axpy4(ti[v], q[b][v], ia[b]);
    }
    sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
    const float pred = map_active((float)sum, P.active_type);
    const float err = cal_grad(label, pred, P.active_type) * 1.0f;
    const float lr = P.lr;
#pragma unroll
    for (int a = 0; a < NU; a++) {
        if (ur[a] == SLOT_ABSENT) continue;
        const float su = lr * err * ua[a];
#pragma unroll
        for (int v = 0; v < V; v++) {
            float4 w = p[a][v];
            axpy4(w, ti[v], su);
            w.x = w.x * dec_u; w.y = w.y * dec_u; w.z = w.z * dec_u; w.w = w.w * dec_u;
            store_row<K / 4>(P.W, P.user_off + ur[a], pitch, m + v * LANES, K, w);
        }
        if (use_ubias) P.bias[P.user_off + ur[a]] = (bu[a] + su) * dec_ub;
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
        if (ir[b] == SLOT_ABSENT) continue;
        const float si = lr * err * ia[b];
#pragma unroll
        for (int v = 0; v < V; v++) {
            float4 w = q[b][v];
            axpy4(w, tu[v], si);
            w.x = w.x * dec_i; w.y = w.y * dec_i; w.z = w.z * dec_i; w.w = w.w * dec_i;
            store_row<K / 4>(P.W, P.item_off + ir[b], pitch, m + v * LANES, K, w);
        }
        P.bias[P.item_off + ir[b]] = (bi[b] + si) * dec_ib;
    }
}
template <int LANES, int V, int NU, int NI>
__global__ __launch_bounds__(256) void k_fewrow_slots(const DevParams P, const FusedSchedule S, long begin, long end) {
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (begin + wave * (long)(64 / LANES) >= end) return;
    fewrow_slots_apply<LANES, V, NU, NI>(P, fewrow_slots_rec<LANES, NU, NI>(S, begin, end, wave));
}
// DEEP, NARROW dependency graphs (a rank pass in the reference's own file order, apex_svd_data.cpp:946-965: user-grouped pairs, 58 693 levels of
// ~54 pairs): a launch per level costs its boundary and a cold start (~5.4 us) for 14 waves of work.  ONE workgroup of 16 waves walks a RUN of
// narrow levels inside one launch instead: level l's instances, a workgroup barrier, level l + 1 -- the waves share the CU's vector L1, so what
// one of them stored before the barrier is what the others load after it (workgroup scope needs no cache action).  Same instances in the same
// level order, same arithmetic: the same bits as one launch per level (tests/test_gpu_sched.py).
template <int LANES, int V, int NU, int NI>
__global__ __launch_bounds__(1024) void k_fewrow_slots_chain(const DevParams P, const FusedSchedule S, const long *level_ptr, long l0, long l1) {
    constexpr int IPW = 64 / LANES;
    const long w0 = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (long l = l0; l < l1; l++) {
        const long begin = level_ptr[l], end = level_ptr[l + 1];
        for (long wave = w0; begin + wave * (long)IPW < end; wave += nw)
            fewrow_slots_apply<LANES, V, NU, NI>(P, fewrow_slots_rec<LANES, NU, NI>(S, begin, end, wave));
        __syncthreads();
    }
}
// (requesting the next level's records before the barrier was built and measured: no gain on the 54-pair levels of a real rank pass -- a level there
// is its rows' latency chain, ~4.5 us, not its launch -- and slower on one-pair levels; the chain pays on streams that are nearly totally ordered,
// 3.4 -> 1.5 us per level, and is neutral otherwise: profiles/r04_chain_probe.txt)

// ---- one user id, one item id, up to four INLINE global slots (the neighbourhood shape of BASELINE configs[3]: FusedSchedule::gsi / gsv),
// the usual configuration of fewrow_fast_applies plus reg_global 0 / 1 through reg_gbias.  A level of this shape is ~280 instances: the
// launch is pure latency, and k_fused's generality (two row slots each side, CSR and inline global paths, relaxed ids, range lookups, per-row
// regulariser switch: ~12 KB of ISA, 754 instruction-cache misses per launch on ~140 CUs, profiles/r03_pmc_neighbourhood_detail.txt) is
// paid on every one of 14 211 levels.  Same arithmetic, same order, same bits: 16 lanes x 2 chunks per row at k = 128 (8 x 2 at k = 64),
// the global biases' update + decay id by id (distinct ids: apex_svd_base.h:384-387, 288-292).
template <int LANES, int V>
__global__ __launch_bounds__(256) void k_fewrow_gslots(const DevParams P, const FusedSchedule S, long begin, long end) {
    constexpr int T = 16 / LANES, IPW = 64 / LANES, K = 4 * LANES * V, GR = 4;
    const int lane = threadIdx.x & 63;
    const int m = (lane & 15) / T;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long s = begin + wave * (long)IPW + gslot;
    if (begin + wave * (long)IPW >= end) return;
    const bool valid = s < end;
    const long sc = valid ? s : begin;
    const int pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;
    const float label = S.label[sc];
    const unsigned ur = P.user_off + S.uidx[0][sc], ir = P.item_off + S.iidx[0][sc];
    const float ua = S.uval[0][sc], ia = S.ival[0][sc];
    unsigned gid[GR];
    float gv[GR], gb[GR];
    bool gon[GR];
#pragma unroll
    for (int j = 0; j < GR; j++) { gid[j] = S.gsi[j][sc]; gv[j] = S.gsv[j][sc]; gon[j] = gid[j] != SLOT_ABSENT; }
    float4 p[V], q[V];
#pragma unroll
    for (int v = 0; v < V; v++) { p[v] = load_row_nt<K / 4>(P.W, ur, pitch, m + v * LANES, K); q[v] = load_row_nt<K / 4>(P.W, ir, pitch, m + v * LANES, K); }
    const float bu = use_ubias ? P.bias[ur] : 0.0f, bi = P.bias[ir];
#pragma unroll
    for (int j = 0; j < GR; j++) gb[j] = gon[j] ? P.g_bias[gid[j]] : 0.0f;
    const float dec_u = snap_to_one(1.0f - P.lr * P.wd_user), dec_i = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;
    double bs = 0.0;
#pragma unroll
    for (int j = 0; j < GR; j++) if (gon[j]) bs += (double)(gv[j] * gb[j]);
    if (use_ubias) bs += (double)(ua * bu);
    bs += (double)(ia * bi);
    double sum = (double)P.base_score + bs;
    float4 tu[V], ti[V];
#pragma unroll
    for (int v = 0; v < V; v++) { tu[v] = f4zero(); ti[v] = f4zero(); axpy4(tu[v], p[v], ua); axpy4(ti[v], q[v], ia); }
    sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
    const float pred = map_active((float)sum, P.active_type);
    const float err = cal_grad(label, pred, P.active_type) * 1.0f;
    const float lr = P.lr;
    if (!valid) return;
#pragma unroll
    for (int j = 0; j < GR; j++) if (gon[j] && m == 0) P.g_bias[gid[j]] = reg_gbias(P, gid[j], gb[j] + lr * err * gv[j]);
    const float su = lr * err * ua, si = lr * err * ia;
#pragma unroll
    for (int v = 0; v < V; v++) {
        float4 w = p[v];
        axpy4(w, ti[v], su);
        w.x = w.x * dec_u; w.y = w.y * dec_u; w.z = w.z * dec_u; w.w = w.w * dec_u;
        store_row<K / 4>(P.W, ur, pitch, m + v * LANES, K, w);
        float4 x = q[v];
        axpy4(x, tu[v], si);
        x.x = x.x * dec_i; x.y = x.y * dec_i; x.z = x.z * dec_i; x.w = x.w * dec_i;
        store_row<K / 4>(P.W, ir, pitch, m + v * LANES, K, x);
    }
    if (m == 0) {
        if (use_ubias) P.bias[ur] = (bu + su) * dec_ub;
        P.bias[ir] = (bi + si) * dec_ib;
    }
}
// data sets this kernel is specialised for: inline global slots, one user id and one item id in EVERY instance (no absent slot), the
// configuration of fewrow_fast_applies on the factor side, no lazy / ranged decay of the global biases, full rows of 64 / 128 factors
bool fewrow_gslots_applies(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, bool dense_slots) {
    return S.gsi[0] != nullptr && dense_slots && max_nu == 1 && max_ni == 1 && !P.relax_global && P.relax_user_from == 0xFFFFFFFFu &&
           P.relax_item_from == 0xFFFFFFFFu && P.reg_method == 0 && P.reg_global < 4 && P.u_rng.n == 0 && P.i_rng.n == 0 && P.user_nonnegative == 0 &&
           P.g_stride == 1 && (P.k == 64 || P.k == 128);
}
void launch_fewrow_gslots(const DevParams &P, const FusedSchedule &S, long begin, long end, hipStream_t st) {
    if (end <= begin) return;
    const long n = end - begin;
    if (P.k == 128) {
        const int grid = (int)((n + 3) / 4);
        hipLaunchKernelGGL((k_fewrow_gslots<16, 2>), dim3(grid), dim3(64), 0, st, P, S, begin, end);
    } else {
        const int grid = (int)((n + 7) / 8);
        hipLaunchKernelGGL((k_fewrow_gslots<8, 2>), dim3(grid), dim3(64), 0, st, P, S, begin, end);
    }
}

template <int LANES, int V, int NU, int NI>
static void launch_fewrow_slots_shape(const DevParams &P, const FusedSchedule &S, long begin, long end, int block_threads, hipStream_t st) {
    const long n = end - begin;
    const long per_block = (long)(block_threads / 64) * (64 / LANES);
    int grid = (int)((n + per_block - 1) / per_block);
    if (P.xcd_remap) grid = (grid + 7) & ~7;
    hipLaunchKernelGGL((k_fewrow_slots<LANES, V, NU, NI>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
}
template <int LANES, int V>
static void launch_fewrow_slots(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int block_threads, hipStream_t st) {
    if (max_nu <= 1 && max_ni <= 1) launch_fewrow_slots_shape<LANES, V, 1, 1>(P, S, begin, end, block_threads, st);
    else if (max_nu <= 1) launch_fewrow_slots_shape<LANES, V, 1, 2>(P, S, begin, end, block_threads, st);
    else if (max_ni <= 1) launch_fewrow_slots_shape<LANES, V, 2, 1>(P, S, begin, end, block_threads, st);
    else launch_fewrow_slots_shape<LANES, V, 2, 2>(P, S, begin, end, block_threads, st);
}

template <int LPI, int NU, int NI>
static void launch_fewrow_shape(const DevParams &P, const FusedSchedule &S, long begin, long end, int block_threads, hipStream_t st) {
    const long n = end - begin;
    const long per_block = (long)(block_threads / 64) * (64 / LPI);
    int grid = (int)((n + per_block - 1) / per_block);
    if (P.xcd_remap) grid = (grid + 7) & ~7;
    if (P.k == 4 * LPI) hipLaunchKernelGGL((k_fewrow_fast<LPI, NU, NI, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    else hipLaunchKernelGGL((k_fewrow_fast<LPI, NU, NI, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
}
template <int LPI>
static void launch_fewrow_lpi(const DevParams &P, const FusedSchedule &S, int nu, int ni, long begin, long end, int bt, hipStream_t st) {
    if (nu <= 1 && ni <= 1) launch_fewrow_shape<LPI, 1, 1>(P, S, begin, end, bt, st);
    else if (nu <= 1) launch_fewrow_shape<LPI, 1, 2>(P, S, begin, end, bt, st);
    else if (ni <= 1) launch_fewrow_shape<LPI, 2, 1>(P, S, begin, end, bt, st);
    else launch_fewrow_shape<LPI, 2, 2>(P, S, begin, end, bt, st);
}
// a run of narrow levels [l0, l1) in one launch (k_fewrow_slots_chain); false when the shape has no chain form (the caller launches level by level)
bool launch_fewrow_chain(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, const long *d_level_ptr, long l0, long l1, hipStream_t st) {
    if (!(P.fewrow_fast && fewrow_fast_applies(P, S) && P.k == 128 && P.fewrow_i16 == 1)) return false;
    if (l1 <= l0) return true;
    if (max_nu <= 1 && max_ni <= 1) hipLaunchKernelGGL((k_fewrow_slots_chain<16, 2, 1, 1>), dim3(1), dim3(1024), 0, st, P, S, d_level_ptr, l0, l1);
    else if (max_nu <= 1) hipLaunchKernelGGL((k_fewrow_slots_chain<16, 2, 1, 2>), dim3(1), dim3(1024), 0, st, P, S, d_level_ptr, l0, l1);
    else if (max_ni <= 1) hipLaunchKernelGGL((k_fewrow_slots_chain<16, 2, 2, 1>), dim3(1), dim3(1024), 0, st, P, S, d_level_ptr, l0, l1);
    else hipLaunchKernelGGL((k_fewrow_slots_chain<16, 2, 2, 2>), dim3(1), dim3(1024), 0, st, P, S, d_level_ptr, l0, l1);
    return true;
}
// true when the configuration / data set is the one this kernel is specialised for
bool fewrow_fast_applies(const DevParams &P, const FusedSchedule &S) {
    return S.gptr == nullptr && !P.relax_global && P.relax_user_from == 0xFFFFFFFFu && P.relax_item_from == 0xFFFFFFFFu &&
           P.reg_method == 0 && P.u_rng.n == 0 && P.i_rng.n == 0 && P.user_nonnegative == 0;
}
void launch_fewrow_fast(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int block_threads, hipStream_t st) {
    if (end <= begin) return;
    // one-wave workgroups: a level of these workloads is a few hundred to a few thousand waves, and 64-thread blocks spread them over
    // four times as many CUs (tools/ab_knob.py: neighbourhood 87.3 vs 91.8 ms per pass with 256; rank pairs 45.2 vs 45.4: flat)
    if (block_threads <= 0) block_threads = 64;
    if (P.k == 128 && P.fewrow_i16 == 1) { launch_fewrow_slots<16, 2>(P, S, max_nu, max_ni, begin, end, block_threads, st); return; }
    // (eight lanes with four chunks each measured slower at k = 128: 46.7 vs 45.2 ms per 50 M pairs; 32 lanes, one chunk: 48.9)
    SVDF_DISPATCH_LPI(lanes_per_instance(P.k), launch_fewrow_lpi<LPI>(P, S, max_nu, max_ni, begin, end, block_threads, st));
}

}  // namespace svdf
