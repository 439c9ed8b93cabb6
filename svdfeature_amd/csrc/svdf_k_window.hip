// svdf_k_window.hip -- the WINDOW-MINIBATCH step of the multi-GPU path: three launches per exchange window
// instead of one launch per conflict-free level.
// (part of the gfx950 kernel set described at the top of svdf_device.h)
//
// With N ranks the contract is |dRMSE| <= 1e-4, not bit parity, and users are private to a rank (rank = user % N).  So inside a window
//   * the USER side stays exact: a lane group walks ONE user's instances of the window in file order with the user's row and bias in
//     registers -- the reference's update_inner (apex_svd_base.h:456-462) instance after instance;
//   * the ITEM side is read as it was at the window start (W_item / i_bias are not written inside a window: they are their own
//     snapshot, L2 / Infinity-Cache resident at 25.6 MB) and what update_inner WOULD have changed on it, (q + s*p)*decay - q, is
//     stored per instance into a contribution slot;
//   * k_window_items sums every item's contributions IN FILE ORDER (slots are laid out item by item, so the sum is a streaming
//     read; no float atomics: the result is deterministic and equals oracle/svdf_oracle.c: svdo_update_csr_batch_stale bit for bit)
//     straight into the wire buffer of the all-reduce, fp32 or fp16;
//   * k_delta_addto adds the all-reduced sum to the replicated ranges on every rank.
// Pack, snapshot copy and the ~15 dependent level launches per window of the level-scheduled shard are gone.
#include "svdf_device.h"

namespace svdf {

// One user's instances of a window: entries [begin, begin + count) of the user-grouped columns.
// Records are in LAUNCH order: users sorted by count (descending), so the lane groups of a wave run the same number of iterations.

// ------------------------------------------------------------------------------------------------- kernel A, any width <= 256
template <int LPI, bool UNITVAL, int NI>
__global__ __launch_bounds__(256) void k_window_users(const DevParams P, const WindowSchedule S) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    const long uidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + gslot;
    const long wave_first = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW;
    if (wave_first >= S.nusers) return;
    const bool valid = uidx < S.nusers;
    const WinUser rec = S.urec[valid ? uidx : wave_first];
    const int maxc = S.urec[wave_first].count;          // records are sorted by count, descending: the wave's first user has the most
    const int pitch = P.pitch, k = P.k;
    const bool use_ubias = P.no_user_bias == 0;
    const unsigned ur = P.user_off + rec.user;
    float4 p = load_row<LPI>(P.W, ur, pitch, L, k);
    float bu = use_ubias ? P.bias[ur] : 0.0f;
    const float wd_u = get_wd(P.u_rng, rec.user, P.wd_user);
    for (int j = 0; j < maxc; j++) {
        const bool act = valid && j < rec.count;
        const long s = (long)rec.begin + (act ? j : 0);
        unsigned item[NI];
        float ia[NI], bi[NI];
        float4 q[NI];
        item[0] = S.item[s];
        ia[0] = (UNITVAL && NI == 1) ? 1.0f : S.ival[s];
        if (NI == 2) { item[1] = S.item1[s]; ia[1] = S.ival1[s]; }
        const float label = S.label ? S.label[s] : 1.0f;
        const float ua = UNITVAL ? 1.0f : S.uval[s];
#pragma unroll
        for (int e = 0; e < NI; e++) {
            q[e] = load_row<LPI>(P.W, P.item_off + item[e], pitch, L, k);
            bi[e] = P.bias[P.item_off + item[e]];
        }
        // calc_bias (:313-353) in double; "+ 0.0" terms are the svdpp / plugin hooks returning 0.0f
        double bs = 0.0;
        if (use_ubias) { bs += (double)(ua * bu); bs += 0.0; }
        bs += 0.0;
#pragma unroll
        for (int e = 0; e < NI; e++) bs += (double)(ia[e] * bi[e]);
        double sum = (double)P.base_score + bs;
        float4 tu = f4zero(), ti = f4zero();
        axpy4(tu, p, ua);
#pragma unroll
        for (int e = 0; e < NI; e++) axpy4(ti, q[e], ia[e]);
        sum += (double)group_dot<LPI>(tu, ti, L, k);
        const float pred = map_active((float)sum, P.active_type);
        const float err = cal_grad(label, pred, P.active_type) * 1.0f;
        const float su = P.lr * err * ua;
        float4 wu = p;
        axpy4(wu, ti, su);
        float nbu = bu + su;
        reg_row<LPI>(P, wu, wd_u, false, L);
        nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
#pragma unroll
        for (int e = 0; e < NI; e++) {
            const float si = P.lr * err * ia[e];
            float4 wi = q[e];
            axpy4(wi, tu, si);
            float nbi = bi[e] + si;
            reg_row<LPI>(P, wi, get_wd(P.i_rng, item[e], P.wd_item), true, L);
            nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
            if (act) {   // what the reference would have changed on the item side
                sub4(wi, q[e]);
                const long slot = e == 0 ? S.slot[s] : S.slot1[s];
                if (NI == 1 && S.hot_sub > 0 && S.iptr[item[0] + 1] - S.iptr[item[0]] > S.hot_sub) {
                    // a hot item of this window: what the change is computed FROM goes to the slot; k_window_apply forms it against the row of its sub-step
                    store_contrib<LPI>(S.contrib, 0, (size_t)slot, pitch, L, k, tu);
                    if (L == 0) { S.cbias[slot] = use_ubias ? bu : 0.0f; S.clabel[slot] = label; }
                } else {
                    store_contrib<LPI>(S.contrib, S.contrib_bf16, (size_t)slot, pitch, L, k, wi);
                    if (L == 0) S.cbias[slot] = nbi - bi[e];
                }
            }
        }
        if (act) {
            p = wu;
            if (use_ubias) bu = nbu;
        }
    }
    if (valid) {
        store_row<LPI>(P.W, ur, pitch, L, k, p);
        if (use_ubias && L == 0) P.bias[ur] = bu;
    }
}

// ------------------------------------------------------------------------------------------------- kernel A, the contract configuration
// k = 4 * LANES * V full rows, unit values, linear link, L2 decay without ranges, user bias on (k_basicmf_slots' configuration): LANES
// lanes per user with V chunks each, the 16 / LANES users of a DPP row interleaved (dot_slots), G user sets per wave, and the NEXT
// instance's item row, bias and record in flight while this one is computed (the item side is read-only inside a window, so the
// prefetch cannot go stale).
// NI = 2: rank pairs (two signed item entries, labels 1); LINK = 0 linear / 3 sigmoid rank loss; UB = user bias on.
template <int LANES, int V, int G, int NI, int LINK, bool UB>
__global__ __launch_bounds__(256) void k_window_users_slots(const DevParams P, const WindowSchedule S) {
    constexpr int T = 16 / LANES;
    constexpr int IPS = 64 / LANES;    // users per user set
    constexpr int K = 4 * LANES * V;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = wave * (long)(G * IPS);
    if (w0 >= S.nusers) return;
    const int m = (lane & 15) / T;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    const int pitch = P.pitch;
    const int maxc = S.urec[w0].count;
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;

    bool valid[G];
    int begin[G], count[G];
    unsigned ur[G];
    float bu[G];
    float4 p[G][V];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long u = w0 + (long)g * IPS + gslot;
        valid[g] = u < S.nusers;
        const WinUser rec = S.urec[valid[g] ? u : w0];
        begin[g] = rec.begin; count[g] = valid[g] ? rec.count : 0;
        ur[g] = P.user_off + rec.user;
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int v = 0; v < V; v++) p[g][v] = load_row_nt<K / 4>(P.W, ur[g], pitch, m + v * LANES, K);
        bu[g] = UB ? P.bias[ur[g]] : 0.0f;
    }
    // software pipeline: record / item rows / item biases of iteration j + 1 are requested before iteration j is computed
    unsigned nir[G][NI];
    float nlabel[G], nbi_[G][NI], nia[G][NI];
    int nslot[G][NI];
    bool nhot[G];
    float4 nq[G][NI][V];
#pragma unroll
    for (int g = 0; g < G; g++) nhot[g] = false;
    auto fetch = [&](int j) {
#pragma unroll
        for (int g = 0; g < G; g++) {
            const long s = (long)begin[g] + (j < count[g] ? j : 0);
            nir[g][0] = P.item_off + S.item[s];
            nslot[g][0] = S.slot[s];
            nia[g][0] = NI == 2 ? S.ival[s] : 1.0f;
            if (NI == 2) { nir[g][NI - 1] = P.item_off + S.item1[s]; nslot[g][NI - 1] = S.slot1[s]; nia[g][NI - 1] = S.ival1[s]; }
            nlabel[g] = NI == 2 ? 1.0f : S.label[s];
            if (NI == 1) nhot[g] = S.hot_sub > 0 && S.iptr[S.item[s] + 1] - S.iptr[S.item[s]] > S.hot_sub;
        }
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int e = 0; e < NI; e++) {
#pragma unroll
                for (int v = 0; v < V; v++) nq[g][e][v] = load_row<K / 4>(P.W, nir[g][e], pitch, m + v * LANES, K);
                nbi_[g][e] = P.bias[nir[g][e]];
            }
        }
    };
    fetch(0);
    for (int j = 0; j < maxc; j++) {
        float label[G], bi[G][NI], ia[G][NI];
        int slot[G][NI];
        bool hot[G];
        float4 q[G][NI][V];
#pragma unroll
        for (int g = 0; g < G; g++) {
            label[g] = nlabel[g]; hot[g] = nhot[g];
#pragma unroll
            for (int e = 0; e < NI; e++) {
                bi[g][e] = nbi_[g][e]; slot[g][e] = nslot[g][e]; ia[g][e] = nia[g][e];
#pragma unroll
                for (int v = 0; v < V; v++) q[g][e][v] = nq[g][e][v];
            }
        }
        if (j + 1 < maxc) fetch(j + 1);
#pragma unroll
        for (int g = 0; g < G; g++) {
            const bool act = j < count[g];
            // the arithmetic of the lane-group kernel above, the user's row and bias carried in registers
            double bs = 0.0;
            if (UB) { bs += (double)(1.0f * bu[g]); bs += 0.0; }
            bs += 0.0;
#pragma unroll
            for (int e = 0; e < NI; e++) bs += (double)(ia[g][e] * bi[g][e]);
            double sum = (double)P.base_score + bs;
            float4 tu[V], ti[V];
#pragma unroll
            for (int v = 0; v < V; v++) {
                tu[v] = f4zero(); ti[v] = f4zero();
                axpy4(tu[v], p[g][v], 1.0f);
#pragma unroll
                for (int e = 0; e < NI; e++) axpy4(ti[v], q[g][e][v], ia[g][e]);
            }
            sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
            const float pred = (float)sum;
            const float err = (LINK == 0 ? label[g] - pred : label[g] - 1.0f / (1.0f + glibc_expf(-pred))) * 1.0f;
            const float su = P.lr * err * 1.0f;
            float nbu = bu[g] + su;
            nbu = nbu * dec_ub;
#pragma unroll
            for (int v = 0; v < V; v++) {
                float4 wu = p[g][v];
                axpy4(wu, ti[v], su);
                wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
                if (act) p[g][v] = wu;
            }
            const float bu_before = bu[g];   // (the hot-item slot below takes the bias the row's pred was formed with)
            if (act && UB) bu[g] = nbu;
#pragma unroll
            for (int e = 0; e < NI; e++) {
                const float si = P.lr * err * ia[g][e];
                float nbi = bi[g][e] + si;
                nbi = nbi * dec_ib;
                float4 c[V];
#pragma unroll
                for (int v = 0; v < V; v++) {
                    float4 wi = q[g][e][v];
                    axpy4(wi, tu[v], si);
                    wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
                    sub4(wi, q[g][e][v]);
                    c[v] = wi;
                }
                if (act && NI == 1 && hot[g]) {   // a hot item of this window (WindowSchedule::hot_sub): the slot takes what the change is computed FROM
#pragma unroll
                    for (int v = 0; v < V; v++) store_contrib<K / 4>(S.contrib, 0, (size_t)slot[g][e], pitch, m + v * LANES, K, tu[v]);
                    S.cbias[slot[g][e]] = UB ? bu_before : 0.0f;
                    S.clabel[slot[g][e]] = label[g];
                } else if (act) {
#pragma unroll
                    for (int v = 0; v < V; v++) store_contrib<K / 4>(S.contrib, S.contrib_bf16, (size_t)slot[g][e], pitch, m + v * LANES, K, c[v]);
                    S.cbias[slot[g][e]] = nbi - bi[g][e];
                }
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        if (valid[g]) {
#pragma unroll
            for (int v = 0; v < V; v++) store_row<K / 4>(P.W, ur[g], pitch, m + v * LANES, K, p[g][v]);
            if (UB) P.bias[ur[g]] = bu[g];
        }
    }
}

// the configurations the slot kernel is instantiated for: ratings (one unit item entry, linear link, user bias) at k = 64 / 128, rank
// pairs (two signed item entries, sigmoid rank loss, no user bias: demo/pairwiseRank, BASELINE configs[4]) at k = 64 / 128
bool window_slots_applies(const DevParams &P, const WindowSchedule &S) {
    if (S.uval != nullptr || P.reg_method != 0 || P.user_nonnegative != 0 || P.u_rng.n != 0 || P.i_rng.n != 0 || !(P.k == 64 || P.k == 128)) return false;
    if (S.item1 == nullptr) return S.ival == nullptr && P.active_type == ACT_LINEAR && P.no_user_bias == 0;
    return P.active_type == ACT_SIGMOID_RANK && P.no_user_bias != 0;
}

void launch_window_users(const DevParams &P, const WindowSchedule &S, int slots, int groups_per_wave, hipStream_t st) {
    if (S.nusers <= 0) return;
    if (slots && window_slots_applies(P, S)) {
        auto go = [&](auto lanes, auto gg, auto ni) {
            constexpr int LANES = decltype(lanes)::value, G = decltype(gg)::value, NI = decltype(ni)::value;
            const long per_wave = (long)G * (64 / LANES);
            const long waves = (S.nusers + per_wave - 1) / per_wave;
            if (NI == 1) hipLaunchKernelGGL((k_window_users_slots<LANES, 2, G, 1, 0, true>), dim3((unsigned)waves), dim3(64), 0, st, P, S);
            else hipLaunchKernelGGL((k_window_users_slots<LANES, 2, G, 2, 3, false>), dim3((unsigned)waves), dim3(64), 0, st, P, S);
        };
        const int g = groups_per_wave > 0 ? groups_per_wave : 1;
        using I1 = std::integral_constant<int, 1>;
        using I2 = std::integral_constant<int, 2>;
        if (S.item1 != nullptr) {
            if (P.k == 64) go(std::integral_constant<int, 8>(), I1(), I2());
            else go(std::integral_constant<int, 16>(), I1(), I2());
        } else if (P.k == 64) {
            if (g >= 2) go(std::integral_constant<int, 8>(), I2(), I1());
            else go(std::integral_constant<int, 8>(), I1(), I1());
        } else {
            if (g >= 2) go(std::integral_constant<int, 16>(), I2(), I1());
            else go(std::integral_constant<int, 16>(), I1(), I1());
        }
        return;
    }
    const int lpi = lanes_per_instance(P.k);
    const long ipw = 64 / lpi;
    const long waves = (S.nusers + ipw - 1) / ipw;
    if (S.item1 != nullptr) {
        if (S.uval == nullptr) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_users<LPI, true, 2>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
        else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_users<LPI, false, 2>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
    } else if (S.uval == nullptr) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_users<LPI, true, 1>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_users<LPI, false, 1>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
}

// position j of the packed layout -> the float it stands for
__device__ __forceinline__ float *addto_slot(const DeltaRanges &R, long j) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < SVDF_MAX_DELTA_RANGES; q++) r += (q < R.n && j >= R.off[q]) ? 1 : 0;
    return R.base[r] + (j - R.off[r]);
}

// ------------------------------------------------------------------------------------------------- kernel B
// Item i's contributions sit in slots [iptr[i], iptr[i + 1]), in file order.  One lane group per item sums them in that order
// (acc = 0 + c_1 + c_2 ..., four rows requested ahead) and writes the item's row and bias of the wire buffer:
// dst = [ (hi - lo) rows of `pitch` | (hi - lo) item biases | nglobal zeros ] for the item range [lo, hi) of the active
// exchange partition (svdf_item_delta_select) -- the packed layout of k_delta_pack.
// LOCAL (stratified schedule): the rank owns the item block exclusively, so there is no sum over ranks and the
// per-item sum is added to the model in place: dst = W_item's first row (row i at dst + i * pitch), dbias = i_bias; fp32, no wire buffer.
// LONG lists (round 5).  The window rule lets a row meet up to 128 updates per window, and on skewed data (Zipf-popular items) the hottest rows do: one
// lane group adding 128 slots eight at a time is sixteen dependent round trips -- 42 us per window of 13 K ratings, when the users' walk takes 6.  A lane
// group that meets a list longer than HOT_MIN therefore queues it (LDS), and after the scan the WHOLE workgroup loads such a list's slots side by side into
// LDS (one round trip per 16 - 64 slots) and its first lane group adds them in slot order: the same additions in the same order, hence the same bits.
#define SVDF_WIN_HOT_MIN 16
#define SVDF_WIN_HOT_QUEUE 32
template <int LPI, bool HALF, bool LOCAL, bool HOT>   // HOT: the window may hold lists that are long relative to its mean (the launcher decides: dense windows keep the plain form)
__global__ __launch_bounds__(256) void k_window_items(const WindowSchedule S, int pitch, int k, long lo, long hi, long nglobal, void *dst, float *dbias, int hot_min) {
    constexpr int IPW = 64 / LPI, G = 256 / LPI;
    constexpr int CHUNK = LPI >= 64 ? 16 : (LPI >= 32 ? 32 : 64);   // slots staged at a time: at most 16 KB of LDS whatever the width
    __shared__ int hq_it[HOT ? SVDF_WIN_HOT_QUEUE : 1], hq_b[HOT ? SVDF_WIN_HOT_QUEUE : 1], hq_e[HOT ? SVDF_WIN_HOT_QUEUE : 1];
    __shared__ int hq_n;
    __shared__ float4 stage[HOT ? CHUNK * LPI : 1];
    __shared__ float stage_b[HOT ? CHUNK : 1];
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int grp = threadIdx.x / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    const long first = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long nitem = hi - lo;
    const bool owns = !(LPI * 4 > k && L * 4 >= k);
    if constexpr (HOT) {
        if (threadIdx.x == 0) hq_n = 0;
        __syncthreads();
    }
    // what becomes of a finished sum: added to the model in place (LOCAL) or written to the wire buffer
    auto finish = [&](long it, int b, int e, const float4 &acc, float accb) {
        const long i = lo + it;
        if (LOCAL) {
            if (b == e) return;   // nobody rated the item in this window
            if (owns) {
                float4 *w = reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + (size_t)i * pitch + (size_t)L * 4);
                float4 c = *w;
                c.x = c.x + acc.x; c.y = c.y + acc.y; c.z = c.z + acc.z; c.w = c.w + acc.w;
                *w = c;
            }
            if (L == 0) dbias[i] = dbias[i] + accb;
            return;
        }
        if (owns) {
            const size_t pos = (size_t)it * pitch + (size_t)L * 4;
            if (HALF) {
                __half2 *h = reinterpret_cast<__half2 *>(reinterpret_cast<__half *>(dst) + pos);
                h[0] = __halves2half2(__float2half_rn(acc.x), __float2half_rn(acc.y));
                h[1] = __halves2half2(__float2half_rn(acc.z), __float2half_rn(acc.w));
            } else {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + pos) = acc;
            }
        }
        if (L == 0) {
            const size_t pos = (size_t)nitem * pitch + (size_t)it;
            if (HALF) reinterpret_cast<__half *>(dst)[pos] = __float2half_rn(accb);
            else reinterpret_cast<float *>(dst)[pos] = accb;
        }
    };
    for (long it = first; it < nitem; it += stride) {
        const long i = lo + it;
        const int b = S.iptr[i], e = S.iptr[i + 1];
        if (HOT && e - b > hot_min) {   // a long list: left to the whole workgroup (below) while the queue has room
            int pos = SVDF_WIN_HOT_QUEUE;
            if (L == 0) pos = atomicAdd(&hq_n, 1);
            pos = __shfl(pos, (lane / LPI) * LPI);
            if (pos < SVDF_WIN_HOT_QUEUE) {
                if (L == 0) { hq_it[pos] = (int)it; hq_b[pos] = b; hq_e[pos] = e; }
                continue;
            }
        }
        float4 acc = f4zero();
        float accb = 0.0f;
        if (S.contrib_bf16) sum_contrib_slots<LPI, true>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
        else sum_contrib_slots<LPI, false>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
        finish(it, b, e, acc, accb);
    }
    if constexpr (HOT) {
    __syncthreads();
    const int nq = min(hq_n, SVDF_WIN_HOT_QUEUE);
    for (int qi = 0; qi < nq; qi++) {
        const int b = hq_b[qi], e = hq_e[qi];
        float4 acc = f4zero();
        float accb = 0.0f;
        for (int c0 = b; c0 < e; c0 += CHUNK) {
            const int cn = min(CHUNK, e - c0);
            // every lane group requests its share of the chunk's slots at once (CHUNK / G per group), then parks them in LDS
            constexpr int PER = (CHUNK + G - 1) / G;
            float4 v[PER];
            float vb[PER];
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int sl = grp + r * G;
                const bool in = sl < cn;
                v[r] = in ? load_contrib<LPI>(S.contrib, S.contrib_bf16, (size_t)(c0 + sl), pitch, L, k) : f4zero();
                vb[r] = (in && L == 0) ? S.cbias[c0 + sl] : 0.0f;
            }
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int sl = grp + r * G;
                if (sl < cn) { stage[sl * LPI + L] = v[r]; if (L == 0) stage_b[sl] = vb[r]; }
            }
            __syncthreads();
            if (grp == 0) {   // slot order, acc = ((0 + c_1) + c_2) + ... as sum_contrib_slots does; eight LDS reads requested ahead of their (ordered) additions
                for (int sl = 0; sl < cn; sl += 8) {
                    float4 t[8];
                    float tb[8];
#pragma unroll
                    for (int q = 0; q < 8; q++) { const int x = min(sl + q, cn - 1); t[q] = stage[x * LPI + L]; tb[q] = stage_b[x]; }
#pragma unroll
                    for (int q = 0; q < 8; q++) if (sl + q < cn) { add_rows(acc, t[q]); accb = accb + tb[q]; }
                }
            }
            __syncthreads();
        }
        if (grp == 0) finish((long)hq_it[qi], b, e, acc, accb);
    }
    }
    if (LOCAL) return;
    // the global biases' part of the wire buffer: a window data set carries no global entry
    const long g0 = nitem * (long)(pitch + 1);
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nglobal; j += (long)gridDim.x * blockDim.x) {
        if (HALF) reinterpret_cast<__half *>(dst)[g0 + j] = __float2half_rn(0.0f);
        else reinterpret_cast<float *>(dst)[g0 + j] = 0.0f;
    }
}
// SPARSE windows (round 5): the windows the max-updates rule cuts on skewed data hold far fewer instances than there are items (13 K ratings over
// 100 K items), and a lane group per ITEM is then 25 K waves that find nothing.  Here a THREAD looks at one item, the workgroup's items that have slots
// go to an LDS queue, the lane groups share the queue (long lists: the cooperative form above).  In place only; same additions in the same order.
template <int LPI>
__global__ __launch_bounds__(256) void k_window_items_sparse(const WindowSchedule S, int pitch, int k, long lo, long hi, float *w_item, float *dbias, int hot_min) {
    constexpr int G = 256 / LPI;
    constexpr int CHUNK = LPI >= 64 ? 16 : (LPI >= 32 ? 32 : 64);
    __shared__ int q_it[256], q_b[256], q_e[256];
    __shared__ int hq_idx[SVDF_WIN_HOT_QUEUE];
    __shared__ int q_n, hq_n;
    __shared__ float4 stage[CHUNK * LPI];
    __shared__ float stage_b[CHUNK];
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int grp = threadIdx.x / LPI;
    const long nitem = hi - lo;
    const bool owns = !(LPI * 4 > k && L * 4 >= k);
    auto finish = [&](long it, const float4 &acc, float accb) {
        const long i = lo + it;
        if (owns) {
            float4 *w = reinterpret_cast<float4 *>(w_item + (size_t)i * pitch + (size_t)L * 4);
            float4 c = *w;
            c.x = c.x + acc.x; c.y = c.y + acc.y; c.z = c.z + acc.z; c.w = c.w + acc.w;
            *w = c;
        }
        if (L == 0) dbias[i] = dbias[i] + accb;
    };
    for (long base = (long)blockIdx.x * 256; base < nitem; base += (long)gridDim.x * 256) {
        if (threadIdx.x == 0) { q_n = 0; hq_n = 0; }
        __syncthreads();
        const long it = base + threadIdx.x;
        if (it < nitem) {
            const int b = S.iptr[lo + it], e = S.iptr[lo + it + 1];
            if (e > b) { const int pos = atomicAdd(&q_n, 1); q_it[pos] = (int)it; q_b[pos] = b; q_e[pos] = e; }
        }
        __syncthreads();
        const int n = q_n;
        for (int idx = grp; idx < n; idx += G) {
            const int b = q_b[idx], e = q_e[idx];
            if (e - b > hot_min) {
                int pos = SVDF_WIN_HOT_QUEUE;
                if (L == 0) pos = atomicAdd(&hq_n, 1);
                pos = __shfl(pos, (lane / LPI) * LPI);
                if (pos < SVDF_WIN_HOT_QUEUE) { if (L == 0) hq_idx[pos] = idx; continue; }
            }
            float4 acc = f4zero();
            float accb = 0.0f;
            if (S.contrib_bf16) sum_contrib_slots<LPI, true, 4>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
            else sum_contrib_slots<LPI, false, 4>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
            finish((long)q_it[idx], acc, accb);
        }
        __syncthreads();
        const int nq = min(hq_n, SVDF_WIN_HOT_QUEUE);
        for (int qi = 0; qi < nq; qi++) {
            const int idx = hq_idx[qi];
            const int b = q_b[idx], e = q_e[idx];
            float4 acc = f4zero();
            float accb = 0.0f;
            for (int c0 = b; c0 < e; c0 += CHUNK) {
                const int cn = min(CHUNK, e - c0);
                constexpr int PER = (CHUNK + G - 1) / G;
                float4 v[PER];
                float vb[PER];
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    const int sl = grp + r * G;
                    const bool in = sl < cn;
                    v[r] = in ? load_contrib<LPI>(S.contrib, S.contrib_bf16, (size_t)(c0 + sl), pitch, L, k) : f4zero();
                    vb[r] = (in && L == 0) ? S.cbias[c0 + sl] : 0.0f;
                }
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    const int sl = grp + r * G;
                    if (sl < cn) { stage[sl * LPI + L] = v[r]; if (L == 0) stage_b[sl] = vb[r]; }
                }
                __syncthreads();
                if (grp == 0) {
                    for (int sl = 0; sl < cn; sl += 8) {
                        float4 t[8];
                        float tb[8];
#pragma unroll
                        for (int q = 0; q < 8; q++) { const int x = min(sl + q, cn - 1); t[q] = stage[x * LPI + L]; tb[q] = stage_b[x]; }
#pragma unroll
                        for (int q = 0; q < 8; q++) if (sl + q < cn) { add_rows(acc, t[q]); accb = accb + tb[q]; }
                    }
                }
                __syncthreads();
            }
            if (grp == 0) finish((long)q_it[idx], acc, accb);
        }
        __syncthreads();
    }
}
// A list is LONG relative to its window: the cooperative form serialises a workgroup's long lists, which pays when they are the exception (the hot rows of a
// skewed window) and costs when every list is long (a dense window of uniform data: 24 slots per item at configs[1], 300 at configs[4] -- there the lane
// groups' own loops, all busy at once, are the parallel form).  Long = more than 16 slots AND more than four times the window's mean list.
static int window_hot_min(long nslots, long nitems) {
    if (nslots < 0 || nitems <= 0) return 0x7FFFFFFF;   // unknown: never
    const long mean4 = 4 * ((nslots + nitems - 1) / nitems);
    return (int)std::min<long>(std::max<long>(SVDF_WIN_HOT_MIN, mean4), 0x7FFFFFFF);
}
// dense windows (8 or more slots per item on average: uniform data) run the plain form of the kernel: no queue, no LDS, no barrier
static bool window_may_hold_long_lists(long nslots, long nitems) { return nslots >= 0 && nitems > 0 && nslots < 8 * nitems; }
void launch_window_items(const WindowSchedule &S, int pitch, int k, long lo, long hi, long nglobal, void *dst, int half, hipStream_t st, long nslots) {
    if (hi <= lo && nglobal <= 0) return;
    const int lpi = lanes_per_instance(k);
    const long ipw = 64 / lpi;
    long waves = (std::max<long>(hi - lo, 1) + ipw - 1) / ipw;
    long grid = (waves + 3) / 4;
    if (grid > 16384) grid = 16384;
    const int hot_min = window_hot_min(nslots, hi - lo);
    const bool hot = window_may_hold_long_lists(nslots, hi - lo);
    if (half && hot) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, true, false, true>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, nglobal, dst, (float *)nullptr, hot_min)); }
    else if (half) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, true, false, false>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, nglobal, dst, (float *)nullptr, hot_min)); }
    else if (hot) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, false, false, true>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, nglobal, dst, (float *)nullptr, hot_min)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, false, false, false>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, nglobal, dst, (float *)nullptr, hot_min)); }
}
// the same sums added in place to the rows / biases of items [lo, hi): W_item + i * pitch, i_bias + i
void launch_window_items_local(const WindowSchedule &S, int pitch, int k, long lo, long hi, float *w_item, float *i_bias, hipStream_t st, long nslots) {
    if (hi <= lo) return;
    const int lpi = lanes_per_instance(k);
    if (nslots >= 0 && nslots * 2 < hi - lo) {   // far fewer contributions than items: most items have none (k_window_items_sparse)
        const long grid = std::min<long>((hi - lo + 255) / 256, 16384);
        SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items_sparse<LPI>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, w_item, i_bias, window_hot_min(nslots, hi - lo)));
        return;
    }
    const long ipw = 64 / lpi;
    long waves = (hi - lo + ipw - 1) / ipw;
    long grid = (waves + 3) / 4;
    if (grid > 16384) grid = 16384;
    if (window_may_hold_long_lists(nslots, hi - lo)) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, false, true, true>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, 0L, (void *)w_item, i_bias, window_hot_min(nslots, hi - lo))); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_items<LPI, false, true, false>), dim3((unsigned)grid), dim3(256), 0, st, S, pitch, k, lo, hi, 0L, (void *)w_item, i_bias, 0x7FFFFFFF)); }
}
// ------------------------------------------------------------------------------------------------- kernel B', windows with hot items (round 6)
// ORDERED SUB-STEPS.  The window rule of round 5 let no row meet more than 128 updates per window, because thousands of changes computed against
// ONE stale value of a row overshoot (NaN on Zipf-popular items) -- so the single hottest item set the number of windows (7 605 at the configs[1]
// size for a top item of 0.97 %), each a few launches over 13 K ratings.  Here a window is cut by what the COLD rows tolerate, and an item with more
// than hot_sub slots in it is applied by ONE workgroup (of 1 024 threads) in file order, hot_sub slots at a time: the slots hold what the users' walk computed the
// change FROM (tmp_u, the user's bias, the label); a sub-step forms every slot's change against the row as the previous sub-step left it --
// update_inner's item side (apex_svd_base.h:456-462 with :383-427): pred from (bias sum in double, dot in the reference's order), err, the axpy,
// the decay -- parks the changes in LDS, its first lane group adds them in slot order (acc = 0 + c_1 + c_2 ...) and the row moves by the sum.
// Equals oracle/svdf_oracle.c: svdo_update_window_substeps bit for bit (tests/test_gpu_window_hot.py).  The workgroup's lane groups hold a copy of
// the row each; the next round's slots are requested before the current round's barrier.
// NT = 1024 threads: a sub-step is 128 dots in the reference's summation order (15 dependent DPP additions per chain: ~1 us of issue per 8 slots of a
// wave) -- sixteen waves share them two slots per lane group (four waves, eight slots each: 8.6 us per sub-step, 138 us per window of 2 048 slots).
// PLAIN: linear link, L2 decay (reg_method 0): the switch over links and regularisers is compiled out of the dependent chain.
// ONE launch applies the whole window (k_window_apply): the first `hot_blocks` workgroups are the hot lane -- workgroup g looks at items g, g + hot_blocks,
// ... (neighbouring ids, which real catalogues often sort by popularity, go to different workgroups), queues the hot ones in LDS and walks them --, the
// other workgroups add the cold items' slots in place, a lane group per item in slot order (sum_contrib_slots: the additions of k_window_items).  Hot and
// cold items share no row, so the hot items' dependent chains run beside the streaming sums instead of after them.
template <int LPI, int NT, bool PLAIN>
__global__ __launch_bounds__(NT) void k_window_apply(const DevParams P, const WindowSchedule S, long num_item, float *w_item, float *i_bias, int hot_blocks) {
    constexpr int G = NT / LPI;
    constexpr int K4 = 4 * LPI;                                       // floats of a (padded) row
    constexpr int CHUNK0 = 2048 / LPI > 128 ? 128 : 2048 / LPI;       // slots per round: at most 32 KB of LDS whatever the width ...
    constexpr int CHUNK = CHUNK0 < G ? G : CHUNK0;                    // ... and at least one per lane group
    constexpr int PER = CHUNK / G;
    constexpr int EPL = (K4 + 63) / 64;                               // row elements per lane of the summing wave
    __shared__ float4 stage[CHUNK * LPI];
    __shared__ float stage_b[CHUNK];
    __shared__ float4 rowq[LPI];
    __shared__ float rowb;
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    const int L = lane & (LPI - 1);
    const int grp = threadIdx.x / LPI;
    const int pitch = P.pitch, k = P.k;
    const bool use_ubias = P.no_user_bias == 0;
    if ((int)blockIdx.x >= hot_blocks) {   // ---- cold items: the in-place sums of k_window_items<LPI, false, true, .> for every list of at most hot_sub slots
        const bool owns = !(LPI * 4 > k && L * 4 >= k);
        const long stride = (long)(gridDim.x - hot_blocks) * G;
        for (long i = (long)(blockIdx.x - hot_blocks) * G + grp; i < num_item; i += stride) {
            const int b = S.iptr[i], e = S.iptr[i + 1];
            if (b == e || e - b > S.hot_sub) continue;
            float4 acc = f4zero();
            float accb = 0.0f;
            sum_contrib_slots<LPI, false>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
            if (owns) {
                float4 *w = reinterpret_cast<float4 *>(w_item + (size_t)i * pitch + (size_t)L * 4);
                float4 c = *w;
                c.x = c.x + acc.x; c.y = c.y + acc.y; c.z = c.z + acc.z; c.w = c.w + acc.w;
                *w = c;
            }
            if (L == 0) i_bias[i] = i_bias[i] + accb;
        }
        return;
    }
    __shared__ int hl_item[NT], hl_b[NT], hl_e[NT];
    __shared__ int hl_n;
    const float *stage_f = reinterpret_cast<const float *>(stage);
    float *rowf = reinterpret_cast<float *>(rowq);
    for (long base = blockIdx.x; base < num_item; base += (long)hot_blocks * NT) {
    if (threadIdx.x == 0) hl_n = 0;
    __syncthreads();
    {
        const long i = base + (long)threadIdx.x * hot_blocks;
        if (i < num_item) {
            const int b = S.iptr[i], e = S.iptr[i + 1];
            if (e - b > S.hot_sub) { const int pos = atomicAdd(&hl_n, 1); hl_item[pos] = (int)i; hl_b[pos] = b; hl_e[pos] = e; }
        }
    }
    __syncthreads();
    const int nhot = hl_n;
    for (int h = 0; h < nhot; h++) {
        const unsigned item = (unsigned)hl_item[h];
        const int b = hl_b[h], e = hl_e[h];
        const unsigned ir = P.item_off + item;
        const float wd_i = get_wd(P.i_rng, item, P.wd_item);
        float4 q = load_row<LPI>(P.W, ir, pitch, L, k);
        float bi = P.bias[ir];
        if (grp == 0) { rowq[L] = q; if (L == 0) rowb = bi; }
        {   // the item's slots were written by other CUs a kernel ago (they sit in HBM / the Infinity Cache): one request per 128-byte line, all in flight at
            // once, brings them into this XCD's L2 -- the sub-steps below are a dependent chain and would otherwise pay that latency once per sub-step
            float warm = 0.0f;
            const size_t w0 = (size_t)b * (size_t)pitch, w1 = (size_t)e * (size_t)pitch;
            for (size_t off = w0 + (size_t)threadIdx.x * 32; off < w1; off += (size_t)NT * 32) warm += S.contrib[off];
            for (int sl = b + (int)threadIdx.x * 32; sl < e; sl += NT * 32) warm += S.cbias[sl] + S.clabel[sl];
            if (warm == 1.2345e-38f) rowb = warm;   // (never true for data that matters: keeps the loads alive)
        }
        // the slots of a round do not depend on the row: the NEXT round's are requested before this round's changes are formed
        float4 ntu[PER];
        float nbu[PER], nlabel[PER];
        auto fetch = [&](int first, int cn) {
#pragma unroll
            for (int r = 0; r < PER; r++) {
                const int sl = grp + r * G;
                const size_t slot = (size_t)(first + (sl < cn ? sl : 0));
                ntu[r] = load_contrib<LPI>(S.contrib, 0, slot, pitch, L, k);
                nbu[r] = S.cbias[slot];
                nlabel[r] = S.clabel[slot];
            }
        };
        fetch(b, min(min(CHUNK, S.hot_sub), e - b));
        for (int s0 = b; s0 < e; s0 += S.hot_sub) {
            const int sn = min(S.hot_sub, e - s0);
            float acc[EPL];
#pragma unroll
            for (int x = 0; x < EPL; x++) acc[x] = 0.0f;
            float accb = 0.0f;
            for (int c0 = 0; c0 < sn; c0 += CHUNK) {
                const int cn = min(CHUNK, sn - c0);
                float4 tu[PER];
                float bu[PER], label[PER];
#pragma unroll
                for (int r = 0; r < PER; r++) { tu[r] = ntu[r]; bu[r] = nbu[r]; label[r] = nlabel[r]; }
                {   // the round after this one: the rest of the sub-step, else the head of the next sub-step
                    int nf = s0 + c0 + CHUNK, nn = sn - c0 - CHUNK;
                    if (nn <= 0) { nf = s0 + S.hot_sub; nn = min(S.hot_sub, e - nf); }
                    if (nn > 0) fetch(nf, min(CHUNK, nn));
                }
#pragma unroll
                for (int r = 0; r < PER; r++) {
                    const int sl = grp + r * G;
                    // calc_bias (:313-353) in double; "+ 0.0" terms are the svdpp / plugin hooks returning 0.0f -- the sums of k_window_users
                    double bs = 0.0;
                    if (use_ubias) { bs += (double)(1.0f * bu[r]); bs += 0.0; }
                    bs += 0.0;
                    bs += (double)(1.0f * bi);
                    double sum = (double)P.base_score + bs;
                    float4 ti = f4zero();
                    axpy4(ti, q, 1.0f);
                    sum += (double)group_dot<LPI>(tu[r], ti, L, k);
                    const float pred = PLAIN ? (float)sum : map_active((float)sum, P.active_type);
                    const float err = (PLAIN ? label[r] - pred : cal_grad(label[r], pred, P.active_type)) * 1.0f;
                    const float si = P.lr * err * 1.0f;
                    float4 wi = q;
                    axpy4(wi, tu[r], si);
                    float nbi = bi + si;
                    if (PLAIN) scale4(wi, 1.0f - P.lr * wd_i);   // (reg_row's method 0)
                    else reg_row<LPI>(P, wi, wd_i, true, L);
                    nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
                    sub4(wi, q);
                    if (sl < cn) { stage[sl * LPI + L] = wi; if (L == 0) stage_b[sl] = nbi - bi; }
                }
                __syncthreads();
                // slot order (acc = ((0 + c_1) + c_2) + ..., the sums of k_window_items), one ROW ELEMENT per lane of the first wave: 128 dependent
                // additions of one float instead of 128 x 4 in a lane group's float4; the bias word by one lane of the second wave beside it
                if (wv == 0) {
                    // whole batches first: AH reads at constant offsets, AH additions, no per-slot bounds test (the clamped form costs ~45 cycles of scalar
                    // bookkeeping per slot: 2.5 us per sub-step of 128), then the tail one slot at a time
                    constexpr int AH = EPL == 1 ? 32 : (EPL == 2 ? 16 : 8);
                    bool on[EPL];
#pragma unroll
                    for (int z = 0; z < EPL; z++) on[z] = lane + 64 * z < K4;
                    int sl = 0;
                    for (; sl + AH <= cn; sl += AH) {
                        const float *src = stage_f + (size_t)sl * K4 + lane;
                        float t[AH][EPL];
#pragma unroll
                        for (int x = 0; x < AH; x++) {
#pragma unroll
                            for (int z = 0; z < EPL; z++) t[x][z] = on[z] ? src[x * K4 + 64 * z] : 0.0f;
                        }
#pragma unroll
                        for (int x = 0; x < AH; x++) {
#pragma unroll
                            for (int z = 0; z < EPL; z++) acc[z] = acc[z] + t[x][z];
                        }
                    }
                    for (; sl < cn; sl++) {
#pragma unroll
                        for (int z = 0; z < EPL; z++) acc[z] = acc[z] + (on[z] ? stage_f[(size_t)sl * K4 + lane + 64 * z] : 0.0f);
                    }
                } else if (wv == 1) {   // (every lane of the second wave, redundantly: 32 broadcast reads in flight, then their ordered additions -- a
                                        // one-lane loop pays an LDS round trip per slot: 7 us per sub-step of 128)
                    int sl = 0;
                    for (; sl + 32 <= cn; sl += 32) {
                        float t[32];
#pragma unroll
                        for (int x = 0; x < 32; x++) t[x] = stage_b[sl + x];
#pragma unroll
                        for (int x = 0; x < 32; x++) accb = accb + t[x];
                    }
                    for (; sl < cn; sl++) accb = accb + stage_b[sl];
                }
                if (c0 + CHUNK < sn) __syncthreads();   // (more rounds of this sub-step: the staging area is written again)
            }
            // the row moves by the sub-step's sum (c = c + acc, as the in-place sums do); every lane group takes the new row
            if (wv == 0) {
#pragma unroll
                for (int z = 0; z < EPL; z++) { const int el = lane + 64 * z; if (el < K4) rowf[el] = rowf[el] + acc[z]; }
            } else if (wv == 1 && lane == 0) {
                rowb = rowb + accb;
            }
            __syncthreads();
            q = rowq[L];
            bi = rowb;
        }
        __syncthreads();   // (the next item of this workgroup writes rowq / rowb)
        if (grp == 0) {
            store_row<LPI>(P.W, ir, pitch, L, k, q);
            if (L == 0) P.bias[ir] = bi;
        }
    }
    __syncthreads();   // (the next scan resets the queue)
    }
}
void launch_window_apply(const DevParams &P, const WindowSchedule &S, long num_item, float *w_item, float *i_bias, hipStream_t st) {
    if (S.hot_sub <= 0 || num_item <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int hot_blocks = 256;
    const long groups = 1024 / lpi;
    const long cold = std::min<long>(std::max<long>((num_item + groups - 1) / groups, 1), 4096);
    const unsigned grid = (unsigned)(hot_blocks + cold);
    if (P.active_type == ACT_LINEAR && P.reg_method == 0) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_apply<LPI, 1024, true>), dim3(grid), dim3(1024), 0, st, P, S, num_item, w_item, i_bias, hot_blocks)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_window_apply<LPI, 1024, false>), dim3(grid), dim3(1024), 0, st, P, S, num_item, w_item, i_bias, hot_blocks)); }
}
// the replicated ranges of the active partition as one packed fp32 buffer and back (the item block a rank hands to the next one)
template <bool SET>
__global__ __launch_bounds__(256) void k_ranges_copy(const DeltaRanges R, float *buf, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        float *cur = addto_slot(R, j);
        if (SET) *cur = buf[j];
        else buf[j] = *cur;
    }
}
// the user id of every instance of a window data set (its instances are grouped by user: urec = user, begin, count), for scoring it
__global__ __launch_bounds__(256) void k_window_user_column(const WinUser *urec, int nusers, unsigned *user_out) {
    const int lane = threadIdx.x & 63;
    const long w = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w >= nusers) return;
    const WinUser u = urec[w];
    for (int j = lane; j < u.count; j += 64) user_out[u.begin + j] = u.user;
}
void launch_window_user_column(const WinUser *urec, int nusers, unsigned *user_out, hipStream_t st) {
    if (nusers <= 0) return;
    const long blocks = ((long)nusers * 64 + 255) / 256;
    hipLaunchKernelGGL(k_window_user_column, dim3((unsigned)blocks), dim3(256), 0, st, urec, nusers, user_out);
}
void launch_ranges_copy(const DeltaRanges &R, float *buf, int set, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (set) hipLaunchKernelGGL(k_ranges_copy<true>, dim3((int)grid), dim3(256), 0, st, R, buf, total);
    else hipLaunchKernelGGL(k_ranges_copy<false>, dim3((int)grid), dim3(256), 0, st, R, buf, total);
}

// ------------------------------------------------------------------------------------------------- kernel C
// replicated ranges += the all-reduced window delta (packed layout of k_delta_pack)
// The first range (item rows: a multiple of four floats, 16-byte aligned in the model and in the wire buffer) goes four floats per
// lane; what follows (item biases, global biases) one float per lane.
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_addto(const DeltaRanges R, const void *src, long total, long vec4) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x;
    float4 *base4 = reinterpret_cast<float4 *>(R.base[0]);
    for (long j = tid; j < vec4; j += stride) {
        float4 d;
        if (HALF) {
            const uint2 raw = reinterpret_cast<const uint2 *>(src)[j];
            const __half2 lo = *reinterpret_cast<const __half2 *>(&raw.x), hi = *reinterpret_cast<const __half2 *>(&raw.y);
            d = make_float4(__low2float(lo), __high2float(lo), __low2float(hi), __high2float(hi));
        } else {
            d = reinterpret_cast<const float4 *>(src)[j];
        }
        float4 c = base4[j];
        c.x = c.x + d.x; c.y = c.y + d.y; c.z = c.z + d.z; c.w = c.w + d.w;
        base4[j] = c;
    }
    for (long j = 4 * vec4 + tid; j < total; j += stride) {
        const float d = HALF ? __half2float(reinterpret_cast<const __half *>(src)[j]) : reinterpret_cast<const float *>(src)[j];
        float *cur = addto_slot(R, j);
        *cur = *cur + d;
    }
}
void launch_delta_addto(const DeltaRanges &R, const void *src, int half, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    // vector path for range 0 when it is 16-byte aligned on both sides and a multiple of four floats
    const long n0 = R.n > 1 ? R.off[1] : total;
    const bool vec_ok = (n0 % 4 == 0) && ((reinterpret_cast<uintptr_t>(R.base[0]) & 15) == 0) && ((reinterpret_cast<uintptr_t>(src) & 15) == 0);
    const long vec4 = vec_ok ? n0 / 4 : 0;
    const long work = std::max(vec4, total - 4 * vec4);
    long grid = (work + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (grid < 1) grid = 1;
    if (half) hipLaunchKernelGGL(k_delta_addto<true>, dim3((int)grid), dim3(256), 0, st, R, src, total, vec4);
    else hipLaunchKernelGGL(k_delta_addto<false>, dim3((int)grid), dim3(256), 0, st, R, src, total, vec4);
}

}  // namespace svdf
