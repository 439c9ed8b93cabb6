// svdf_k_wunit.hip -- the WINDOW-MINIBATCH step for user units: user-group (SVD++) blocks and rows with global features
// (part of the gfx950 kernel set described at the top of svdf_device.h).
//
// Exact sequential semantics leave these shapes a handful of users per launch (BASELINE configs[3]: ~6 SVD++ users or ~280 neighbourhood
// rows per conflict-free level), because every instance reads and writes SHARED rows: W_item / i_bias, the implicit-feedback rows
// W_ufeedback / ufeedback_bias (apex_svd_base.h:523-554) and the global biases (:188-210, :313-353).  The window step defers exactly those:
//   * a user's unit runs exact on its PRIVATE state -- its W_user row and bias and, for SVD++, tmp_ufeedback / old_ufeedback
//     (SVDPPFeature's members, :486-488) -- instance after instance in file order, one lane group per user: k_wunit_walk;
//   * the shared rows are read as they were at the window start (nothing writes them inside a window: they are their own snapshot) and
//     what the reference WOULD have changed on them -- (q + s p) decay - q per item entry, reg(g + s) - g per global entry,
//     (w + d val) - w per feedback row at the unit's end -- goes to a contribution slot;
//   * k_wunit_sum adds every shared row's contributions IN FILE ORDER (slots are laid out target by target; no float atomics: the result
//     is deterministic and equals oracle/svdf_oracle.c: svdo_update_block_stale / svdo_update_csr_batch_stale bit for bit) and either adds
//     the sum to the model in place (one GPU, `amd:step = minibatch`) or writes the wire buffer of the N-rank exchange.
#include <algorithm>
#include <cstdlib>

#include "svdf_instance.h"

namespace svdf {

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------- kernel A: the users' exact walks
// FB: the trainer is SVDPPFeature (user-group format): a segment prepares tmp_ufeedback from its feedback list (:523-538), every row
// goes through the update_svdpp hook (:512-520), the segment's end is update_ufeedback (:539-554) against the window-start rows.
// General form: any width <= 256, any row shape (one user entry per row), every link and regulariser of the base solver.
template <int LPI, bool FB>
__global__ __launch_bounds__(256) void k_wunit_walk(const DevParams P, const WUnitSchedule S) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long uidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    if (uidx >= S.nunits) return;
    const WinUnit un = S.units[uidx];
    const int pitch = P.pitch, k = P.k;
    const bool ub = P.no_user_bias == 0;
    const float lr = P.lr;
    const unsigned ur = P.user_off + un.user;
    float4 p = load_row<LPI>(P.W, ur, pitch, L, k);
    float bu = ub ? P.bias[ur] : 0.0f;
    const float wd_u = get_wd(P.u_rng, un.user, P.wd_user);
    for (int sg = 0; sg < un.seg_count; sg++) {
        const WinSeg seg = sg == 0 ? un.first : S.segs[un.seg_begin + sg];
        SvdppRegs pp;
        pp.tmp_fb = f4zero(); pp.old_fb = f4zero(); pp.norm = 0.0f; pp.tmp_bias = 0.0f; pp.old_bias = 0.0f;
        if (FB) {   // prepare_ufeedback + the backup of update(block) (:568-574)
            for (int j = seg.fb_begin; j < seg.fb_begin + seg.fb_count; j++) {
                const WinEnt f = S.fbent[j];
                const unsigned row = P.fb_off + f.idx;
                axpy4(pp.tmp_fb, load_row<LPI>(P.W, row, pitch, L, k), f.val);
                pp.norm = pp.norm + f.val * f.val;
                if (ub) pp.tmp_bias = pp.tmp_bias + P.bias[row] * f.val;
            }
            pp.old_bias = pp.tmp_bias;
            pp.old_fb = pp.tmp_fb;
        }
        for (int r = seg.row_begin; r < seg.row_begin + seg.row_count; r++) {
            int e0, e1, e2;
            if (S.rptr) { e0 = S.rptr[2 * (long)r]; e1 = S.rptr[2 * (long)r + 1]; e2 = S.rptr[2 * (long)r + 2]; }
            else { e0 = r * S.estride; e1 = e0 + S.estride - 1; e2 = e1 + 1; }
            const float label = S.label[r];
            const float ua = S.uval ? S.uval[r] : 1.0f;
            // ---- pred (:445-454): calc_bias in double, the hooks' terms where the reference adds them
            double bs = 0.0;
            for (int j = e0; j < e1; j++) { const WinEnt e = S.ent[j]; bs += (double)(e.val * P.g_bias[e.idx]); }
            if (ub) {
                bs += (double)(ua * bu);
                bs += (double)(FB ? pp.tmp_bias : 0.0f);
            }
            bs += 0.0;
            for (int j = e1; j < e2; j++) { const WinEnt e = S.ent[j]; bs += (double)(e.val * P.bias[P.item_off + e.idx]); }
            double sum = (double)P.base_score + bs;
            float4 tu = FB ? pp.tmp_fb : f4zero();
            axpy4(tu, p, ua);
            float4 ti = f4zero();
            for (int j = e1; j < e2; j++) { const WinEnt e = S.ent[j]; axpy4(ti, load_row<LPI>(P.W, P.item_off + e.idx, pitch, L, k), e.val); }
            sum += (double)group_dot<LPI>(tu, ti, L, k);
            const float pred = map_active((float)sum, P.active_type);
            const float err = cal_grad(label, pred, P.active_type) * 1.0f;
            // ---- update_no_decay (:383-427) + regularize(after) (:286-311), the shared rows' part as contributions
            for (int j = e0; j < e1; j++) {
                const WinEnt e = S.ent[j];
                const float g = P.g_bias[e.idx];
                float g2 = g + lr * err * e.val;
                g2 = reg_gbias(P, e.idx, g2);
                if (L == 0) S.gcontrib[e.slot] = g2 - g;
            }
            const float su = lr * err * ua;
            float4 wu = p;
            axpy4(wu, ti, su);
            float nbu = bu + su;
            for (int j = e1; j < e2; j++) {
                const WinEnt e = S.ent[j];
                const float si = lr * err * e.val;
                const float4 q = load_row<LPI>(P.W, P.item_off + e.idx, pitch, L, k);
                const float bi = P.bias[P.item_off + e.idx];
                float4 wi = q;
                axpy4(wi, tu, si);
                float nbi = bi + si;
                reg_row<LPI>(P, wi, get_wd(P.i_rng, e.idx, P.wd_item), true, L);
                nbi = nbi * (1.0f - lr * P.wd_item_bias);
                sub4(wi, q);
                if (e.slot < 0) {   // the row's only contribution of this window: applied here (apply_single, svdf_device.h)
                    store_row<LPI>(P.W, P.item_off + e.idx, pitch, L, k, apply_single(q, wi, S.contrib_bf16 != 0));
                    if (L == 0) P.bias[P.item_off + e.idx] = apply_single(bi, nbi - bi, false);
                } else {
                    store_contrib<LPI>(S.contrib, S.contrib_bf16, (size_t)e.slot, pitch, L, k, wi);
                    if (L == 0) S.cbias[e.slot] = nbi - bi;
                }
            }
            if (FB) pp.update(P, err, ti, ub);
            reg_row<LPI>(P, wu, wd_u, false, L);
            nbu = nbu * (1.0f - lr * P.wd_user_bias);
            p = wu;
            if (ub) bu = nbu;
        }
        if (FB && seg.fb_count > 0) {   // update_ufeedback (:539-554) against the window-start feedback rows
            float4 d = pp.tmp_fb;
            sub4(d, pp.old_fb);
            float db = pp.tmp_bias - pp.old_bias;
            const float inv = 1.0f / pp.norm;
            scale4(d, inv);
            db = db * inv;
            if (S.fbrec) {   // deferred scatter: the segment's delta; k_wunit_sum forms (w + d val) - w against the rows it updates
                if (!(LPI * 4 > k && L * 4 >= k)) *reinterpret_cast<float4 *>(S.dvec + (size_t)(un.seg_begin + sg) * pitch + (size_t)L * 4) = d;
                if (L == 0) S.dbias[un.seg_begin + sg] = db;
            } else
            for (int j = seg.fb_begin; j < seg.fb_begin + seg.fb_count; j++) {
                const WinEnt f = S.fbent[j];
                const unsigned row = P.fb_off + f.idx;
                const float4 w = load_row<LPI>(P.W, row, pitch, L, k);
                float4 w2 = w;
                axpy4(w2, d, f.val);
                sub4(w2, w);
                if (f.slot < 0) store_row<LPI>(P.W, row, pitch, L, k, apply_single(w, w2, S.contrib_bf16 != 0));
                else store_contrib<LPI>(S.contrib, S.contrib_bf16, (size_t)f.slot, pitch, L, k, w2);
                if (L == 0) {
                    const float b = (ub || f.slot < 0) ? P.bias[row] : 0.0f;
                    float cb = 0.0f;
                    if (ub) { const float b2 = b + db * f.val; cb = b2 - b; }
                    if (f.slot < 0) P.bias[row] = apply_single(b, cb, false);
                    else S.cbias[f.slot] = cb;
                }
            }
        }
    }
    store_row<LPI>(P.W, ur, pitch, L, k, p);
    if (ub && L == 0) P.bias[ur] = bu;
}

// ------------------------------------------------------------------------------------------------- kernel A, the BASELINE configs[3] shapes
// Full rows of k = 4 * LANES * V floats (64 / 128) held V chunks per lane by LANES lanes -- 64 / LANES units per wave instead of two --,
// the T = 16 / LANES units of a DPP row interleaved (dot_slots reproduces the reference's four SSE chains, svdf_device.h), fixed row
// layout (NG global entries + one item entry), unit user values, reg_method 0 / 1 / 3 (elementwise).  The arithmetic is the general
// kernel's, statement for statement.  What differs is latency:
//   * NG == 0 (SVD++ and plain rows: LONG units -- 100 rows per user at configs[3]): the shared side is read-only inside a window, so the
//     rows ahead cannot go stale: a ring of D rows is in flight (records two cycles ahead, item rows one cycle ahead), and the feedback
//     gather / scatter requests eight rows at a time.  A window's launch lasts as long as its longest unit: 100 dependent row steps.
//   * NG > 0 (neighbourhood rows: SHORT units, about one row per user and window): D = 1; the gain is units in flight.
__device__ __forceinline__ void reg_chunk(const DevParams &P, float4 &w, float wd, bool is_item) {   // reg_row without the projection mode
    const float lambda = P.lr * wd;
    int method = P.reg_method;
    if (method == 3) method = is_item ? 0 : 1;
    if (method == 0) scale4(w, 1.0f - lambda);
    else l1_row(w, lambda);
    if (!is_item && P.user_nonnegative) clamp_nonneg(w);
}
template <int LANES, int V, bool FB, int NG, int D>
__global__ __launch_bounds__(256) void k_wunit_fast(const DevParams P, const WUnitSchedule S) {
    constexpr int T = 16 / LANES, IPW = 64 / LANES, K = 4 * LANES * V, E = NG + 1;
    static_assert(NG == 0 || D == 1, "the row ring is for rows without global entries");
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int m = (lane & 15) / T;
    const int gslot = (lane >> 4) * T + (lane & (T - 1));
    const long uidx = wave * IPW + gslot;
    if (uidx >= S.nunits) return;
    const WinUnit un = S.units[uidx];
    const int pitch = P.pitch;
    const bool ub = P.no_user_bias == 0;
    const float lr = P.lr;
    const unsigned ur = P.user_off + un.user;
    float4 p[V];
#pragma unroll
    for (int v = 0; v < V; v++) p[v] = load_row_nt<K / 4>(P.W, ur, pitch, m + v * LANES, K);
    float bu = ub ? P.bias[ur] : 0.0f;
    const float wd_u = get_wd(P.u_rng, un.user, P.wd_user);
    const float dec_ub = 1.0f - lr * P.wd_user_bias, dec_ib = 1.0f - lr * P.wd_item_bias;
    for (int sg = 0; sg < un.seg_count; sg++) {
        const WinSeg seg = sg == 0 ? un.first : S.segs[un.seg_begin + sg];
        float4 tmp_fb[V], old_fb[V];
        float norm = 0.0f, tmp_bias = 0.0f, old_bias = 0.0f;
#pragma unroll
        for (int v = 0; v < V; v++) { tmp_fb[v] = f4zero(); old_fb[v] = f4zero(); }
        if (FB) {   // prepare_ufeedback (:523-538), eight rows requested at a time, accumulated in list order
            for (int j0 = 0; j0 < seg.fb_count; j0 += 8) {
                WinEnt f[8];
                float4 w[8][V];
                float b[8];
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = S.fbent[seg.fb_begin + (j0 + q < seg.fb_count ? j0 + q : j0)];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned row = P.fb_off + f[q].idx;
#pragma unroll
                    for (int v = 0; v < V; v++) w[q][v] = load_row<K / 4>(P.W, row, pitch, m + v * LANES, K);
                    b[q] = ub ? P.bias[row] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (j0 + q < seg.fb_count) {
#pragma unroll
                        for (int v = 0; v < V; v++) axpy4(tmp_fb[v], w[q][v], f[q].val);
                        norm = norm + f[q].val * f[q].val;
                        if (ub) tmp_bias = tmp_bias + b[q] * f[q].val;
                    }
                }
            }
            old_bias = tmp_bias;
#pragma unroll
            for (int v = 0; v < V; v++) old_fb[v] = tmp_fb[v];
        }
        // ---- one row: the general kernel's statements on (p, bu, tmp_fb; the row's record, item row q, item bias bi)
        auto step = [&](float label, const WinEnt (&ge)[NG > 0 ? NG : 1], const float (&gb)[NG > 0 ? NG : 1], const WinEnt &ie, const float4 (&q)[V], float bi) {
            double bs = 0.0;
#pragma unroll
            for (int j = 0; j < NG; j++) bs += (double)(ge[j].val * gb[j]);
            if (ub) {
                bs += (double)(1.0f * bu);
                bs += (double)(FB ? tmp_bias : 0.0f);
            }
            bs += 0.0;
            bs += (double)(ie.val * bi);
            double sum = (double)P.base_score + bs;
            float4 tu[V], ti[V];
#pragma unroll
            for (int v = 0; v < V; v++) {
                tu[v] = FB ? tmp_fb[v] : f4zero();
                axpy4(tu[v], p[v], 1.0f);
                ti[v] = f4zero();
                axpy4(ti[v], q[v], ie.val);
            }
            sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
            const float pred = map_active((float)sum, P.active_type);
            const float err = cal_grad(label, pred, P.active_type) * 1.0f;
#pragma unroll
            for (int j = 0; j < NG; j++) {
                float g2 = gb[j] + lr * err * ge[j].val;
                g2 = reg_gbias(P, ge[j].idx, g2);
                if (m == 0) S.gcontrib[ge[j].slot] = g2 - gb[j];
            }
            const float su = lr * err * 1.0f;
            float nbu = bu + su;
            const float si = lr * err * ie.val;
            float nbi = bi + si;
            nbi = nbi * dec_ib;
            const float wd_i = get_wd(P.i_rng, ie.idx, P.wd_item);
#pragma unroll
            for (int v = 0; v < V; v++) {
                float4 wi = q[v];
                axpy4(wi, tu[v], si);
                reg_chunk(P, wi, wd_i, true);
                sub4(wi, q[v]);
                if (ie.slot < 0) store_row<K / 4>(P.W, P.item_off + ie.idx, pitch, m + v * LANES, K, apply_single(q[v], wi, S.contrib_bf16 != 0));   // the row's only contribution of the window
                else store_contrib<K / 4>(S.contrib, S.contrib_bf16, (size_t)ie.slot, pitch, m + v * LANES, K, wi);
            }
            if (m == 0) {
                if (ie.slot < 0) P.bias[P.item_off + ie.idx] = apply_single(bi, nbi - bi, false);
                else S.cbias[ie.slot] = nbi - bi;
            }
            if (FB) {   // update_svdpp (:512-520)
                const float lr2 = lr * P.scale_lr_ufeedback;
#pragma unroll
                for (int v = 0; v < V; v++) { axpy4(tmp_fb[v], ti[v], lr2 * err * norm); scale4(tmp_fb[v], 1.0f - lr2 * P.wd_ufeedback); }
                if (ub) {
                    tmp_bias = tmp_bias + lr2 * err * norm;
                    tmp_bias = tmp_bias * (1.0f - lr2 * P.wd_ufeedback_bias);
                }
            }
#pragma unroll
            for (int v = 0; v < V; v++) {
                float4 wu = p[v];
                axpy4(wu, ti[v], su);
                reg_chunk(P, wu, wd_u, false);
                p[v] = wu;
            }
            nbu = nbu * dec_ub;
            if (ub) bu = nbu;
        };
        const int n = seg.row_count, r0 = seg.row_begin;
        if constexpr (NG == 0 && D > 1) {
            // rows r0 + j: the entry index IS the row index (estride 1).  rec = this cycle's records, nrec = the next cycle's, q / bi = this
            // cycle's item rows; slot d is refilled right after its row was computed, so a row load has D row steps to arrive
            float lab[D], nlab[D], bi[D];
            WinEnt rec[D], nrec[D];
            float4 q[D][V];
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int r = r0 + (d < n ? d : 0);
                rec[d] = S.ent[r]; lab[d] = S.label[r];
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                const unsigned row = P.item_off + rec[d].idx;
#pragma unroll
                for (int v = 0; v < V; v++) q[d][v] = load_row<K / 4>(P.W, row, pitch, m + v * LANES, K);
                bi[d] = P.bias[row];
            }
#pragma unroll
            for (int d = 0; d < D; d++) {
                const int r = r0 + (d + D < n ? d + D : 0);
                nrec[d] = S.ent[r]; nlab[d] = S.label[r];
            }
            const WinEnt none[1] = {WinEnt{0u, 0.0f, 0, 0}};
            const float nog[1] = {0.0f};
            for (int base = 0; base < n; base += D) {
#pragma unroll
                for (int d = 0; d < D; d++) {
                    const int j = base + d;
                    if (j < n) {
                        step(lab[d], none, nog, rec[d], q[d], bi[d]);
                        rec[d] = nrec[d]; lab[d] = nlab[d];
                        const unsigned row = P.item_off + rec[d].idx;
#pragma unroll
                        for (int v = 0; v < V; v++) q[d][v] = load_row<K / 4>(P.W, row, pitch, m + v * LANES, K);
                        bi[d] = P.bias[row];
                        const int r = r0 + (j + 2 * D < n ? j + 2 * D : 0);
                        nrec[d] = S.ent[r]; nlab[d] = S.label[r];
                    }
                }
            }
        } else {
            for (int j = 0; j < n; j++) {
                const int r = r0 + j;
                WinEnt ge[NG > 0 ? NG : 1];
                float gb[NG > 0 ? NG : 1];
                ge[0] = WinEnt{0u, 0.0f, 0, 0}; gb[0] = 0.0f;
#pragma unroll
                for (int g = 0; g < NG; g++) ge[g] = S.ent[(long)r * E + g];
                const WinEnt ie = S.ent[(long)r * E + NG];
                const float label = S.label[r];
#pragma unroll
                for (int g = 0; g < NG; g++) gb[g] = P.g_bias[ge[g].idx];
                float4 q[V];
                const unsigned row = P.item_off + ie.idx;
#pragma unroll
                for (int v = 0; v < V; v++) q[v] = load_row<K / 4>(P.W, row, pitch, m + v * LANES, K);
                const float bi = P.bias[row];
                step(label, ge, gb, ie, q, bi);
            }
        }
        if (FB && seg.fb_count > 0) {   // update_ufeedback (:539-554) against the window-start feedback rows, eight rows at a time
            float4 d[V];
#pragma unroll
            for (int v = 0; v < V; v++) { d[v] = tmp_fb[v]; sub4(d[v], old_fb[v]); }
            float db = tmp_bias - old_bias;
            const float inv = 1.0f / norm;
#pragma unroll
            for (int v = 0; v < V; v++) scale4(d[v], inv);
            db = db * inv;
            if (S.fbrec) {   // deferred scatter (see k_wunit_walk)
#pragma unroll
                for (int v = 0; v < V; v++) *reinterpret_cast<float4 *>(S.dvec + (size_t)(un.seg_begin + sg) * pitch + (size_t)(m + v * LANES) * 4) = d[v];
                if (m == 0) S.dbias[un.seg_begin + sg] = db;
            } else
            for (int j0 = 0; j0 < seg.fb_count; j0 += 8) {
                WinEnt f[8];
                float4 w[8][V];
                float b[8];
#pragma unroll
                for (int q = 0; q < 8; q++) f[q] = S.fbent[seg.fb_begin + (j0 + q < seg.fb_count ? j0 + q : j0)];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const unsigned row = P.fb_off + f[q].idx;
#pragma unroll
                    for (int v = 0; v < V; v++) w[q][v] = load_row<K / 4>(P.W, row, pitch, m + v * LANES, K);
                    b[q] = ub ? P.bias[row] : 0.0f;
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    if (j0 + q < seg.fb_count) {
#pragma unroll
                        for (int v = 0; v < V; v++) {
                            float4 w2 = w[q][v];
                            axpy4(w2, d[v], f[q].val);
                            sub4(w2, w[q][v]);
                            if (f[q].slot < 0) store_row<K / 4>(P.W, P.fb_off + f[q].idx, pitch, m + v * LANES, K, apply_single(w[q][v], w2, S.contrib_bf16 != 0));
                            else store_contrib<K / 4>(S.contrib, S.contrib_bf16, (size_t)f[q].slot, pitch, m + v * LANES, K, w2);
                        }
                        if (m == 0) {
                            float cb = 0.0f;
                            if (ub) { const float b2 = b[q] + db * f[q].val; cb = b2 - b[q]; }
                            if (f[q].slot < 0) { const float b0 = ub ? b[q] : P.bias[P.fb_off + f[q].idx]; P.bias[P.fb_off + f[q].idx] = apply_single(b0, cb, false); }
                            else S.cbias[f[q].slot] = cb;
                        }
                    }
                }
            }
        }
    }
#pragma unroll
    for (int v = 0; v < V; v++) store_row<K / 4>(P.W, ur, pitch, m + v * LANES, K, p[v]);
    if (ub && m == 0) P.bias[ur] = bu;
}

// ------------------------------------------------------------------------------------------------- kernel B: per-target sums
// Target t = replicated row t of [W_ufeedback rows | W_item rows]; its contributions sit in slots [tptr[t], tptr[t + 1]) in file order.
// One lane group per target adds them in that order (acc = 0 + c_1 + c_2 ..., eight rows requested at a time).
//   LOCAL: the sum is added to the model in place (one GPU: nobody else holds a part of it);
//   else:  the wire buffer of the exchange, dst = [T rows of `pitch` | T biases | nglobal global biases] (the packed layout of
//          Engine::delta_ranges for the whole item range), fp32 or fp16.
// The global biases' sums (gptr over gcontrib) are taken by the same launch, one thread per global id.
template <int LPI, bool HALF, bool LOCAL>
__global__ __launch_bounds__(256) void k_wunit_sum(const WUnitSchedule S, float *W, float *bias, float *g_bias, unsigned fb_off, unsigned item_off,
                                                   int pitch, int k, void *dst) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    const long first = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long T = S.nfb_rows + S.nitem_rows;
    // a target is a chain of dependent loads (its slot range, the records or rows of its slots, for deferred feedback rows the segments' deltas) and the
    // kernel is bound by how many such chains the resident waves hold, not by bytes: the next target's slot range is requested one iteration ahead
    // One-GPU windows bring the list of targets that have slots (a window touches a fraction of the rows, and a row's only contribution has been applied by
    // the walk): every lane group of a wave then has work in every iteration.
    const bool compact = LOCAL && S.touched != nullptr;
    const long N = compact ? S.ntouched : T;
    long pt = first;
    int pb = 0, pe = 0;
    auto request = [&](long i) {
        if (compact) { const WinTouched x = S.touched[i]; pt = x.t; pb = x.b; pe = x.e; }
        else { pt = i; pb = S.tptr[i]; pe = S.tptr[i + 1]; }
    };
    if (first < N) request(first);
    for (long i = first; i < N; i += stride) {
        const long t = pt;
        const int b = pb, e = pe;
        if (i + stride < N) request(i + stride);
        if (LOCAL && b == e) continue;
        float4 acc = f4zero();
        float accb = 0.0f;
        const bool owns = !(LPI * 4 > k && L * 4 >= k);
        if (S.fbrec && t < S.nfb_rows) {
            // deferred feedback scatter: slot s names a segment and the entry's value; the contribution is what the unit's walk would have stored --
            // (w + d val) - w against the window-start row, rounded like a stored contribution row -- and the sum runs in slot (= file) order
            const size_t row = (size_t)fb_off + (size_t)t;
            const float4 w = owns ? *reinterpret_cast<const float4 *>(W + row * pitch + (size_t)L * 4) : f4zero();
            const float bw = S.user_bias ? bias[row] : 0.0f;
            const bool bf = S.contrib_bf16 != 0;
            constexpr int DB = 4;   // records / deltas requested together (a feedback row of the configs[3] windows meets 1.5 contributions on average)
            for (int s0 = b; s0 < e; s0 += DB) {
                WinFbRec r[DB];
                float4 d[DB];
                float dbv[DB];
#pragma unroll
                for (int q = 0; q < DB; q++) r[q] = S.fbrec[min(s0 + q, e - 1)];
#pragma unroll
                for (int q = 0; q < DB; q++) {
                    d[q] = owns ? *reinterpret_cast<const float4 *>(S.dvec + (size_t)r[q].seg * pitch + (size_t)L * 4) : f4zero();
                    dbv[q] = S.dbias[r[q].seg];
                }
#pragma unroll
                for (int q = 0; q < DB; q++) {
                    if (s0 + q < e) {
                        float4 w2 = w;
                        axpy4(w2, d[q], r[q].val);
                        sub4(w2, w);
                        add_rows(acc, make_float4(contrib_as_stored(w2.x, bf), contrib_as_stored(w2.y, bf), contrib_as_stored(w2.z, bf), contrib_as_stored(w2.w, bf)));
                        float cb = 0.0f;
                        if (S.user_bias) { const float b2 = bw + dbv[q] * r[q].val; cb = b2 - bw; }
                        accb = accb + cb;
                    }
                }
            }
        } else if (S.contrib_bf16) sum_contrib_slots<LPI, true, (LOCAL ? 4 : 8)>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
        else sum_contrib_slots<LPI, false, (LOCAL ? 4 : 8)>(S.contrib, S.cbias, b, e, pitch, L, k, acc, accb);
        if (LOCAL) {
            const size_t row = t < S.nfb_rows ? (size_t)fb_off + (size_t)t : (size_t)item_off + (size_t)(t - S.nfb_rows);
            if (owns) {
                float4 *w = reinterpret_cast<float4 *>(W + row * pitch + (size_t)L * 4);
                float4 c = *w;
                c.x = c.x + acc.x; c.y = c.y + acc.y; c.z = c.z + acc.z; c.w = c.w + acc.w;
                *w = c;
            }
            if (L == 0) bias[row] = bias[row] + accb;
            continue;
        }
        if (owns) {
            const size_t pos = (size_t)t * pitch + (size_t)L * 4;
            if (HALF) {
                __half2 *h = reinterpret_cast<__half2 *>(reinterpret_cast<__half *>(dst) + pos);
                h[0] = __halves2half2(__float2half_rn(acc.x), __float2half_rn(acc.y));
                h[1] = __halves2half2(__float2half_rn(acc.z), __float2half_rn(acc.w));
            } else {
                *reinterpret_cast<float4 *>(reinterpret_cast<float *>(dst) + pos) = acc;
            }
        }
        if (L == 0) {
            const size_t pos = (size_t)T * pitch + (size_t)t;
            if (HALF) reinterpret_cast<__half *>(dst)[pos] = __float2half_rn(accb);
            else reinterpret_cast<float *>(dst)[pos] = accb;
        }
    }
    const long g0 = T * (long)(pitch + 1);
    for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < S.nglobal; g += (long)gridDim.x * blockDim.x) {
        const int b = S.gptr ? S.gptr[g] : 0, e = S.gptr ? S.gptr[g + 1] : 0;
        float acc = 0.0f;
        for (int s = b; s < e; s++) acc = acc + S.gcontrib[s];
        if (LOCAL) { if (b < e) g_bias[g] = g_bias[g] + acc; }
        else if (HALF) reinterpret_cast<__half *>(dst)[g0 + g] = __float2half_rn(acc);
        else reinterpret_cast<float *>(dst)[g0 + g] = acc;
    }
}

bool wunit_fast_applies(const DevParams &P, const WUnitSchedule &S, bool feedback) {
    if (!(P.k == 64 || P.k == 128) || S.rptr != nullptr || S.uval != nullptr || P.reg_method == 2) return false;
    const int ng = S.estride - 1;
    return ng == 0 || (ng == 4 && !feedback);
}
void launch_wunit_walk(const DevParams &P, const WUnitSchedule &S, bool feedback, int fast, hipStream_t st) {
    if (S.nunits <= 0) return;
    // fast: 0 = the general lane-group kernel, 1 = the slot kernel where it applies, 2 (default) = in addition one WAVE per unit for user-group
    // windows whose launch does not fill the chip anyway (its time is the longest unit's latency: svdf_k_wave.hip, k_wunit_wave)
    if (fast >= 2 && wunit_wave_applies(P, S, feedback) && S.nunits <= 16384) { launch_wunit_wave(P, S, st); return; }
    if (fast && wunit_fast_applies(P, S, feedback)) {
        auto go = [&](auto lanes) {
            constexpr int LANES = decltype(lanes)::value;
            const long per_wave = 64 / LANES;
            const long waves = (S.nunits + per_wave - 1) / per_wave;
            if (S.estride == 1) {
                if (feedback) hipLaunchKernelGGL((k_wunit_fast<LANES, 2, true, 0, 4>), dim3((unsigned)waves), dim3(64), 0, st, P, S);
                else hipLaunchKernelGGL((k_wunit_fast<LANES, 2, false, 0, 4>), dim3((unsigned)waves), dim3(64), 0, st, P, S);
            } else {
                hipLaunchKernelGGL((k_wunit_fast<LANES, 2, false, 4, 1>), dim3((unsigned)waves), dim3(64), 0, st, P, S);
            }
        };
        if (P.k == 64) go(std::integral_constant<int, 8>());
        else go(std::integral_constant<int, 16>());
        return;
    }
    const int lpi = lanes_per_instance(P.k);
    const long ipw = 64 / lpi;
    const long waves = (S.nunits + ipw - 1) / ipw;
    if (feedback) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_walk<LPI, true>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_walk<LPI, false>), dim3((unsigned)waves), dim3(64), 0, st, P, S)); }
}
// dst == nullptr: add the sums to the model in place; else the wire buffer (half: fp16)
void launch_wunit_sum(const DevParams &P, const WUnitSchedule &S, void *dst, int half, hipStream_t st) {
    const long T = S.nfb_rows + S.nitem_rows;
    if (T <= 0 && S.nglobal <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const long ipw = 64 / lpi;
    const long work = (!dst && S.touched) ? S.ntouched : T;   // in place over the list of targets with slots, else every target
    long waves = (std::max<long>(work, 1) + ipw - 1) / ipw;
    long grid = (waves + 3) / 4;
    // (many short-lived waves beat one resident set walking several targets each: grid cap 2 048 -> 53.5 us, 4 096 -> 45.8, 16 384 -> 43.8 per SVD++ window)
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    if (!dst) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, false, true>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, (void *)nullptr)); }
    else if (half) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, true, false>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, dst)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, false, false>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, dst)); }
}

}  // namespace svdf
