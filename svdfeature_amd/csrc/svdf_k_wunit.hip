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
#include "svdf_instance.h"

namespace svdf {

#pragma clang fp contract(off)

// ------------------------------------------------------------------------------------------------- kernel A: the users' exact walks
// FB: the trainer is SVDPPFeature (user-group format): a segment prepares tmp_ufeedback from its feedback list (:523-538), every row
// goes through the update_svdpp hook (:512-520), the segment's end is update_ufeedback (:539-554) against the window-start rows.
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
        const WinSeg seg = S.segs[un.seg_begin + sg];
        SvdppRegs pp;
        pp.tmp_fb = f4zero(); pp.old_fb = f4zero(); pp.norm = 0.0f; pp.tmp_bias = 0.0f; pp.old_bias = 0.0f;
        if (FB) {   // prepare_ufeedback + the backup of update(block) (:568-574)
            for (int j = seg.fb_begin; j < seg.fb_begin + seg.fb_count; j++) {
                const unsigned row = P.fb_off + S.fbidx[j];
                const float v = S.fbval[j];
                axpy4(pp.tmp_fb, load_row<LPI>(P.W, row, pitch, L, k), v);
                pp.norm = pp.norm + v * v;
                if (ub) pp.tmp_bias = pp.tmp_bias + P.bias[row] * v;
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
            for (int j = e0; j < e1; j++) bs += (double)(S.eval[j] * P.g_bias[S.eidx[j]]);
            if (ub) {
                bs += (double)(ua * bu);
                bs += (double)(FB ? pp.tmp_bias : 0.0f);
            }
            bs += 0.0;
            for (int j = e1; j < e2; j++) bs += (double)(S.eval[j] * P.bias[P.item_off + S.eidx[j]]);
            double sum = (double)P.base_score + bs;
            float4 tu = FB ? pp.tmp_fb : f4zero();
            axpy4(tu, p, ua);
            float4 ti = f4zero();
            for (int j = e1; j < e2; j++) axpy4(ti, load_row<LPI>(P.W, P.item_off + S.eidx[j], pitch, L, k), S.eval[j]);
            sum += (double)group_dot<LPI>(tu, ti, L, k);
            const float pred = map_active((float)sum, P.active_type);
            const float err = cal_grad(label, pred, P.active_type) * 1.0f;
            // ---- update_no_decay (:383-427) + regularize(after) (:286-311), the shared rows' part as contributions
            for (int j = e0; j < e1; j++) {
                const unsigned gid = S.eidx[j];
                const float g = P.g_bias[gid];
                float g2 = g + lr * err * S.eval[j];
                g2 = reg_gbias(P, gid, g2);
                if (L == 0) S.gcontrib[S.eslot[j]] = g2 - g;
            }
            const float su = lr * err * ua;
            float4 wu = p;
            axpy4(wu, ti, su);
            float nbu = bu + su;
            for (int j = e1; j < e2; j++) {
                const unsigned iid = S.eidx[j];
                const float si = lr * err * S.eval[j];
                const float4 q = load_row<LPI>(P.W, P.item_off + iid, pitch, L, k);
                const float bi = P.bias[P.item_off + iid];
                float4 wi = q;
                axpy4(wi, tu, si);
                float nbi = bi + si;
                reg_row<LPI>(P, wi, get_wd(P.i_rng, iid, P.wd_item), true, L);
                nbi = nbi * (1.0f - lr * P.wd_item_bias);
                sub4(wi, q);
                const int slot = S.eslot[j];
                store_row<LPI>(S.contrib, (size_t)slot, pitch, L, k, wi);
                if (L == 0) S.cbias[slot] = nbi - bi;
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
            for (int j = seg.fb_begin; j < seg.fb_begin + seg.fb_count; j++) {
                const unsigned row = P.fb_off + S.fbidx[j];
                const float v = S.fbval[j];
                const float4 w = load_row<LPI>(P.W, row, pitch, L, k);
                float4 w2 = w;
                axpy4(w2, d, v);
                sub4(w2, w);
                const int slot = S.fbslot[j];
                store_row<LPI>(S.contrib, (size_t)slot, pitch, L, k, w2);
                if (L == 0) {
                    float cb = 0.0f;
                    if (ub) { const float b = P.bias[row]; const float b2 = b + db * v; cb = b2 - b; }
                    S.cbias[slot] = cb;
                }
            }
        }
    }
    store_row<LPI>(P.W, ur, pitch, L, k, p);
    if (ub && L == 0) P.bias[ur] = bu;
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
    for (long t = first; t < T; t += stride) {
        const int b = S.tptr[t], e = S.tptr[t + 1];
        if (LOCAL && b == e) continue;
        float4 acc = f4zero();
        float accb = 0.0f;
        for (int s = b; s < e; s += 8) {
            float4 c[8];
            float cb[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const bool in = s + q < e;
                c[q] = in ? load_row<LPI>(S.contrib, (size_t)(s + q), pitch, L, k) : f4zero();
                cb[q] = in ? S.cbias[s + q] : 0.0f;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) { add_rows(acc, c[q]); accb = accb + cb[q]; }
        }
        const bool owns = !(LPI * 4 > k && L * 4 >= k);
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

void launch_wunit_walk(const DevParams &P, const WUnitSchedule &S, bool feedback, hipStream_t st) {
    if (S.nunits <= 0) return;
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
    long waves = (std::max<long>(T, 1) + ipw - 1) / ipw;
    long grid = (waves + 3) / 4;
    if (grid > 16384) grid = 16384;
    if (grid < 1) grid = 1;
    if (!dst) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, false, true>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, (void *)nullptr)); }
    else if (half) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, true, false>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, dst)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_wunit_sum<LPI, false, false>), dim3((unsigned)grid), dim3(256), 0, st, S, P.W, P.bias, P.g_bias, P.fb_off, P.item_off, P.pitch, P.k, dst)); }
}

}  // namespace svdf
