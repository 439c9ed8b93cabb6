// svdf_k_general.hip -- general sparse kernels (k_general, k_predict) and the lane-group SVD++ kernels (k_svdpp, k_svdpp_predict)
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_device.h"

namespace svdf {

// =====================================================================================
// General sparse instance (any number of global / user / item features, side-feature children,
// every regulariser).  Rows are read-modify-written through memory in the reference's order, so
// an id that appears twice in one instance is updated and decayed twice like the reference does.
// =====================================================================================
template <typename R>
struct SvdppRegsT {   // SVDPPFeature members (apex_svd_base.h:486-488) held in registers
    R tmp_fb, old_fb;
    float norm, tmp_bias, old_bias;
};
using SvdppRegs = SvdppRegsT<float4>;

// pred() (:445-454): fills tmp_u / tmp_i, returns the score before the link function (double)
template <int LPI, typename R>
__device__ __forceinline__ double instance_score(const DevParams &P, int ng, int nu, int ni, const unsigned *idx,
                                                 const float *val, int L, const SvdppRegsT<R> *pp, R &tu, R &ti) {
    using io = row_io<LPI, R>;
    const int k = P.k, pitch = P.pitch;
    const unsigned *ig = idx, *iu = idx + ng, *ii = idx + ng + nu;
    const float *vg = val, *vu = val + ng, *vi = val + ng + nu;
    double bs = 0.0;
    for (int j = 0; j < ng; j++) bs += (double)(vg[j] * P.g_bias[gpos(P, ig[j])]);
    if (P.no_user_bias == 0) {
        for (int j = 0; j < nu; j++) {
            const unsigned uid = iu[j];
            bs += (double)(vu[j] * P.bias[P.user_off + uid]);
            if (uid < P.feat_user.num_row)
                for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                    bs += (double)(P.bias[P.user_off + P.feat_user.index[c]] * P.feat_user.value[c]);
        }
        bs += (double)(pp ? pp->tmp_bias : 0.0f);
    }
    bs += 0.0;
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        const float ival = vi[j];
        bs += (double)(ival * P.bias[P.item_off + iid]);
        if (iid < P.feat_item.num_row)
            for (unsigned c = P.feat_item.row_ptr[iid]; c < P.feat_item.row_ptr[iid + 1]; c++)
                bs += (double)(P.bias[P.item_off + P.feat_item.index[c]] * P.feat_item.value[c] * ival);
    }
    double sum = (double)P.base_score + bs;
    tu = pp ? pp->tmp_fb : row_traits<R>::zero();
    ti = row_traits<R>::zero();
    for (int j = 0; j < nu; j++) {
        const unsigned uid = iu[j];
        axpy4(tu, io::load(P.W, P.user_off + uid, pitch, L, k), vu[j]);
        if (uid < P.feat_user.num_row)
            for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                axpy4(tu, io::load(P.W, P.user_off + P.feat_user.index[c], pitch, L, k), P.feat_user.value[c]);
    }
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        const float ival = vi[j];
        axpy4(ti, io::load(P.W, P.item_off + iid, pitch, L, k), ival);
        if (iid < P.feat_item.num_row)
            for (unsigned c = P.feat_item.row_ptr[iid]; c < P.feat_item.row_ptr[iid + 1]; c++)  // scalar formed in double
                axpy4(ti, io::load(P.W, P.item_off + P.feat_item.index[c], pitch, L, k),
                      (float)((double)P.feat_item.value[c] * (double)ival));
    }
    sum += (double)group_dot<LPI>(tu, ti, L, k);
    return sum;
}

// W[row] += tmp*sc ; bias[row] += sc   (every lane of the group stores the same bias value so
// each thread later reads back its own write)
template <int LPI, typename R>
__device__ __forceinline__ void rmw_row(const DevParams &P, unsigned row, const R &tmp, float sc, bool with_bias, int L) {
    R w = row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k);
    axpy4(w, tmp, sc);
    row_io<LPI, R>::store(P.W, row, P.pitch, L, P.k, w);
    if (with_bias) { float b = P.bias[row]; b = b + sc; P.bias[row] = b; }
}
template <int LPI, typename R>
__device__ __forceinline__ void reg_user(const DevParams &P, unsigned uid, int L, unsigned counter) {  // :211-250
    const unsigned row = P.user_off + uid;
    R w = row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k);
    reg_row<LPI>(P, w, get_wd(P.u_rng, uid, P.wd_user), false, L, lazy_span(P, row, counter));
    row_io<LPI, R>::store(P.W, row, P.pitch, L, P.k, w);
    if (P.no_user_bias == 0) { float b = P.bias[row]; b = b * (1.0f - P.lr * P.wd_user_bias); P.bias[row] = b; }
}
template <int LPI, typename R>
__device__ __forceinline__ void reg_item(const DevParams &P, unsigned iid, int L, unsigned counter) {  // :251-283
    const unsigned row = P.item_off + iid;
    R w = row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k);
    reg_row<LPI>(P, w, get_wd(P.i_rng, iid, P.wd_item), true, L, lazy_span(P, row, counter));
    row_io<LPI, R>::store(P.W, row, P.pitch, L, P.k, w);
    float b = P.bias[row]; b = b * (1.0f - P.lr * P.wd_item_bias); P.bias[row] = b;
}

// regularize(feature, is_after_update) (:286-311): globals and factor rows each run either before the step
// (lazy modes 4/5, with the sample counter of BEFORE the step) or after it (modes 0..3)
template <int LPI, typename R>
__device__ __forceinline__ void instance_regularize(const DevParams &P, int ng, int nu, int ni, const unsigned *idx, int L,
                                                    bool after, unsigned counter) {
    const unsigned *ig = idx, *iu = idx + ng, *ii = idx + ng + nu;
    if (after == (P.reg_global < 4))
        for (int j = 0; j < ng; j++) { const unsigned gid = ig[j]; float g = reg_gbias(P, gid, P.g_bias[gpos(P, gid)], counter); P.g_bias[gpos(P, gid)] = g; }
    if (after == (P.reg_method < 4)) {
        for (int j = 0; j < nu; j++) {
            const unsigned uid = iu[j];
            reg_user<LPI, R>(P, uid, L, counter);
            if (uid < P.feat_user.num_row)
                for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++) reg_user<LPI, R>(P, P.feat_user.index[c], L, counter);
        }
        for (int j = 0; j < ni; j++) {
            const unsigned iid = ii[j];
            reg_item<LPI, R>(P, iid, L, counter);
            if (iid < P.feat_item.num_row)
                for (unsigned c = P.feat_item.row_ptr[iid]; c < P.feat_item.row_ptr[iid + 1]; c++) reg_item<LPI, R>(P, P.feat_item.index[c], L, counter);
        }
    }
}

// update_inner (:456-462); counter = sample_counter before this instance (only the lazy modes look at it)
template <int LPI, typename R>
__device__ __forceinline__ void instance_update(const DevParams &P, float label, int ng, int nu, int ni,
                                                const unsigned *idx, const float *val, int L, SvdppRegsT<R> *pp, unsigned counter) {
    const unsigned *ig = idx, *iu = idx + ng, *ii = idx + ng + nu;
    const float *vg = val, *vu = val + ng, *vi = val + ng + nu;
    if (P.reg_method >= 4 || P.reg_global >= 4) instance_regularize<LPI, R>(P, ng, nu, ni, idx, L, false, counter);
    R tu, ti;
    const double sum = instance_score<LPI, R>(P, ng, nu, ni, idx, val, L, pp, tu, ti);
    const float pred = map_active((float)sum, P.active_type);
    const float err = cal_grad(label, pred, P.active_type) * 1.0f;
    const float lr = P.lr;
    const bool ub = P.no_user_bias == 0;
    // ---- update_no_decay (:383-427)
    for (int j = 0; j < ng; j++) { float g = P.g_bias[gpos(P, ig[j])]; g = g + lr * err * vg[j]; P.g_bias[gpos(P, ig[j])] = g; }
    for (int j = 0; j < nu; j++) {
        const unsigned uid = iu[j];
        rmw_row<LPI, R>(P, P.user_off + uid, ti, lr * err * vu[j], ub, L);
        if (uid < P.feat_user.num_row)
            for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                rmw_row<LPI, R>(P, P.user_off + P.feat_user.index[c], ti, lr * err * P.feat_user.value[c], ub, L);
    }
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        const float ival = vi[j];
        rmw_row<LPI, R>(P, P.item_off + iid, tu, lr * err * ival, true, L);
        if (iid < P.feat_item.num_row)
            for (unsigned c = P.feat_item.row_ptr[iid]; c < P.feat_item.row_ptr[iid + 1]; c++)
                rmw_row<LPI, R>(P, P.item_off + P.feat_item.index[c], tu, lr * err * P.feat_item.value[c] * ival, true, L);
    }
    if (pp) {  // update_svdpp (:512-520)
        const float lr2 = lr * P.scale_lr_ufeedback;
        axpy4(pp->tmp_fb, ti, lr2 * err * pp->norm);
        scale4(pp->tmp_fb, 1.0f - lr2 * P.wd_ufeedback);
        if (ub) {
            pp->tmp_bias = pp->tmp_bias + lr2 * err * pp->norm;
            pp->tmp_bias = pp->tmp_bias * (1.0f - lr2 * P.wd_ufeedback_bias);
        }
    }
    // ---- sample_counter++ ; regularize(feature, true)
    instance_regularize<LPI, R>(P, ng, nu, ni, idx, L, true, counter + 1u);
}

// Kernel 2: one conflict-free batch of general instances; order[] lists instance ids of the batch.
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_general(const DevParams P, const DevCSR D, const int *order, long begin, long end,
                                                 unsigned counter_base) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = begin + gidx; s < end; s += stride) {
        const int r = order ? order[s] : (int)s;
        const int p0 = D.row_ptr[3 * (long)r], p1 = D.row_ptr[3 * (long)r + 1], p2 = D.row_ptr[3 * (long)r + 2], p3 = D.row_ptr[3 * (long)r + 3];
        instance_update<LPI, R>(P, D.row_label[r], p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, nullptr,
                                counter_base + (unsigned)r);
    }
}

// Kernel 3: predictions for a CSR stream (read-only, every instance independent)
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_predict(const DevParams P, const DevCSR D, long n, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long r = gidx; r < n; r += stride) {
        const int p0 = D.row_ptr[3 * r], p1 = D.row_ptr[3 * r + 1], p2 = D.row_ptr[3 * r + 2], p3 = D.row_ptr[3 * r + 3];
        R tu, ti;
        const double sum = instance_score<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, nullptr, tu, ti);
        if (L == 0) out[r] = map_active((float)sum, P.active_type);
    }
}

// ---- SVD++ user units (SVDPPFeature, apex_svd_base.h:484-592) --------------------------------
template <int LPI, typename R>
__device__ __forceinline__ void svdpp_load_state(const DevParams &P, SvdppRegsT<R> &pp, int L) {
    const float *st = P.svdpp_state;
    pp.tmp_fb = row_io<LPI, R>::load(st, 0, P.pitch, L, P.k);
    pp.old_fb = row_io<LPI, R>::load(st, 1, P.pitch, L, P.k);
    pp.norm = st[2 * P.pitch]; pp.tmp_bias = st[2 * P.pitch + 1]; pp.old_bias = st[2 * P.pitch + 2];
}
template <int LPI, typename R>
__device__ __forceinline__ void svdpp_save_state(const DevParams &P, const SvdppRegsT<R> &pp, int L) {
    float *st = P.svdpp_state;
    row_io<LPI, R>::store(st, 0, P.pitch, L, P.k, pp.tmp_fb);
    row_io<LPI, R>::store(st, 1, P.pitch, L, P.k, pp.old_fb);
    if (L == 0) { st[2 * P.pitch] = pp.norm; st[2 * P.pitch + 1] = pp.tmp_bias; st[2 * P.pitch + 2] = pp.old_bias; }
}
// prepare_ufeedback (:523-538)
template <int LPI, typename R>
__device__ __forceinline__ void svdpp_prepare(const DevParams &P, SvdppRegsT<R> &pp, const unsigned *fidx, const float *fval, int nfb, int L) {
    pp.norm = 0.0f; pp.tmp_fb = row_traits<R>::zero(); pp.tmp_bias = 0.0f;
    for (int j = 0; j < nfb; j++) {
        const unsigned row = P.fb_off + fidx[j];
        const float v = fval[j];
        axpy4(pp.tmp_fb, row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k), v);
        pp.norm = pp.norm + v * v;
        if (P.no_user_bias == 0) pp.tmp_bias = pp.tmp_bias + P.bias[row] * v;
    }
}
// update_ufeedback (:539-554)
template <int LPI, typename R>
__device__ __forceinline__ void svdpp_scatter(const DevParams &P, SvdppRegsT<R> &pp, const unsigned *fidx, const float *fval, int nfb, int L) {
    if (nfb == 0) return;
    R d = pp.tmp_fb;
    sub4(d, pp.old_fb);  // K5
    float db = pp.tmp_bias - pp.old_bias;
    const float inv = 1.0f / pp.norm;
    scale4(d, inv);
    db = db * inv;
    pp.tmp_fb = d; pp.tmp_bias = db;  // the reference leaves the scaled delta in tmp_ufeedback
    for (int j = 0; j < nfb; j++) {
        const unsigned row = P.fb_off + fidx[j];
        const float v = fval[j];
        R w = row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k);
        axpy4(w, d, v);
        row_io<LPI, R>::store(P.W, row, P.pitch, L, P.k, w);
        if (P.no_user_bias == 0) { float b = P.bias[row]; b = b + db * v; P.bias[row] = b; }
    }
}

// Kernel 4b: the other units of a conflict-free batch (any row shape); one lane group walks one user's rows in order
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_svdpp(const DevParams P, const DevCSR D, const DevUnit *units, const unsigned *fb_index,
                                               const float *fb_value, const int *order, long begin, long end, unsigned counter_base) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = begin + gidx; s < end; s += stride) {
        const DevUnit u = units[order ? order[s] : (int)s];
        SvdppRegsT<R> pp;
        if (u.flags & UNIT_LOAD) svdpp_load_state<LPI, R>(P, pp, L);
        if (u.flags & UNIT_START) {
            svdpp_prepare<LPI, R>(P, pp, fb_index + u.fb_begin, fb_value + u.fb_begin, u.fb_end - u.fb_begin, L);
            pp.old_bias = pp.tmp_bias;
            pp.old_fb = pp.tmp_fb;
        }
        {
            for (int r = u.row_begin; r < u.row_end; r++) {
                const int p0 = D.row_ptr[3 * (long)r], p1 = D.row_ptr[3 * (long)r + 1], p2 = D.row_ptr[3 * (long)r + 2], p3 = D.row_ptr[3 * (long)r + 3];
                instance_update<LPI, R>(P, D.row_label[r], p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, &pp,
                                        counter_base + (unsigned)r);
            }
        }
        if (u.flags & UNIT_END) svdpp_scatter<LPI, R>(P, pp, fb_index + u.fb_begin, fb_value + u.fb_begin, u.fb_end - u.fb_begin, L);
        if (u.flags & UNIT_SAVE) svdpp_save_state<LPI, R>(P, pp, L);
    }
}
// Kernel 5: predictions for user units (predict(vector<float>&, SVDPlusBlock), :583-591)
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_svdpp_predict(const DevParams P, const DevCSR D, const DevUnit *units, const unsigned *fb_index,
                                                       const float *fb_value, long nunit, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = gidx; s < nunit; s += stride) {
        const DevUnit u = units[s];
        SvdppRegsT<R> pp;
        if (u.flags & UNIT_LOAD) svdpp_load_state<LPI, R>(P, pp, L);
        if (u.flags & UNIT_START) svdpp_prepare<LPI, R>(P, pp, fb_index + u.fb_begin, fb_value + u.fb_begin, u.fb_end - u.fb_begin, L);
        for (int r = u.row_begin; r < u.row_end; r++) {
            const int p0 = D.row_ptr[3 * (long)r], p1 = D.row_ptr[3 * (long)r + 1], p2 = D.row_ptr[3 * (long)r + 2], p3 = D.row_ptr[3 * (long)r + 3];
            R tu, ti;
            const double sum = instance_score<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, &pp, tu, ti);
            if (L == 0) out[r] = map_active((float)sum, P.active_type);
        }
        if (u.flags & UNIT_SAVE) svdpp_save_state<LPI, R>(P, pp, L);
    }
}

void launch_general(const DevParams &P, const DevCSR &D, const int *order, long begin, long end, unsigned counter_base, hipStream_t st) {
    if (end <= begin) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(end - begin, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_general<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, order, begin, end, counter_base));
}
void launch_predict(const DevParams &P, const DevCSR &D, long n, float *out, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_predict<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, n, out));
}
void launch_svdpp(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                  const int *order, long begin, long end, unsigned counter_base, hipStream_t st) {
    if (end <= begin) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(end - begin, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, units, fb_index, fb_value, order, begin, end, counter_base));
}
void launch_svdpp_predict(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                          long nunit, float *out, hipStream_t st) {
    if (nunit <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(nunit, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp_predict<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, units, fb_index, fb_value, nunit, out));
}

}  // namespace svdf
