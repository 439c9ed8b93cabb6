// svdf_instance.h -- the per-instance code of the general kernels (pred / update_no_decay / regularize in the reference's
// order, apex_svd_base.h:286-462) written once over a row type R (float4 lane groups or WideRow) and an implicit-feedback
// policy PP; shared by svdf_k_general.hip (k_general, k_svdpp) and svdf_k_imfb.hip (k_imfb).
#ifndef SVDF_INSTANCE_H_
#define SVDF_INSTANCE_H_
#include "svdf_device.h"

#pragma clang fp contract(off)

namespace svdf {

// =====================================================================================
// General sparse instance (any number of global / user / item features, side-feature children,
// every regulariser).  Rows are read-modify-written through memory in the reference's order, so
// an id that appears twice in one instance is updated and decayed twice like the reference does.
// =====================================================================================
// The solver's implicit-feedback hooks (prepare_svdpp / get_bias_svdpp / update_svdpp, apex_svd_base.h:429-441) are a
// compile-time POLICY of the per-instance code: NoFeedback = SVDFeature's defaults, SvdppRegsT = SVDPPFeature (:506-520),
// ImfbRegsT (svdf_k_imfb.hip) = the multi-level variant solver (solvers/multi-imfb/apex_multi_imfb.h:70-98).
template <typename R>
struct NoFeedback {
    __device__ __forceinline__ void prepare(R &tu) const { tu = row_traits<R>::zero(); }   // :430-432
    __device__ __forceinline__ float bias() const { return 0.0f; }                        // :433-435
    __device__ __forceinline__ void update(const DevParams &, float, const R &, bool) {}  // :442-444
};
template <typename R>
struct SvdppRegsT {   // SVDPPFeature members (apex_svd_base.h:486-488) held in registers
    R tmp_fb, old_fb;
    float norm, tmp_bias, old_bias;
    __device__ __forceinline__ void prepare(R &tu) const { tu = tmp_fb; }                 // :506-508
    __device__ __forceinline__ float bias() const { return tmp_bias; }                    // :509-511
    __device__ __forceinline__ void update(const DevParams &P, float err, const R &ti, bool ub) {   // update_svdpp (:512-520)
        const float lr2 = P.lr * P.scale_lr_ufeedback;
        axpy4(tmp_fb, ti, lr2 * err * norm);
        scale4(tmp_fb, 1.0f - lr2 * P.wd_ufeedback);
        if (ub) {
            tmp_bias = tmp_bias + lr2 * err * norm;
            tmp_bias = tmp_bias * (1.0f - lr2 * P.wd_ufeedback_bias);
        }
    }
};
using SvdppRegs = SvdppRegsT<float4>;

// pred() (:445-454): fills tmp_u / tmp_i, returns the score before the link function (double)
template <int LPI, typename R, typename PP>
__device__ __forceinline__ double instance_score(const DevParams &P, int ng, int nu, int ni, const unsigned *idx,
                                                 const float *val, int L, const PP &pp, R &tu, R &ti) {
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
        bs += (double)pp.bias();
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
    pp.prepare(tu);
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
template <int LPI, typename R, typename PP>
__device__ __forceinline__ void instance_update(const DevParams &P, float label, int ng, int nu, int ni,
                                                const unsigned *idx, const float *val, int L, PP &pp, unsigned counter) {
    const unsigned *ig = idx, *iu = idx + ng, *ii = idx + ng + nu;
    const float *vg = val, *vu = val + ng, *vi = val + ng + nu;
    if (P.reg_method >= 4 || P.reg_global >= 4) instance_regularize<LPI, R>(P, ng, nu, ni, idx, L, false, counter);
    R tu, ti;
    const double sum = instance_score<LPI, R, PP>(P, ng, nu, ni, idx, val, L, pp, tu, ti);
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
    pp.update(P, err, ti, ub);   // update_svdpp hook
    // ---- sample_counter++ ; regularize(feature, true)
    instance_regularize<LPI, R>(P, ng, nu, ni, idx, L, true, counter + 1u);
}

}  // namespace svdf
#endif
