// svdf_k_imfb.hip -- the multi-level implicit-feedback variant solver (extend_type 2): SVDPPMultiIMFB,
// solvers/multi-imfb/apex_multi_imfb.h:30-210, on the lane-group layout of the general kernels.
//
// The reference keeps a STACK of implicit-feedback levels: a DEFAULT / START block pushes a level (prepare_ufeedback over the
// block's feedback list, :121-137,161-171), every row sees the SUM of the open levels (prepare_svdpp :70-79, get_bias_svdpp
// :80-86) and updates each of them (update_svdpp :87-98, disabled or empty levels skipped), a DEFAULT / END block pops the top
// level and scatters its change through that block's own list (update_ufeedback :138-153) unless the level is disabled.
// Here a scheduling unit is a run of consecutive blocks; one lane group walks it with the stack in registers: DEPTH = IMFB_DEPTH (4)
// levels in the usual build of the kernel, IMFB_DEPTH_MAX (16) in a second one the engine switches to once the data nests deeper (the
// compiler spills what does not fit; the reference's stack is an unbounded std::vector, apex_multi_imfb.h:41-58, 161-171).  A unit that starts or ends with open levels loads / saves the whole stack
// from the trainer's state slot, the way the reference keeps it in the trainer object between calls.
#include "svdf_instance.h"

namespace svdf {

template <typename R>
struct ImfbLevel { R tmp, old; float norm, tmp_bias, old_bias; int nfb; };

template <typename R, int DEPTH>
struct ImfbRegsT {
    ImfbLevel<R> lv[DEPTH];
    int top;
    unsigned disable;   // bit l: ufeedback_disable_level = l
    __device__ __forceinline__ void prepare(R &tu) const {   // :70-79: copy of level 0, then += level 1.. in order (plain adds)
        tu = top > 0 ? lv[0].tmp : row_traits<R>::zero();
#pragma unroll
        for (int l = 1; l < DEPTH; l++) if (l < top) add_rows(tu, lv[l].tmp);
    }
    __device__ __forceinline__ float bias() const {          // :80-86: float sum starting at 0.0f
        float s = 0.0f;
#pragma unroll
        for (int l = 0; l < DEPTH; l++) if (l < top) s = s + lv[l].tmp_bias;
        return s;
    }
    __device__ __forceinline__ void update(const DevParams &P, float err, const R &ti, bool ub) {   // :87-98
        const float lr2 = P.lr * P.scale_lr_ufeedback;
#pragma unroll
        for (int l = 0; l < DEPTH; l++) {
            if (l >= top || ((disable >> l) & 1u) || lv[l].nfb == 0) continue;
            axpy4(lv[l].tmp, ti, lr2 * err * lv[l].norm);
            scale4(lv[l].tmp, 1.0f - lr2 * P.wd_ufeedback);
            if (ub) {
                lv[l].tmp_bias = lv[l].tmp_bias + lr2 * err * lv[l].norm;
                lv[l].tmp_bias = lv[l].tmp_bias * (1.0f - lr2 * P.wd_ufeedback_bias);
            }
        }
    }
};

// state slot layout (floats): [0] top, then per level 2*pitch row floats + norm, tmp_bias, old_bias, nfb
__device__ __forceinline__ size_t imfb_level_off(const DevParams &P, int l) { return 4 + (size_t)l * (2 * (size_t)P.pitch + 4); }
template <int LPI, typename R, int DEPTH>
__device__ __forceinline__ void imfb_load(const DevParams &P, ImfbRegsT<R, DEPTH> &pp, int L) {
    const float *st = P.svdpp_state;
    pp.top = __float_as_int(st[0]);
#pragma unroll
    for (int l = 0; l < DEPTH; l++) {
        if (l >= pp.top) continue;
        const float *b = st + imfb_level_off(P, l);
        pp.lv[l].tmp = row_io<LPI, R>::load(b, 0, P.pitch, L, P.k);
        pp.lv[l].old = row_io<LPI, R>::load(b, 1, P.pitch, L, P.k);
        pp.lv[l].norm = b[2 * P.pitch]; pp.lv[l].tmp_bias = b[2 * P.pitch + 1]; pp.lv[l].old_bias = b[2 * P.pitch + 2];
        pp.lv[l].nfb = __float_as_int(b[2 * P.pitch + 3]);
    }
}
template <int LPI, typename R, int DEPTH>
__device__ __forceinline__ void imfb_save(const DevParams &P, const ImfbRegsT<R, DEPTH> &pp, int L) {
    float *st = P.svdpp_state;
    if (L == 0) st[0] = __int_as_float(pp.top);
#pragma unroll
    for (int l = 0; l < DEPTH; l++) {
        if (l >= pp.top) continue;
        float *b = st + imfb_level_off(P, l);
        row_io<LPI, R>::store(b, 0, P.pitch, L, P.k, pp.lv[l].tmp);
        row_io<LPI, R>::store(b, 1, P.pitch, L, P.k, pp.lv[l].old);
        if (L == 0) { b[2 * P.pitch] = pp.lv[l].norm; b[2 * P.pitch + 1] = pp.lv[l].tmp_bias; b[2 * P.pitch + 2] = pp.lv[l].old_bias;
                      b[2 * P.pitch + 3] = __int_as_float(pp.lv[l].nfb); }
    }
}
// push_ufeedback + prepare_ufeedback (:121-137, 161-171) into level `top`
template <int LPI, typename R, int DEPTH>
__device__ __forceinline__ void imfb_push(const DevParams &P, ImfbRegsT<R, DEPTH> &pp, const unsigned *fidx, const float *fval, int nfb, int L) {
    R tmp = row_traits<R>::zero();
    float norm = 0.0f, bias = 0.0f;
    for (int j = 0; j < nfb; j++) {
        const unsigned row = P.fb_off + fidx[j];
        const float v = fval[j];
        axpy4(tmp, row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k), v);
        norm = norm + v * v;
        if (P.no_user_bias == 0) bias = bias + P.bias[row] * v;
    }
#pragma unroll
    for (int l = 0; l < DEPTH; l++)
        if (l == pp.top) { pp.lv[l].tmp = tmp; pp.lv[l].old = tmp; pp.lv[l].norm = norm; pp.lv[l].tmp_bias = bias; pp.lv[l].old_bias = bias; pp.lv[l].nfb = nfb; }
    pp.top++;
}
// --top; update_ufeedback (:138-153) unless the level is disabled (:186-190)
template <int LPI, typename R, int DEPTH>
__device__ __forceinline__ void imfb_pop(const DevParams &P, ImfbRegsT<R, DEPTH> &pp, const unsigned *fidx, const float *fval, int nfb, int L, bool scatter) {
    pp.top--;
    if (!scatter || ((pp.disable >> pp.top) & 1u) || nfb == 0) return;
    R d = row_traits<R>::zero(), old = row_traits<R>::zero();
    float tb = 0.0f, ob = 0.0f, norm = 1.0f;
#pragma unroll
    for (int l = 0; l < DEPTH; l++)
        if (l == pp.top) { d = pp.lv[l].tmp; old = pp.lv[l].old; tb = pp.lv[l].tmp_bias; ob = pp.lv[l].old_bias; norm = pp.lv[l].norm; }
    sub4(d, old);
    float db = tb - ob;
    const float inv = 1.0f / norm;
    scale4(d, inv);
    db = db * inv;
    for (int j = 0; j < nfb; j++) {
        const unsigned row = P.fb_off + fidx[j];
        const float v = fval[j];
        R w = row_io<LPI, R>::load(P.W, row, P.pitch, L, P.k);
        axpy4(w, d, v);
        row_io<LPI, R>::store(P.W, row, P.pitch, L, P.k, w);
        if (P.no_user_bias == 0) { float b = P.bias[row]; b = b + db * v; P.bias[row] = b; }
    }
}

// one conflict-free batch of units; a unit = blocks [u.fb_begin, u.fb_end) of blks[] (DevUnit reused: the fb_* fields hold the
// block range, row_* the rows of all its blocks for the counter of the lazy modes).  PREDICT: scores only, no scatter.
template <int LPI, typename R, bool PREDICT, int DEPTH>
__global__ __launch_bounds__(256) void k_imfb(const DevParams P, const DevCSR D, const DevUnit *units, const DevBlk *blks, const unsigned *fb_index,
                                              const float *fb_value, const int *order, long begin, long end, unsigned counter_base, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = begin + gidx; s < end; s += stride) {
        const DevUnit u = units[order ? order[s] : (int)s];
        ImfbRegsT<R, DEPTH> pp;
        pp.top = 0;
        pp.disable = P.imfb_disable;
#pragma unroll
        for (int l = 0; l < DEPTH; l++) { pp.lv[l].tmp = row_traits<R>::zero(); pp.lv[l].old = row_traits<R>::zero(); pp.lv[l].norm = 0.0f; pp.lv[l].tmp_bias = 0.0f; pp.lv[l].old_bias = 0.0f; pp.lv[l].nfb = 0; }
        if (u.flags & UNIT_LOAD) imfb_load<LPI, R, DEPTH>(P, pp, L);
        for (int b = u.fb_begin; b < u.fb_end; b++) {
            const DevBlk k = blks[b];
            const bool starts = k.tag == TAG_DEFAULT || k.tag == TAG_START, ends = k.tag == TAG_DEFAULT || k.tag == TAG_END;
            if (starts) imfb_push<LPI, R, DEPTH>(P, pp, fb_index + k.fb_begin, fb_value + k.fb_begin, k.fb_end - k.fb_begin, L);
            for (int r = k.row_begin; r < k.row_end; r++) {
                const int p0 = D.row_ptr[3 * (long)r], p1 = D.row_ptr[3 * (long)r + 1], p2 = D.row_ptr[3 * (long)r + 2], p3 = D.row_ptr[3 * (long)r + 3];
                if (PREDICT) {
                    R tu, ti;
                    const double sum = instance_score<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, pp, tu, ti);
                    if (L == 0) out[r] = map_active((float)sum, P.active_type);
                } else {
                    instance_update<LPI, R>(P, D.row_label[r], p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, pp,
                                            counter_base + (unsigned)r);
                }
            }
            if (ends) imfb_pop<LPI, R, DEPTH>(P, pp, fb_index + k.sc_begin, fb_value + k.sc_begin, k.sc_end - k.sc_begin, L, !PREDICT);
        }
        if (u.flags & UNIT_SAVE) imfb_save<LPI, R, DEPTH>(P, pp, L);
    }
}

void launch_imfb(const DevParams &P, const DevCSR &D, const DevUnit *units, const DevBlk *blks, const unsigned *fb_index, const float *fb_value,
                 const int *order, long begin, long end, unsigned counter_base, float *predict_out, hipStream_t st) {
    if (end <= begin) return;
    const int lpi = lanes_per_instance(P.k);
    int grid, block;
    launch_shape(end - begin, lpi, 256 * 8, P.small_blocks != 0 && predict_out == nullptr, grid, block);
    if (P.imfb_deep) {   // the data nests deeper than IMFB_DEPTH levels: the build with IMFB_DEPTH_MAX levels (spills, correct)
        if (predict_out) {
            SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_imfb<LPI, R, true, IMFB_DEPTH_MAX>), dim3(grid), dim3(block), 0, st, P, D, units, blks, fb_index, fb_value, order, begin, end, counter_base, predict_out));
        } else {
            SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_imfb<LPI, R, false, IMFB_DEPTH_MAX>), dim3(grid), dim3(block), 0, st, P, D, units, blks, fb_index, fb_value, order, begin, end, counter_base, predict_out));
        }
        return;
    }
    if (predict_out) {
        SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_imfb<LPI, R, true, IMFB_DEPTH>), dim3(grid), dim3(block), 0, st, P, D, units, blks, fb_index, fb_value, order, begin, end, counter_base, predict_out));
    } else {
        SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_imfb<LPI, R, false, IMFB_DEPTH>), dim3(grid), dim3(block), 0, st, P, D, units, blks, fb_index, fb_value, order, begin, end, counter_base, predict_out));
    }
}

}  // namespace svdf
