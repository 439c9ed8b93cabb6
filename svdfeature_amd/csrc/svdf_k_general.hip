// svdf_k_general.hip -- general sparse kernels (k_general, k_predict) and the lane-group SVD++ kernels (k_svdpp, k_svdpp_predict)
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_instance.h"

namespace svdf {

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
        NoFeedback<R> none;
        instance_update<LPI, R>(P, D.row_label[r], p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, none,
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
        const double sum = instance_score<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, NoFeedback<R>(), tu, ti);
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
                instance_update<LPI, R>(P, D.row_label[r], p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, pp,
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
            const double sum = instance_score<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, pp, tu, ti);
            if (L == 0) out[r] = map_active((float)sum, P.active_type);
        }
        if (u.flags & UNIT_SAVE) svdpp_save_state<LPI, R>(P, pp, L);
    }
}

void launch_general(const DevParams &P, const DevCSR &D, const int *order, long begin, long end, unsigned counter_base, hipStream_t st) {
    if (end <= begin) return;
    const int lpi = lanes_per_instance(P.k);
    int grid, block;
    launch_shape(end - begin, lpi, 256 * 8, P.small_blocks != 0, grid, block);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_general<LPI, R>), dim3(grid), dim3(block), 0, st, P, D, order, begin, end, counter_base));
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
    int grid, block;
    launch_shape(end - begin, lpi, 256 * 8, P.small_blocks != 0, grid, block);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp<LPI, R>), dim3(grid), dim3(block), 0, st, P, D, units, fb_index, fb_value, order, begin, end, counter_base));
}
void launch_svdpp_predict(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                          long nunit, float *out, hipStream_t st) {
    if (nunit <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(nunit, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp_predict<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, units, fb_index, fb_value, nunit, out));
}

}  // namespace svdf
