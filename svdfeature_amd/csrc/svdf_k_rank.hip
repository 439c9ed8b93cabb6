// svdf_k_rank.hip -- SVDFeatureRanker (solvers/base-solver/apex_svd_base.h:597-813) and the evaluator's squared-error sum
// (svd_feature_infer.cpp:38-56) on the device.
//
// The ranker scores every candidate item of a prepared item set against one user: score_i = spec_i + (bias_i + <p_u, q_i>)
// (:754-765).  That is a dense matrix-vector product, but its RESULT is index work -- the sorted order of the scores -- and
// near-ties flip with the rounding order, so the dot product keeps the reference's 4-chain SSE order (group_dot) instead of an
// MFMA tile whose accumulation order is the hardware's.  The item matrix is streamed once per user: HBM bound, one lane group
// per candidate, rows of the prepared matrix are contiguous.
#include "svdf_instance.h"

namespace svdf {

// prepare_ifactor (:676-700): factor part and FLOAT bias of one candidate / special sample
template <int LPI, typename R>
__device__ __forceinline__ void rank_prepare_item(const DevParams &P, int ng, int nu, int ni, const unsigned *idx, const float *val, int L,
                                                  R &f, float &bias) {
    using io = row_io<LPI, R>;
    const unsigned *ig = idx, *ii = idx + ng + nu;
    const float *vg = val, *vi = val + ng + nu;
    f = row_traits<R>::zero();
    bias = 0.0f;
    for (int j = 0; j < ni; j++) {
        const unsigned iid = ii[j];
        const float ival = vi[j];
        axpy4(f, io::load(P.W, P.item_off + iid, P.pitch, L, P.k), ival);
        bias = bias + P.bias[P.item_off + iid] * ival;
        if (iid < P.feat_item.num_row)
            for (unsigned c = P.feat_item.row_ptr[iid]; c < P.feat_item.row_ptr[iid + 1]; c++) {
                const unsigned cid = P.feat_item.index[c];
                const float cv = P.feat_item.value[c];
                axpy4(f, io::load(P.W, P.item_off + cid, P.pitch, L, P.k), (float)((double)cv * (double)ival));
                bias = bias + P.bias[P.item_off + cid] * cv * ival;
            }
    }
    for (int j = 0; j < ng; j++) bias = bias + vg[j] * P.g_bias[gpos(P, ig[j])];
}

// proc_item (:702-707): candidate `r` of the item set -> row r of tmp_ifactors, bias_ifactors[r]
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_rank_items(const DevParams P, const DevCSR D, long first, long n, float *ifactors, float *ibias) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long r = first + gidx; r < n; r += stride) {
        const int p0 = D.row_ptr[3 * r], p1 = D.row_ptr[3 * r + 1], p2 = D.row_ptr[3 * r + 2], p3 = D.row_ptr[3 * r + 3];
        R f;
        float b;
        rank_prepare_item<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, f, b);
        row_io<LPI, R>::store(ifactors, (size_t)r, P.pitch, L, P.k, f);
        if (L == 0) ibias[r] = b;
    }
}

// process(block) prologue (:796-806): tmp_ufeedback = sum W_ufeedback[fid] * val
template <int LPI, typename R>
__global__ __launch_bounds__(64) void k_rank_feedback(const DevParams P, const unsigned *fidx, const float *fval, int nfb, float *fb_out) {
    const int lane = threadIdx.x & 63;
    if (lane >= LPI) return;
    R f = row_traits<R>::zero();
    for (int j = 0; j < nfb; j++) axpy4(f, row_io<LPI, R>::load(P.W, P.fb_off + fidx[j], P.pitch, lane, P.k), fval[j]);
    row_io<LPI, R>::store(fb_out, 0, P.pitch, lane, P.k, f);
}

// proc_user (:709-728): tmp_ufactor = (tmp_ufeedback | 0) + sum of the user's rows (and side-table children)
template <int LPI, typename R>
__global__ __launch_bounds__(64) void k_rank_user(const DevParams P, const unsigned *uidx, const float *uval, int nu, const float *fb_in, float *tu_out) {
    const int lane = threadIdx.x & 63;
    if (lane >= LPI) return;
    using io = row_io<LPI, R>;
    R tu = fb_in ? io::load(fb_in, 0, P.pitch, lane, P.k) : row_traits<R>::zero();
    for (int j = 0; j < nu; j++) {
        const unsigned uid = uidx[j];
        axpy4(tu, io::load(P.W, P.user_off + uid, P.pitch, lane, P.k), uval[j]);
        if (uid < P.feat_user.num_row)
            for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                axpy4(tu, io::load(P.W, P.user_off + P.feat_user.index[c], P.pitch, lane, P.k), P.feat_user.value[c]);
    }
    io::store(tu_out, 0, P.pitch, lane, P.k, tu);
}

// proc_spec (:739-747) for the special samples of one user section (at most one per candidate, the last one given):
// item_score[idx] = bias + <tmp_ufactor, prepared factor>
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_rank_spec(const DevParams P, const DevCSR D, long n, const int *spec_idx, const float *tu_in, float *item_score) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    const R tu = row_io<LPI, R>::load(tu_in, 0, P.pitch, L, P.k);
    for (long r = gidx; r < n; r += stride) {
        const int p0 = D.row_ptr[3 * r], p1 = D.row_ptr[3 * r + 1], p2 = D.row_ptr[3 * r + 2], p3 = D.row_ptr[3 * r + 3];
        R f;
        float b;
        rank_prepare_item<LPI, R>(P, p1 - p0, p2 - p1, p3 - p2, D.feat_index + p0, D.feat_value + p0, L, f, b);
        const float s = b + group_dot<LPI>(tu, f, L, P.k);
        if (L == 0) item_score[spec_idx[r]] = s;
    }
}

// proc_rank (:754-765): item_score[i] += bias_ifactors[i] + <tmp_ufactor, tmp_ifactors[i]> for every candidate that is not
// banned; banned candidates get -inf keys later (they are not ranked at all)
template <int LPI, typename R>
__global__ __launch_bounds__(256) void k_rank_score(const DevParams P, long n, const float *tu_in, const float *ifactors, const float *ibias,
                                                    const signed char *tag, float *item_score) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    const R tu = row_io<LPI, R>::load(tu_in, 0, P.pitch, L, P.k);
    for (long i = gidx; i < n; i += stride) {
        if (tag[i] < 0) continue;   // BAN_SAMPLE
        const R q = row_io<LPI, R>::load(ifactors, (size_t)i, P.pitch, L, P.k);
        const float t = ibias[i] + group_dot<LPI>(tu, q, L, P.k);
        if (L == 0) item_score[i] = item_score[i] + t;
    }
}

// rank positions of the positive samples (:777-784): position = number of ranked candidates with a strictly higher score;
// ties[p] counts OTHER ranked candidates with exactly the positive's score (their relative order is the host sort's business)
__global__ __launch_bounds__(256) void k_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos,
                                                        int *greater, int *ties) {
    extern __shared__ float ps[];   // scores of the positives
    for (int j = threadIdx.x; j < npos; j += blockDim.x) ps[j] = score[pos_item[j]];
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        if (tag[i] < 0) continue;
        const float s = score[i];
        for (int j = 0; j < npos; j++) {
            if (s > ps[j]) atomicAdd(&greater[j], 1);
            else if (s == ps[j] && pos_item[j] != (int)i) atomicAdd(&ties[j], 1);
            else if (!(s <= ps[j])) atomicAdd(&ties[j], 1);   // NaN on either side: leave the decision to the host sort
        }
    }
}

// squared error of predictions (svd_feature_infer.cpp:43-47: diff = (pred - label) * scale in fp32, diff*diff in fp64): one
// fp64 partial sum per workgroup, fixed tree order
__global__ __launch_bounds__(256) void k_sqerr_partials(const float *pred, const float *label, long n, float scale, double *partials) {
    __shared__ double sh[256];
    double acc = 0.0;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const double d = (double)((pred[i] - label[i]) * scale);
        acc += d * d;
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) sh[threadIdx.x] += sh[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = sh[0];
}

void launch_rank_items(const DevParams &P, const DevCSR &D, long first, long n, float *ifactors, float *ibias, hipStream_t st) {
    if (n <= first) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n - first, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_items<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, first, n, ifactors, ibias));
}
void launch_rank_feedback(const DevParams &P, const unsigned *fidx, const float *fval, int nfb, float *fb_out, hipStream_t st) {
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_feedback<LPI, R>), dim3(1), dim3(64), 0, st, P, fidx, fval, nfb, fb_out));
}
void launch_rank_user(const DevParams &P, const unsigned *uidx, const float *uval, int nu, const float *fb_in, float *tu_out, hipStream_t st) {
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_user<LPI, R>), dim3(1), dim3(64), 0, st, P, uidx, uval, nu, fb_in, tu_out));
}
void launch_rank_spec(const DevParams &P, const DevCSR &D, long n, const int *spec_idx, const float *tu, float *item_score, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_spec<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, n, spec_idx, tu, item_score));
}
void launch_rank_score(const DevParams &P, long n, const float *tu, const float *ifactors, const float *ibias, const signed char *tag,
                       float *item_score, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_score<LPI, R>), dim3(grid), dim3(256), 0, st, P, n, tu, ifactors, ibias, tag, item_score));
}
void launch_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos, int *greater, int *ties, hipStream_t st) {
    if (n <= 0 || npos <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 2048) grid = 2048;
    hipLaunchKernelGGL(k_rank_positions, dim3((int)grid), dim3(256), (size_t)npos * sizeof(float), st, n, score, tag, pos_item, npos, greater, ties);
}
int sqerr_partials_grid(long n) {
    long grid = (n + 255) / 256;
    if (grid > 1024) grid = 1024;
    if (grid < 1) grid = 1;
    return (int)grid;
}
void launch_sqerr_partials(const float *pred, const float *label, long n, float scale, double *partials, hipStream_t st) {
    hipLaunchKernelGGL(k_sqerr_partials, dim3(sqerr_partials_grid(n)), dim3(256), 0, st, pred, label, n, scale, partials);
}

}  // namespace svdf
