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

// One candidate's bias_ifactors + <tmp_ufactor, tmp_ifactors> (:754-765) computed by ONE lane from the chunk-major matrix
// (see k_rank_transpose): the reference's dot product (apex_tensor_sse.h:289-317) is four serial chains over the 4-float
// chunks, then (a0 + a2) + (a1 + a3), then the scalar tail in index order.  Same operations in the same order, bit for bit.
template <int UNROLL>
__device__ __forceinline__ float rank_lane_score(int k, long cap, const float4 *p4, const float4 *q, float bias) {
    const int nfull = k >> 2, ntail = k & 3;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int j = 0;
    for (; j + UNROLL <= nfull; j += UNROLL) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = q[(size_t)(j + u) * cap];   // plain loads: every section re-reads the matrix and it fits the 256 MB Infinity Cache
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const float4 p = p4[j + u];
            a0 = a0 + p.x * v[u].x; a1 = a1 + p.y * v[u].y; a2 = a2 + p.z * v[u].z; a3 = a3 + p.w * v[u].w;
        }
    }
    for (; j < nfull; j++) {
        const float4 v = q[(size_t)j * cap], p = p4[j];
        a0 = a0 + p.x * v.x; a1 = a1 + p.y * v.y; a2 = a2 + p.z * v.z; a3 = a3 + p.w * v.w;
    }
    float sum = (a0 + a2) + (a1 + a3);   // sum_all: movehl add, then shuffle add_ss
    if (ntail) {                         // scalar tail, in index order
        const float4 v = q[(size_t)nfull * cap], p = p4[nfull];
        sum = sum + p.x * v.x;
        if (ntail > 1) sum = sum + p.y * v.y;
        if (ntail > 2) sum = sum + p.z * v.z;
    }
    return bias + sum;
}

// proc_user (:709-728): tmp_ufactor = (tmp_ufeedback | 0) + sum of the user's rows (and side-table children).
// The same single-wave launch opens the section on the device: everything the host staged for it arrives in ONE pinned
// upload (RankSection describes the words), so the tags of the previous section are cleared and this section's
// POS_SAMPLE / BAN_SAMPLE tags (proc_tag, :729-738) applied here, and the position counters / NaN flag zeroed.
template <int LPI, typename R>
__global__ __launch_bounds__(64) void k_rank_user(const DevParams P, const unsigned *stage, const RankSection S, const float *fb_in, float *tu_out,
                                                  signed char *tag, int *cnt, unsigned *flag, long cap, const float4 *ifT, const float *ibias,
                                                  float *pos_score) {
    extern __shared__ float tus[];   // the section's tmp_ufactor, for the positives' scores below
    const int lane = threadIdx.x & 63;
    const unsigned *prev = stage + 2 * S.nu + S.npos, *nidx = prev + S.nprev, *ntag = nidx + S.nnew;
    for (int j = lane; j < S.nprev; j += 64) tag[prev[j]] = 0;
    __syncthreads();
    for (int j = lane; j < S.nnew; j += 64) tag[nidx[j]] = (signed char)(int)ntag[j];
    for (int j = lane; j < 2 * S.npos; j += 64) cnt[j] = 0;
    if (lane == 0) *flag = 0u;
    using io = row_io<LPI, R>;
    if (lane < LPI) {
        const unsigned *uidx = stage;
        const float *uval = reinterpret_cast<const float *>(stage + S.nu);
        R tu = fb_in ? io::load(fb_in, 0, P.pitch, lane, P.k) : row_traits<R>::zero();
        for (int j = 0; j < S.nu; j++) {
            const unsigned uid = uidx[j];
            axpy4(tu, io::load(P.W, P.user_off + uid, P.pitch, lane, P.k), uval[j]);
            if (uid < P.feat_user.num_row)
                for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                    axpy4(tu, io::load(P.W, P.user_off + P.feat_user.index[c], P.pitch, lane, P.k), P.feat_user.value[c]);
        }
        io::store(tu_out, 0, P.pitch, lane, P.k, tu);
        if (pos_score) io::store(tus, 0, P.pitch, lane, P.k, tu);
    }
    if (!pos_score) return;
    // No special sample in this section: a positive's final score is 0 + (bias + dot), exactly what k_rank_score will store for
    // it.  Computing the few positives here lets the scoring pass count the rank positions (:777-784) while it streams.
    __syncthreads();
    const int *pos = reinterpret_cast<const int *>(stage + 2 * S.nu);
    for (int j = lane; j < S.npos; j += 64)
        pos_score[j] = 0.0f + rank_lane_score<1>(P.k, cap, reinterpret_cast<const float4 *>(tus), ifT + pos[j], ibias[pos[j]]);
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

// The prepared candidate matrix is OURS to lay out (tmp_ifactors, :676-700, is private to the ranker), and the scoring pass
// is the only reader: store it chunk-major -- chunk j (4 floats, the reference's SSE register) of candidate i at
// ifT[j * cap + i] -- so that ONE LANE owns one candidate.  The reference's dot product (apex_tensor_sse.h:289-317) is four
// serial chains over the chunks; across lanes (group_dot) that is 32 dependent DPP steps per candidate and the pass was
// VALU-latency bound at 2.0 TB/s; inside one lane it is 2 VALU ops per element, every load of a wave is 1 KB contiguous,
// and the user's factor is wave-uniform (scalar loads).  Same operations in the same order, bit for bit.
__global__ __launch_bounds__(256) void k_rank_transpose(int k, int pitch, long first, long n, long cap, const float *ifactors, float4 *ifT) {
    const int nchunk = pitch >> 2;
    const long total = (n - first) * nchunk;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += stride) {
        const long r = first + t / nchunk;
        const int j = (int)(t % nchunk);
        const float *src = ifactors + (size_t)r * pitch + 4 * j;
        float4 v;
        v.x = 4 * j + 0 < k ? src[0] : 0.0f;
        v.y = 4 * j + 1 < k ? src[1] : 0.0f;
        v.z = 4 * j + 2 < k ? src[2] : 0.0f;
        v.w = 4 * j + 3 < k ? src[3] : 0.0f;
        ifT[(size_t)j * cap + r] = v;
    }
}

// proc_rank (:754-765): item_score[i] += bias_ifactors[i] + <tmp_ufactor, tmp_ifactors[i]> for every candidate that is not
// banned (BAN_SAMPLE candidates are not ranked at all).  One lane per candidate.
//   MODE 0: scores only (a special sample wrote scores this section; k_rank_positions / k_rank_keys follow)
//   MODE 1: + rank positions of the positives against pos_score[] (from k_rank_user): wave ballots -> LDS -> one global
//           atomic per workgroup and counter.  greater = ranked candidates with a strictly higher score; ties = OTHER ranked
//           candidates with exactly the positive's score, or NaN on either side (the host's std::sort decides those)
//   MODE 2: + top_k sort keys: descending score == ascending key (order-preserving bit transform of the float, inverted);
//           banned candidates and NaN scores get the largest keys (NaN additionally raises *flag: the host decides)
__device__ __forceinline__ unsigned rank_sort_key(float s, bool banned, unsigned *flag) {
    unsigned u = __float_as_uint(s);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending in the float order
    unsigned key = ~u;                                 // descending score == ascending key
    if (s == 0.0f) key = ~0x80000000u;                 // -0 and +0 compare equal in the reference's comparator
    if (s != s) { atomicOr(flag, 1u); key = 0xFFFFFFFEu; }
    if (banned) key = 0xFFFFFFFFu;
    return key;
}
template <int UNROLL, int MODE>
__global__ __launch_bounds__(256) void k_rank_score(int k, long n, long cap, const float *__restrict__ tu, const float4 *__restrict__ ifT,
                                                    const float *__restrict__ ibias, const signed char *__restrict__ tag, float *item_score, int fresh,
                                                    const int *pos_item, const float *pos_score, int npos, int *greater, int *ties,
                                                    unsigned *keys, unsigned *vals, unsigned *flag) {
    extern __shared__ int cnt[];   // MODE 1: 2 * npos counters
    if (MODE == 1) {
        for (int j = threadIdx.x; j < 2 * npos; j += blockDim.x) cnt[j] = 0;
        __syncthreads();
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool on = i < n && tag[i < n ? i : 0] >= 0;
    float s = 0.0f;
    if (on) {
        s = (fresh ? 0.0f : item_score[i]) + rank_lane_score<UNROLL>(k, cap, reinterpret_cast<const float4 *>(tu), ifT + i, ibias[i]);
        item_score[i] = s;   // fresh: no special sample wrote a score this section, item_score is the 0 of proc_user (:726)
    }
    if (MODE == 1) {
        for (int j = 0; j < npos; j++) {
            const float ps = pos_score[j];
            const bool gt = on && s > ps;
            const bool tie = on && !gt && ((s == ps && pos_item[j] != (int)i) || !(s <= ps));
            const int ngt = __popcll(__ballot(gt)), ntie = __popcll(__ballot(tie));
            if ((threadIdx.x & 63) == 0) {
                if (ngt) atomicAdd(&cnt[j], ngt);
                if (ntie) atomicAdd(&cnt[npos + j], ntie);
            }
        }
        __syncthreads();
        for (int j = threadIdx.x; j < npos; j += blockDim.x) {
            if (cnt[j]) atomicAdd(&greater[j], cnt[j]);
            if (cnt[npos + j]) atomicAdd(&ties[j], cnt[npos + j]);
        }
    }
    if (MODE == 2 && i < n) {
        keys[i] = rank_sort_key(s, !on, flag);
        vals[i] = (unsigned)i;
    }
}

// rank positions of the positive samples (:777-784): position = number of ranked candidates with a strictly higher score;
// ties[p] counts OTHER ranked candidates with exactly the positive's score (their relative order is the host sort's business).
// Counts are accumulated per workgroup in LDS and added to the global counters once per workgroup: one same-address
// device-scope atomic per candidate and positive made this kernel 2.8 ms for 100 K candidates, 125x the scoring pass.
__global__ __launch_bounds__(256) void k_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos,
                                                        int *greater, int *ties) {
    extern __shared__ float ps[];   // npos scores of the positives, then 2*npos int counters
    int *cnt = reinterpret_cast<int *>(ps + npos);
    for (int j = threadIdx.x; j < npos; j += blockDim.x) ps[j] = score[pos_item[j]];
    for (int j = threadIdx.x; j < 2 * npos; j += blockDim.x) cnt[j] = 0;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    const long rounds = (n + stride - 1) / stride;   // whole waves stay in the loop: the counts are wave ballots
    for (long r = 0; r < rounds; r++) {
        const long i = r * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        const bool on = i < n && tag[i < n ? i : 0] >= 0;
        const float s = on ? score[i] : 0.0f;
        for (int j = 0; j < npos; j++) {
            const bool gt = on && s > ps[j];
            // equal scores, and NaN on either side: leave the decision to the host sort
            const bool tie = on && !gt && ((s == ps[j] && pos_item[j] != (int)i) || !(s <= ps[j]));
            const int ngt = __popcll(__ballot(gt)), ntie = __popcll(__ballot(tie));
            if ((threadIdx.x & 63) == 0) {
                if (ngt) atomicAdd(&cnt[j], ngt);
                if (ntie) atomicAdd(&cnt[npos + j], ntie);
            }
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < npos; j += blockDim.x) {
        if (cnt[j]) atomicAdd(&greater[j], cnt[j]);
        if (cnt[npos + j]) atomicAdd(&ties[j], cnt[npos + j]);
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
void launch_rank_user(const DevParams &P, const unsigned *stage, const RankSection &S, const float *fb_in, float *tu_out, signed char *tag, int *cnt,
                      unsigned *flag, long cap, const float *ifT, const float *ibias, float *pos_score, hipStream_t st) {
    const size_t lds = pos_score ? ((size_t)P.pitch + 4) * sizeof(float) : 0;
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_user<LPI, R>), dim3(1), dim3(64), lds, st, P, stage, S, fb_in, tu_out, tag, cnt, flag, cap,
                                              reinterpret_cast<const float4 *>(ifT), ibias, pos_score));
}
void launch_rank_spec(const DevParams &P, const DevCSR &D, long n, const int *spec_idx, const float *tu, float *item_score, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_spec<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, n, spec_idx, tu, item_score));
}
void launch_rank_transpose(const DevParams &P, long first, long n, long cap, const float *ifactors, float *ifT, hipStream_t st) {
    if (n <= first) return;
    long grid = ((n - first) * (P.pitch >> 2) + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_rank_transpose, dim3((int)grid), dim3(256), 0, st, P.k, P.pitch, first, n, cap, ifactors, reinterpret_cast<float4 *>(ifT));
}
void launch_rank_score(const DevParams &P, long n, long cap, const float *tu, const float *ifT, const float *ibias, const signed char *tag,
                       float *item_score, int fresh, const RankFused &F, hipStream_t st) {
    if (n <= 0) return;
    const unsigned grid = (unsigned)((n + 255) / 256);
    const float4 *q = reinterpret_cast<const float4 *>(ifT);
    if (F.mode == 1)
        hipLaunchKernelGGL((k_rank_score<8, 1>), dim3(grid), dim3(256), (size_t)2 * F.npos * sizeof(int), st, P.k, n, cap, tu, q, ibias, tag, item_score, fresh,
                           F.pos_item, F.pos_score, F.npos, F.greater, F.ties, nullptr, nullptr, nullptr);
    else if (F.mode == 2)
        hipLaunchKernelGGL((k_rank_score<8, 2>), dim3(grid), dim3(256), 0, st, P.k, n, cap, tu, q, ibias, tag, item_score, fresh, nullptr, nullptr, 0, nullptr,
                           nullptr, F.keys, F.vals, F.flag);
    else
        hipLaunchKernelGGL((k_rank_score<8, 0>), dim3(grid), dim3(256), 0, st, P.k, n, cap, tu, q, ibias, tag, item_score, fresh, nullptr, nullptr, 0, nullptr,
                           nullptr, nullptr, nullptr, nullptr);
}
void launch_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos, int *greater, int *ties, hipStream_t st) {
    if (n <= 0 || npos <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 128) grid = 128;
    hipLaunchKernelGGL(k_rank_positions, dim3((int)grid), dim3(256), (size_t)npos * (sizeof(float) + 2 * sizeof(int)), st, n, score, tag, pos_item, npos, greater, ties);
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
