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
                                                  float *pos_score, unsigned *zero_words, int nzero) {
    extern __shared__ float tus[];   // the section's tmp_ufactor, for the positives' scores below
    const int lane = threadIdx.x & 63;
    const unsigned *prev = stage + 2 * S.nu + S.npos, *nidx = prev + S.nprev, *ntag = nidx + S.nnew;
    for (int j = lane; j < S.nprev; j += 64) tag[prev[j]] = 0;
    __syncthreads();
    for (int j = lane; j < S.nnew; j += 64) tag[nidx[j]] = (signed char)(int)ntag[j];
    for (int j = lane; j < 2 * S.npos; j += 64) cnt[j] = 0;
    for (int j = lane; j < nzero; j += 64) zero_words[j] = 0u;   // top_k: histograms and counters of the radix selection
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
        pos_score[j] = 0.0f + rank_lane_score<8>(P.k, cap, reinterpret_cast<const float4 *>(tus), ifT + pos[j], ibias[pos[j]]);
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

// ---- top_k: the top_k+1 smallest sort keys, in order, by radix selection instead of a full sort (a rocPRIM merge sort of
// 100 K (key, candidate) pairs is 55 us, 5x the scoring pass).  Three histogram passes over the key digits (11 + 11 + 10 bits,
// most significant first) find the K-th smallest key T exactly; one pass appends every key <= T to a small buffer, one
// workgroup sorts that buffer.  Work area (unsigned words): hist[3][2048], then state: 0 b1, 1 K2, 2 b2, 3 K3, 4 appended.
constexpr int RSEL_BINS = 2048;
constexpr int RSEL_CAP = 4096;
__device__ __forceinline__ unsigned rsel_digit(unsigned key, int pass) { return pass == 1 ? key >> 21 : (pass == 2 ? (key >> 10) & 0x7FFu : key & 0x3FFu); }
// bin holding the K-th smallest entry (K >= 1) of a 2048-bin histogram, and K minus the entries in the bins below it; 256 threads
__device__ __forceinline__ void rsel_find(const unsigned *hist, unsigned K, unsigned *sh, unsigned &bin, unsigned &Krem) {
    const int t = threadIdx.x;
    unsigned h[8], mine = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) { h[j] = hist[8 * t + j]; mine += h[j]; }
    unsigned incl = mine;   // inclusive scan over the 256 threads: inside each wave by shuffles, wave totals through LDS
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const unsigned up = __shfl_up(incl, d);
        if ((t & 63) >= d) incl += up;
    }
    if ((t & 63) == 63) sh[t >> 6] = incl;
    __syncthreads();
    unsigned base = 0;
    for (int w = 0; w < (t >> 6); w++) base += sh[w];
    incl += base;
    const unsigned excl = incl - mine;
    if (excl < K && K <= incl) {
        unsigned c = excl;
        for (int j = 0; j < 8; j++) {
            if (K <= c + h[j]) { sh[8] = (unsigned)(8 * t + j); sh[9] = K - c; break; }
            c += h[j];
        }
    }
    __syncthreads();
    bin = sh[8];
    Krem = sh[9];
    __syncthreads();
}
// plain LDS atomics: a wave whose keys share a bin serialises inside the LDS unit, which measured cheaper than aggregating the
// lanes with ballots first (k_rank_score<.,2> 17.5 us with the ballot loop)
__device__ __forceinline__ void rsel_lds_add(unsigned *lh, bool on, unsigned bin) {
    if (on) atomicAdd(&lh[bin], 1u);
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
                                                    unsigned *keys, unsigned *vals, unsigned *flag, unsigned *hist1) {
    extern __shared__ int cnt[];   // MODE 1: 2 * npos counters; MODE 2 with hist1: RSEL_BINS bins of the selection's first pass
    if (MODE == 2 && hist1) {
        for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x) cnt[j] = 0;
        __syncthreads();
    }
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
    if (MODE == 2) {
        const unsigned key = i < n ? rank_sort_key(s, !on, flag) : 0u;
        if (i < n) { keys[i] = key; vals[i] = (unsigned)i; }
        if (hist1) {   // first histogram pass of the radix selection, while the keys are in registers
            unsigned *lh = reinterpret_cast<unsigned *>(cnt);
            rsel_lds_add(lh, i < n, key >> 21);
            __syncthreads();
            for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x)
                if (lh[j]) atomicAdd(&hist1[j], lh[j]);
        }
    }
}

// (blockIdx.y = section of a tile: keys / work areas of consecutive sections lie key_stride / work_stride words apart; one section: y = 0)
template <int PASS>
__global__ __launch_bounds__(256) void k_rsel_hist(long n, const unsigned *keys, unsigned *work, const RselSecs Ks, long key_stride, long work_stride) {
    __shared__ unsigned lh[RSEL_BINS];
    __shared__ unsigned sh[16];
    keys += (size_t)blockIdx.y * key_stride;
    work += (size_t)blockIdx.y * work_stride;
    const unsigned K1 = Ks.K1[blockIdx.y];
    unsigned *hist = work + (PASS - 1) * RSEL_BINS, *state = work + 3 * RSEL_BINS;
    unsigned prefix = 0;
    int shift = 32;   // keys match when key >> shift == prefix
    if (PASS == 2) {
        unsigned b1, K2;
        rsel_find(work, K1, sh, b1, K2);
        if (blockIdx.x == 0 && threadIdx.x == 0) { state[0] = b1; state[1] = K2; }
        prefix = b1; shift = 21;
    } else if (PASS == 3) {
        unsigned b2, K3;
        rsel_find(work + RSEL_BINS, state[1], sh, b2, K3);
        if (blockIdx.x == 0 && threadIdx.x == 0) { state[2] = b2; state[3] = K3; }
        prefix = (state[0] << 11) | b2; shift = 10;
    }
    for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x) lh[j] = 0;
    __syncthreads();
    const long stride = (long)gridDim.x * blockDim.x;
    const long rounds = (n + stride - 1) / stride;
    for (long r = 0; r < rounds; r++) {
        const long i = r * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned key = i < n ? keys[i] : 0u;
        const bool on = i < n && (PASS == 1 || (key >> shift) == prefix);
        rsel_lds_add(lh, on, rsel_digit(key, PASS));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x)
        if (lh[j]) atomicAdd(&hist[j], lh[j]);
}
__global__ __launch_bounds__(256) void k_rsel_compact(long n, const unsigned *keys, const unsigned *vals, unsigned *work, unsigned *ck, unsigned *cv,
                                                      long key_stride, long work_stride) {
    __shared__ unsigned sh[16];
    keys += (size_t)blockIdx.y * key_stride;
    work += (size_t)blockIdx.y * work_stride;
    ck += (size_t)blockIdx.y * RSEL_CAP;
    cv += (size_t)blockIdx.y * RSEL_CAP;
    unsigned *state = work + 3 * RSEL_BINS;
    unsigned b3, K4;
    rsel_find(work + 2 * RSEL_BINS, state[3], sh, b3, K4);
    const unsigned T = (state[0] << 21) | (state[2] << 10) | b3;   // the K1-th smallest key
    const long stride = (long)gridDim.x * blockDim.x;
    const long rounds = (n + stride - 1) / stride;
    for (long r = 0; r < rounds; r++) {
        const long i = r * stride + (long)blockIdx.x * blockDim.x + threadIdx.x;
        const unsigned key = i < n ? keys[i] : 0xFFFFFFFFu;
        const bool on = i < n && key <= T;
        const unsigned long long m = __ballot(on);
        if (m) {
            const int lane = threadIdx.x & 63;
            unsigned base = 0;
            if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&state[4], (unsigned)__popcll(m));
            base = (unsigned)__builtin_amdgcn_readlane((int)base, __ffsll((long long)m) - 1);
            const unsigned pos = base + (unsigned)__popcll(m & ((1ull << lane) - 1ull));
            if (on && pos < (unsigned)RSEL_CAP) { ck[pos] = key; cv[pos] = vals ? vals[i] : (unsigned)i; }   // a tile's values are the candidate indices
        }
    }
}
// one workgroup: bitonic sort of the appended (key, candidate) pairs, the first K1 written out; more than RSEL_CAP keys <= T
// (thousands of candidates tied at the threshold) raises flag bit 1: the host's sort takes the section
// out: K1 keys, K1 candidates, then the section's flag word (1 = a NaN score, 2 = overflow here): ONE readback
__global__ __launch_bounds__(1024) void k_rsel_sort(const unsigned *work, const unsigned *ck, const unsigned *cv, const RselSecs Ks, unsigned *out,
                                                    const unsigned *flag, long work_stride, long out_stride) {
    __shared__ unsigned sk[RSEL_CAP], sv[RSEL_CAP];
    work += (size_t)blockIdx.x * work_stride;   // one workgroup per section
    ck += (size_t)blockIdx.x * RSEL_CAP;
    cv += (size_t)blockIdx.x * RSEL_CAP;
    out += (size_t)blockIdx.x * out_stride;
    flag += blockIdx.x;
    const unsigned K1 = Ks.K1[blockIdx.x];
    unsigned *out_keys = out, *out_vals = out + K1;
    const unsigned appended = work[3 * RSEL_BINS + 4];
    if (threadIdx.x == 0) out[2 * K1] = *flag | (appended > (unsigned)RSEL_CAP ? 2u : 0u);
    const unsigned count = appended < (unsigned)RSEL_CAP ? appended : (unsigned)RSEL_CAP;
    unsigned P = 2;
    while (P < count) P <<= 1;
    for (unsigned i = threadIdx.x; i < P; i += blockDim.x) {
        sk[i] = i < count ? ck[i] : 0xFFFFFFFFu;
        sv[i] = i < count ? cv[i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    for (unsigned k = 2; k <= P; k <<= 1)
        for (unsigned j = k >> 1; j > 0; j >>= 1) {
            for (unsigned i = threadIdx.x; i < P; i += blockDim.x) {
                const unsigned x = i ^ j;
                if (x > i) {
                    const bool asc = (i & k) == 0;
                    const unsigned a = sk[i], b = sk[x];
                    if ((a > b) == asc && a != b) { sk[i] = b; sk[x] = a; const unsigned t = sv[i]; sv[i] = sv[x]; sv[x] = t; }
                }
            }
            __syncthreads();
        }
    for (unsigned i = threadIdx.x; i < K1; i += blockDim.x) {
        out_keys[i] = i < P ? sk[i] : 0xFFFFFFFFu;
        out_vals[i] = i < P ? sv[i] : 0xFFFFFFFFu;
    }
}

// top_k on a tile, long prefixes: the first histogram of section blockIdx.y's radix selection from the keys the scoring pass wrote
__global__ __launch_bounds__(256) void k_rank_tile_keys(long n, long cap, const unsigned *keys, unsigned *work, long work_stride) {
    __shared__ unsigned lh[RSEL_BINS];
    const int u = blockIdx.y;
    for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x) lh[j] = 0;
    __syncthreads();
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&lh[keys[(size_t)u * cap + i] >> 21], 1u);
    __syncthreads();
    unsigned *hist1 = work + (size_t)u * work_stride;
    for (int j = threadIdx.x; j < RSEL_BINS; j += blockDim.x)
        if (lh[j]) atomicAdd(&hist1[j], lh[j]);
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
// =====================================================================================================================================
// A TILE of user sections per pass over the candidate matrix (svdf_ranker_process_rows, no special samples) -- the ranker as a BATCHED
// PRODUCT.  The reference scores every candidate against ONE user per PROCESS line (apex_svd_base.h:754-765): 52 MB of prepared
// candidates streamed per section.  Sections that follow each other without new candidates score the SAME matrix, so up to RANK_TILE = 32
// of them share one pass: candidates x sections x k.  It is not an MFMA product: the ORDER of the scores is the result, near-ties flip
// with the rounding order, so per (candidate, section) the arithmetic is the reference's dot product (apex_tensor_sse.h:289-317: four
// serial chains over the 4-float chunks, (a0 + a2) + (a1 + a3), scalar tail, unfused multiply and add) -- 2 VALU operations per element,
// which makes the pass VALU-bound at 32 sections per sweep (8 192 multiply-adds' worth of VALU per lane and tile against 16 KB read).
//   * one LANE owns one candidate (chunk-major matrix: every load of a wave is 1 KB contiguous), 4 accumulators per section in
//     registers (128 VGPRs at 32 sections);
//   * the tile's user factors are WAVE-UNIFORM: chunk-major too (tuT[chunk][section][4], written by the opening kernel), read through the
//     constant address space = scalar loads, and used as SGPR operands of the multiplies -- no LDS traffic, no VGPRs;
//   * what is per user stays per user: the ban bits (bit u of banmask[i]: candidate i is BAN_SAMPLE in section u), the positives with
//     their scores and greater / tie counters (positions mode), the sort keys and their per-wave minima (top_k mode).
// stage words of section u at stage + T.off[u]: uidx[nu] uval[nu] pos[npos] ban[nban]; cnt / pos_score entries at T.pos0[u].
// =====================================================================================================================================
typedef float rk_f4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(4))) rk_f4 rk_cf4;
__device__ __forceinline__ rk_cf4 *rk_constant(const float *p) {
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    return (rk_cf4 *)p;
#pragma clang diagnostic pop
}

template <int LPI, typename R>
__global__ __launch_bounds__(1024) void k_rank_tile_open(const DevParams P, const unsigned *stage, const RankTile T, const float *fb_in, float *tuT,
                                                         unsigned *banmask, const unsigned *prev_ban, int nprev, int *cnt, unsigned *flag, long cap,
                                                         const float4 *ifT, const float *ibias, float *pos_score, unsigned *zero_words, long nzero) {
    extern __shared__ float tus[];   // [RANK_TILE][pitch + 4]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwave = blockDim.x >> 6;
    for (long j = threadIdx.x; j < nzero; j += blockDim.x) zero_words[j] = 0u;           // top_k, long prefixes: the radix selection's work areas
    for (int j = threadIdx.x; j < nprev; j += blockDim.x) banmask[prev_ban[j]] = 0u;   // the previous tile's bans
    __syncthreads();
    using io = row_io<LPI, R>;
    for (int u = wave; u < RANK_TILE; u += nwave) {
        const bool live = u < T.nsec;
        const unsigned *st = stage + (live ? T.off[u] : 0);
        const int nu = live ? T.nu[u] : 0, npos = live ? T.npos[u] : 0, nban = live ? T.nban[u] : 0;
        const unsigned *ban = st + 2 * nu + npos;
        for (int j = lane; j < nban; j += 64) atomicOr(&banmask[ban[j]], 1u << u);
        int *mycnt = cnt + 2 * (live ? T.pos0[u] : 0);
        for (int j = lane; j < 2 * npos; j += 64) mycnt[j] = 0;
        if (live && lane == 0) flag[u] = 0u;
        float *mytu = tus + (size_t)u * (P.pitch + 4);
        if (lane < LPI) {
            const unsigned *uidx = st;
            const float *uval = reinterpret_cast<const float *>(st + nu);
            R tu = (live && fb_in) ? io::load(fb_in, 0, P.pitch, lane, P.k) : row_traits<R>::zero();
            for (int j = 0; j < nu; j++) {
                const unsigned uid = uidx[j];
                axpy4(tu, io::load(P.W, P.user_off + uid, P.pitch, lane, P.k), uval[j]);
                if (uid < P.feat_user.num_row)
                    for (unsigned c = P.feat_user.row_ptr[uid]; c < P.feat_user.row_ptr[uid + 1]; c++)
                        axpy4(tu, io::load(P.W, P.user_off + P.feat_user.index[c], P.pitch, lane, P.k), P.feat_user.value[c]);
            }
            io::store(mytu, 0, P.pitch, lane, P.k, tu);   // (dead sections: zeros)
        }
    }
    __syncthreads();
    // the tile's factors chunk-major for the scoring pass: tuT[(chunk * RANK_TILE + section) * 4 + c]; lanes past k hold the zero padding
    const int nchunk = P.pitch >> 2;
    for (int e = threadIdx.x; e < nchunk * RANK_TILE * 4; e += blockDim.x) {
        const int c = e & 3, u = (e >> 2) % RANK_TILE, j = (e >> 2) / RANK_TILE;
        const int col = 4 * j + c;
        tuT[e] = col < P.k ? tus[(size_t)u * (P.pitch + 4) + col] : 0.0f;
    }
    for (int u = wave; u < T.nsec; u += nwave) {
        const unsigned *st = stage + T.off[u];
        const int *pos = reinterpret_cast<const int *>(st + 2 * T.nu[u]);
        const float *mytu = tus + (size_t)u * (P.pitch + 4);
        for (int j = lane; j < T.npos[u]; j += 64)
            pos_score[T.pos0[u] + j] = 0.0f + rank_lane_score<8>(P.k, cap, reinterpret_cast<const float4 *>(mytu), ifT + pos[j], ibias[pos[j]]);
    }
}

// MODE 0: rank positions of the tile's positives (greater / tie counters); scores of the ranked candidates to out[u * cap + i]
// MODE 1: top_k: the sort KEY of every candidate (banned: 0xFFFFFFFF, NaN: 0xFFFFFFFE + the section's flag) to out[u * cap + i] -- a key
//         is its score up to the sign of zero and NaN payloads, neither of which the reference's comparator sees -- and the minimum key
//         of every wave to wmin[u * nwave_total + wave]: the K-th smallest of those minima bounds the K-th smallest key from above
//         (k_rank_tile_select)
// A wave scores 64 candidates against NSEC sections: sections sec0 = blockIdx.y * NSEC ... of the tile.  The user factors of 4 sections
// and one chunk are ONE 64-byte scalar load; the loads are software-pipelined one group ahead and fenced (sched_barrier), or the
// scheduler hoists all of an iteration's scalar loads to its top and spills SGPRs through v_writelane.
__device__ __forceinline__ unsigned rk_umin(unsigned a, unsigned b) { return a < b ? a : b; }
typedef float rk_f16 __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(4))) rk_f16 rk_cf16;
// scalar load of one group / wait for it: asm volatile keeps the load where it is written; the compiler's own wait-count pass does not see
// it, so the wait is ours and is tied to the registers ("+s") ahead of their first use
#define RK_SLOAD(DST, PTR) asm volatile("s_load_dwordx16 %0, %1, 0x0" : "=&s"(DST) : "s"(PTR))
#define RK_SWAIT1(A) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A))
#define RK_SWAIT2(A, B) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(A), "+s"(B))
// chains 0, 1 and chains 2, 3 of one section as two packed pairs: written as 2-vectors so that the packed multiplies / adds are formed
// HERE (left to the SLP vectoriser the whole tile becomes one tree that is emitted behind the last load, every factor spilled on the way)
typedef float rk_f2 __attribute__((ext_vector_type(2)));
// one group = 4 sections x one chunk: the 8 packed multiplies, then the 8 packed adds (independent neighbours; multiply-add pairs back to back
// cost a wait state each)
#define RK_GROUP(A, P, VLO, VHI)                                                                                                          \
    do {                                                                                                                                  \
        const rk_f2 t0_ = (rk_f2){(P).s0, (P).s1} * (VLO), t1_ = (rk_f2){(P).s2, (P).s3} * (VHI), t2_ = (rk_f2){(P).s4, (P).s5} * (VLO),  \
                    t3_ = (rk_f2){(P).s6, (P).s7} * (VHI), t4_ = (rk_f2){(P).s8, (P).s9} * (VLO), t5_ = (rk_f2){(P).sa, (P).sb} * (VHI),  \
                    t6_ = (rk_f2){(P).sc, (P).sd} * (VLO), t7_ = (rk_f2){(P).se, (P).sf} * (VHI);                                         \
        (A)[0][0] = (A)[0][0] + t0_; (A)[0][1] = (A)[0][1] + t1_; (A)[1][0] = (A)[1][0] + t2_; (A)[1][1] = (A)[1][1] + t3_;               \
        (A)[2][0] = (A)[2][0] + t4_; (A)[2][1] = (A)[2][1] + t5_; (A)[3][0] = (A)[3][0] + t6_; (A)[3][1] = (A)[3][1] + t7_;               \
    } while (0)
template <int NSEC, int MODE, int LD, int DBG = 0>   // LD: chunks (16-byte loads) per lane and iteration, requested one iteration ahead
__global__ __launch_bounds__(256) void k_rank_score_tile(int k, int nchunk, long n, long cap, const float *__restrict__ tuT, const float4 *__restrict__ ifT,
                                                         const float *__restrict__ ibias, const unsigned *__restrict__ banmask, unsigned *out,
                                                         const unsigned *stage, const RankTile T, const float *pos_score, int *cnt, unsigned *wmin,
                                                         unsigned *flag) {
    static_assert(NSEC % 4 == 0 && RANK_TILE % NSEC == 0, "sections per wave");
    extern __shared__ int lcnt[];   // MODE 0: 2 * total positives counters
    const int totpos = T.pos0[T.nsec - 1] + T.npos[T.nsec - 1];
    const int sec0 = blockIdx.y * NSEC;
    if (MODE == 0) {
        for (int j = threadIdx.x; j < 2 * totpos; j += blockDim.x) lcnt[j] = 0;
        __syncthreads();
    }
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int lane = threadIdx.x & 63;
    const bool in = i < n;
    const long ic = in ? i : 0;
    const int nfull = k >> 2, ntail = k & 3;
    // positions: the first 64 positives of every section of this wave, one per lane (their scores and candidate ids), requested before the pass
    float psreg[MODE == 0 ? NSEC : 1];
    int pireg[MODE == 0 ? NSEC : 1];
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < NSEC; u++) {
            const int su = sec0 + u;
            const bool have = su < T.nsec && lane < T.npos[su < T.nsec ? su : 0];
            psreg[u] = have ? pos_score[T.pos0[su] + lane] : 0.0f;
            pireg[u] = have ? (int)stage[T.off[su] + 2 * T.nu[su] + lane] : -1;
        }
    }
    rk_f2 a[NSEC][2];
#pragma unroll
    for (int u = 0; u < NSEC; u++) { a[u][0] = (rk_f2){0.0f, 0.0f}; a[u][1] = (rk_f2){0.0f, 0.0f}; }
    const float4 *q = ifT + ic;
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wold-style-cast"
    rk_cf16 *tg = (rk_cf16 *)tuT;   // group g of chunk j: 4 sections x 4 floats at tg[j * (RANK_TILE / 4) + g]
    rk_cf4 *tc = (rk_cf4 *)tuT;
#pragma clang diagnostic pop
    // The factors of SG groups (4 sections each) are requested together, one STEP ahead of their use, into the other one of two register sets
    // (the unrolled steps alternate: no copies).  Scalar loads return out of order, so a wait is always for all of them -- it stands ahead of
    // the next request.  Nothing is in flight across a loop edge: the compiler does not know that these registers are written behind its back,
    // and a copy it places between a request and its wait (a loop-carried value changing registers) would let the load land in registers
    // it has given to something else.  One exposed scalar latency per iteration, covered by the other waves.
    constexpr int GPC = RANK_TILE / 4, NG = NSEC / 4, SG = NSEC >= 8 ? 2 : 1, NSG = NG / SG;
    const int g0 = sec0 / 4;
    float4 vn[LD];   // the candidate's chunks one iteration ahead (clamped to the matrix: the last iteration re-reads its final chunk)
#pragma unroll
    for (int c = 0; c < LD; c++) vn[c] = q[(size_t)(c < nchunk ? c : nchunk - 1) * cap];
    int j = 0;
    for (; j + LD <= nfull; j += LD) {
        rk_f16 buf[2][SG];
#pragma unroll
        for (int e = 0; e < SG; e++) RK_SLOAD(buf[0][e], tg + j * GPC + g0 + e);
        float4 v[LD];
#pragma unroll
        for (int c = 0; c < LD; c++) {
            v[c] = vn[c];
            const int jn = j + LD + c;
            if (DBG != 2) vn[c] = q[(size_t)(jn < nchunk ? jn : nchunk - 1) * cap];
        }
#pragma unroll
        for (int st = 0; st < LD * NSG; st++) {
            const int c = st / NSG, sg = st % NSG;
            if (SG == 2) RK_SWAIT2(buf[st & 1][0], buf[st & 1][SG - 1]); else RK_SWAIT1(buf[st & 1][0]);
            if (st + 1 < LD * NSG) {
                const int c1 = (st + 1) / NSG, sg1 = (st + 1) % NSG;
#pragma unroll
                for (int e = 0; e < SG; e++)
                    if (DBG != 1) RK_SLOAD(buf[(st + 1) & 1][e], tg + (j + c1) * GPC + g0 + sg1 * SG + e);
            }
            const rk_f2 vlo = {v[c].x, v[c].y}, vhi = {v[c].z, v[c].w};
#pragma unroll
            for (int e = 0; e < SG; e++) RK_GROUP(a + 4 * (sg * SG + e), buf[st & 1][e], vlo, vhi);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    for (; j < nfull; j++) {
        rk_f16 buf[2][SG];
#pragma unroll
        for (int e = 0; e < SG; e++) RK_SLOAD(buf[0][e], tg + j * GPC + g0 + e);
        const float4 v = q[(size_t)j * cap];
        const rk_f2 vlo = {v.x, v.y}, vhi = {v.z, v.w};
#pragma unroll
        for (int sg = 0; sg < NSG; sg++) {
            if (SG == 2) RK_SWAIT2(buf[sg & 1][0], buf[sg & 1][SG - 1]); else RK_SWAIT1(buf[sg & 1][0]);
            if (sg + 1 < NSG) {
#pragma unroll
                for (int e = 0; e < SG; e++) RK_SLOAD(buf[(sg + 1) & 1][e], tg + j * GPC + g0 + (sg + 1) * SG + e);
            }
#pragma unroll
            for (int e = 0; e < SG; e++) RK_GROUP(a + 4 * (sg * SG + e), buf[sg & 1][e], vlo, vhi);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float4 vt = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if (ntail) vt = q[(size_t)nfull * cap];
    const float bias = ibias[ic];
    const unsigned bm = banmask[ic];
    const long nwave_total = (long)gridDim.x * (blockDim.x >> 6);
    const long wave_id = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
#pragma unroll
    for (int u = 0; u < NSEC; u++) {
        const int su = sec0 + u;
        if (su >= T.nsec) continue;
        float sum = (a[u][0].x + a[u][1].x) + (a[u][0].y + a[u][1].y);   // (a0 + a2) + (a1 + a3)
        if (ntail) {
            const rk_f4 p = tc[nfull * RANK_TILE + su];
            sum = sum + p.x * vt.x;
            if (ntail > 1) sum = sum + p.y * vt.y;
            if (ntail > 2) sum = sum + p.z * vt.z;
        }
        const bool on = in && !((bm >> su) & 1u);
        const float s = 0.0f + (bias + sum);    // item_score is the 0 of proc_user (:726) plus bias + dot (:762-764)
        if (MODE == 0) {
            if (on) out[(size_t)su * cap + i] = __float_as_uint(s);
            // greater[j] = ranked candidates with a strictly higher score than positive j; ties[j] != 0: the host's sort decides the section
            // (another ranked candidate with exactly a positive's score, or NaN on either side).  Banned lanes compare as -inf.
            const int npos = T.npos[su];
            const float sc = on ? s : -__builtin_huge_valf();
            unsigned long long tmask = __ballot(on && s != s);
            for (int base = 0; base < npos; base += 64) {
                float psl = psreg[u];
                int pil = pireg[u];
                if (base > 0) {   // more than 64 positives in a section: the next 64
                    const bool have = base + lane < npos;
                    psl = have ? pos_score[T.pos0[su] + base + lane] : 0.0f;
                    pil = have ? (int)stage[T.off[su] + 2 * T.nu[su] + base + lane] : -1;
                }
                const int m = npos - base < 64 ? npos - base : 64;
                tmask |= __ballot(lane < m && psl != psl);
                int cg = 0;   // lane jj: this wave's count for positive base + jj
                for (int jj = 0; jj < m; jj++) {
                    const float ps = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, psl), jj));
                    const int pi = __builtin_amdgcn_readlane(pil, jj);
                    const int ngt = __popcll(__ballot(sc > ps));
                    tmask |= __ballot(sc == ps && pi != (int)i);
                    cg = lane == jj ? ngt : cg;
                }
                if (lane < m && cg) atomicAdd(&lcnt[2 * T.pos0[su] + base + lane], cg);
            }
            if (tmask != 0ull && lane == 0 && npos > 0) atomicAdd(&lcnt[2 * T.pos0[su] + npos], 1);   // (ties[0] of the section; any non-zero entry sends it to the host)
        } else {
            unsigned key = 0xFFFFFFFFu;
            if (in) {
                key = rank_sort_key(on ? s : 0.0f, !on, flag + su);
                out[(size_t)su * cap + i] = key;
            }
            // the wave's minimum: inside each row of 16 lanes by DPP (quad swaps, half-row and row mirrors), the four rows through SGPRs
            unsigned m = key;
            m = rk_umin(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0xB1, 0xF, 0xF, false));
            m = rk_umin(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x4E, 0xF, 0xF, false));
            m = rk_umin(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x141, 0xF, 0xF, false));
            m = rk_umin(m, (unsigned)__builtin_amdgcn_update_dpp((int)m, (int)m, 0x140, 0xF, 0xF, false));
            const unsigned m0 = (unsigned)__builtin_amdgcn_readlane((int)m, 0), m1 = (unsigned)__builtin_amdgcn_readlane((int)m, 16),
                           m2 = (unsigned)__builtin_amdgcn_readlane((int)m, 32), m3 = (unsigned)__builtin_amdgcn_readlane((int)m, 48);
            if (lane == 0) wmin[(size_t)su * (size_t)nwave_total + (size_t)wave_id] = rk_umin(rk_umin(m0, m1), rk_umin(m2, m3));
        }
    }
    if (MODE == 0) {
        __syncthreads();
        for (int jj = threadIdx.x; jj < 2 * totpos; jj += blockDim.x)
            if (lcnt[jj]) atomicAdd(&cnt[jj], lcnt[jj]);
    }
}

// top_k of a tile's section (blockIdx.x) from its keys and the per-wave minima of k_rank_score_tile<., 1>: one workgroup.
//   1. T0 = the K1-th smallest wave minimum (with multiplicity): at least K1 candidates have a key <= T0, so every one of the K1 smallest
//      keys is <= T0 -- and few others are (the minima of 64 candidates each are an almost sorted sample of the best keys);
//   2. one sweep over the section's keys appends every (key, candidate) with key <= T0 to LDS (more than RSEL_TILE_CAP of them -- masses of
//      candidates tied at the threshold -- raises flag bit 1: the host's sort takes the section);
//   3. bitonic sort of the appended pairs, the first K1 written out: K1 keys, K1 candidates, then the flag word (1 = a NaN score, 2 = overflow).
// needs nminima <= RSEL_TILE_MINIMA and (nminima >= K1 or n <= RSEL_TILE_CAP)
constexpr int RSEL_TILE_CAP = 2048;
constexpr int RSEL_TILE_MINIMA = 8192;
__global__ __launch_bounds__(1024) void k_rank_tile_select(long n, long cap, const unsigned *keys, const unsigned *wmin, long nminima, const RselSecs Ks,
                                                           unsigned *outp, long out_stride, const unsigned *flag) {
    __shared__ unsigned mins[RSEL_TILE_MINIMA];
    __shared__ unsigned sk[RSEL_TILE_CAP], sv[RSEL_TILE_CAP];
    __shared__ unsigned count, T0s;
    const int u = blockIdx.x, t = threadIdx.x, lane = t & 63, wave = t >> 6;
    keys += (size_t)u * cap;
    wmin += (size_t)u * nminima;
    unsigned *out = outp + (size_t)u * out_stride;
    const unsigned K1 = Ks.K1[u];
    for (long j = t; j < nminima; j += blockDim.x) mins[j] = wmin[j];
    if (t == 0) { count = 0u; T0s = 0xFFFFFFFFu; }
    __syncthreads();
    // 1. T0: an upper bound of the K1-th smallest key that few keys stay under.  The minima are folded into 64 groups (lane l of wave 0 ends up
    // with the minimum of the candidates of every 64th wave); the K1-th smallest of the 64 group minima has K1 candidates at or under it
    // (K1 <= 32 groups), and at 100 K candidates about 1.2 x K1 keys in all.  Fewer than K1 minima: T0 stays "every key" (n <= RSEL_TILE_CAP).
    for (long j = (long)t + blockDim.x; j < nminima; j += blockDim.x) mins[t] = rk_umin(mins[t], mins[j]);
    __syncthreads();
    if (wave == 0 && (unsigned)nminima >= K1) {
        const long live = nminima < (long)blockDim.x ? nminima : (long)blockDim.x;
        unsigned m[4] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
        for (int r = 0; r < 16; r += 4) {
#pragma unroll
            for (int e = 0; e < 4; e++) {
                const long j = lane + 64L * (r + e);
                if (j < live) m[e] = rk_umin(m[e], mins[j]);
            }
        }
        const unsigned mine = rk_umin(rk_umin(m[0], m[1]), rk_umin(m[2], m[3]));
        unsigned rank = 0u;
        for (int o = 0; o < 64; o++) {
            const unsigned other = (unsigned)__builtin_amdgcn_readlane((int)mine, o);
            rank += (other < mine || (other == mine && o < lane)) ? 1u : 0u;
        }
        if (rank == K1 - 1u) T0s = mine;
    }
    __syncthreads();
    const unsigned T0 = T0s;
    // 2. the sweep: 4 keys per thread and step
    const long n4 = n & ~3L;
    const bool aligned = ((cap & 3L) == 0);
    if (aligned) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        for (long b = (long)t * 4; b < n4; b += (long)blockDim.x * 4) {
            const uint4 kk = k4[b >> 2];
            const unsigned kv[4] = {kk.x, kk.y, kk.z, kk.w};
#pragma unroll
            for (int e = 0; e < 4; e++)
                if (kv[e] <= T0) {
                    const unsigned pos = atomicAdd(&count, 1u);
                    if (pos < (unsigned)RSEL_TILE_CAP) { sk[pos] = kv[e]; sv[pos] = (unsigned)(b + e); }
                }
        }
    }
    for (long i = (aligned ? n4 : 0L) + t; i < n; i += blockDim.x) {
        const unsigned kx = keys[i];
        if (kx <= T0) {
            const unsigned pos = atomicAdd(&count, 1u);
            if (pos < (unsigned)RSEL_TILE_CAP) { sk[pos] = kx; sv[pos] = (unsigned)i; }
        }
    }
    __syncthreads();
    const unsigned appended = count;
    if (t == 0) out[2 * K1] = flag[u] | (appended > (unsigned)RSEL_TILE_CAP ? 2u : 0u);
    const unsigned cnt = appended < (unsigned)RSEL_TILE_CAP ? appended : (unsigned)RSEL_TILE_CAP;
    // 3. ascending by (key, candidate): the appending order is the atomics', the output is not.  Up to 64 pairs (the usual case: a handful more
    // than K1): wave 0 ranks every pair against the others; more: bitonic sort by the workgroup
    if (cnt <= 64u) {
        if (wave == 0) {
            const unsigned ka = lane < (int)cnt ? sk[lane] : 0xFFFFFFFFu, va = lane < (int)cnt ? sv[lane] : 0xFFFFFFFFu;
            unsigned rank = 0u;
            for (unsigned o = 0; o < cnt; o++) {
                const unsigned kb = (unsigned)__builtin_amdgcn_readlane((int)ka, (int)o), vb = (unsigned)__builtin_amdgcn_readlane((int)va, (int)o);
                rank += (kb < ka || (kb == ka && vb < va)) ? 1u : 0u;
            }
            if (lane < (int)cnt && rank < K1) { out[rank] = ka; out[K1 + rank] = va; }
            for (unsigned r = cnt + lane; r < K1; r += 64u) { out[r] = 0xFFFFFFFFu; out[K1 + r] = 0xFFFFFFFFu; }
        }
        return;
    }
    unsigned Pw = 2;
    while (Pw < cnt) Pw <<= 1;
    for (unsigned i = cnt + t; i < Pw; i += blockDim.x) { sk[i] = 0xFFFFFFFFu; sv[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (unsigned kk = 2; kk <= Pw; kk <<= 1)
        for (unsigned jj = kk >> 1; jj > 0; jj >>= 1) {
            for (unsigned i = t; i < Pw; i += blockDim.x) {
                const unsigned x = i ^ jj;
                if (x > i) {
                    const bool asc = (i & kk) == 0;
                    const unsigned ka = sk[i], kb = sk[x], va = sv[i], vb = sv[x];
                    const bool gt = ka > kb || (ka == kb && va > vb);
                    if (gt == asc && !(ka == kb && va == vb)) { sk[i] = kb; sk[x] = ka; sv[i] = vb; sv[x] = va; }
                }
            }
            __syncthreads();
        }
    for (unsigned i = t; i < K1; i += blockDim.x) {
        out[i] = i < Pw ? sk[i] : 0xFFFFFFFFu;
        out[K1 + i] = i < Pw ? sv[i] : 0xFFFFFFFFu;
    }
}

void launch_rank_tile_open(const DevParams &P, const unsigned *stage, const RankTile &T, const float *fb_in, float *tuT, unsigned *banmask,
                           const unsigned *prev_ban, int nprev, int *cnt, unsigned *flag, long cap, const float *ifT, const float *ibias, float *pos_score,
                           unsigned *zero_words, long nzero, hipStream_t st) {
    const size_t lds = (size_t)RANK_TILE * ((size_t)P.pitch + 4) * sizeof(float);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_tile_open<LPI, R>), dim3(1), dim3(1024), lds, st, P, stage, T, fb_in, tuT, banmask, prev_ban,
                                              nprev, cnt, flag, cap, reinterpret_cast<const float4 *>(ifT), ibias, pos_score, zero_words, nzero));
}
long rank_tile_minima(long n) { return ((n + 255) / 256) * 4; }   // per-wave minima one scoring pass writes per section
bool rank_tile_select_applies(long n, long cap, long K1max) {
    return rank_tile_minima(n) <= RSEL_TILE_MINIMA && K1max <= 32 && (rank_tile_minima(n) >= K1max || n <= RSEL_TILE_CAP) && cap < (1L << 31);
}
void launch_rank_score_tile(const DevParams &P, long n, long cap, const float *tuT, const float *ifT, const float *ibias, const unsigned *banmask, unsigned *out,
                            const unsigned *stage, const RankTile &T, const float *pos_score, int *cnt, int mode, unsigned *wmin, unsigned *flag, hipStream_t st) {
    if (n <= 0 || T.nsec <= 0) return;
    static const int env_spw = [] { const char *e = getenv("SVDF_RANK_SECS_PER_WAVE"); return e ? atoi(e) : 0; }();
    // measured (100 K candidates, k = 128, 32 sections): top_k 34.7 us at 16 sections per wave (42.1 at 8, 57.6 at 32); positions 57.9 us at 8 (71.5 at 16)
    const int spw = env_spw == 4 || env_spw == 8 || env_spw == 16 || env_spw == 32 ? env_spw : (mode == 0 ? 8 : 16);
    const unsigned grid = (unsigned)((n + 255) / 256);
    const int totpos = T.pos0[T.nsec - 1] + T.npos[T.nsec - 1];
    const float4 *q = reinterpret_cast<const float4 *>(ifT);
    static const int env_ld = [] { const char *e = getenv("SVDF_RANK_LD"); return e ? atoi(e) : 0; }();
    auto go = [&](auto nsec, auto md) {
        constexpr int NS = decltype(nsec)::value;
        constexpr int MD = decltype(md)::value;
        const size_t lds = MD == 0 ? (size_t)2 * totpos * sizeof(int) : 0;
        const unsigned gy = (unsigned)((T.nsec + NS - 1) / NS);
        static const int env_dbg = [] { const char *e = getenv("SVDF_RANK_DBG"); return e ? atoi(e) : 0; }();
        if (env_dbg == 1 && NS == 16 && MD == 1)
            hipLaunchKernelGGL((k_rank_score_tile<NS, MD, 2, (NS == 16 && MD == 1 ? 1 : 0)>), dim3(grid, gy), dim3(256), lds, st, P.k, P.pitch >> 2, n, cap, tuT, q, ibias, banmask, out, stage, T, pos_score, cnt, wmin, flag);
        else if (env_dbg == 2 && NS == 16 && MD == 1)
            hipLaunchKernelGGL((k_rank_score_tile<NS, MD, 2, (NS == 16 && MD == 1 ? 2 : 0)>), dim3(grid, gy), dim3(256), lds, st, P.k, P.pitch >> 2, n, cap, tuT, q, ibias, banmask, out, stage, T, pos_score, cnt, wmin, flag);
        else if (env_ld == 4 && NS <= 16)
            hipLaunchKernelGGL((k_rank_score_tile<NS, MD, (NS <= 16 ? 4 : 2)>), dim3(grid, gy), dim3(256), lds, st, P.k, P.pitch >> 2, n, cap, tuT, q, ibias, banmask, out, stage, T, pos_score, cnt, wmin, flag);
        else
        hipLaunchKernelGGL((k_rank_score_tile<NS, MD, 2>), dim3(grid, gy), dim3(256), lds, st, P.k, P.pitch >> 2, n, cap, tuT, q, ibias, banmask, out, stage, T, pos_score, cnt, wmin, flag);
    };
    auto pick = [&](auto md) {
        const int w = T.nsec < spw ? T.nsec : spw;
        if (w <= 4) go(std::integral_constant<int, 4>(), md);
        else if (w <= 8) go(std::integral_constant<int, 8>(), md);
        else if (w <= 16) go(std::integral_constant<int, 16>(), md);
        else go(std::integral_constant<int, RANK_TILE>(), md);
    };
    if (mode == 0) pick(std::integral_constant<int, 0>());
    else pick(std::integral_constant<int, 1>());
}
void launch_rank_tile_select(long n, long cap, int nsec, const unsigned *keys, const unsigned *wmin, const RselSecs &Ks, unsigned *out, long out_stride,
                             const unsigned *flag, hipStream_t st) {
    if (n <= 0 || nsec <= 0) return;
    hipLaunchKernelGGL(k_rank_tile_select, dim3((unsigned)nsec), dim3(1024), 0, st, n, cap, keys, wmin, rank_tile_minima(n), Ks, out, out_stride, flag);
}

void launch_rank_user(const DevParams &P, const unsigned *stage, const RankSection &S, const float *fb_in, float *tu_out, signed char *tag, int *cnt,
                      unsigned *flag, long cap, const float *ifT, const float *ibias, float *pos_score, unsigned *zero_words, int nzero, hipStream_t st) {
    const size_t lds = pos_score ? ((size_t)P.pitch + 4) * sizeof(float) : 0;
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_rank_user<LPI, R>), dim3(1), dim3(64), lds, st, P, stage, S, fb_in, tu_out, tag, cnt, flag, cap,
                                              reinterpret_cast<const float4 *>(ifT), ibias, pos_score, zero_words, nzero));
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
                           F.pos_item, F.pos_score, F.npos, F.greater, F.ties, nullptr, nullptr, nullptr, nullptr);
    else if (F.mode == 2)
        hipLaunchKernelGGL((k_rank_score<8, 2>), dim3(grid), dim3(256), F.hist1 ? (size_t)RSEL_BINS * sizeof(unsigned) : 0, st, P.k, n, cap, tu, q, ibias, tag,
                           item_score, fresh, nullptr, nullptr, 0, nullptr, nullptr, F.keys, F.vals, F.flag, F.hist1);
    else
        hipLaunchKernelGGL((k_rank_score<8, 0>), dim3(grid), dim3(256), 0, st, P.k, n, cap, tu, q, ibias, tag, item_score, fresh, nullptr, nullptr, 0, nullptr,
                           nullptr, nullptr, nullptr, nullptr, nullptr);
}
void launch_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos, int *greater, int *ties, hipStream_t st) {
    if (n <= 0 || npos <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 128) grid = 128;
    hipLaunchKernelGGL(k_rank_positions, dim3((int)grid), dim3(256), (size_t)npos * (sizeof(float) + 2 * sizeof(int)), st, n, score, tag, pos_item, npos, greater, ties);
}
long rank_select_work_words() { return 3L * RSEL_BINS + 8; }
long rank_select_cap() { return RSEL_CAP; }
// the K1 smallest keys of keys[0..n) and their values, ascending, then the flag word, into out[0 .. 2*K1] (K1 <= RSEL_CAP / 2,
// K1 <= n).  work must be zero and its first histogram filled: k_rank_user zeroes it, k_rank_score<.,2> fills hist[0].
void launch_rank_select(long n, const unsigned *keys, const unsigned *vals, unsigned K1, unsigned *work, unsigned *ck, unsigned *cv, unsigned *out,
                        const unsigned *flag, hipStream_t st) {
    long grid = (n + 255) / 256;
    if (grid > 256) grid = 256;
    RselSecs Ks = {};
    Ks.K1[0] = K1;
    hipLaunchKernelGGL((k_rsel_hist<2>), dim3((int)grid), dim3(256), 0, st, n, keys, work, Ks, 0L, 0L);
    hipLaunchKernelGGL((k_rsel_hist<3>), dim3((int)grid), dim3(256), 0, st, n, keys, work, Ks, 0L, 0L);
    hipLaunchKernelGGL(k_rsel_compact, dim3((int)grid), dim3(256), 0, st, n, keys, vals, work, ck, cv, 0L, 0L);
    hipLaunchKernelGGL(k_rsel_sort, dim3(1), dim3(1024), 0, st, work, ck, cv, Ks, out, flag, 0L, 0L);
}
// The same selection for the nsec sections of a tile at once (grid y = section; prefixes too long for k_rank_tile_select): keys[u * cap + i]
// from k_rank_score_tile<., 1>, the Ks.K1[u] smallest (key, candidate) pairs of section u ascending, then its flag word, at
// out + u * out_stride.  work: nsec * rank_select_work_words() words, ZEROED by the caller; ck / cv: nsec * rank_select_cap() words each;
// flag: one word per section (zeroed by k_rank_tile_open).
void launch_rank_select_tile(long n, long cap, int nsec, const unsigned *keys, const RselSecs &Ks, unsigned *work, unsigned *ck, unsigned *cv, unsigned *out,
                             long out_stride, unsigned *flag, hipStream_t st) {
    if (n <= 0 || nsec <= 0) return;
    const long ww = rank_select_work_words();
    hipLaunchKernelGGL(k_rank_tile_keys, dim3((unsigned)((n + 255) / 256), (unsigned)nsec), dim3(256), 0, st, n, cap, keys, work, ww);
    long grid = (n + 255) / 256;
    if (grid > 128) grid = 128;
    hipLaunchKernelGGL((k_rsel_hist<2>), dim3((int)grid, (unsigned)nsec), dim3(256), 0, st, n, keys, work, Ks, cap, ww);
    hipLaunchKernelGGL((k_rsel_hist<3>), dim3((int)grid, (unsigned)nsec), dim3(256), 0, st, n, keys, work, Ks, cap, ww);
    hipLaunchKernelGGL(k_rsel_compact, dim3((int)grid, (unsigned)nsec), dim3(256), 0, st, n, keys, (const unsigned *)nullptr, work, ck, cv, cap, ww);
    hipLaunchKernelGGL(k_rsel_sort, dim3((unsigned)nsec), dim3(1024), 0, st, work, ck, cv, Ks, out, flag, ww, out_stride);
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
