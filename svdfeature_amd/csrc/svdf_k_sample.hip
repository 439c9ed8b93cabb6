// svdf_k_sample.hip -- PairwiseRankGenerator (apex_svd_data.cpp:812-1025) ON THE DEVICE for the common shape of rank input
// (SURVEY.md 8f2): user blocks whose rows carry one user entry and one item entry, positives against negatives
// (rank_sample_method = 0, :942-962), no pointwise output.
//
// The reference samples one user block after the other with libc rand().  Per block the number of draws is known before
// anything is drawn -- shuffle(neg) takes |neg|-1, shuffle(pos) |pos|-1 (apex_random.h:119-130), nothing else is random --
// so an exclusive scan over the blocks gives every block its slice of the rand() stream, the stream itself is expanded in
// parallel chunks from jump-ahead tables (svdf_randstream.cpp), and every block then runs the reference's two Fisher-Yates
// shuffles and its round-robin pairing in a thread of its own.  Same draws, same pairs, byte for byte.
#include <hip/hip_runtime.h>

#include "svdf_kernels.h"

#pragma clang fp contract(off)

namespace svdf {

// x[n] = x[n-31] + x[n-3] (mod 2^32): chunk c expands its table into raw[c*C .. c*C + C) (clipped to D)
__global__ __launch_bounds__(64) void k_rand_expand(const unsigned *tables, long nchunks, long C, long D, unsigned *raw) {
    const long c = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= nchunks) return;
    unsigned buf[31];
    for (int j = 0; j < 31; j++) buf[j] = tables[c * 31 + j];
    long lim = D - c * C;
    if (lim > C) lim = C;
    int f = 0;
    for (long i = 0; i < lim; i++) {
        int r = f + 28;
        if (r >= 31) r -= 31;
        const unsigned v = buf[f] + buf[r];
        buf[f] = v;
        raw[c * C + i] = v;
        f = f + 1 == 31 ? 0 : f + 1;
    }
}

// sample_posneg's membership tests (:945-949), float arithmetic as written there
__device__ __forceinline__ bool is_pos(float label, float lowerb) { return label - lowerb > -1e-6f; }
__device__ __forceinline__ bool is_neg(float label, float upperb) { return label - upperb < 1e-6f; }

// per block: number of rand() draws and number of generated pairs
__global__ __launch_bounds__(256) void k_sample_counts(const RankSourceDev S, SamplerParams sp, long *draws, long *pairs) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.num_block) return;
    long npos = 0, nneg = 0;
    for (long r = S.block_row_ptr[b]; r < S.block_row_ptr[b + 1]; r++) {
        const float l = S.label[r];
        npos += is_pos(l, sp.pos_lowerb) ? 1 : 0;
        nneg += is_neg(l, sp.neg_upperb) ? 1 : 0;
    }
    long d = 0, p = 0;
    if (npos > 0 && nneg > 0) {
        d = (nneg - 1) + (npos - 1);
        unsigned long snum = (unsigned long)nneg;
        if (sp.sample_num > 0) snum = (unsigned long)sp.sample_num;
        if (snum > (unsigned)sp.sample_max) snum = (unsigned long)(long)sp.sample_max;   // :955-956, the cast of the reference
        p = (long)snum;
    }
    draws[b] = d;
    pairs[b] = p;
}

// apex_random.h:48-50,65-67: floor( rand() / (RAND_MAX + 1.0) * n ), rand() = x >> 1
__device__ __forceinline__ unsigned next_uint32(unsigned raw, unsigned n) {
    const double u = (double)(int)(raw >> 1) / 2147483648.0;
    return (unsigned)floor(u * (double)n);
}
__device__ __forceinline__ void shuffle(int *d, long sz, const unsigned *raw, long &cur) {   // apex_random.h:119-124
    if (sz == 0) return;
    for (unsigned i = (unsigned)sz - 1; i > 0; i--) {
        const unsigned j = next_uint32(raw[cur++], i + 1);
        const int t = d[i]; d[i] = d[j]; d[j] = t;
    }
}

// one thread per user block: sample_posneg (:942-962) + genpair (:887-913) for one-user-one-item rows
__global__ __launch_bounds__(64) void k_sample_posneg(const RankSourceDev S, SamplerParams sp, const long *draw_off, const long *pair_off,
                                                      const unsigned *raw, int *pos_list, int *neg_list, PairColumns out) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.num_block) return;
    const long r0 = S.block_row_ptr[b], r1 = S.block_row_ptr[b + 1];
    int *pos = pos_list + r0, *neg = neg_list + r0;
    long npos = 0, nneg = 0;
    for (long r = r0; r < r1; r++) {
        const float l = S.label[r];
        if (is_pos(l, sp.pos_lowerb)) pos[npos++] = (int)r;
        if (is_neg(l, sp.neg_upperb)) neg[nneg++] = (int)r;
    }
    if (npos == 0 || nneg == 0) return;
    long cur = draw_off[b];
    shuffle(neg, nneg, raw, cur);
    shuffle(pos, npos, raw, cur);
    const long snum = pair_off[b + 1] - pair_off[b];
    for (long i = 0; i < snum; i++) {
        const int p = pos[i % npos], n = neg[i % nneg];
        const long o = pair_off[b] + i;
        out.label[o] = 1.0f;                 // rank_sample_method / 10 == 0 (:907-909)
        out.uidx[o] = S.uidx[p];             // the positive's user entry (:894-903); kept: |value| > 1e-6 checked on the host
        out.uval[o] = S.uval[p];
        const unsigned ip = S.iidx[p], in = S.iidx[n];
        const float vp = S.ival[p], vn = S.ival[n];
        if (ip < in) { out.i0[o] = ip; out.v0[o] = vp; out.i1[o] = in; out.v1[o] = -vn; }          // merge (:828-860)
        else if (in < ip) { out.i0[o] = in; out.v0[o] = -vn; out.i1[o] = ip; out.v1[o] = vp; }
        else { out.i0[o] = ip; out.v0[o] = vp - vn; out.i1[o] = SLOT_ABSENT; out.v1[o] = 0.0f; }
    }
}

void launch_rand_expand(const unsigned *tables, long nchunks, long C, long D, unsigned *raw, hipStream_t st) {
    if (nchunks <= 0) return;
    hipLaunchKernelGGL(k_rand_expand, dim3((int)((nchunks + 63) / 64)), dim3(64), 0, st, tables, nchunks, C, D, raw);
}
void launch_sample_counts(const RankSourceDev &S, const SamplerParams &sp, long *draws, long *pairs, hipStream_t st) {
    if (S.num_block <= 0) return;
    hipLaunchKernelGGL(k_sample_counts, dim3((int)((S.num_block + 255) / 256)), dim3(256), 0, st, S, sp, draws, pairs);
}
void launch_sample_posneg(const RankSourceDev &S, const SamplerParams &sp, const long *draw_off, const long *pair_off, const unsigned *raw,
                          int *pos_list, int *neg_list, const PairColumns &out, hipStream_t st) {
    if (S.num_block <= 0) return;
    hipLaunchKernelGGL(k_sample_posneg, dim3((int)((S.num_block + 63) / 64)), dim3(64), 0, st, S, sp, draw_off, pair_off, raw, pos_list, neg_list, out);
}

}  // namespace svdf
