// svdf_k_gsample.hip -- PairwiseRankGenerator (apex_svd_data.cpp:812-1025) ON THE DEVICE, the GENERAL form:
// rows of any shape (global entries, several user / item entries), blocks with implicit feedback, both sampling methods
// (rank_sample_method 0 = positives against negatives :946-965, 1 = every row against a random row whose label differs by more than
// the gap :920-944), pairwise or pointwise output (:862-915).  svdf_k_sample.hip keeps the specialised path for plain rows, whose
// output stays in HBM as schedule columns; this file produces the generated blocks in the reference's own CSR layout:
//   k_gsample_counts   per block: rand() draws and generated pairs -- both known before anything is drawn (below)
//   [host scan]        every block's slice of the rand() stream (svdf_randstream.cpp jump-ahead) and of the output
//   k_gsample_pairs    per block: the reference's shuffles / std::sort / pairing -> (positive row, negative row) per pair
//   k_gpair_sizes      per pair: section lengths of the generated row(s)  -> rocPRIM exclusive scan -> row_ptr
//   k_gpair_write      per pair: merged entries (index order, the negative's sign flipped, :828-860) and label
// sample_cmp: shuffle(neg) takes n - 1 draws; row i of the shuffled list draws once iff rng = left + n - right > 0, where left /
// right are the lower_bound positions of (label - gap) and ((label - gap) + 2 gap) among ALL labels of the block -- a property of
// the row's label, not of the shuffle -- so a block's number of draws is the same whatever the shuffle produces.  The order of rows
// with EQUAL labels in the sorted list is libstdc++'s introsort's (svdf_stdsort.h).
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>

#include <stdexcept>

#include "svdf_kernels.h"
#include "svdf_stdsort.h"

#pragma clang fp contract(off)

namespace svdf {

__device__ __forceinline__ bool g_is_pos(float label, float lowerb) { return label - lowerb > -1e-6f; }   // :949
__device__ __forceinline__ bool g_is_neg(float label, float upperb) { return label - upperb < 1e-6f; }    // :950
// apex_random.h:48-50,65-67: floor( rand() / (RAND_MAX + 1.0) * n ), rand() = x >> 1
__device__ __forceinline__ unsigned g_next_uint32(unsigned raw, unsigned n) {
    const double u = (double)(int)(raw >> 1) / 2147483648.0;
    return (unsigned)floor(u * (double)n);
}
__device__ __forceinline__ void g_shuffle(int *d, long sz, const unsigned *raw, long &cur) {   // apex_random.h:119-124
    if (sz == 0) return;
    for (unsigned i = (unsigned)sz - 1; i > 0; i--) {
        const unsigned j = g_next_uint32(raw[cur++], i + 1);
        const int t = d[i]; d[i] = d[j]; d[j] = t;
    }
}

__global__ __launch_bounds__(64) void k_gsample_counts(const RankRowsDev S, GSamplerParams sp, int *scratch, long *draws, long *pairs) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.num_block) return;
    const long r0 = S.block_row_ptr[b], r1 = S.block_row_ptr[b + 1], n = r1 - r0;
    long d = 0, p = 0;
    if (sp.method == 0) {
        long npos = 0, nneg = 0;
        for (long r = r0; r < r1; r++) {
            const float l = S.label[r];
            npos += g_is_pos(l, sp.pos_lowerb) ? 1 : 0;
            nneg += g_is_neg(l, sp.neg_upperb) ? 1 : 0;
        }
        if (npos > 0 && nneg > 0) {
            d = (nneg - 1) + (npos - 1);
            unsigned long snum = (unsigned long)nneg;
            if (sp.sample_num > 0) snum = (unsigned long)sp.sample_num;
            if (snum > (unsigned)sp.sample_max) snum = (unsigned long)(long)sp.sample_max;   // :958-959, the cast of the reference
            p = (long)snum;
        }
    } else if (n > 0) {
        int *ids = scratch + r0;
        for (long j = 0; j < n; j++) ids[j] = (int)(r0 + j);
        stdsort::sort(ids, n, stdsort::ByLabel{S.label});   // any order among equal labels will do for counting
        d = n - 1;
        for (long r = r0; r < r1; r++) {
            float l = S.label[r];
            l -= sp.gap;
            const long left = stdsort::lower_bound_label(ids, n, S.label, l);
            l += sp.gap * 2;
            const long right = stdsort::lower_bound_label(ids, n, S.label, l);
            const unsigned rng = (unsigned)(left + n - right);
            if (rng > 0) { d++; p++; }
        }
    }
    draws[b] = d;
    pairs[b] = p;
}

__global__ __launch_bounds__(64) void k_gsample_pairs(const RankRowsDev S, GSamplerParams sp, const long *draw_off, const long *pair_off, const unsigned *raw,
                                                      int *pos_list, int *neg_list, int *pair_p, int *pair_n) {
    const long b = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.num_block) return;
    const long r0 = S.block_row_ptr[b], r1 = S.block_row_ptr[b + 1], n = r1 - r0;
    int *pos = pos_list + r0, *neg = neg_list + r0;
    long cur = draw_off[b];
    long o = pair_off[b];
    if (sp.method == 0) {   // sample_posneg
        long npos = 0, nneg = 0;
        for (long r = r0; r < r1; r++) {
            const float l = S.label[r];
            if (g_is_pos(l, sp.pos_lowerb)) pos[npos++] = (int)r;
            if (g_is_neg(l, sp.neg_upperb)) neg[nneg++] = (int)r;
        }
        if (npos == 0 || nneg == 0) return;
        g_shuffle(neg, nneg, raw, cur);
        g_shuffle(pos, npos, raw, cur);
        const long snum = pair_off[b + 1] - pair_off[b];
        for (long i = 0; i < snum; i++) { pair_p[o + i] = pos[i % npos]; pair_n[o + i] = neg[i % nneg]; }
        return;
    }
    if (n == 0) return;
    // sample_cmp
    for (long j = 0; j < n; j++) { pos[j] = (int)(r0 + j); neg[j] = (int)(r0 + j); }
    g_shuffle(neg, n, raw, cur);
    stdsort::sort(pos, n, stdsort::ByLabel{S.label});
    for (long i = 0; i < n; i++) {
        float l = S.label[neg[i]];
        l -= sp.gap;
        const long left = stdsort::lower_bound_label(pos, n, S.label, l);
        l += sp.gap * 2;
        const long right = stdsort::lower_bound_label(pos, n, S.label, l);
        const unsigned rng = (unsigned)(left + n - right);
        if (rng > 0) {
            const unsigned idx = g_next_uint32(raw[cur++], rng);
            if ((long)idx < left) { pair_p[o] = neg[i]; pair_n[o] = pos[idx]; }                 // genpair(neg[i], pos[idx])
            else { pair_p[o] = pos[right + (long)idx - left]; pair_n[o] = neg[i]; }             // genpair(pos[right + idx - left], neg[i])
            o++;
        }
    }
}

// merge of two index-sorted entry lists with the second one's sign flipped (:828-860): number of entries / the entries themselves
__device__ __forceinline__ int merge_count(const unsigned *i1, int n1, const unsigned *i2, int n2) {
    int num = 0, i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (i1[i] < i2[j]) { i++; num++; continue; }
        if (i2[j] < i1[i]) { j++; num++; continue; }
        i++; j++; num++;
    }
    return num + (n1 - i) + (n2 - j);
}
__device__ __forceinline__ void merge_write(const unsigned *i1, const float *v1, int n1, const unsigned *i2, const float *v2, int n2, unsigned *oi, float *ov) {
    int i = 0, j = 0, o = 0;
    while (i < n1 && j < n2) {
        if (i1[i] < i2[j]) { oi[o] = i1[i]; ov[o] = v1[i]; i++; o++; continue; }
        if (i2[j] < i1[i]) { oi[o] = i2[j]; ov[o] = -v2[j]; j++; o++; continue; }
        oi[o] = i1[i]; ov[o] = v1[i] - v2[j]; i++; j++; o++;
    }
    while (i < n1) { oi[o] = i1[i]; ov[o] = v1[i]; i++; o++; }
    while (j < n2) { oi[o] = i2[j]; ov[o] = -v2[j]; j++; o++; }
}
__device__ __forceinline__ bool user_entry_kept(float v) { return v > 1e-6f || v < -1e-6f; }   // :868, :897
// lens[3 * row + s]: length of section s (global / user / item) of generated row `row`; one row per pair, two with pointwise output
__global__ __launch_bounds__(256) void k_gpair_sizes(const RankRowsDev S, GSamplerParams sp, long npairs, const int *pair_p, const int *pair_n, int *lens) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += stride) {
        const int *pp = S.row_ptr + 3L * pair_p[t], *pn = S.row_ptr + 3L * pair_n[t];
        int ku = 0;
        for (int j = pp[1]; j < pp[2]; j++) ku += user_entry_kept(S.value[j]) ? 1 : 0;
        if (sp.pointwise) {
            int kn = 0;
            for (int j = pn[1]; j < pn[2]; j++) kn += user_entry_kept(S.value[j]) ? 1 : 0;
            int *o = lens + 6 * t;
            o[0] = pp[1] - pp[0]; o[1] = ku; o[2] = pp[3] - pp[2];
            o[3] = pn[1] - pn[0]; o[4] = kn; o[5] = pn[3] - pn[2];
        } else {
            int *o = lens + 3 * t;
            o[0] = merge_count(S.index + pp[0], pp[1] - pp[0], S.index + pn[0], pn[1] - pn[0]);
            o[1] = ku;
            o[2] = merge_count(S.index + pp[2], pp[3] - pp[2], S.index + pn[2], pn[3] - pn[2]);
        }
    }
}
__device__ __forceinline__ void pointwise_write(const RankRowsDev &S, const int *p, const int *optr, unsigned *oi, float *ov) {   // genpair_pointwise (:862-885)
    int o = optr[0];
    for (int j = p[0]; j < p[1]; j++) { oi[o] = S.index[j]; ov[o] = S.value[j]; o++; }
    for (int j = p[1]; j < p[2]; j++) if (user_entry_kept(S.value[j])) { oi[o] = S.index[j]; ov[o] = S.value[j]; o++; }
    for (int j = p[2]; j < p[3]; j++) { oi[o] = S.index[j]; ov[o] = S.value[j]; o++; }
}
__global__ __launch_bounds__(256) void k_gpair_write(const RankRowsDev S, GSamplerParams sp, long npairs, const int *pair_p, const int *pair_n, const int *optr,
                                                     float *olabel, unsigned *oi, float *ov) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < npairs; t += stride) {
        const int *pp = S.row_ptr + 3L * pair_p[t], *pn = S.row_ptr + 3L * pair_n[t];
        if (sp.pointwise) {
            pointwise_write(S, pp, optr + 6 * t, oi, ov);
            pointwise_write(S, pn, optr + 6 * t + 3, oi, ov);
            olabel[2 * t] = 1.0f;
            olabel[2 * t + 1] = 0.0f;
            continue;
        }
        const int *q = optr + 3 * t;
        merge_write(S.index + pp[0], S.value + pp[0], pp[1] - pp[0], S.index + pn[0], S.value + pn[0], pn[1] - pn[0], oi + q[0], ov + q[0]);
        int o = q[1];
        for (int j = pp[1]; j < pp[2]; j++) if (user_entry_kept(S.value[j])) { oi[o] = S.index[j]; ov[o] = S.value[j]; o++; }
        merge_write(S.index + pp[2], S.value + pp[2], pp[3] - pp[2], S.index + pn[2], S.value + pn[2], pn[3] - pn[2], oi + q[2], ov + q[2]);
        olabel[t] = (sp.method_raw / 10 == 0) ? 1.0f : S.label[pair_p[t]] - S.label[pair_n[t]];   // :907-911
    }
}

void launch_gsample_counts(const RankRowsDev &S, const GSamplerParams &sp, int *scratch, long *draws, long *pairs, hipStream_t st) {
    if (S.num_block <= 0) return;
    hipLaunchKernelGGL(k_gsample_counts, dim3((int)((S.num_block + 63) / 64)), dim3(64), 0, st, S, sp, scratch, draws, pairs);
}
void launch_gsample_pairs(const RankRowsDev &S, const GSamplerParams &sp, const long *draw_off, const long *pair_off, const unsigned *raw, int *pos_list,
                          int *neg_list, int *pair_p, int *pair_n, hipStream_t st) {
    if (S.num_block <= 0) return;
    hipLaunchKernelGGL(k_gsample_pairs, dim3((int)((S.num_block + 63) / 64)), dim3(64), 0, st, S, sp, draw_off, pair_off, raw, pos_list, neg_list, pair_p, pair_n);
}
void launch_gpair_sizes(const RankRowsDev &S, const GSamplerParams &sp, long npairs, const int *pair_p, const int *pair_n, int *lens, hipStream_t st) {
    if (npairs <= 0) return;
    long grid = (npairs + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_gpair_sizes, dim3((int)grid), dim3(256), 0, st, S, sp, npairs, pair_p, pair_n, lens);
}
void launch_gpair_write(const RankRowsDev &S, const GSamplerParams &sp, long npairs, const int *pair_p, const int *pair_n, const int *optr, float *olabel,
                        unsigned *oi, float *ov, hipStream_t st) {
    if (npairs <= 0) return;
    long grid = (npairs + 255) / 256;
    if (grid > 8192) grid = 8192;
    hipLaunchKernelGGL(k_gpair_write, dim3((int)grid), dim3(256), 0, st, S, sp, npairs, pair_p, pair_n, optr, olabel, oi, ov);
}
// out[j] = sum of in[0 .. j) for j = 0 .. n (out[n] = total; in[n] is read but does not matter); in and out must not overlap
void device_exclusive_scan_i32(const int *in, int *out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st) {
    size_t need = 0;
    (void)rocprim::exclusive_scan(nullptr, need, in, out, 0, (size_t)(n + 1), rocprim::plus<int>(), st);
    if (need > *tmp_bytes) {
        if (*tmp) (void)hipFree(*tmp);
        *tmp = nullptr;
        if (hipMalloc(tmp, need) != hipSuccess) { *tmp_bytes = 0; throw std::runtime_error("device_exclusive_scan_i32: out of device memory"); }
        *tmp_bytes = need;
    }
    if (rocprim::exclusive_scan(*tmp, need, in, out, 0, (size_t)(n + 1), rocprim::plus<int>(), st) != hipSuccess)
        throw std::runtime_error("device_exclusive_scan_i32 failed");
}

// host instantiation of the restated std::sort for tests: ids 0 .. n-1 sorted by label
void host_sort_by_label(const float *label, long n, int *ids) {
    for (long j = 0; j < n; j++) ids[j] = (int)j;
    stdsort::sort(ids, n, stdsort::ByLabel{label});
}

}  // namespace svdf
