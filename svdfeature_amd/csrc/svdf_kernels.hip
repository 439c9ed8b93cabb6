// svdf_kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the apex_svd SGD hot path.
//
// What is computed, and in which order, follows the reference's base solver
// (solvers/base-solver/apex_svd_base.h: pred :445-454, calc_bias :313-353, prepare_tmp :354-381,
// update_no_decay :383-427, regularize :188-311, SVD++ hooks :506-554) and its SSE2 tensor ops
// (apex-tensor/apex_tensor_sse.h: scalar_map :261-272 with the |s-1|<=1e-6 skip :231-242,
// sdot :289-317 + sum_all :88-97).  HOW it is computed is CDNA4-first:
//
//  * a factor row (k fp32, pitch ceil(k/4)*4) is owned by a LANE GROUP of LPI lanes, one float4
//    per lane, so a wave issues one 16 B/lane load per row set: 64/LPI rows x (LPI*16) bytes,
//    e.g. 4 rows x 256 B per instruction at k=64 -- fully coalesced gathers, no LDS round trip;
//  * the dot product reproduces the reference's 4-lane SSE accumulation order bit for bit with a
//    DPP scan: lane m of the group holds chunk m's four products, and
//    acc[m] = acc[m-1] + prod[m] is applied LPI-1 times through row_shr:1 / wave_shr:1 DPP adds
//    (no LDS, no bpermute); lanes < step index are already final and are recomputed to the
//    same value, so no select is needed;
//  * instances of one launch are CONFLICT-FREE (no shared parameter row; the host scheduler
//    builds such batches in file order), so every read-modify-write is a plain load/store and
//    the result equals the reference's one-instance-at-a-time SGD exactly;
//  * each wave keeps G independent row sets in flight (G*2 KiB of gathers per wave at k=64)
//    to cover HBM latency without relying on occupancy alone.
//
// Arithmetic contract: fp32, unfused (-ffp-contract=off and the pragma below), correctly rounded
// divide/sqrt, fp64 bias/score accumulation.  Only expf (sigmoid links) differs from glibc by
// ulps; everything else is bit-exact against oracle/svdf_oracle.c.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "svdf_kernels.h"

#pragma clang fp contract(off)

namespace svdf {

// ------------------------------------------------------------------ small helpers
__device__ __forceinline__ float4 f4zero() { return make_float4(0.0f, 0.0f, 0.0f, 0.0f); }

// apex_tensor_sse.h:231-242: multiply is skipped when |s-1| <= 1e-6.  The reference compares (double)fabsf(s - 1.0f)
// with the double 1e-6; no float lies between (float)1e-6 = 9.99999997e-7 and that double, so the float compare
// below decides identically for every input (NaN included: both say "one") without fp64 instructions.
__device__ __forceinline__ bool scalar_is_one(float s) { return !(fabsf(s - 1.0f) > 1e-6f); }

// The skipped multiply, branch-free: x * 1.0f is x for every x (IEEE multiplication by one is exact, signs of zero and
// infinities included), so "skip the multiply when |s-1| <= 1e-6" is the same as multiplying by exactly 1.0f then --
// one select on the scalar instead of one per element.
__device__ __forceinline__ float snap_to_one(float s) { return scalar_is_one(s) ? 1.0f : s; }
// K1: dst += src*s (separate mul and add)
__device__ __forceinline__ void axpy4(float4 &d, const float4 s, float a) {
    const float a1 = snap_to_one(a);
    float mx = s.x * a1, my = s.y * a1, mz = s.z * a1, mw = s.w * a1;
    d.x = d.x + mx; d.y = d.y + my; d.z = d.z + mz; d.w = d.w + mw;
}
// K2: dst *= s
__device__ __forceinline__ void scale4(float4 &d, float a) {
    const float a1 = snap_to_one(a);
    d.x = d.x * a1; d.y = d.y * a1; d.z = d.z * a1; d.w = d.w * a1;
}
__device__ __forceinline__ float l1(float w, float eps) {  // K6
    if (w > eps) return w - eps;
    if (w < -eps) return w + eps;
    return 0.0f;
}

// previous lane's value inside the lane group (0 for the group's first lane)
template <int LPI>
__device__ __forceinline__ float prev_lane(float v, int L) {
    int iv = __float_as_int(v);
    int r;
    if constexpr (LPI <= 16) {
        r = __builtin_amdgcn_update_dpp(0, iv, 0x111 /* row_shr:1 */, 0xf, 0xf, true);
        if constexpr (LPI < 16) r = (L == 0) ? 0 : r;
    } else {
        r = __builtin_amdgcn_update_dpp(0, iv, 0x138 /* wave_shr:1 */, 0xf, 0xf, true);
        if constexpr (LPI < 64) r = (L == 0) ? 0 : r;
    }
    return __int_as_float(r);
}
template <int LPI>
__device__ __forceinline__ float group_bcast(float v, int src_L) {
    if constexpr (LPI == 1) return v;
    const int lane = (int)(threadIdx.x & 63);
    return __shfl(v, (lane & ~(LPI - 1)) + src_L, 64);
}

// K3: apex_tensor_sse.h:289-317.  a,b: this lane's chunk (lane L holds elements 4L..4L+3; pad and
// out-of-range chunks are 0).  Returns the dot product, identical on every lane of the group.
template <int LPI>
__device__ __forceinline__ float group_dot(const float4 a, const float4 b, int L, int k) {
    const int nfull = k >> 2;
    const int ntail = k & 3;
    float m0 = a.x * b.x, m1 = a.y * b.y, m2 = a.z * b.z, m3 = a.w * b.w;
    const bool full = L < nfull;
    // lanes beyond the full chunks feed +0 so the running sums just travel on to the last lane
    float c0 = full ? m0 : 0.0f, c1 = full ? m1 : 0.0f, c2 = full ? m2 : 0.0f, c3 = full ? m3 : 0.0f;
    float a0 = 0.0f + c0, a1 = 0.0f + c1, a2 = 0.0f + c2, a3 = 0.0f + c3;
    if constexpr (LPI <= 16 || LPI == 64) {   // 64: wave_shr:1 needs no select (lane 0 reads 0), measured faster than row scans
#pragma unroll
        for (int s = 1; s < LPI; s++) {
            a0 = prev_lane<LPI>(a0, L) + c0;
            a1 = prev_lane<LPI>(a1, L) + c1;
            a2 = prev_lane<LPI>(a2, L) + c2;
            a3 = prev_lane<LPI>(a3, L) + c3;
        }
    } else {
        // LPI == 32, two 16-lane DPP rows per group: scan one row at a time with the cheap fused
        // v_add_f32_dpp row_shr:1 (a row's first lane reads 0), and carry the finished sum of row r-1 into
        // row r by FOLDING it into the addend of that row's first lane: a[16r] = 0 + (carry + c[16r]).
        // Same additions in the same order as the chain a[m] = a[m-1] + c[m]; no per-step select
        // (measured 23 ns per step with wave_shr + select vs 8 ns per step for a plain row_shr add).
#pragma unroll
        for (int s = 1; s < 16; s++) {
            a0 = prev_lane<16>(a0, L) + c0; a1 = prev_lane<16>(a1, L) + c1;
            a2 = prev_lane<16>(a2, L) + c2; a3 = prev_lane<16>(a3, L) + c3;
        }
#pragma unroll
        for (int r = 1; r < LPI / 16; r++) {
            const float s0 = group_bcast<LPI>(a0, 16 * r - 1), s1 = group_bcast<LPI>(a1, 16 * r - 1);
            const float s2 = group_bcast<LPI>(a2, 16 * r - 1), s3 = group_bcast<LPI>(a3, 16 * r - 1);
            if (L == 16 * r) { c0 = s0 + c0; c1 = s1 + c1; c2 = s2 + c2; c3 = s3 + c3; }
#pragma unroll
            for (int s = 0; s < 16; s++) {
                a0 = prev_lane<16>(a0, L) + c0; a1 = prev_lane<16>(a1, L) + c1;
                a2 = prev_lane<16>(a2, L) + c2; a3 = prev_lane<16>(a3, L) + c3;
            }
        }
    }
    float h = (a0 + a2) + (a1 + a3);  // sum_all: movehl add, then shuffle add_ss
    float sum = group_bcast<LPI>(h, LPI - 1);
    if (ntail) {  // scalar tail, in index order
        float t0 = group_bcast<LPI>(m0, nfull);
        sum = sum + t0;
        if (ntail > 1) { float t1 = group_bcast<LPI>(m1, nfull); sum = sum + t1; }
        if (ntail > 2) { float t2 = group_bcast<LPI>(m2, nfull); sum = sum + t2; }
    }
    return sum;
}

// ------------------------------------------------------------------ wide rows (256 < num_factor <= 1024)
// A whole wave owns the row, VPL float4 per lane: chunk c (elements 4c..4c+3) sits in lane c % 64, slot c / 64, so a
// row gather is VPL fully coalesced 1 KiB loads.  Only the general kernels are instantiated for wide rows; the
// helpers below are overloads of the float4 ones, so the per-instance code is written once for both (typename R).
template <int VPL>
struct WideRow { float4 v[VPL]; };
template <typename R> struct row_traits;
template <> struct row_traits<float4> {
    static constexpr int VPL = 1;
    static __device__ __forceinline__ float4 zero() { return f4zero(); }
};
template <int V> struct row_traits<WideRow<V>> {
    static constexpr int VPL = V;
    static __device__ __forceinline__ WideRow<V> zero() {
        WideRow<V> r;
#pragma unroll
        for (int v = 0; v < V; v++) r.v[v] = f4zero();
        return r;
    }
};
template <int V> __device__ __forceinline__ void axpy4(WideRow<V> &d, const WideRow<V> &s, float a) {
#pragma unroll
    for (int v = 0; v < V; v++) axpy4(d.v[v], s.v[v], a);
}
template <int V> __device__ __forceinline__ void scale4(WideRow<V> &d, float a) {
#pragma unroll
    for (int v = 0; v < V; v++) scale4(d.v[v], a);
}
__device__ __forceinline__ void sub4(float4 &d, const float4 s) { d.x = d.x - s.x; d.y = d.y - s.y; d.z = d.z - s.z; d.w = d.w - s.w; }  // K5
template <int V> __device__ __forceinline__ void sub4(WideRow<V> &d, const WideRow<V> &s) {
#pragma unroll
    for (int v = 0; v < V; v++) sub4(d.v[v], s.v[v]);
}
__device__ __forceinline__ void l1_row(float4 &w, float th) { w.x = l1(w.x, th); w.y = l1(w.y, th); w.z = l1(w.z, th); w.w = l1(w.w, th); }
template <int V> __device__ __forceinline__ void l1_row(WideRow<V> &w, float th) {
#pragma unroll
    for (int v = 0; v < V; v++) l1_row(w.v[v], th);
}
__device__ __forceinline__ void clamp_nonneg(float4 &w) {  // K7 smaller_then_fill(w, 0)
    if (w.x <= 0.0f) w.x = 0.0f;
    if (w.y <= 0.0f) w.y = 0.0f;
    if (w.z <= 0.0f) w.z = 0.0f;
    if (w.w <= 0.0f) w.w = 0.0f;
}
template <int V> __device__ __forceinline__ void clamp_nonneg(WideRow<V> &w) {
#pragma unroll
    for (int v = 0; v < V; v++) clamp_nonneg(w.v[v]);
}
// K3 for wide rows: the same chain a[c] = a[c-1] + prod[c] over all chunks in index order; slot v is scanned across the
// 64 lanes with wave_shr:1 adds, and the finished sum of slot v-1 (lane 63) is folded into the addend of slot v's
// lane 0 -- the carry trick of the 32-lane groups above, one slot at a time.
template <int LPI, int V>
__device__ __forceinline__ float group_dot(const WideRow<V> &a, const WideRow<V> &b, int L, int k) {
    static_assert(LPI == 64, "wide rows are owned by a whole wave");
    const int nfull = k >> 2;
    const int ntail = k & 3;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    float t0 = 0.0f, t1 = 0.0f, t2 = 0.0f;   // products of the tail chunk
#pragma unroll
    for (int v = 0; v < V; v++) {
        const float m0 = a.v[v].x * b.v[v].x, m1 = a.v[v].y * b.v[v].y, m2 = a.v[v].z * b.v[v].z, m3 = a.v[v].w * b.v[v].w;
        const bool full = L + 64 * v < nfull;
        float c0 = full ? m0 : 0.0f, c1 = full ? m1 : 0.0f, c2 = full ? m2 : 0.0f, c3 = full ? m3 : 0.0f;
        if (v > 0) {
            const float s0 = group_bcast<64>(a0, 63), s1 = group_bcast<64>(a1, 63), s2 = group_bcast<64>(a2, 63), s3 = group_bcast<64>(a3, 63);
            if (L == 0) { c0 = s0 + c0; c1 = s1 + c1; c2 = s2 + c2; c3 = s3 + c3; }
        }
        a0 = 0.0f + c0; a1 = 0.0f + c1; a2 = 0.0f + c2; a3 = 0.0f + c3;
#pragma unroll
        for (int s = 1; s < 64; s++) {
            a0 = prev_lane<64>(a0, L) + c0; a1 = prev_lane<64>(a1, L) + c1;
            a2 = prev_lane<64>(a2, L) + c2; a3 = prev_lane<64>(a3, L) + c3;
        }
        if (ntail && (nfull >> 6) == v) {   // the tail chunk lives in this slot, lane nfull % 64
            t0 = group_bcast<64>(m0, nfull & 63); t1 = group_bcast<64>(m1, nfull & 63); t2 = group_bcast<64>(m2, nfull & 63);
        }
    }
    const float h = (a0 + a2) + (a1 + a3);
    float sum = group_bcast<64>(h, 63);
    if (ntail) {
        sum = sum + t0;
        if (ntail > 1) sum = sum + t1;
        if (ntail > 2) sum = sum + t2;
    }
    return sum;
}

// apex_svd_model.h:112-123
__device__ __forceinline__ float map_active(float sum, int type) {
    if (type == ACT_SIGMOID_L2 || type == ACT_SIGMOID_LIKELIHOOD) return 1.0f / (1.0f + expf(-sum));
    return sum;
}
__device__ __forceinline__ float smooth_hinge_grad(float z) {
    if (z > 1.0f) return 0.0f;
    if (z < 0.0f) return 1.0f;
    return 1.0f - z;
}
// apex_svd_model.h:132-156
__device__ __forceinline__ float cal_grad(float r, float pred, int type) {
    switch (type) {
    case ACT_LINEAR: return r - pred;
    case ACT_SIGMOID_L2: return (r - pred) * pred * (1 - pred);
    case ACT_SIGMOID_LIKELIHOOD: return r - pred;
    case ACT_SIGMOID_QSGRAD:
    case ACT_SIGMOID_RANK: return r - 1.0f / (1.0f + expf(-pred));
    case ACT_HINGE_SMOOTH:
        if (r > 0.5f) return smooth_hinge_grad(pred - 0.5f);
        return -smooth_hinge_grad(0.5f - pred);
    case ACT_HINGE_L2:
        if (r > 0.5f) { if (pred > 1.0f) return 0.0f; return r - pred; }
        if (pred < 0.0f) return 0.0f;
        return r - pred;
    default: return 0.0f;
    }
}
// ParameterSet::get_wd (apex_svd_base.h:69-74); ranges are validated on the host
__device__ __forceinline__ float get_wd(const DevRanges &rg, unsigned id, float dflt) {
    if (rg.n == 0) return dflt;
    int lo = 0, hi = rg.n;
    while (lo < hi) { int mid = (lo + hi) >> 1; if (rg.bound[mid] < id) lo = mid + 1; else hi = mid; }
    return rg.wd[lo < rg.n ? lo : rg.n - 1];
}

// position of a global bias in device memory: contiguous (stride 1) normally; in relaxed-global mode one per 128-byte
// line (stride 32), because atomics to the same line serialise
__device__ __forceinline__ size_t gpos(const DevParams &P, unsigned gid) { return (size_t)gid * (size_t)P.g_stride; }
// factor-row regularisation (reg_user / reg_item, apex_svd_base.h:211-283) on a row held in
// registers.  is_item selects the item flavour of reg_method 3 (L2) and skips the nonneg clamp.
// Lazy modes 4/5 (:225-238, :265-278) take kk = (float)(ref[id] - sample_counter): the reference subtracts two
// UNSIGNED counters, so kk is 0 for an id touched in this very instance and about 4.29e9 otherwise; restated as is.
template <int LPI, typename R>
__device__ __forceinline__ void reg_row(const DevParams &P, R &w, float wd, bool is_item, int L, float kk = 0.0f) {
    const float lambda = P.lr * wd;
    int method = P.reg_method;
    if (method == 3) method = is_item ? 0 : 1;
    if (method == 0) {
        scale4(w, 1.0f - lambda);
    } else if (method == 1) {
        l1_row(w, lambda);
    } else if (method == 2) {  // project(): ||w||^2 <= wd
        float sum = group_dot<LPI>(w, w, L, P.k);
        if (sum > wd) scale4(w, sqrtf(wd / sum));
    } else if (method == 4) {  // lazy L2
        scale4(w, expf(logf(1.0f - lambda) * kk));
    } else if (method == 5) {  // lazy L1
        l1_row(w, lambda * kk);
    }
    if (!is_item && P.user_nonnegative) clamp_nonneg(w);
}
__device__ __forceinline__ float reg_gbias(const DevParams &P, unsigned gid, float g, unsigned counter = 0) {  // :188-210
    float lambda = P.lr * get_wd(P.g_rng, gid, P.wd_global);
    if (gid >= P.num_regfree_global) {
        if (P.reg_global == 0) g = g * (1.0f - lambda);
        else if (P.reg_global == 1) g = l1(g, lambda);
        else {  // 4 lazy L2, 5 lazy L1 (:194-205); regfree ids keep their ref untouched like the reference
            const float kk = (float)(unsigned)(P.ref_global[gid] - counter);
            P.ref_global[gid] = counter;
            if (P.reg_global == 4) g = g * expf(logf(1.0f - lambda) * kk);
            else g = l1(g, lambda * kk);
        }
    }
    return g;
}
// kk of a factor row for the lazy modes; every lane of the group reads the same ref word, then writes the same value
__device__ __forceinline__ float lazy_span(const DevParams &P, unsigned row, unsigned counter) {
    if (P.reg_method < 4) return 0.0f;
    const float kk = (float)(unsigned)(P.ref_ui[row] - counter);
    P.ref_ui[row] = counter;
    return kk;
}

template <int LPI>
__device__ __forceinline__ float4 load_row(const float *W, size_t row, int pitch, int L, int k) {
    if (LPI * 4 > k && L * 4 >= k) return f4zero();
    return *reinterpret_cast<const float4 *>(W + row * (size_t)pitch + (size_t)L * 4);
}
template <int LPI>
__device__ __forceinline__ void store_row(float *W, size_t row, int pitch, int L, int k, const float4 v) {
    if (LPI * 4 > k && L * 4 >= k) return;
    *reinterpret_cast<float4 *>(W + row * (size_t)pitch + (size_t)L * 4) = v;
}

// row store with a cache policy: 0 plain (line stays dirty in the XCD's L2 until the kernel ends),
// 1 nontemporal hint, 2 sc1 write-through (the line leaves L2 as soon as it is written)
typedef float svdf_f4 __attribute__((ext_vector_type(4)));
template <int LPI>
__device__ __forceinline__ void store_row_policy(float *W, size_t row, int pitch, int L, int k, const float4 v, int mode) {
    if (LPI * 4 > k && L * 4 >= k) return;
    float *ptr = W + row * (size_t)pitch + (size_t)L * 4;
    if (mode == 0) {
        *reinterpret_cast<float4 *>(ptr) = v;
    } else {
        svdf_f4 x = {v.x, v.y, v.z, v.w};
        if (mode == 1) __builtin_nontemporal_store(x, reinterpret_cast<svdf_f4 *>(ptr));
        else asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(ptr), "v"(x) : "memory");
    }
}

// row load / store by row type (float4: one lane group per row; WideRow: the whole wave, VPL slots)
template <int LPI, typename R> struct row_io;
template <int LPI> struct row_io<LPI, float4> {
    static __device__ __forceinline__ float4 load(const float *W, size_t row, int pitch, int L, int k) { return load_row<LPI>(W, row, pitch, L, k); }
    static __device__ __forceinline__ void store(float *W, size_t row, int pitch, int L, int k, const float4 &v) { store_row<LPI>(W, row, pitch, L, k, v); }
};
template <int LPI, int V> struct row_io<LPI, WideRow<V>> {
    static_assert(LPI == 64, "wide rows are owned by a whole wave");
    static __device__ __forceinline__ WideRow<V> load(const float *W, size_t row, int pitch, int L, int k) {
        WideRow<V> r;
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int e = 4 * (L + 64 * v);
            r.v[v] = e < k ? *reinterpret_cast<const float4 *>(W + row * (size_t)pitch + (size_t)e) : f4zero();
        }
        return r;
    }
    static __device__ __forceinline__ void store(float *W, size_t row, int pitch, int L, int k, const WideRow<V> &r) {
#pragma unroll
        for (int v = 0; v < V; v++) {
            const int e = 4 * (L + 64 * v);
            if (e < k) *reinterpret_cast<float4 *>(W + row * (size_t)pitch + (size_t)e) = r.v[v];
        }
    }
};

// =====================================================================================
// Kernel 1: basicMF fused SGD step -- no global feature, one user id, one item id, no side
// tables, distinct user/item rows.  One lane group per instance, G groups per wave in flight.
// Traffic per instance (k=64): 2 x 256 B row reads + 2 x 256 B row writes + 2 bias RMW +
// 12..20 B of schedule = the 1072 B/instance algorithmic figure of SURVEY.md 8(d4).
// =====================================================================================
// One wave's G x (64/LPI) instances.  Compile-time switches strip what the contract workload never needs from the
// instruction stream (the kernel is not only HBM- but also VALU-issue-bound: a level is about one occupancy wave, so every
// SIMD runs its ~5 waves' instructions once per launch, ~40 % of the launch time):
//   FULL      num_factor == 4*LPI: no per-lane bounds test around row loads / stores, no masked chunks / tail in the dot
//   FAST      linear link, reg_method 0, user bias on, no per-range decay, no clamp, plain stores: no per-row switches
// (A separate unpredicated path for waves whose slots are all occupied was tried too: it costs 103 instead of 74 VGPRs at
// G=4 and measured slower.)
template <int LPI, int G, bool UNITVAL, bool FULL, bool FAST>
__device__ __forceinline__ void basicmf_wave(const DevParams &P, const BasicSchedule &S, long begin, long end, long w0, int L, int gslot) {
    constexpr int IPW = 64 / LPI;  // instances per wave per group slot
    const int pitch = P.pitch;
    const int k = FULL ? 4 * LPI : P.k;
    const bool use_ubias = FAST ? true : P.no_user_bias == 0;

    bool valid[G];
    unsigned ur[G], ir[G];
    float label[G], ua[G], ia[G], bu[G], bi[G];
    float4 p[G], q[G];

    // ---- stage 1: schedule records (coalesced, one address per lane group)
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long s = w0 + (long)g * IPW + gslot;
        valid[g] = s < end;
        const long sc = valid[g] ? s : begin;
        ur[g] = P.user_off + S.user[sc];
        ir[g] = P.item_off + S.item[sc];
        label[g] = S.label[sc];
        ua[g] = UNITVAL ? 1.0f : S.uval[sc];
        ia[g] = UNITVAL ? 1.0f : S.ival[sc];
    }
    // ---- stage 2: all row gathers of the wave issued back to back
#pragma unroll
    for (int g = 0; g < G; g++) {
        p[g] = f4zero(); q[g] = f4zero(); bu[g] = 0.0f; bi[g] = 0.0f;
        if (valid[g]) {
            p[g] = load_row<LPI>(P.W, ur[g], pitch, L, k);
            q[g] = load_row<LPI>(P.W, ir[g], pitch, L, k);
            if (use_ubias) bu[g] = P.bias[ur[g]];
            bi[g] = P.bias[ir[g]];
        }
    }
    // row-invariant decay factors of the FAST configuration (L2: W *= 1 - lr*wd, multiply skipped when that is 1)
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
    // ---- stage 3: score, gradient, fused update + decay, scatter
#pragma unroll
    for (int g = 0; g < G; g++) {
        // calc_bias (:313-353) in double; "+ 0.0" terms are the svdpp / plugin hooks returning 0.0f
        double bs = 0.0;
        if (use_ubias) { bs += (double)(ua[g] * bu[g]); bs += 0.0; }
        bs += 0.0;
        bs += (double)(ia[g] * bi[g]);
        double sum = (double)P.base_score + bs;
        // prepare_tmp (:354-381): tmp = 0 + row*val
        float4 tu = f4zero(), ti = f4zero();
        axpy4(tu, p[g], ua[g]);
        axpy4(ti, q[g], ia[g]);
        sum += (double)group_dot<LPI>(tu, ti, L, k);
        const float pred = FAST ? (float)sum : map_active((float)sum, P.active_type);
        const float err = (FAST ? label[g] - pred : cal_grad(label[g], pred, P.active_type)) * 1.0f;
        // update_no_decay (:383-427): both rows use the pre-update snapshots
        const float su = P.lr * err * ua[g];
        const float si = P.lr * err * ia[g];
        float4 wu = p[g], wi = q[g];
        axpy4(wu, ti, su);
        axpy4(wi, tu, si);
        float nbu = bu[g] + su, nbi = bi[g] + si;
        // regularize(feature, true) (:286-311)
        if (FAST) {
            wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
            wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
        } else {
            reg_row<LPI>(P, wu, get_wd(P.u_rng, ur[g] - P.user_off, P.wd_user), false, L);
            reg_row<LPI>(P, wi, get_wd(P.i_rng, ir[g] - P.item_off, P.wd_item), true, L);
        }
        nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
        nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
        if (valid[g]) {
            if (FAST) {
                store_row<LPI>(P.W, ur[g], pitch, L, k, wu);
                store_row<LPI>(P.W, ir[g], pitch, L, k, wi);
                // every lane of the group writes the same bias word: one request, no exec-mask branch
                P.bias[ur[g]] = nbu;
                P.bias[ir[g]] = nbi;
            } else {
                store_row_policy<LPI>(P.W, ur[g], pitch, L, k, wu, P.store_mode);
                store_row_policy<LPI>(P.W, ir[g], pitch, L, k, wi, P.store_mode);
                if (L == 0) {
                    if (use_ubias) P.bias[ur[g]] = nbu;
                    P.bias[ir[g]] = nbi;
                }
            }
        }
    }
}

template <int LPI, int G, bool UNITVAL, bool FULL, bool FAST>
__global__ __launch_bounds__(256) void k_basicmf(const DevParams P, const BasicSchedule S, long begin, long end) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    // Hardware deals workgroups to the 8 XCDs round-robin (blockIdx % 8).  With xcd_remap the grid is a multiple
    // of 8 and XCD x works on the x-th contiguous eighth of the batch, so (the batch being sorted by item id)
    // neighbouring item rows / bias sectors meet in ONE XCD's L2 instead of eight.
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = begin + wave * (long)(G * IPW);
    if (w0 < end) basicmf_wave<LPI, G, UNITVAL, FULL, FAST>(P, S, begin, end, w0, L, gslot);
}

// =====================================================================================
// Kernel 1b: fused step for "few-row" instances (<= NU user ids, <= NI item ids, any number of global
// features, no id repeated inside an instance, no side tables).  Same structure as k_basicmf: every row
// of the wave's G instances is gathered up front, the update and all regularisers are applied in
// registers, rows are written once.  Because the ids of an instance are distinct, "update every row,
// then regularise every row" (apex_svd_base.h:456-462) equals "update+regularise row by row".
// =====================================================================================
// relaxed shared ids: W[row] += (now - was) with hardware float atomics (agent scope: coherent across the XCDs' L2s)
template <int LPI>
__device__ __forceinline__ void atomic_row_add(float *W, size_t row, int pitch, int L, int k, const float4 now, const float4 was) {
    if (LPI * 4 > k && L * 4 >= k) return;
    float *ptr = W + row * (size_t)pitch + (size_t)L * 4;
    unsafeAtomicAdd(ptr + 0, now.x - was.x);
    unsafeAtomicAdd(ptr + 1, now.y - was.y);
    unsafeAtomicAdd(ptr + 2, now.z - was.z);
    unsafeAtomicAdd(ptr + 3, now.w - was.w);
}
// HOTU (relaxed mode, NU == 2, G == 1, up to 1024 threads): the LAST user slot holds a shared side-feature id.  A batch is
// sorted by that id, so the instances of a workgroup mostly share it: their changes of the shared row are summed in LDS
// (fixed order inside the workgroup) and the leader of each run issues ONE set of atomics -- 32x fewer contended atomics
// per shared row at k=128 than one set per instance.
template <int LPI, int NU, int NI, int G, bool FULL, bool HOTU = false>   // FULL: num_factor == 4*LPI, see basicmf_wave
__global__ __launch_bounds__(HOTU ? 1024 : 256) void k_fused(const DevParams P, const FusedSchedule S, long begin, long end) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    long tile = blockIdx.x;   // XCD-aware tile mapping, see k_basicmf
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = begin + wave * (long)(G * IPW);
    const int k = FULL ? 4 * LPI : P.k, pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;

    bool valid[G];
    unsigned ur[G][NU], ir[G][NI];
    float label[G], ua[G][NU], ia[G][NI], bu[G][NU], bi[G][NI];
    int g0[G], g1[G];
    float4 p[G][NU], q[G][NI];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long s = w0 + (long)g * IPW + gslot;
        valid[g] = s < end;
        const long sc = valid[g] ? s : begin;
        label[g] = S.label[sc];
#pragma unroll
        for (int a = 0; a < NU; a++) { ur[g][a] = valid[g] ? S.uidx[a][sc] : (unsigned)SLOT_ABSENT; ua[g][a] = S.uval[a][sc]; }
#pragma unroll
        for (int b = 0; b < NI; b++) { ir[g][b] = valid[g] ? S.iidx[b][sc] : (unsigned)SLOT_ABSENT; ia[g][b] = S.ival[b][sc]; }
        g0[g] = 0; g1[g] = 0;
        if (S.gptr && valid[g]) { g0[g] = S.gptr[sc]; g1[g] = S.gptr[sc + 1]; }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
            p[g][a] = f4zero(); bu[g][a] = 0.0f;
            if (ur[g][a] != SLOT_ABSENT) {
                p[g][a] = load_row<LPI>(P.W, P.user_off + ur[g][a], pitch, L, k);
                if (use_ubias) bu[g][a] = P.bias[P.user_off + ur[g][a]];
            }
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            q[g][b] = f4zero(); bi[g][b] = 0.0f;
            if (ir[g][b] != SLOT_ABSENT) {
                q[g][b] = load_row<LPI>(P.W, P.item_off + ir[g][b], pitch, L, k);
                bi[g][b] = P.bias[P.item_off + ir[g][b]];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        double bs = 0.0;
        for (int j = g0[g]; j < g1[g]; j++) bs += (double)(S.gval[j] * P.g_bias[gpos(P, S.gidx[j])]);
        if (use_ubias) {
#pragma unroll
            for (int a = 0; a < NU; a++) if (ur[g][a] != SLOT_ABSENT) bs += (double)(ua[g][a] * bu[g][a]);
        }
#pragma unroll
        for (int b = 0; b < NI; b++) if (ir[g][b] != SLOT_ABSENT) bs += (double)(ia[g][b] * bi[g][b]);
        double sum = (double)P.base_score + bs;
        float4 tu = f4zero(), ti = f4zero();
#pragma unroll
        for (int a = 0; a < NU; a++) if (ur[g][a] != SLOT_ABSENT) axpy4(tu, p[g][a], ua[g][a]);
#pragma unroll
        for (int b = 0; b < NI; b++) if (ir[g][b] != SLOT_ABSENT) axpy4(ti, q[g][b], ia[g][b]);
        sum += (double)group_dot<LPI>(tu, ti, L, k);
        const float pred = map_active((float)sum, P.active_type);
        const float err = cal_grad(label[g], pred, P.active_type) * 1.0f;
        const float lr = P.lr;
        // global biases go through memory in the reference's order (all updates, then all decays), so a
        // global id listed twice behaves like the reference; every lane stores the same value
        if (P.relax_global) {
            // relaxed shared ids: other instances of this launch may be updating the same global -- add this instance's
            // change (update, then decay of the value it read) atomically; one lane per group
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gid = S.gidx[j];
                const float gb = P.g_bias[gpos(P, gid)];
                const float nb = reg_gbias(P, gid, gb + lr * err * S.gval[j]);
                if (L == 0) unsafeAtomicAdd(&P.g_bias[gpos(P, gid)], nb - gb);
            }
        } else {
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gid = S.gidx[j];
                float gb = P.g_bias[gpos(P, gid)];
                gb = gb + lr * err * S.gval[j];
                P.g_bias[gpos(P, gid)] = gb;
            }
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gid = S.gidx[j];
                P.g_bias[gpos(P, gid)] = reg_gbias(P, gid, P.g_bias[gpos(P, gid)]);
            }
        }
        // HOTU: this instance's change of the shared row in the last user slot, handed to the workgroup reduction below
        bool hot_here = false;
        unsigned hot_row = SLOT_ABSENT;
        float4 hot_delta = f4zero();
        float hot_db = 0.0f;
#pragma unroll
        for (int a = 0; a < NU; a++) {
            if (ur[g][a] == SLOT_ABSENT) continue;
            const float su = lr * err * ua[g][a];
            float4 w = p[g][a];
            axpy4(w, ti, su);
            float nb = bu[g][a] + su;
            reg_row<LPI>(P, w, get_wd(P.u_rng, ur[g][a], P.wd_user), false, L);
            nb = nb * (1.0f - lr * P.wd_user_bias);
            if (HOTU && a == NU - 1 && ur[g][a] >= P.relax_user_from) {
                hot_here = true;
                hot_row = P.user_off + ur[g][a];
                hot_delta = make_float4(w.x - p[g][a].x, w.y - p[g][a].y, w.z - p[g][a].z, w.w - p[g][a].w);
                hot_db = nb - bu[g][a];
                continue;
            }
            if (ur[g][a] >= P.relax_user_from) {   // relaxed shared id: add the change, element by element
                atomic_row_add<LPI>(P.W, P.user_off + ur[g][a], pitch, L, k, w, p[g][a]);
                if (use_ubias && L == 0) unsafeAtomicAdd(&P.bias[P.user_off + ur[g][a]], nb - bu[g][a]);
                continue;
            }
            store_row<LPI>(P.W, P.user_off + ur[g][a], pitch, L, k, w);
            if (use_ubias && L == 0) P.bias[P.user_off + ur[g][a]] = nb;
        }
        if constexpr (HOTU) {
            extern __shared__ float4 hot_lds[];                       // [BI][LPI] deltas, then BI row ids, BI bias deltas
            const int BI = (int)(blockDim.x >> 6) * IPW;              // instances per workgroup
            unsigned *hot_ids = reinterpret_cast<unsigned *>(hot_lds + BI * LPI);
            float *hot_dbs = reinterpret_cast<float *>(hot_ids + BI);
            const int e = (int)(threadIdx.x >> 6) * IPW + gslot;
            hot_lds[e * LPI + L] = hot_delta;
            if (L == 0) { hot_ids[e] = hot_here ? hot_row : (unsigned)SLOT_ABSENT; hot_dbs[e] = hot_db; }
            __syncthreads();
            if (hot_here && (e == 0 || hot_ids[e - 1] != hot_row)) {   // first instance of a run of the same shared row
                float4 sum = hot_delta;
                float sb = hot_db;
                for (int e2 = e + 1; e2 < BI && hot_ids[e2] == hot_row; e2++) {
                    const float4 d = hot_lds[e2 * LPI + L];
                    sum.x = sum.x + d.x; sum.y = sum.y + d.y; sum.z = sum.z + d.z; sum.w = sum.w + d.w;
                    sb = sb + hot_dbs[e2];
                }
                if (!(LPI * 4 > k && L * 4 >= k)) {
                    float *ptr = P.W + (size_t)hot_row * pitch + (size_t)L * 4;
                    unsafeAtomicAdd(ptr + 0, sum.x); unsafeAtomicAdd(ptr + 1, sum.y);
                    unsafeAtomicAdd(ptr + 2, sum.z); unsafeAtomicAdd(ptr + 3, sum.w);
                }
                if (use_ubias && L == 0) unsafeAtomicAdd(&P.bias[hot_row], sb);
            }
            __syncthreads();
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            if (ir[g][b] == SLOT_ABSENT) continue;
            const float si = lr * err * ia[g][b];
            float4 w = q[g][b];
            axpy4(w, tu, si);
            float nb = bi[g][b] + si;
            reg_row<LPI>(P, w, get_wd(P.i_rng, ir[g][b], P.wd_item), true, L);
            nb = nb * (1.0f - lr * P.wd_item_bias);
            if (ir[g][b] >= P.relax_item_from) {
                atomic_row_add<LPI>(P.W, P.item_off + ir[g][b], pitch, L, k, w, q[g][b]);
                if (L == 0) unsafeAtomicAdd(&P.bias[P.item_off + ir[g][b]], nb - bi[g][b]);
                continue;
            }
            store_row<LPI>(P.W, P.item_off + ir[g][b], pitch, L, k, w);
            if (L == 0) P.bias[P.item_off + ir[g][b]] = nb;
        }
    }
}

// read-only scoring of a fused schedule (out[s] in schedule order)
template <int LPI, int NU, int NI>
__global__ __launch_bounds__(256) void k_predict_fused(const DevParams P, const FusedSchedule S, long n, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = gidx; s < n; s += stride) {
        double bs = 0.0;
        if (S.gptr) for (int j = S.gptr[s]; j < S.gptr[s + 1]; j++) bs += (double)(S.gval[j] * P.g_bias[gpos(P, S.gidx[j])]);
        float4 tu = f4zero(), ti = f4zero();
#pragma unroll
        for (int a = 0; a < NU; a++) {
            const unsigned u = S.uidx[a][s];
            if (u == SLOT_ABSENT) continue;
            const float v = S.uval[a][s];
            if (P.no_user_bias == 0) bs += (double)(v * P.bias[P.user_off + u]);
            axpy4(tu, load_row<LPI>(P.W, P.user_off + u, P.pitch, L, P.k), v);
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            const unsigned i = S.iidx[b][s];
            if (i == SLOT_ABSENT) continue;
            const float v = S.ival[b][s];
            bs += (double)(v * P.bias[P.item_off + i]);
            axpy4(ti, load_row<LPI>(P.W, P.item_off + i, P.pitch, L, P.k), v);
        }
        double sum = (double)P.base_score + bs;
        sum += (double)group_dot<LPI>(tu, ti, L, P.k);
        if (L == 0) out[s] = map_active((float)sum, P.active_type);
    }
}

// read-only scoring of a basicMF schedule (out[s] in schedule order)
template <int LPI, bool UNITVAL>
__global__ __launch_bounds__(256) void k_predict_basic(const DevParams P, const BasicSchedule S, long n, float *out) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const long gidx = ((long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * IPW + lane / LPI;
    const long stride = (long)gridDim.x * (blockDim.x >> 6) * IPW;
    for (long s = gidx; s < n; s += stride) {
        const unsigned ur = P.user_off + S.user[s], ir = P.item_off + S.item[s];
        const float ua = UNITVAL ? 1.0f : S.uval[s], ia = UNITVAL ? 1.0f : S.ival[s];
        double bs = 0.0;
        if (P.no_user_bias == 0) bs += (double)(ua * P.bias[ur]);
        bs += (double)(ia * P.bias[ir]);
        double sum = (double)P.base_score + bs;
        float4 tu = f4zero(), ti = f4zero();
        axpy4(tu, load_row<LPI>(P.W, ur, P.pitch, L, P.k), ua);
        axpy4(ti, load_row<LPI>(P.W, ir, P.pitch, L, P.k), ia);
        sum += (double)group_dot<LPI>(tu, ti, L, P.k);
        if (L == 0) out[s] = map_active((float)sum, P.active_type);
    }
}

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

// ---- fast path for "simple" units (host-verified, UNIT_SIMPLE): every row is (no global, ONE user id -- the same for
// the whole unit --, one item id), the unit's item ids are pairwise distinct, its feedback ids are pairwise distinct,
// no side tables, separate feedback/user/item row spaces.
//
// A user's rows are a strict recurrence (p_u, tmp_ufeedback -> err -> p_u, tmp_ufeedback), and exact sequential
// semantics leave only a handful of users per conflict-free batch, so this path is LATENCY-bound: what counts is the
// number of dependent instructions per row, not bytes.  Layout for that: ONE WAVE PER USER, one element per lane and
// register, arranged so that the reference's four SSE accumulation chains (elements j, j+4, j+8, ... for j = 0..3)
// each live in their own 16-lane DPP row:
//      lane = 16*j + m,  register q   <->   element 4*(NR*m + q) + j        (NR = registers per row = ceil(k/64))
// i.e. lane m of a DPP row holds the NR consecutive chunks NR*m .. NR*m+NR-1 of its chain.  The whole dot product
// is then 15 steps of ONE v_add_f32_dpp row_shr:1 (all four chains at once) followed by NR-1 plain adds inside the
// lane -- a dependent DPP add costs ~19 cycles, a plain one ~8, so wide rows pay 15 slow steps, not 16*NR-1 --
// against 4 instructions per step and bpermute carries in the float4-per-lane layout; every elementwise op (axpy,
// decay, L1 ...) is k/64 instructions instead of 4.  Measured on MI355X (tools/svdpp_latency2.py): DESIGN.md section 5.
//   * the user's factor row, bias and the feedback state stay in registers for the whole unit,
//   * item rows (and their records, via scalar loads: everything about a row is wave-uniform) are fetched
//     SVDPP_PFW rows ahead, item rows are written once, fire and forget,
//   * feedback rows are gathered / scattered a batch (16 or 32) at a time, the next batch in flight meanwhile
//     (accumulation order unchanged).
constexpr int SVDPP_PFW = 8;   // rows fetched ahead (double-buffered: 8..16 rows = 2..4 us of lookahead)
// feedback rows per gather / scatter batch; two batches are in flight (HBM + translation latency is ~2 us, a batch of
// 16 accumulates in ~0.4 us)
template <int NR> struct svdpp_fbw { static constexpr int value = 16; };   // 32 measured slower (VGPRs spill to AGPRs)
// batches of feedback rows in flight ahead of the one being accumulated / scattered (2 measured no faster than 1:
// 64.3 vs 62.1 us for a unit of 100 rows + 100 ids at k=128 -- the phase is bound by the loads' issue, not their latency)
template <int NR> struct svdpp_fbdepth { static constexpr int value = 1; };

template <int NR>
struct ChainRow { float r[NR]; };

__device__ __forceinline__ float dpp_row_shr1(float v) {   // lane m <- lane m-1 of its 16-lane row, 0 into m = 0
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane)); }
__device__ __forceinline__ float wave_uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Load of read-only launch data (schedule records, unit descriptors, feedback lists) at a wave-uniform address through
// the constant address space: the backend may then use the scalar unit (s_load, lgkmcnt) instead of a vector load
// with 64 identical addresses -- the kernel never writes these arrays, which it cannot prove by itself because the
// parameter tables it does write are reachable through plain pointers too.  Keeps vmcnt for the row traffic.
template <typename T> __device__ __forceinline__ T uniform_load(const T *p) {
    typedef const T __attribute__((address_space(4))) * cptr;
    return *reinterpret_cast<cptr>(reinterpret_cast<uintptr_t>(p));
}
template <int NR> __device__ __forceinline__ ChainRow<NR> chain_zero() {
    ChainRow<NR> z;
#pragma unroll
    for (int q = 0; q < NR; q++) z.r[q] = 0.0f;
    return z;
}
// k < 0 tells the row is FULL (num_factor == 64*NR, the usual case): no per-lane bounds test, hence no exec-mask branch
// around every load and store of the instruction-bound row loop
template <int NR> __device__ __forceinline__ ChainRow<NR> chain_load(const float *W, size_t row, int pitch, int lane, int k) {
    const float *base = W + row * (size_t)pitch;
    const int e0 = 4 * NR * (lane & 15) + (lane >> 4);
    ChainRow<NR> x;
#pragma unroll
    for (int q = 0; q < NR; q++) { const int e = e0 + 4 * q; x.r[q] = (k < 0 || e < k) ? base[e] : 0.0f; }
    return x;
}
template <int NR> __device__ __forceinline__ void chain_store(float *W, size_t row, int pitch, int lane, int k, const ChainRow<NR> &x) {
    float *base = W + row * (size_t)pitch;
    const int e0 = 4 * NR * (lane & 15) + (lane >> 4);
#pragma unroll
    for (int q = 0; q < NR; q++) { const int e = e0 + 4 * q; if (k < 0 || e < k) base[e] = x.r[q]; }
}
// relaxed shared rows: W[row] += x, element by element, with hardware float atomics (chain layout addresses)
template <int NR> __device__ __forceinline__ void chain_atomic_add(float *W, size_t row, int pitch, int lane, int k, const ChainRow<NR> &x) {
    float *base = W + row * (size_t)pitch;
    const int e0 = 4 * NR * (lane & 15) + (lane >> 4);
#pragma unroll
    for (int q = 0; q < NR; q++) { const int e = e0 + 4 * q; if (k < 0 || e < k) unsafeAtomicAdd(base + e, x.r[q]); }
}
// The same registers in LINEAR layout: lane l holds the NR consecutive elements NR*l .. NR*l+NR-1, i.e. a row is ONE fully
// contiguous 256*NR-byte load or store per wave.  The feedback phases (prepare_ufeedback / update_ufeedback) only do
// elementwise work on hundreds of rows, which is layout-agnostic: they run in linear layout and convert the one row that
// crosses into the chain-layout row loop (tmp_ufeedback, the scatter delta) with NR*NR ds_bpermutes per phase.
template <int NR> __device__ __forceinline__ ChainRow<NR> lin_load(const float *W, size_t row, int pitch, int lane, int k) {
    const float *base = W + row * (size_t)pitch + NR * lane;
    ChainRow<NR> x;
    if (k < 0) {
        if constexpr (NR == 1) x.r[0] = base[0];
        else if constexpr (NR == 2) { const float2 t = *reinterpret_cast<const float2 *>(base); x.r[0] = t.x; x.r[1] = t.y; }
        else if constexpr (NR == 4) { const float4 t = *reinterpret_cast<const float4 *>(base); x.r[0] = t.x; x.r[1] = t.y; x.r[2] = t.z; x.r[3] = t.w; }
        else {
#pragma unroll
            for (int c = 0; c < NR; c++) x.r[c] = base[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < NR; c++) x.r[c] = (NR * lane + c < k) ? base[c] : 0.0f;
    }
    return x;
}
template <int NR> __device__ __forceinline__ void lin_store(float *W, size_t row, int pitch, int lane, int k, const ChainRow<NR> &x) {
    float *base = W + row * (size_t)pitch + NR * lane;
    if (k < 0) {
        if constexpr (NR == 1) base[0] = x.r[0];
        else if constexpr (NR == 2) *reinterpret_cast<float2 *>(base) = make_float2(x.r[0], x.r[1]);
        else if constexpr (NR == 4) *reinterpret_cast<float4 *>(base) = make_float4(x.r[0], x.r[1], x.r[2], x.r[3]);
        else {
#pragma unroll
            for (int c = 0; c < NR; c++) base[c] = x.r[c];
        }
    } else {
#pragma unroll
        for (int c = 0; c < NR; c++) if (NR * lane + c < k) base[c] = x.r[c];
    }
}
template <int NR> __device__ __forceinline__ void lin_atomic_add(float *W, size_t row, int pitch, int lane, int k, const ChainRow<NR> &x) {
    float *base = W + row * (size_t)pitch + NR * lane;
#pragma unroll
    for (int c = 0; c < NR; c++) if (k < 0 || NR * lane + c < k) unsafeAtomicAdd(base + c, x.r[c]);
}
__device__ __forceinline__ float lane_gather(float v, int src_lane) {
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
// chain layout (lane 16j+m, register q <-> element 4*(NR*m+q)+j)  <->  linear layout (lane l, slot c <-> element NR*l+c)
template <int NR> __device__ __forceinline__ ChainRow<NR> lin_to_chain(const ChainRow<NR> &lin, int lane) {
    ChainRow<NR> out;
#pragma unroll
    for (int q = 0; q < NR; q++) {
        const int e = 4 * (NR * (lane & 15) + q) + (lane >> 4);
        const int src = e / NR, comp = e % NR;
        float v = lane_gather(lin.r[0], src);
#pragma unroll
        for (int c = 1; c < NR; c++) { const float t = lane_gather(lin.r[c], src); v = (comp == c) ? t : v; }
        out.r[q] = v;
    }
    return out;
}
template <int NR> __device__ __forceinline__ ChainRow<NR> chain_to_lin(const ChainRow<NR> &ch, int lane) {
    ChainRow<NR> out;
#pragma unroll
    for (int c = 0; c < NR; c++) {
        const int e = NR * lane + c;
        const int chunk = e >> 2;
        const int src = 16 * (e & 3) + chunk / NR, qsel = chunk % NR;
        float v = lane_gather(ch.r[0], src);
#pragma unroll
        for (int q = 1; q < NR; q++) { const float t = lane_gather(ch.r[q], src); v = (qsel == q) ? t : v; }
        out.r[c] = v;
    }
    return out;
}
// K1 / K2 with a wave-uniform scalar
template <int NR> __device__ __forceinline__ void chain_axpy(ChainRow<NR> &d, const ChainRow<NR> &s, float a) {
    const float a1 = snap_to_one(a);
#pragma unroll
    for (int q = 0; q < NR; q++) { const float m = s.r[q] * a1; d.r[q] = d.r[q] + m; }
}
template <int NR> __device__ __forceinline__ void chain_scale(ChainRow<NR> &d, float a) {
    const float a1 = snap_to_one(a);
#pragma unroll
    for (int q = 0; q < NR; q++) d.r[q] = d.r[q] * a1;
}
// K3 in the chain layout; the result is wave-uniform.  Lane m adds its NR chunks, in order, to the running sum handed
// over by lane m-1; all lanes run the 15 hand-over steps (lanes below the step index are final already and recompute the
// same value), so there is no select and no cross-register carry.
template <int NR> __device__ __forceinline__ float chain_dot(const ChainRow<NR> &a, const ChainRow<NR> &b, int lane, int k) {
    const int nfull = k >> 2, ntail = k & 3, m = lane & 15;
    float prod[NR], c[NR];
#pragma unroll
    for (int q = 0; q < NR; q++) {
        prod[q] = a.r[q] * b.r[q];
        c[q] = (NR * m + q < nfull) ? prod[q] : 0.0f;   // chunks beyond the full ones feed +0, the sums travel on
    }
    float acc = 0.0f + c[0];
#pragma unroll
    for (int q = 1; q < NR; q++) acc = acc + c[q];
#pragma unroll
    for (int s = 1; s < 16; s++) {
        acc = dpp_row_shr1(acc) + c[0];
#pragma unroll
        for (int q = 1; q < NR; q++) acc = acc + c[q];
    }
    const float s0 = lane_value(acc, 15), s1 = lane_value(acc, 31), s2 = lane_value(acc, 47), s3 = lane_value(acc, 63);
    float sum = (s0 + s2) + (s1 + s3);   // sum_all: movehl add, then shuffle add_ss
    if (ntail) {                         // scalar tail, in index order: chunk nfull = lane nfull / NR, register nfull % NR
        float pt = prod[0];
#pragma unroll
        for (int q = 1; q < NR; q++) pt = (nfull % NR) == q ? prod[q] : pt;
        const int tm = nfull / NR;
        sum = sum + lane_value(pt, tm);
        if (ntail > 1) sum = sum + lane_value(pt, 16 + tm);
        if (ntail > 2) sum = sum + lane_value(pt, 32 + tm);
    }
    return sum;
}
// reg_user / reg_item on a row in registers (reg modes 0..3; lazy modes never reach the fast path)
template <int NR> __device__ __forceinline__ void chain_reg(const DevParams &P, ChainRow<NR> &w, float wd, bool is_item, int lane, int k) {
    const float lambda = P.lr * wd;
    int method = P.reg_method;
    if (method == 3) method = is_item ? 0 : 1;
    if (method == 0) {
        chain_scale(w, 1.0f - lambda);
    } else if (method == 1) {
#pragma unroll
        for (int q = 0; q < NR; q++) w.r[q] = l1(w.r[q], lambda);
    } else if (method == 2) {
        const float sum = chain_dot(w, w, lane, k);
        if (sum > wd) chain_scale(w, sqrtf(wd / sum));
    }
    if (!is_item && P.user_nonnegative) {
#pragma unroll
        for (int q = 0; q < NR; q++) if (w.r[q] <= 0.0f) w.r[q] = 0.0f;
    }
}
// Records of a user's rows, 64 rows at a time, one row per lane (same scheme as FbBlock below): label, item id, the
// re-read flag and -- unless the unit-value specialisation applies -- the two feature values
struct RowBlock {
    float label, uv, iv;
    unsigned item;
    int fresh;
};
template <bool UV>
__device__ __forceinline__ RowBlock row_block(const DevCSR &D, int row_begin, int e0, int first, int nrow, int lane) {
    const int j = min(first + lane, nrow - 1);   // rows beyond the unit's end repeat its last row (fetched, never used)
    RowBlock b;
    b.label = D.row_label[row_begin + j];
    b.item = D.feat_index[e0 + 2 * j + 1];
    b.fresh = D.row_fresh ? (int)D.row_fresh[row_begin + j] : 0;
    b.uv = UV ? 1.0f : D.feat_value[e0 + 2 * j];
    b.iv = UV ? 1.0f : D.feat_value[e0 + 2 * j + 1];
    return b;
}
__device__ __forceinline__ float pick(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int pick(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ unsigned pick(unsigned v, int l) { return (unsigned)__builtin_amdgcn_readlane((int)v, l); }
template <int NR>
struct ChainRowPF {   // the fetched-ahead part of one row: the item's factor row and bias (its id for the store)
    ChainRow<NR> q;
    float bi;
    unsigned irow;
};
// the SVDPP_PFW rows at offsets off .. off+SVDPP_PFW-1 of block b
template <int NR>
__device__ __forceinline__ void chain_fetch_rows(const DevParams &P, const RowBlock &b, int off, int lane, int kio, ChainRowPF<NR> (&o)[SVDPP_PFW]) {
#pragma unroll
    for (int c = 0; c < SVDPP_PFW; c++) {
        o[c].irow = P.item_off + pick(b.item, off + c);
        o[c].q = chain_load<NR>(P.W, o[c].irow, P.pitch, lane, kio);
        o[c].bi = P.bias[o[c].irow];
    }
}

// Feedback ids and values of a user, 64 at a time: lane l of the wave holds entry first + l (clamped to the last one), loaded
// with ONE coalesced vector load per 64 entries; an entry is picked with v_readlane right where it is used.  (Fetching
// them one by one through the scalar unit serialises: each s_load result was spilled to a VGPR lane behind its own
// lgkmcnt(0) wait -- three batches of ids do not fit the SGPR file -- 0.12 us per id.)
struct FbBlock {
    unsigned id;   // this lane's feedback id
    float v;       // and its value
};
__device__ __forceinline__ FbBlock fb_block(const unsigned *fidx, const float *fval, int first, int nfb, int lane) {
    const int j = min(first + lane, nfb - 1);
    FbBlock b;
    b.id = fidx[j];
    b.v = fval[j];
    return b;
}
__device__ __forceinline__ unsigned fb_id(const FbBlock &b, int l) { return (unsigned)__builtin_amdgcn_readlane((int)b.id, l); }
__device__ __forceinline__ float fb_val(const FbBlock &b, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b.v), l)); }
// rows of one batch (entries off .. off+FBW-1 of block b); indices past the end were clamped when the block was loaded:
// such rows are fetched but never accumulated or stored
template <int NR> struct FbRows { ChainRow<NR> w[svdpp_fbw<NR>::value]; float b[svdpp_fbw<NR>::value]; };
template <int NR>
__device__ __forceinline__ void fb_fetch_rows(const DevParams &P, const FbBlock &blk, int off, bool ub, int lane, int kio, FbRows<NR> &o) {
#pragma unroll
    for (int c = 0; c < svdpp_fbw<NR>::value; c++) {
        const unsigned row = P.fb_off + fb_id(blk, off + c);
        o.w[c] = lin_load<NR>(P.W, row, P.pitch, lane, kio);   // LINEAR layout: one contiguous load per row
        o.b[c] = ub ? P.bias[row] : 0.0f;
    }
}

// one simple unit, start to end, by one wave (u and everything derived from it is wave-uniform).
// FAST: the configuration of every BASELINE run -- linear link, L2 decay (reg_method 0), user bias on, no per-range
// decay, no nonnegativity clamp -- compiled without the per-row switches; anything else takes the general instantiation.
// FULL: num_factor == 64*NR, so no lane is ever out of the row and the dot has no masked chunks and no tail.
// UV: every feature value of the fast-path units is 1.0 (the usual rating data): values are compile-time constants.
// RX: relaxed mode compiled in (atomic adds to item / feedback rows); the exact kernels carry none of it.
template <int NR, bool FAST, bool FULL, bool UV, bool RX>
__device__ __forceinline__ void svdpp_unit_wave(const DevParams &P, const DevCSR &D, const DevUnit &u, const unsigned *fb_index,
                                                const float *fb_value, int lane) {
    const int pitch = P.pitch;
    const int k = FULL ? 64 * NR : P.k;      // dot / projection width
    const int kio = FULL ? -1 : P.k;         // bound of row loads / stores (-1: none)
    const bool ub = FAST ? true : P.no_user_bias == 0;
    const bool rx_item = RX && P.relax_item_from == 0u, rx_fb = RX && P.relax_feedback != 0;   // wave-uniform
    const unsigned *fidx = fb_index + u.fb_begin;
    const float *fval = fb_value + u.fb_begin;
    const int nfb = u.fb_end - u.fb_begin;
    ChainRow<NR> tmp_fb = chain_zero<NR>(), old_fb = chain_zero<NR>();
    float norm = 0.0f, tmp_bias = 0.0f, old_bias = 0.0f;
    float *st = P.svdpp_state;
    if (u.flags & UNIT_LOAD) {
        tmp_fb = chain_load<NR>(st, 0, pitch, lane, kio);
        old_fb = chain_load<NR>(st, 1, pitch, lane, kio);
        norm = st[2 * pitch]; tmp_bias = st[2 * pitch + 1]; old_bias = st[2 * pitch + 2];
    }
    if (u.flags & UNIT_START) {   // prepare_ufeedback (:523-538)
        norm = 0.0f; tmp_fb = chain_zero<NR>(); tmp_bias = 0.0f;
        if (nfb > 0) {
            static_assert(64 % svdpp_fbw<NR>::value == 0, "a batch must not straddle two id blocks");
            // queue slot 0: the batch being accumulated; slots 1..DEPTH: batches whose rows are in flight; blkn: the id
            // block after the newest slot's, loaded a block ahead
            constexpr int FBW = svdpp_fbw<NR>::value, DEPTH = svdpp_fbdepth<NR>::value;
            FbBlock blk[DEPTH + 1], blkn;
            int off[DEPTH + 1];
            FbRows<NR> rq[DEPTH + 1];
            blk[0] = fb_block(fidx, fval, 0, nfb, lane);
            blkn = fb_block(fidx, fval, 64, nfb, lane);
            off[0] = 0;
            fb_fetch_rows<NR>(P, blk[0], 0, ub, lane, kio, rq[0]);
#pragma unroll
            for (int d = 1; d <= DEPTH; d++) {   // (DEPTH * FBW < 64: the first batches all sit in the first block)
                blk[d] = blk[0]; off[d] = d * FBW;
                if (d < DEPTH) fb_fetch_rows<NR>(P, blk[d], off[d], ub, lane, kio, rq[d]);
            }
            for (int j0 = 0; j0 < nfb; j0 += FBW) {
                off[DEPTH] = (j0 + DEPTH * FBW) & 63;
                if (off[DEPTH] == 0) { blk[DEPTH] = blkn; blkn = fb_block(fidx, fval, j0 + DEPTH * FBW + 64, nfb, lane); }
                fb_fetch_rows<NR>(P, blk[DEPTH], off[DEPTH], ub, lane, kio, rq[DEPTH]);
#pragma unroll
                for (int c = 0; c < FBW; c++) {
                    if (j0 + c < nfb) {
                        const float v = fb_val(blk[0], off[0] + c);
                        chain_axpy(tmp_fb, rq[0].w[c], v);   // (tmp_fb is in LINEAR layout during this phase)
                        norm = norm + v * v;
                        if (ub) tmp_bias = tmp_bias + rq[0].b[c] * v;
                    }
                }
#pragma unroll
                for (int d = 0; d < DEPTH; d++) { rq[d] = rq[d + 1]; blk[d] = blk[d + 1]; off[d] = off[d + 1]; }
            }
            tmp_fb = lin_to_chain<NR>(tmp_fb, lane);   // elementwise sums are layout-agnostic: convert the result once
        }
        old_bias = tmp_bias;
        old_fb = tmp_fb;
    }
    const int nrow = u.row_end - u.row_begin;
    if (nrow > 0) {
        const int e0 = uniform_load(D.row_ptr + 3 * (long)u.row_begin);   // rows are (0,1,1): entries of row j start at e0 + 2j
        const unsigned urow = P.user_off + uniform_load(D.feat_index + e0);
        ChainRow<NR> p = chain_load<NR>(P.W, urow, pitch, lane, kio);
        float bu = ub ? P.bias[urow] : 0.0f;
        const float wd_u = FAST ? P.wd_user : get_wd(P.u_rng, urow - P.user_off, P.wd_user);
        const float lr = P.lr;
        // row-invariant scalars of update_svdpp and of the L2 decays, and whether their multiply is skipped
        const float lr2 = lr * P.scale_lr_ufeedback;
        const float dec_fb = 1.0f - lr2 * P.wd_ufeedback, dec_fbb = 1.0f - lr2 * P.wd_ufeedback_bias;
        const float dec_u = 1.0f - lr * wd_u, dec_i = 1.0f - lr * P.wd_item;
        const float dec_ub = 1.0f - lr * P.wd_user_bias, dec_ib = 1.0f - lr * P.wd_item_bias;
        const float dec_fb1 = snap_to_one(dec_fb), dec_u1 = snap_to_one(dec_u), dec_i1 = snap_to_one(dec_i);
        static_assert(64 % SVDPP_PFW == 0, "a group of rows must not straddle two record blocks");
        // rb_cur / rb_pre: record blocks of the group being processed / being fetched ahead; rb_next: the block after rb_pre's
        RowBlock rb_cur = row_block<UV>(D, u.row_begin, e0, 0, nrow, lane), rb_pre = rb_cur;
        RowBlock rb_next = row_block<UV>(D, u.row_begin, e0, 64, nrow, lane);
        ChainRowPF<NR> cur[SVDPP_PFW], nxt[SVDPP_PFW];
        chain_fetch_rows<NR>(P, rb_cur, 0, lane, kio, cur);
        for (int j0 = 0; j0 < nrow; j0 += SVDPP_PFW) {
            const int off_pre = (j0 + SVDPP_PFW) & 63, off_cur = j0 & 63;
            if (off_pre == 0) { rb_pre = rb_next; rb_next = row_block<UV>(D, u.row_begin, e0, j0 + SVDPP_PFW + 64, nrow, lane); }
            chain_fetch_rows<NR>(P, rb_pre, off_pre, lane, kio, nxt);
#pragma unroll
            for (int c = 0; c < SVDPP_PFW; c++) {
                if (j0 + c < nrow) {
                    struct { ChainRow<NR> q; float bi, label, uv, iv; unsigned irow; } x;
                    x.q = cur[c].q; x.bi = cur[c].bi; x.irow = cur[c].irow;
                    x.label = pick(rb_cur.label, off_cur + c);
                    x.uv = UV ? 1.0f : pick(rb_cur.uv, off_cur + c);
                    x.iv = UV ? 1.0f : pick(rb_cur.iv, off_cur + c);
                    if (pick(rb_cur.fresh, off_cur + c)) {   // this item was written by an earlier row of the unit after (or while) it was fetched ahead
                        // The re-read must be COMPLETE before this block is left: loads and stores share one in-order
                        // counter on gfx9, and a load still pending at the join would make every row of the common
                        // path wait for everything in flight (measured 0.54 instead of 0.41 us per row at k=128).  The
                        // empty asm statements consume the loaded values here, so the wait lands inside the block.
                        ChainRow<NR> t = chain_load<NR>(P.W, x.irow, pitch, lane, kio);
                        float tb = P.bias[x.irow];
#pragma unroll
                        for (int q = 0; q < NR; q++) asm volatile("" : "+v"(t.r[q]));
                        asm volatile("" : "+v"(tb));
                        x.q = t;
                        x.bi = tb;
                    }
                    double bs = 0.0;                                   // calc_bias (:313-353)
                    if (ub) { bs += (double)(x.uv * bu); bs += (double)tmp_bias; }
                    bs += (double)(x.iv * x.bi);
                    double sum = (double)P.base_score + bs;
                    ChainRow<NR> tu = tmp_fb, ti = chain_zero<NR>();   // prepare_tmp (:354-381, :506-508)
                    chain_axpy(tu, p, x.uv);
                    chain_axpy(ti, x.q, x.iv);
                    sum += (double)chain_dot(tu, ti, lane, k);
                    const float pred = FAST ? (float)sum : map_active((float)sum, P.active_type);
                    const float err = (FAST ? x.label - pred : cal_grad(x.label, pred, P.active_type)) * 1.0f;
                    const float su = lr * err * x.uv;                  // update_no_decay (:383-427)
                    chain_axpy(p, ti, su);
                    if (ub) bu = bu + su;
                    const float si = lr * err * x.iv;
                    ChainRow<NR> w = x.q;
                    chain_axpy(w, tu, si);
                    float nbi = x.bi + si;
                    chain_axpy(tmp_fb, ti, lr2 * err * norm);          // update_svdpp (:512-520)
#pragma unroll
                    for (int q = 0; q < NR; q++) tmp_fb.r[q] = tmp_fb.r[q] * dec_fb1;
                    if (ub) {
                        tmp_bias = tmp_bias + lr2 * err * norm;
                        tmp_bias = tmp_bias * dec_fbb;
                    }
                    if (FAST) {                                        // regularize(feature, true) (:286-311), L2 form
#pragma unroll
                        for (int q = 0; q < NR; q++) p.r[q] = p.r[q] * dec_u1;
#pragma unroll
                        for (int q = 0; q < NR; q++) w.r[q] = w.r[q] * dec_i1;
                    } else {
                        chain_reg(P, p, wd_u, false, lane, k);
                        chain_reg(P, w, get_wd(P.i_rng, x.irow - P.item_off, P.wd_item), true, lane, k);
                    }
                    if (ub) bu = bu * dec_ub;
                    nbi = nbi * dec_ib;
                    if (rx_item) {   // relaxed item rows: other users of this launch may be updating the same item -- add the change
                        ChainRow<NR> dw;
#pragma unroll
                        for (int q = 0; q < NR; q++) dw.r[q] = w.r[q] - x.q.r[q];
                        chain_atomic_add<NR>(P.W, x.irow, pitch, lane, kio, dw);
                        if (lane == 0) unsafeAtomicAdd(&P.bias[x.irow], nbi - x.bi);
                    } else {
                        chain_store<NR>(P.W, x.irow, pitch, lane, kio, w);
                        P.bias[x.irow] = nbi;   // every lane writes the same word: one request, and no exec-mask branch
                    }
                }
            }
#pragma unroll
            for (int c = 0; c < SVDPP_PFW; c++) cur[c] = nxt[c];
            rb_cur = rb_pre;
        }
        chain_store<NR>(P.W, urow, pitch, lane, kio, p);
        if (ub && lane == 0) P.bias[urow] = bu;
    }
    if ((u.flags & UNIT_END) && nfb > 0) {   // update_ufeedback (:539-554)
        ChainRow<NR> d = tmp_fb;
#pragma unroll
        for (int q = 0; q < NR; q++) d.r[q] = d.r[q] - old_fb.r[q];   // K5
        float db = tmp_bias - old_bias;
        const float inv = 1.0f / norm;
        chain_scale(d, inv);
        db = db * inv;
        tmp_fb = d; tmp_bias = db;   // the reference leaves the scaled delta in tmp_ufeedback
        const ChainRow<NR> dl = chain_to_lin<NR>(d, lane);   // the scatter below runs in LINEAR layout
        constexpr int FBW = svdpp_fbw<NR>::value, DEPTH = svdpp_fbdepth<NR>::value;
        FbBlock blk[DEPTH + 1], blkn;
        int off[DEPTH + 1];
        FbRows<NR> rq[DEPTH + 1];
        blk[0] = fb_block(fidx, fval, 0, nfb, lane);
        blkn = fb_block(fidx, fval, 64, nfb, lane);
        off[0] = 0;
        fb_fetch_rows<NR>(P, blk[0], 0, ub, lane, kio, rq[0]);
#pragma unroll
        for (int d = 1; d <= DEPTH; d++) {
            blk[d] = blk[0]; off[d] = d * FBW;
            if (d < DEPTH) fb_fetch_rows<NR>(P, blk[d], off[d], ub, lane, kio, rq[d]);
        }
        for (int j0 = 0; j0 < nfb; j0 += FBW) {
            off[DEPTH] = (j0 + DEPTH * FBW) & 63;
            if (off[DEPTH] == 0) { blk[DEPTH] = blkn; blkn = fb_block(fidx, fval, j0 + DEPTH * FBW + 64, nfb, lane); }
            fb_fetch_rows<NR>(P, blk[DEPTH], off[DEPTH], ub, lane, kio, rq[DEPTH]);   // distinct ids: nothing fetched here is written below
#pragma unroll
            for (int c = 0; c < FBW; c++) {
                if (j0 + c < nfb) {
                    const float v = fb_val(blk[0], off[0] + c);
                    const unsigned row = P.fb_off + fb_id(blk[0], off[0] + c);
                    if (rx_fb) {   // relaxed feedback rows: the scatter is an addition anyway -- make it atomic
                        const float v1 = snap_to_one(v);
                        ChainRow<NR> dw;
#pragma unroll
                        for (int q = 0; q < NR; q++) dw.r[q] = dl.r[q] * v1;
                        lin_atomic_add<NR>(P.W, row, pitch, lane, kio, dw);
                        if (ub && lane == 0) unsafeAtomicAdd(&P.bias[row], db * v);
                    } else {
                        chain_axpy(rq[0].w[c], dl, v);
                        lin_store<NR>(P.W, row, pitch, lane, kio, rq[0].w[c]);
                        if (ub) P.bias[row] = rq[0].b[c] + db * v;
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < DEPTH; q++) { rq[q] = rq[q + 1]; blk[q] = blk[q + 1]; off[q] = off[q + 1]; }
        }
    }
    if (u.flags & UNIT_SAVE) {
        chain_store<NR>(st, 0, pitch, lane, kio, tmp_fb);
        chain_store<NR>(st, 1, pitch, lane, kio, old_fb);
        if (lane == 0) { st[2 * pitch] = norm; st[2 * pitch + 1] = tmp_bias; st[2 * pitch + 2] = old_bias; }
    }
}

// Kernel 4a: the simple units of one conflict-free batch, one wave per user
template <int NR, bool FAST, bool FULL, bool UV, bool RX>
__global__ __launch_bounds__(64) void k_svdpp_wave(const DevParams P, const DevCSR D, const DevUnit *units, const unsigned *fb_index,
                                                   const float *fb_value, const int *order, long begin, long end) {
    const int lane = threadIdx.x & 63;
    for (long s = begin + blockIdx.x; s < end; s += gridDim.x) {
        const int uid = __builtin_amdgcn_readfirstlane(order ? order[s] : (int)s);
        const DevUnit *up = units + uid;
        DevUnit u;
        u.fb_begin = __builtin_amdgcn_readfirstlane(up->fb_begin); u.fb_end = __builtin_amdgcn_readfirstlane(up->fb_end);
        u.row_begin = __builtin_amdgcn_readfirstlane(up->row_begin); u.row_end = __builtin_amdgcn_readfirstlane(up->row_end);
        u.flags = __builtin_amdgcn_readfirstlane(up->flags);
        svdpp_unit_wave<NR, FAST, FULL, UV, RX>(P, D, u, fb_index, fb_value, lane);
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

// ---- multi-GPU item-side delta exchange -------------------------------------
__global__ void k_delta_sub(const float *cur, const float *snap, float *delta, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = i; j < n; j += stride) delta[j] = cur[j] - snap[j];
}
__global__ void k_delta_add(float *cur, const float *snap, const float *delta, long n) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = i; j < n; j += stride) cur[j] = snap[j] + delta[j];
}

// =====================================================================================
// launchers
// =====================================================================================
int lanes_per_instance(int k) {
    int chunks = (k + 3) / 4;
    int lpi = 1;
    while (lpi < chunks && lpi < 64) lpi <<= 1;
    return lpi;   // 64 also for wide rows (k > 256: several float4 slots per lane)
}
int max_supported_factor() { return 1024; }
int max_fast_path_factor() { return 256; }

static inline int grid_for(long groups, int lpi, int cap) {
    const long per_block = 4L * (64 / lpi);
    long g = (groups + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

template <int LPI>
static void launch_basicmf_lpi(const DevParams &P, const BasicSchedule &S, long begin, long end, int G, int block_threads, hipStream_t st) {
    const long n = end - begin;
    const bool unit = S.uval == nullptr;
    auto go = [&](auto gtag) {
        constexpr int GG = decltype(gtag)::value;
        const long per_block = (long)(block_threads / 64) * GG * (64 / LPI);
        int grid = (int)((n + per_block - 1) / per_block);
        if (P.xcd_remap) grid = (grid + 7) & ~7;
        // specialised instruction stream for the configuration of the contract workload, general one otherwise
        const bool fast = P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 &&
                          P.u_rng.n == 0 && P.i_rng.n == 0 && P.store_mode == 0;
        if (unit && fast && P.k == 4 * LPI) hipLaunchKernelGGL((k_basicmf<LPI, GG, true, true, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else if (unit) hipLaunchKernelGGL((k_basicmf<LPI, GG, true, false, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else hipLaunchKernelGGL((k_basicmf<LPI, GG, false, false, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    };
    switch (G) {
    case 1: go(std::integral_constant<int, 1>()); break;
    case 2: go(std::integral_constant<int, 2>()); break;
    case 8: go(std::integral_constant<int, 8>()); break;
    default: go(std::integral_constant<int, 4>()); break;
    }
}

// (variadic: a launch expands to kernel<<<a, b, c, d>>>(...), whose bare commas must survive being passed on)
#define SVDF_DISPATCH_LPI(lpi, ...)                          \
    switch (lpi) {                                           \
    case 1: { constexpr int LPI = 1; __VA_ARGS__; } break;   \
    case 2: { constexpr int LPI = 2; __VA_ARGS__; } break;   \
    case 4: { constexpr int LPI = 4; __VA_ARGS__; } break;   \
    case 8: { constexpr int LPI = 8; __VA_ARGS__; } break;   \
    case 16: { constexpr int LPI = 16; __VA_ARGS__; } break; \
    case 32: { constexpr int LPI = 32; __VA_ARGS__; } break; \
    default: { constexpr int LPI = 64; __VA_ARGS__; } break; \
    }
// general-path kernels: LPI as above with float4 rows up to 256 factors, a whole wave with 2..4 float4 slots beyond
#define SVDF_DISPATCH_ROW(k, ...)                                                          \
    if ((k) <= 256) {                                                                      \
        using R = float4;                                                                  \
        SVDF_DISPATCH_LPI(lanes_per_instance(k), __VA_ARGS__)                              \
    } else if ((k) <= 512) { constexpr int LPI = 64; using R = WideRow<2>; __VA_ARGS__; }  \
    else if ((k) <= 768) { constexpr int LPI = 64; using R = WideRow<3>; __VA_ARGS__; }    \
    else { constexpr int LPI = 64; using R = WideRow<4>; __VA_ARGS__; }

void launch_basicmf(const DevParams &P, const BasicSchedule &S, long begin, long end, int groups_per_wave, int block_threads, hipStream_t st) {
    if (end <= begin) return;
    // 0 = tuned default (tools/sweep_knobs.py on MI355X with the FULL/FAST specialisation): k=64 (16 lanes per row) wants
    // 4 row sets in flight per wave in 64-thread blocks (24.9 ms/pass; 26.3 with one row set), k=256 two row sets, every
    // other width one row set per wave in 256-thread blocks
    const int lpi_ = lanes_per_instance(P.k);
    if (groups_per_wave <= 0) groups_per_wave = lpi_ == 16 ? 4 : (lpi_ == 64 ? 2 : 1);
    if (block_threads <= 0) block_threads = lpi_ == 16 ? 64 : 256;
    SVDF_DISPATCH_LPI(lanes_per_instance(P.k), launch_basicmf_lpi<LPI>(P, S, begin, end, groups_per_wave, block_threads, st));
}
template <int LPI, int NU, int NI>
static void launch_fused_shape(const DevParams &P, const FusedSchedule &S, long begin, long end, int G, int block_threads, hipStream_t st) {
    const long n = end - begin;
    if (G >= 2 && NU + NI <= 3 && !(NU == 2 && P.relax_user_from != 0xFFFFFFFFu && P.hot_reduce)) {
        const long per_block = (long)(block_threads / 64) * 2 * (64 / LPI);
        int grid = (int)((n + per_block - 1) / per_block);
        if (P.xcd_remap) grid = (grid + 7) & ~7;
        if (P.k == 4 * LPI) hipLaunchKernelGGL((k_fused<LPI, NU, NI, 2, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else hipLaunchKernelGGL((k_fused<LPI, NU, NI, 2, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    } else if (NU == 2 && P.relax_user_from != 0xFFFFFFFFu && P.hot_reduce) {
        // relaxed shared user feature in the last user slot: big workgroups that pre-reduce its changes in LDS
        constexpr int HB = 1024;
        const long per_block = (long)(HB / 64) * (64 / LPI);
        int grid = (int)((n + per_block - 1) / per_block);
        if (P.xcd_remap) grid = (grid + 7) & ~7;
        const size_t lds = (size_t)per_block * (LPI * 16 + 8);
        if (P.k == 4 * LPI) hipLaunchKernelGGL((k_fused<LPI, NU, NI, 1, true, NU == 2>), dim3(grid), dim3(HB), lds, st, P, S, begin, end);
        else hipLaunchKernelGGL((k_fused<LPI, NU, NI, 1, false, NU == 2>), dim3(grid), dim3(HB), lds, st, P, S, begin, end);
    } else {
        const long per_block = (long)(block_threads / 64) * (64 / LPI);
        int grid = (int)((n + per_block - 1) / per_block);
        if (P.xcd_remap) grid = (grid + 7) & ~7;
        if (P.k == 4 * LPI) hipLaunchKernelGGL((k_fused<LPI, NU, NI, 1, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
        else hipLaunchKernelGGL((k_fused<LPI, NU, NI, 1, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    }
}
template <int LPI>
static void launch_fused_lpi(const DevParams &P, const FusedSchedule &S, int nu, int ni, long begin, long end, int G, int bt, hipStream_t st) {
    if (nu <= 1 && ni <= 1) launch_fused_shape<LPI, 1, 1>(P, S, begin, end, G, bt, st);
    else if (nu <= 1) launch_fused_shape<LPI, 1, 2>(P, S, begin, end, G, bt, st);
    else if (ni <= 1) launch_fused_shape<LPI, 2, 1>(P, S, begin, end, G, bt, st);
    else launch_fused_shape<LPI, 2, 2>(P, S, begin, end, G, bt, st);
}
void launch_fused(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int groups_per_wave,
                  int block_threads, hipStream_t st) {
    if (end <= begin) return;
    const int lpi_ = lanes_per_instance(P.k);
    if (groups_per_wave <= 0) groups_per_wave = lpi_ == 16 ? 2 : 1;
    if (block_threads <= 0) block_threads = lpi_ == 16 ? 128 : 256;
    SVDF_DISPATCH_LPI(lanes_per_instance(P.k), launch_fused_lpi<LPI>(P, S, max_nu, max_ni, begin, end, groups_per_wave, block_threads, st));
}
template <int LPI>
static void launch_predict_fused_lpi(const DevParams &P, const FusedSchedule &S, int nu, int ni, long n, float *out, int grid, hipStream_t st) {
    if (nu <= 1 && ni <= 1) hipLaunchKernelGGL((k_predict_fused<LPI, 1, 1>), dim3(grid), dim3(256), 0, st, P, S, n, out);
    else if (nu <= 1) hipLaunchKernelGGL((k_predict_fused<LPI, 1, 2>), dim3(grid), dim3(256), 0, st, P, S, n, out);
    else if (ni <= 1) hipLaunchKernelGGL((k_predict_fused<LPI, 2, 1>), dim3(grid), dim3(256), 0, st, P, S, n, out);
    else hipLaunchKernelGGL((k_predict_fused<LPI, 2, 2>), dim3(grid), dim3(256), 0, st, P, S, n, out);
}
void launch_predict_fused(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long n, float *out, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    SVDF_DISPATCH_LPI(lpi, launch_predict_fused_lpi<LPI>(P, S, max_nu, max_ni, n, out, grid, st));
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
void launch_predict_basic(const DevParams &P, const BasicSchedule &S, long n, float *out, hipStream_t st) {
    if (n <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(n, lpi, 256 * 8);
    if (S.uval == nullptr) { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_predict_basic<LPI, true>), dim3(grid), dim3(256), 0, st, P, S, n, out)); }
    else { SVDF_DISPATCH_LPI(lpi, hipLaunchKernelGGL((k_predict_basic<LPI, false>), dim3(grid), dim3(256), 0, st, P, S, n, out)); }
}
void launch_svdpp(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                  const int *order, long begin, long end, unsigned counter_base, hipStream_t st) {
    if (end <= begin) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(end - begin, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, units, fb_index, fb_value, order, begin, end, counter_base));
}
void launch_svdpp_wave(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                       const int *order, long begin, long end, hipStream_t st) {
    if (end <= begin) return;
    long grid = end - begin;           // one 64-thread workgroup (= one wave) per user: a batch rarely holds more users than CUs
    if (grid > 16384) grid = 16384;
    const int nr = (P.k + 63) / 64;
    const bool fast = P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 &&
                      P.u_rng.n == 0 && P.i_rng.n == 0;
#define SVDF_WAVE_LAUNCH(NR_, FAST_, FULL_, UV_, RX_) \
    hipLaunchKernelGGL((k_svdpp_wave<NR_, FAST_, FULL_, UV_, RX_>), dim3((int)grid), dim3(64), 0, st, P, D, units, fb_index, fb_value, order, begin, end)
#define SVDF_WAVE_CASE(NR_)                                                                   \
    case NR_:                                                                                 \
        if (relaxed && fast && full && D.unit_values) SVDF_WAVE_LAUNCH(NR_, true, true, true, true);   \
        else if (relaxed) SVDF_WAVE_LAUNCH(NR_, false, false, false, true);                   \
        else if (fast && full && D.unit_values) SVDF_WAVE_LAUNCH(NR_, true, true, true, false); \
        else if (fast && full) SVDF_WAVE_LAUNCH(NR_, true, true, false, false);               \
        else if (fast) SVDF_WAVE_LAUNCH(NR_, true, false, false, false);                      \
        else SVDF_WAVE_LAUNCH(NR_, false, false, false, false);                               \
        break;
    const bool relaxed = P.relax_item_from == 0u || P.relax_feedback != 0;
    const bool full = P.k == 64 * nr;
    switch (nr) {
        SVDF_WAVE_CASE(1)
        SVDF_WAVE_CASE(2)
        SVDF_WAVE_CASE(3)
    default:
        SVDF_WAVE_CASE(4)
    }
#undef SVDF_WAVE_LAUNCH
#undef SVDF_WAVE_CASE
}
void launch_svdpp_predict(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                          long nunit, float *out, hipStream_t st) {
    if (nunit <= 0) return;
    const int lpi = lanes_per_instance(P.k);
    const int grid = grid_for(nunit, lpi, 256 * 8);
    SVDF_DISPATCH_ROW(P.k, hipLaunchKernelGGL((k_svdpp_predict<LPI, R>), dim3(grid), dim3(256), 0, st, P, D, units, fb_index, fb_value, nunit, out));
}
// One launch over ALL replicated ranges (W_item, biases, globals ...): pack = (current - snapshot) in the wire type,
// unpack = current <- snapshot + delta, optionally snapshot <- current so that the next window needs no copy.
// fp16 conversion is round-to-nearest-even (what a separate .half() pass would do).
__device__ __forceinline__ float *delta_slot(const DeltaRanges &R, long j) {
    int r = 0;
#pragma unroll
    for (int q = 1; q < SVDF_MAX_DELTA_RANGES; q++) r += (q < R.n && j >= R.off[q]) ? 1 : 0;
    return R.base[r] + (j - R.off[r]);
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_pack(const DeltaRanges R, const float *snap, void *dst, long total) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const float d = *delta_slot(R, j) - snap[j];
        if (HALF) reinterpret_cast<__half *>(dst)[j] = __float2half_rn(d);
        else reinterpret_cast<float *>(dst)[j] = d;
    }
}
template <bool HALF>
__global__ __launch_bounds__(256) void k_delta_unpack(const DeltaRanges R, float *snap, const void *src, long total, int refresh) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < total; j += stride) {
        const float d = HALF ? __half2float(reinterpret_cast<const __half *>(src)[j]) : reinterpret_cast<const float *>(src)[j];
        const float v = snap[j] + d;
        *delta_slot(R, j) = v;
        if (refresh) snap[j] = v;
    }
}
void launch_delta_pack(const DeltaRanges &R, const float *snap, void *dst, int half, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (half) hipLaunchKernelGGL(k_delta_pack<true>, dim3((int)grid), dim3(256), 0, st, R, snap, dst, total);
    else hipLaunchKernelGGL(k_delta_pack<false>, dim3((int)grid), dim3(256), 0, st, R, snap, dst, total);
}
void launch_delta_unpack(const DeltaRanges &R, float *snap, const void *src, int half, int refresh, hipStream_t st) {
    const long total = R.off[R.n];
    if (total <= 0) return;
    long grid = (total + 255) / 256;
    if (grid > 8192) grid = 8192;
    if (half) hipLaunchKernelGGL(k_delta_unpack<true>, dim3((int)grid), dim3(256), 0, st, R, snap, src, total, refresh);
    else hipLaunchKernelGGL(k_delta_unpack<false>, dim3((int)grid), dim3(256), 0, st, R, snap, src, total, refresh);
}
void launch_delta_sub(const float *cur, const float *snap, float *delta, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_delta_sub, dim3((int)grid), dim3(256), 0, st, cur, snap, delta, n);
}
void launch_delta_add(float *cur, const float *snap, const float *delta, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 4096) grid = 4096;
    hipLaunchKernelGGL(k_delta_add, dim3((int)grid), dim3(256), 0, st, cur, snap, delta, n);
}

}  // namespace svdf
