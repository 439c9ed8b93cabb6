// svdf_device.h -- hand-written gfx950 (CDNA4, wave64) kernels of the apex_svd SGD hot path.
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
// divide/sqrt, fp64 bias/score accumulation, glibc's expf restated (glibc_expf below): every configuration is
// bit-exact against oracle/svdf_oracle.c.
#ifndef SVDF_DEVICE_H_
#define SVDF_DEVICE_H_
// Device-side helpers shared by the kernel translation units (svdf_k_*.hip): arithmetic in the reference's order, the
// DPP scans of the bit-exact dot product, row loads / stores, regularisers, and the host-side launch dispatch macros.
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
__device__ __forceinline__ void add_rows(float4 &d, const float4 s) { d.x = d.x + s.x; d.y = d.y + s.y; d.z = d.z + s.z; d.w = d.w + s.w; }  // tensor += tensor
template <int V> __device__ __forceinline__ void add_rows(WideRow<V> &d, const WideRow<V> &s) {
#pragma unroll
    for (int v = 0; v < V; v++) add_rows(d.v[v], s.v[v]);
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

// K3 for full rows held V chunks per lane by LANES <= 16 lanes (k = 4 * LANES * V): lane m of an instance holds chunks m, m + LANES,
// m + 2 LANES, ...; the T = 16 / LANES instances of a 16-lane DPP row are interleaved lane by lane (instance a on lanes T m + a).
// The chain is the reference's: slot 0's chunks in lane order (LANES - 1 row_shr:T adds), the finished sums rotate from the row's
// last lanes to its first (row_ror:T) and are folded into the addend of the next slot's first chunk -- a[first] = 0 + (carry + c),
// the carry trick of group_dot<32> above --, and so on through the slots.  Used by k_basicmf_slots and k_fewrow_slots.
template <int T> __device__ __forceinline__ float dpp_shr_t(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + T, 0xf, 0xf, true));
}
template <int T> __device__ __forceinline__ float dpp_ror_t(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x120 + T, 0xf, 0xf, false));
}
template <int LANES, int V>
__device__ __forceinline__ float dot_slots(const float4 (&a)[V], const float4 (&b)[V], int m, int lane) {
    constexpr int T = 16 / LANES;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
#pragma unroll
    for (int v = 0; v < V; v++) {
        float c0 = a[v].x * b[v].x, c1 = a[v].y * b[v].y, c2 = a[v].z * b[v].z, c3 = a[v].w * b[v].w;   // chunk m + v * LANES
        if (v > 0) {
            const float k0 = dpp_ror_t<T>(s0), k1 = dpp_ror_t<T>(s1), k2 = dpp_ror_t<T>(s2), k3 = dpp_ror_t<T>(s3);
            if (m == 0) { c0 = k0 + c0; c1 = k1 + c1; c2 = k2 + c2; c3 = k3 + c3; }
        }
        s0 = 0.0f + c0; s1 = 0.0f + c1; s2 = 0.0f + c2; s3 = 0.0f + c3;
#pragma unroll
        for (int t = 1; t < LANES; t++) {
            s0 = dpp_shr_t<T>(s0) + c0; s1 = dpp_shr_t<T>(s1) + c1; s2 = dpp_shr_t<T>(s2) + c2; s3 = dpp_shr_t<T>(s3) + c3;
        }
    }
    const float h = (s0 + s2) + (s1 + s3);
    return __shfl(h, (lane & ~15) + T * (LANES - 1) + (lane & (T - 1)), 64);
}

// glibc's expf restated for the device (sysdeps/ieee754/flt-32/e_expf.c of glibc >= 2.27, the libm the reference links
// against; glibc is a system library, not part of the reference tree): x*32/ln2 = k + r, exp(x) = 2^(k/32) * p(r) with a
// 32-entry table of 2^(i/32) and a cubic in fp64, result rounded to fp32 once.  On x86-64 hosts with FMA the dynamic
// linker selects the build of that file in which the range reduction r = x*InvLn2N - k is ONE fused operation (the other
// steps stay unfused): restated like that, this function returns the host libm's result for every one of the 2^32 float
// inputs (tools/check_expf.c compares them all on the CPU: 0 mismatches); the unfused build differs on exactly two
// inputs (0x4202422f, 0xc27c65d9).  The table entries are 2^(i/32) correctly rounded to fp64 minus (i << 47).
static __constant__ unsigned long long kExp2fTab[32] = {
0x3ff0000000000000ULL,
0x3fefd9b0d3158574ULL,
0x3fefb5586cf9890fULL,
0x3fef9301d0125b51ULL,
0x3fef72b83c7d517bULL,
0x3fef54873168b9aaULL,
0x3fef387a6e756238ULL,
0x3fef1e9df51fdee1ULL,
0x3fef06fe0a31b715ULL,
0x3feef1a7373aa9cbULL,
0x3feedea64c123422ULL,
0x3feece086061892dULL,
0x3feebfdad5362a27ULL,
0x3feeb42b569d4f82ULL,
0x3feeab07dd485429ULL,
0x3feea47eb03a5585ULL,
0x3feea09e667f3bcdULL,
0x3fee9f75e8ec5f74ULL,
0x3feea11473eb0187ULL,
0x3feea589994cce13ULL,
0x3feeace5422aa0dbULL,
0x3feeb737b0cdc5e5ULL,
0x3feec49182a3f090ULL,
0x3feed503b23e255dULL,
0x3feee89f995ad3adULL,
0x3feeff76f2fb5e47ULL,
0x3fef199bdd85529cULL,
0x3fef3720dcef9069ULL,
0x3fef5818dcfba487ULL,
0x3fef7c97337b9b5fULL,
0x3fefa4afa2a490daULL,
0x3fefd0765b6e4540ULL

};
__device__ __forceinline__ float glibc_expf(float x) {
    const unsigned ux = __float_as_uint(x);
    const unsigned abstop = (ux >> 20) & 0x7ffu;
    if (abstop >= 0x42bu) {                             // |x| >= 88 or NaN
        if (ux == 0xff800000u) return 0.0f;             // -inf
        if (abstop >= 0x7f8u) return x + x;             // +inf, NaN
        if (x > 0x1.62e42ep6f) return __uint_as_float(0x7f800000u);   // overflow
        if (x < -0x1.9fe368p6f) return 0.0f;            // underflow
    }
    const double xd = (double)x;
    const double inv_ln2_n = 0x1.71547652b82fep+5;      // 32 / ln 2
    const double shift = 0x1.8p+52;
    const double z = inv_ln2_n * xd;
    double kd = z + shift;
    const unsigned long long ki = (unsigned long long)__double_as_longlong(kd);
    kd = kd - shift;
    const double r = __fma_rn(inv_ln2_n, xd, -kd);
    unsigned long long t = kExp2fTab[ki & 31];
    t += ki << 47;
    const double s = __longlong_as_double((long long)t);
    const double zz = 0x1.c6af84b912394p-20 * r + 0x1.ebfce50fac4f3p-13;
    const double r2 = r * r;
    double y = 0x1.62e42ff0c52d6p-6 * r + 1.0;
    y = zz * r2 + y;
    y = y * s;
    return (float)y;
}

// apex_svd_model.h:112-123
__device__ __forceinline__ float map_active(float sum, int type) {
    if (type == ACT_SIGMOID_L2 || type == ACT_SIGMOID_LIKELIHOOD) return 1.0f / (1.0f + glibc_expf(-sum));
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
    case ACT_SIGMOID_RANK: return r - 1.0f / (1.0f + glibc_expf(-pred));
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
        scale4(w, glibc_expf(logf(1.0f - lambda) * kk));
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
            if (P.reg_global == 4) g = g * glibc_expf(logf(1.0f - lambda) * kk);
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

// row load with the nontemporal hint (load_mode knob, k_basicmf only: rows of a level are read once)
typedef float svdf_f4_ld __attribute__((ext_vector_type(4)));
template <int LPI>
__device__ __forceinline__ float4 load_row_nt(const float *W, size_t row, int pitch, int L, int k) {
    if (LPI * 4 > k && L * 4 >= k) return f4zero();
    const svdf_f4_ld v = __builtin_nontemporal_load(reinterpret_cast<const svdf_f4_ld *>(W + row * (size_t)pitch + (size_t)L * 4));
    return make_float4(v.x, v.y, v.z, v.w);
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

// ---- contribution rows of the window-minibatch step (svdf_k_window.hip, svdf_k_wunit.hip): fp32, or -- opt-in, `amd:contrib = bf16` --
// bfloat16 (round to nearest even of the fp32 contribution; sums are taken in fp32): a contribution is written once and read once, so
// half the bytes is half the traffic of both.  The checker applies the same rounding (oracle/svdf_oracle.c: svdo_set_stale_rounding).
__device__ __forceinline__ unsigned bf16_rne(float x) {
    const unsigned u = __float_as_uint(x);
    return (u + 0x7FFFu + ((u >> 16) & 1u)) >> 16;
}
template <int LPI>
__device__ __forceinline__ void store_contrib(float *base, int bf16, size_t slot, int pitch, int L, int k, const float4 v) {
    if (LPI * 4 > k && L * 4 >= k) return;
    if (bf16) {
        uint2 pk;
        pk.x = bf16_rne(v.x) | (bf16_rne(v.y) << 16);
        pk.y = bf16_rne(v.z) | (bf16_rne(v.w) << 16);
        *reinterpret_cast<uint2 *>(reinterpret_cast<unsigned short *>(base) + slot * (size_t)pitch + (size_t)L * 4) = pk;
    } else {
        *reinterpret_cast<float4 *>(base + slot * (size_t)pitch + (size_t)L * 4) = v;
    }
}
template <int LPI>
__device__ __forceinline__ float4 load_contrib(const float *base, int bf16, size_t slot, int pitch, int L, int k) {
    if (LPI * 4 > k && L * 4 >= k) return f4zero();
    if (bf16) {
        const uint2 pk = *reinterpret_cast<const uint2 *>(reinterpret_cast<const unsigned short *>(base) + slot * (size_t)pitch + (size_t)L * 4);
        return make_float4(__uint_as_float(pk.x << 16), __uint_as_float(pk.x & 0xFFFF0000u), __uint_as_float(pk.y << 16), __uint_as_float(pk.y & 0xFFFF0000u));
    }
    return *reinterpret_cast<const float4 *>(base + slot * (size_t)pitch + (size_t)L * 4);
}

// A shared row's ONLY contribution of a window (slot < 0 in its entry; one-GPU window sequences, svdf_wunit.cpp): nobody else reads or
// writes the row inside the window, so the unit applies it where it is computed -- with the sum kernel's own operations, acc = +0 + c on the
// contribution as the slot would have stored it, then row + acc -- instead of writing a slot that k_wunit_sum reads back: the same bits,
// a row write instead of a slot write + a slot read + a row read + a row write.
__device__ __forceinline__ float contrib_as_stored(float x, bool bf16) { return bf16 ? __uint_as_float(bf16_rne(x) << 16) : x; }
__device__ __forceinline__ float apply_single(float w, float c, bool bf16) {
    const float acc = 0.0f + contrib_as_stored(c, bf16);
    return w + acc;
}
__device__ __forceinline__ float4 apply_single(const float4 w, const float4 c, bool bf16) {
    return make_float4(apply_single(w.x, c.x, bf16), apply_single(w.y, c.y, bf16), apply_single(w.z, c.z, bf16), apply_single(w.w, c.w, bf16));
}
// sum of contribution slots [b, e) in slot order: eight rows requested at a time; slots past the segment's end feed +0.0f, which leaves the running
// sum unchanged bit for bit (the sum starts at +0.0f and x + y is -0 only when both are, so acc is never -0).  The storage format is a TEMPLATE
// parameter here: with the format test inside the unrolled block the compiler kept a branch between the eight loads and they were issued one
// by one (the stratified step's small windows: 11.6 -> 26 us per launch).
template <int LPI, bool BF16, int NB = 8>   // NB rows requested at a time (8; 4 where lists are short and registers buy resident waves: the in-place sums of one-GPU windows)
__device__ __forceinline__ void sum_contrib_slots(const float *contrib, const float *cbias, int b, int e, int pitch, int L, int k, float4 &acc, float &accb) {
    for (int t = b; t < e; t += NB) {
        float4 c[NB];
        float cb[NB];
#pragma unroll
        for (int q = 0; q < NB; q++) {
            const bool in = t + q < e;
            c[q] = in ? load_contrib<LPI>(contrib, BF16 ? 1 : 0, (size_t)(t + q), pitch, L, k) : f4zero();
            cb[q] = in ? cbias[t + q] : 0.0f;
        }
#pragma unroll
        for (int q = 0; q < NB; q++) { add_rows(acc, c[q]); accb = accb + cb[q]; }
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


// ---- launch helpers (host side)
static inline int grid_for(long groups, int lpi, int cap) {
    const long per_block = 4L * (64 / lpi);
    long g = (groups + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

// Launch shape of a level for the grid-stride kernels: levels of the general / user-group kernels are usually a few hundred to a few
// thousand waves, and one-wave workgroups spread those over four times as many CUs as 256-thread ones (measured on k_fused:
// neighbourhood data 87.3 vs 91.8 ms per pass); big levels keep 256-thread workgroups under the grid cap.
static inline void launch_shape(long groups, int lpi, int cap, bool small_blocks, int &grid, int &block) {
    const long ipw = 64 / lpi;
    const long waves = (groups + ipw - 1) / ipw;
    if (small_blocks && waves <= 8192) { block = 64; grid = (int)(waves < 1 ? 1 : waves); }
    else { block = 256; grid = grid_for(groups, lpi, cap); }
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


}  // namespace svdf
#endif
