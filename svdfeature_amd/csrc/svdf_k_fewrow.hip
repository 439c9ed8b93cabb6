// svdf_k_fewrow.hip -- the few-row fused step specialised for its usual configuration (rank pairs, BASELINE configs[4]; basicMF
// with side ids): no global feature in the data set, no relaxed ids, L2 decay (reg_method 0), no per-range decay, no clamp.
// Same arithmetic, same order, same bits as k_fused (svdf_k_fused.hip) -- which stays the general few-row kernel -- with the
// instruction stream stripped of everything this configuration never executes: the global-bias paths, the per-row decay
// switch and range lookup, the relaxed-id tests, the exec-mask branch around the bias stores.  k_fused<32,1,2> spends 448 VALU
// + 165 SALU instructions per wave on a pair (profiles/r02_pmc_pairwise.txt) and a level of 11.5 K pairs is one occupancy
// wave of the chip, so instruction issue is ~40 % of a launch: the same lever as k_basicmf's FULL / FAST specialisation.
#include "svdf_device.h"

namespace svdf {

template <int LPI, int NU, int NI, bool FULL>
__global__ __launch_bounds__(256) void k_fewrow_fast(const DevParams P, const FusedSchedule S, long begin, long end) {
    constexpr int IPW = 64 / LPI;
    const int lane = threadIdx.x & 63;
    const int L = lane & (LPI - 1);
    const int gslot = lane / LPI;
    long tile = blockIdx.x;   // XCD-aware tile mapping, see k_basicmf
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long s = begin + wave * (long)IPW + gslot;
    if (begin + wave * (long)IPW >= end) return;
    const bool valid = s < end;
    const long sc = valid ? s : begin;
    const int k = FULL ? 4 * LPI : P.k, pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;

    unsigned ur[NU], ir[NI];
    float ua[NU], ia[NI], bu[NU], bi[NI];
    float4 p[NU], q[NI];
    const float label = S.label[sc];
#pragma unroll
    for (int a = 0; a < NU; a++) { ur[a] = valid ? S.uidx[a][sc] : (unsigned)SLOT_ABSENT; ua[a] = S.uval[a][sc]; }
#pragma unroll
    for (int b = 0; b < NI; b++) { ir[b] = valid ? S.iidx[b][sc] : (unsigned)SLOT_ABSENT; ia[b] = S.ival[b][sc]; }
#pragma unroll
    for (int a = 0; a < NU; a++) {
        p[a] = f4zero(); bu[a] = 0.0f;
        if (ur[a] != SLOT_ABSENT) {
            p[a] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.user_off + ur[a], pitch, L, k) : load_row<LPI>(P.W, P.user_off + ur[a], pitch, L, k);
            if (use_ubias) bu[a] = P.bias[P.user_off + ur[a]];
        }
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
        q[b] = f4zero(); bi[b] = 0.0f;
        if (ir[b] != SLOT_ABSENT) {
            q[b] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.item_off + ir[b], pitch, L, k) : load_row<LPI>(P.W, P.item_off + ir[b], pitch, L, k);
            bi[b] = P.bias[P.item_off + ir[b]];
        }
    }
    // row-invariant decay factors (L2: W *= 1 - lr*wd, the multiply skipped when that rounds to one; biases are plain multiplies)
    const float dec_u = snap_to_one(1.0f - P.lr * P.wd_user), dec_i = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;
    // calc_bias (:313-353) in double, prepare_tmp (:354-381), dot, link, gradient -- as in k_fused
    double bs = 0.0;
    if (use_ubias) {
#pragma unroll
        for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) bs += (double)(ua[a] * bu[a]);
    }
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) bs += (double)(ia[b] * bi[b]);
    double sum = (double)P.base_score + bs;
    float4 tu = f4zero(), ti = f4zero();
#pragma unroll
    for (int a = 0; a < NU; a++) if (ur[a] != SLOT_ABSENT) axpy4(tu, p[a], ua[a]);
#pragma unroll
    for (int b = 0; b < NI; b++) if (ir[b] != SLOT_ABSENT) axpy4(ti, q[b], ia[b]);
    sum += (double)group_dot<LPI>(tu, ti, L, k);
    const float pred = map_active((float)sum, P.active_type);
    const float err = cal_grad(label, pred, P.active_type) * 1.0f;
    const float lr = P.lr;
    // update_no_decay (:383-427) + regularize (:286-311), rows written once; every lane of the group writes the same bias word
#pragma unroll
    for (int a = 0; a < NU; a++) {
        if (ur[a] == SLOT_ABSENT) continue;
        const float su = lr * err * ua[a];
        float4 w = p[a];
        axpy4(w, ti, su);
        w.x = w.x * dec_u; w.y = w.y * dec_u; w.z = w.z * dec_u; w.w = w.w * dec_u;
        store_row<LPI>(P.W, P.user_off + ur[a], pitch, L, k, w);
        if (use_ubias) P.bias[P.user_off + ur[a]] = (bu[a] + su) * dec_ub;
    }
#pragma unroll
    for (int b = 0; b < NI; b++) {
        if (ir[b] == SLOT_ABSENT) continue;
        const float si = lr * err * ia[b];
        float4 w = q[b];
        axpy4(w, tu, si);
        w.x = w.x * dec_i; w.y = w.y * dec_i; w.z = w.z * dec_i; w.w = w.w * dec_i;
        store_row<LPI>(P.W, P.item_off + ir[b], pitch, L, k, w);
        P.bias[P.item_off + ir[b]] = (bi[b] + si) * dec_ib;
    }
}

// ---- k = 128 with SIXTEEN lanes per row (fewrow_i16 knob): a lane holds chunks m and m + 16 of a row (two coalesced 256-byte pieces
// per instance and load instruction), one instance per 16-lane DPP row, four per wave instead of two -- the layout idea of
// k_basicmf_i8 (svdf_k_basic.hip): every wave instruction serves twice the instances.  The dot product is the chain of
// group_dot<32>: chunks 0..15 through the first slots (15 row_shr:1 steps), the finished sums rotate from lane 15 to lane 0
// (row_ror:1) and are folded into chunk 16's addend, then chunks 16..31 through the second slots.  Same additions, same order.
__device__ __forceinline__ float dpp_row_shr1f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, true)); }
__device__ __forceinline__ float dpp_row_ror1f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x121, 0xf, 0xf, false)); }
__device__ __forceinline__ float dot128_i16(const float4 a0, const float4 a1, const float4 b0, const float4 b1, int m, int lane) {
    const float c0 = a0.x * b0.x, c1 = a0.y * b0.y, c2 = a0.z * b0.z, c3 = a0.w * b0.w;   // chunk m
    float d0 = a1.x * b1.x, d1 = a1.y * b1.y, d2 = a1.z * b1.z, d3 = a1.w * b1.w;         // chunk m + 16
    float s0 = 0.0f + c0, s1 = 0.0f + c1, s2 = 0.0f + c2, s3 = 0.0f + c3;
#pragma unroll
    for (int t = 1; t < 16; t++) {
        s0 = dpp_row_shr1f(s0) + c0; s1 = dpp_row_shr1f(s1) + c1; s2 = dpp_row_shr1f(s2) + c2; s3 = dpp_row_shr1f(s3) + c3;
    }
    const float k0 = dpp_row_ror1f(s0), k1 = dpp_row_ror1f(s1), k2 = dpp_row_ror1f(s2), k3 = dpp_row_ror1f(s3);
    if (m == 0) { d0 = k0 + d0; d1 = k1 + d1; d2 = k2 + d2; d3 = k3 + d3; }
    s0 = 0.0f + d0; s1 = 0.0f + d1; s2 = 0.0f + d2; s3 = 0.0f + d3;
#pragma unroll
    for (int t = 1; t < 16; t++) {
        s0 = dpp_row_shr1f(s0) + d0; s1 = dpp_row_shr1f(s1) + d1; s2 = dpp_row_shr1f(s2) + d2; s3 = dpp_row_shr1f(s3) + d3;
    }
    const float h = (s0 + s2) + (s1 + s3);
    return __shfl(h, (lane & ~15) + 15, 64);
}

template <int NU, int NI, int G>
__global__ __launch_bounds__(256) void k_fewrow_i16(const DevParams P, const FusedSchedule S, long begin, long end) {
    const int lane = threadIdx.x & 63;
    const int m = lane & 15;
    const int gslot = lane >> 4;
    long tile = blockIdx.x;
    if (P.xcd_remap) tile = (long)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const long wave = tile * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long w0 = begin + wave * (4L * G);
    if (w0 >= end) return;
    const int pitch = P.pitch;
    const bool use_ubias = P.no_user_bias == 0;

    unsigned ur[G][NU], ir[G][NI];
    float label[G], ua[G][NU], ia[G][NI], bu[G][NU], bi[G][NI];
    float4 p0[G][NU], p1[G][NU], q0[G][NI], q1[G][NI];
#pragma unroll
    for (int g = 0; g < G; g++) {
        const long s = w0 + 4L * g + gslot;
        const bool valid = s < end;
        const long sc = valid ? s : begin;
        label[g] = S.label[sc];
#pragma unroll
        for (int a = 0; a < NU; a++) { ur[g][a] = valid ? S.uidx[a][sc] : (unsigned)SLOT_ABSENT; ua[g][a] = S.uval[a][sc]; }
#pragma unroll
        for (int b = 0; b < NI; b++) { ir[g][b] = valid ? S.iidx[b][sc] : (unsigned)SLOT_ABSENT; ia[g][b] = S.ival[b][sc]; }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
            p0[g][a] = f4zero(); p1[g][a] = f4zero(); bu[g][a] = 0.0f;
            if (ur[g][a] != SLOT_ABSENT) {
                p0[g][a] = load_row_nt<32>(P.W, P.user_off + ur[g][a], pitch, m, 128);
                p1[g][a] = load_row_nt<32>(P.W, P.user_off + ur[g][a], pitch, m + 16, 128);
                if (use_ubias) bu[g][a] = P.bias[P.user_off + ur[g][a]];
            }
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            q0[g][b] = f4zero(); q1[g][b] = f4zero(); bi[g][b] = 0.0f;
            if (ir[g][b] != SLOT_ABSENT) {
                q0[g][b] = load_row_nt<32>(P.W, P.item_off + ir[g][b], pitch, m, 128);
                q1[g][b] = load_row_nt<32>(P.W, P.item_off + ir[g][b], pitch, m + 16, 128);
                bi[g][b] = P.bias[P.item_off + ir[g][b]];
            }
        }
    }
    const float dec_u = snap_to_one(1.0f - P.lr * P.wd_user), dec_i = snap_to_one(1.0f - P.lr * P.wd_item);
    const float dec_ub = 1.0f - P.lr * P.wd_user_bias, dec_ib = 1.0f - P.lr * P.wd_item_bias;
    const float lr = P.lr;
#pragma unroll
    for (int g = 0; g < G; g++) {
        // the arithmetic of k_fewrow_fast<32, NU, NI, true>, two chunks per lane
        double bs = 0.0;
        if (use_ubias) {
#pragma unroll
            for (int a = 0; a < NU; a++) if (ur[g][a] != SLOT_ABSENT) bs += (double)(ua[g][a] * bu[g][a]);
        }
#pragma unroll
        for (int b = 0; b < NI; b++) if (ir[g][b] != SLOT_ABSENT) bs += (double)(ia[g][b] * bi[g][b]);
        double sum = (double)P.base_score + bs;
        float4 tu0 = f4zero(), tu1 = f4zero(), ti0 = f4zero(), ti1 = f4zero();
#pragma unroll
        for (int a = 0; a < NU; a++) if (ur[g][a] != SLOT_ABSENT) { axpy4(tu0, p0[g][a], ua[g][a]); axpy4(tu1, p1[g][a], ua[g][a]); }
#pragma unroll
        for (int b = 0; b < NI; b++) if (ir[g][b] != SLOT_ABSENT) { axpy4(ti0, q0[g][b], ia[g][b]); axpy4(ti1, q1[g][b], ia[g][b]); }
        sum += (double)dot128_i16(tu0, tu1, ti0, ti1, m, lane);
        const float pred = map_active((float)sum, P.active_type);
        const float err = cal_grad(label[g], pred, P.active_type) * 1.0f;
#pragma unroll
        for (int a = 0; a < NU; a++) {
            if (ur[g][a] == SLOT_ABSENT) continue;
            const float su = lr * err * ua[g][a];
            float4 w0 = p0[g][a], w1 = p1[g][a];
            axpy4(w0, ti0, su); axpy4(w1, ti1, su);
            w0.x = w0.x * dec_u; w0.y = w0.y * dec_u; w0.z = w0.z * dec_u; w0.w = w0.w * dec_u;
            w1.x = w1.x * dec_u; w1.y = w1.y * dec_u; w1.z = w1.z * dec_u; w1.w = w1.w * dec_u;
            store_row<32>(P.W, P.user_off + ur[g][a], pitch, m, 128, w0);
            store_row<32>(P.W, P.user_off + ur[g][a], pitch, m + 16, 128, w1);
            if (use_ubias) P.bias[P.user_off + ur[g][a]] = (bu[g][a] + su) * dec_ub;
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            if (ir[g][b] == SLOT_ABSENT) continue;
            const float si = lr * err * ia[g][b];
            float4 w0 = q0[g][b], w1 = q1[g][b];
            axpy4(w0, tu0, si); axpy4(w1, tu1, si);
            w0.x = w0.x * dec_i; w0.y = w0.y * dec_i; w0.z = w0.z * dec_i; w0.w = w0.w * dec_i;
            w1.x = w1.x * dec_i; w1.y = w1.y * dec_i; w1.z = w1.z * dec_i; w1.w = w1.w * dec_i;
            store_row<32>(P.W, P.item_off + ir[g][b], pitch, m, 128, w0);
            store_row<32>(P.W, P.item_off + ir[g][b], pitch, m + 16, 128, w1);
            P.bias[P.item_off + ir[g][b]] = (bi[g][b] + si) * dec_ib;
        }
    }
}

template <int NU, int NI>
static void launch_fewrow_i16_shape(const DevParams &P, const FusedSchedule &S, long begin, long end, int block_threads, int G, hipStream_t st) {
    const long n = end - begin;
    const long per_block = (long)(block_threads / 64) * 4 * G;
    int grid = (int)((n + per_block - 1) / per_block);
    if (P.xcd_remap) grid = (grid + 7) & ~7;
    hipLaunchKernelGGL((k_fewrow_i16<NU, NI, 1>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
}

template <int LPI, int NU, int NI>
static void launch_fewrow_shape(const DevParams &P, const FusedSchedule &S, long begin, long end, int block_threads, hipStream_t st) {
    const long n = end - begin;
    const long per_block = (long)(block_threads / 64) * (64 / LPI);
    int grid = (int)((n + per_block - 1) / per_block);
    if (P.xcd_remap) grid = (grid + 7) & ~7;
    if (P.k == 4 * LPI) hipLaunchKernelGGL((k_fewrow_fast<LPI, NU, NI, true>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
    else hipLaunchKernelGGL((k_fewrow_fast<LPI, NU, NI, false>), dim3(grid), dim3(block_threads), 0, st, P, S, begin, end);
}
template <int LPI>
static void launch_fewrow_lpi(const DevParams &P, const FusedSchedule &S, int nu, int ni, long begin, long end, int bt, hipStream_t st) {
    if (nu <= 1 && ni <= 1) launch_fewrow_shape<LPI, 1, 1>(P, S, begin, end, bt, st);
    else if (nu <= 1) launch_fewrow_shape<LPI, 1, 2>(P, S, begin, end, bt, st);
    else if (ni <= 1) launch_fewrow_shape<LPI, 2, 1>(P, S, begin, end, bt, st);
    else launch_fewrow_shape<LPI, 2, 2>(P, S, begin, end, bt, st);
}
// true when the configuration / data set is the one this kernel is specialised for
bool fewrow_fast_applies(const DevParams &P, const FusedSchedule &S) {
    return S.gptr == nullptr && !P.relax_global && P.relax_user_from == 0xFFFFFFFFu && P.relax_item_from == 0xFFFFFFFFu &&
           P.reg_method == 0 && P.u_rng.n == 0 && P.i_rng.n == 0 && P.user_nonnegative == 0;
}
void launch_fewrow_fast(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int block_threads, hipStream_t st) {
    if (end <= begin) return;
    if (block_threads <= 0) block_threads = 256;
    if (P.k == 128 && P.fewrow_i16) {
        const int G = 1;   // row sets per wave (4 instances each); 2 measured slower: 53.0 vs 45.4 ms per 50 M pairs (32-lane layout: 49.0)
        if (max_nu <= 1 && max_ni <= 1) launch_fewrow_i16_shape<1, 1>(P, S, begin, end, block_threads, G, st);
        else if (max_nu <= 1) launch_fewrow_i16_shape<1, 2>(P, S, begin, end, block_threads, G, st);
        else if (max_ni <= 1) launch_fewrow_i16_shape<2, 1>(P, S, begin, end, block_threads, G, st);
        else launch_fewrow_i16_shape<2, 2>(P, S, begin, end, block_threads, G, st);
        return;
    }
    SVDF_DISPATCH_LPI(lanes_per_instance(P.k), launch_fewrow_lpi<LPI>(P, S, max_nu, max_ni, begin, end, block_threads, st));
}

}  // namespace svdf
