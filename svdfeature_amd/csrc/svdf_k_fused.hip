// svdf_k_fused.hip -- few-row fused kernels (k_fused, k_predict_fused) and their launchers
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_device.h"

namespace svdf {

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
    // global features of an instance, kept in registers when it has at most GR of them and no id twice (the usual case);
    // otherwise they are walked through memory in the reference's order below
    constexpr int GR = 4;
    bool greg[G], gon[G][GR];
    unsigned gid[G][GR];
    float gv[G][GR], gb[G][GR];
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
        greg[g] = false;
        if (S.gptr && valid[g] && (!S.gsi[0] || P.relax_global)) { g0[g] = S.gptr[sc]; g1[g] = S.gptr[sc + 1]; }
    }
    if (S.gptr && !P.relax_global) {   // kernel arguments: wave-uniform, data sets without global features skip all of it
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (S.gsi[0]) {
                // inline slots (every instance of the data set has <= GR distinct global ids): ids and values sit next to
                // the user / item slots of the record, one dependent load less in front of the bias gather
                const long s = w0 + (long)g * IPW + gslot;
                const long sc = valid[g] ? s : begin;
                greg[g] = valid[g];
#pragma unroll
                for (int j = 0; j < GR; j++) {
                    gid[g][j] = valid[g] ? S.gsi[j][sc] : (unsigned)SLOT_ABSENT;
                    gv[g][j] = S.gsv[j][sc];
                    gon[g][j] = gid[g][j] != SLOT_ABSENT;
                }
            } else {
                const int ngl = g1[g] - g0[g];
                greg[g] = ngl <= GR;
#pragma unroll
                for (int j = 0; j < GR; j++) {
                    gon[g][j] = j < ngl && ngl <= GR;
                    gid[g][j] = gon[g][j] ? S.gidx[g0[g] + j] : 0xFFFFFFF0u + (unsigned)j;
                    gv[g][j] = gon[g][j] ? S.gval[g0[g] + j] : 0.0f;
                }
#pragma unroll
                for (int a = 0; a < GR; a++)
#pragma unroll
                    for (int b = a + 1; b < GR; b++) if (gid[g][a] == gid[g][b]) greg[g] = false;
            }
#pragma unroll
            for (int j = 0; j < GR; j++) gb[g][j] = (greg[g] && gon[g][j]) ? P.g_bias[gpos(P, gid[g][j])] : 0.0f;
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
#pragma unroll
        for (int a = 0; a < NU; a++) {
            p[g][a] = f4zero(); bu[g][a] = 0.0f;
            if (ur[g][a] != SLOT_ABSENT) {
                p[g][a] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.user_off + ur[g][a], pitch, L, k) : load_row<LPI>(P.W, P.user_off + ur[g][a], pitch, L, k);
                if (use_ubias) bu[g][a] = P.bias[P.user_off + ur[g][a]];
            }
        }
#pragma unroll
        for (int b = 0; b < NI; b++) {
            q[g][b] = f4zero(); bi[g][b] = 0.0f;
            if (ir[g][b] != SLOT_ABSENT) {
                q[g][b] = (P.load_mode & 1) ? load_row_nt<LPI>(P.W, P.item_off + ir[g][b], pitch, L, k) : load_row<LPI>(P.W, P.item_off + ir[g][b], pitch, L, k);
                bi[g][b] = P.bias[P.item_off + ir[g][b]];
            }
        }
    }
#pragma unroll
    for (int g = 0; g < G; g++) {
        double bs = 0.0;
        if (greg[g]) {
#pragma unroll
            for (int j = 0; j < GR; j++) if (gon[g][j]) bs += (double)(gv[g][j] * gb[g][j]);
        } else {
            for (int j = g0[g]; j < g1[g]; j++) bs += (double)(S.gval[j] * P.g_bias[gpos(P, S.gidx[j])]);
        }
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
        // global biases: registers when the instance's ids are distinct (greg), otherwise through memory in the reference's
        // order (all updates, then all decays), so a global id listed twice behaves like the reference; every lane stores
        // the same value
        if (P.relax_global) {
            // relaxed shared ids: other instances of this launch may be updating the same global -- add this instance's
            // change (update, then decay of the value it read) atomically; one lane per group
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gi = S.gidx[j];
                const float was = P.g_bias[gpos(P, gi)];
                const float nb = reg_gbias(P, gi, was + lr * err * S.gval[j]);
                if (L == 0) unsafeAtomicAdd(&P.g_bias[gpos(P, gi)], nb - was);
            }
        } else if (greg[g]) {
            // distinct ids: "all updates, then all decays" (apex_svd_base.h:384-387, 288-292) is update+decay id by id
#pragma unroll
            for (int j = 0; j < GR; j++)
                if (gon[g][j]) P.g_bias[gpos(P, gid[g][j])] = reg_gbias(P, gid[g][j], gb[g][j] + lr * err * gv[g][j]);
        } else {
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gi = S.gidx[j];
                float gbm = P.g_bias[gpos(P, gi)];
                gbm = gbm + lr * err * S.gval[j];
                P.g_bias[gpos(P, gi)] = gbm;
            }
            for (int j = g0[g]; j < g1[g]; j++) {
                const unsigned gi = S.gidx[j];
                P.g_bias[gpos(P, gi)] = reg_gbias(P, gi, P.g_bias[gpos(P, gi)]);
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
        if (S.gsi[0]) {   // inline slots hold the instance's global ids in file order
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const unsigned gi = S.gsi[j][s];
                if (gi != SLOT_ABSENT) bs += (double)(S.gsv[j][s] * P.g_bias[gpos(P, gi)]);
            }
        } else if (S.gptr) for (int j = S.gptr[s]; j < S.gptr[s + 1]; j++) bs += (double)(S.gval[j] * P.g_bias[gpos(P, S.gidx[j])]);
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
    if (P.fewrow_fast && fewrow_fast_applies(P, S)) { launch_fewrow_fast(P, S, max_nu, max_ni, begin, end, block_threads, st); return; }
    const int lpi_ = lanes_per_instance(P.k);
    if (groups_per_wave <= 0) groups_per_wave = lpi_ == 16 ? 2 : 1;
    if (block_threads <= 0) block_threads = 64;   // one-wave workgroups, see launch_fewrow_fast
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

}  // namespace svdf
