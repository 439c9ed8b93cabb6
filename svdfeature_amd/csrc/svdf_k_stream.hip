// svdf_k_stream.hip -- the conflict DAG executed inside ONE launch per pass (knob "stream_exec").
//
// What it replaces: the launch-per-level loop of Engine::train_dataset.  A level schedule (section 2) orders the instances so that every
// predecessor of an instance -- the previous toucher of each of its parameter rows in FILE order -- sits in an earlier level; launching
// level after level turns that partial order into ~1 900 (ratings) to millions (user-grouped rank passes) of kernel boundaries, each a
// drain + 1.8 us + a latency chain.  Here the same level-sorted arrays are cut into TILES (<= TS consecutive positions of ONE level, so
// the instances of a tile share no row), a persistent grid takes tile tickets IN ORDER, and every instance carries the tile ids of the
// previous touchers of its rows:
//   * an instance may run as soon as those tiles carry this pass's stamp in done[] -- its exact predecessors, not "the whole level
//     before" -- so wide levels stream across their boundaries and narrow levels run as the dependency DAG they are;
//   * forward progress: tickets are handed out in position order and every dependency points to a LOWER tile, so the lowest unfinished
//     tile is always held by a running wave that waits for nothing;
//   * visibility across the 8 XCDs (private L2s): all row / bias traffic of the pass is `sc0 sc1` (write-through stores, loads that do
//     not trust a cached copy), a wave drains its stores (`s_waitcnt vmcnt(0)`) before ONE lane stores the tile's stamp (agent scope),
//     waiters poll that word with relaxed agent-scope loads (/opt/skills/guides/cdna_hip_programming.md guideline 16, form "sc0 sc1
//     stores and loads both sides"); tools/dag_probe prices one such hand-over at 0.63 - 1.08 us on an idle chip;
//   * every spin is bounded: a timeout raises the host-mapped error word, every wave leaves at its next poll, the host fails loudly.
// Per instance the arithmetic is the level kernels' (same device functions, same order): the result equals the level-by-level pass --
// and through it the reference's sequential pass -- bit for bit (tests/test_gpu_stream.py).
#include "svdf_device.h"

namespace svdf {

typedef unsigned int st_u4 __attribute__((ext_vector_type(4)));
static constexpr int ST_AUX = 17;   // sc0 | sc1

__device__ __forceinline__ unsigned ld_agent(const unsigned *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(unsigned *p, unsigned v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float ldf_agent(const float *p) { return __uint_as_float(ld_agent(reinterpret_cast<const unsigned *>(p))); }
__device__ __forceinline__ void stf_agent(float *p, float v) { st_agent(reinterpret_cast<unsigned *>(p), __float_as_uint(v)); }
template <int AUX = ST_AUX>
__device__ __forceinline__ float4 ld_row_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
    const st_u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, AUX);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}
template <int AUX = ST_AUX>
__device__ __forceinline__ void st_row_wt(__amdgpu_buffer_rsrc_t r, unsigned byte_off, const float4 w) {
    st_u4 v = {__float_as_uint(w.x), __float_as_uint(w.y), __float_as_uint(w.z), __float_as_uint(w.w)};
    __builtin_amdgcn_raw_buffer_store_b128(v, r, (int)byte_off, 0, AUX);
}

// ------------------------------------------------------------------------------------------------- plan building
// tile t of level l (tile_base[l] <= t < tile_base[l + 1]) covers positions level_ptr[l] + (t - tile_base[l]) * TS ... (at most TS of them)
__global__ __launch_bounds__(256) void k_stream_tiles(const unsigned *tile_base, const unsigned *level_ptr, long nlevels, int TS, unsigned ntiles,
                                                      uint2 *tile_hdr, unsigned *tile_of_pos) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long t = (long)blockIdx.x * blockDim.x + threadIdx.x; t < (long)ntiles; t += stride) {
        long lo = 0, hi = nlevels;   // last level with tile_base[l] <= t
        while (hi - lo > 1) {
            const long mid = (lo + hi) >> 1;
            if (tile_base[mid] <= (unsigned)t) lo = mid; else hi = mid;
        }
        const unsigned b = level_ptr[lo] + (unsigned)(t - tile_base[lo]) * (unsigned)TS;
        const unsigned e = level_ptr[lo + 1];
        const unsigned c = e - b < (unsigned)TS ? e - b : (unsigned)TS;
        tile_hdr[t] = make_uint2(b, c);
        for (unsigned j = 0; j < c; j++) tile_of_pos[b + j] = (unsigned)t;
    }
}
void launch_stream_tiles(const unsigned *tile_base, const unsigned *level_ptr, long nlevels, int TS, unsigned ntiles, uint2 *tile_hdr,
                         unsigned *tile_of_pos, hipStream_t st) {
    if (ntiles == 0) return;
    long grid = ((long)ntiles + 255) / 256;
    if (grid > 65536) grid = 65536;
    hipLaunchKernelGGL(k_stream_tiles, dim3((int)grid), dim3(256), 0, st, tile_base, level_ptr, nlevels, TS, ntiles, tile_hdr, tile_of_pos);
}
__global__ __launch_bounds__(256) void k_stream_iota(unsigned *v, long n) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < n; j += stride) v[j] = (unsigned)j;
}
void launch_stream_iota(unsigned *v, long n, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_stream_iota, dim3((int)grid), dim3(256), 0, st, v, n);
}
// entries of one id space: entry e = position * members + member (position-major, so ascending e = ascending position), keys[e] = the row
// id of that slot.  After a STABLE sort by key the entries of a row are ascending in e: the previous toucher of entries[j]'s row is
// entries[j - 1] when the keys agree -- unless both belong to the SAME position (one instance listing a row in two slots of a space
// cannot happen: the schedule builders refuse duplicate ids in a row).  absent: key of "no row in this slot".
__global__ __launch_bounds__(256) void k_stream_interleave(const unsigned *c0, const unsigned *c1, const unsigned *c2, int members, long n, unsigned *keys,
                                                           unsigned *entries) {
    const long stride = (long)gridDim.x * blockDim.x;
    const long m = n * members;
    for (long e = (long)blockIdx.x * blockDim.x + threadIdx.x; e < m; e += stride) {
        const long pos = e / members;
        const int mem = (int)(e - pos * members);
        keys[e] = (mem == 0 ? c0 : (mem == 1 ? c1 : c2))[pos];
        entries[e] = (unsigned)e;
    }
}
void launch_stream_interleave(const unsigned *const *cols, int members, long n, unsigned *keys, unsigned *entries, hipStream_t st) {
    if (n <= 0) return;
    long grid = (n * members + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_stream_interleave, dim3((int)grid), dim3(256), 0, st, cols[0], members > 1 ? cols[1] : nullptr, members > 2 ? cols[2] : nullptr, members, n,
                       keys, entries);
}
__global__ __launch_bounds__(256) void k_stream_preds(const unsigned *keys, const unsigned *entries, long m, unsigned absent, const unsigned *tile_of_pos,
                                                      unsigned *pred, int members, int member) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < m; j += stride) {
        const unsigned e = entries[j];
        const unsigned pos = e / (unsigned)members;
        if ((int)(e - pos * (unsigned)members) != member) continue;
        const unsigned key = keys[j];
        pred[pos] = (j > 0 && key != absent && keys[j - 1] == key) ? tile_of_pos[entries[j - 1] / (unsigned)members] : 0xFFFFFFFFu;
    }
}
void launch_stream_preds(const unsigned *keys, const unsigned *entries, long m, unsigned absent, const unsigned *tile_of_pos, unsigned *pred, int members,
                         int member, hipStream_t st) {
    if (m <= 0) return;
    long grid = (m + 255) / 256;
    if (grid > 16384) grid = 16384;
    hipLaunchKernelGGL(k_stream_preds, dim3((int)grid), dim3(256), 0, st, keys, entries, m, absent, tile_of_pos, pred, members, member);
}

// ------------------------------------------------------------------------------------------------- waiting
// every lane checks the stamps of its own instances' predecessor tiles; the wave leaves the loop together.  Returns false on a timeout or
// when another wave has raised the error word.
template <int NP>
__device__ __forceinline__ bool stream_wait(const unsigned (&pred)[NP], const unsigned *done, unsigned pass, unsigned *err, unsigned spin_limit) {
    unsigned spins = 0;
    for (;;) {
        bool ok = true;
#pragma unroll
        for (int j = 0; j < NP; j++) ok &= pred[j] == 0xFFFFFFFFu || ld_agent(done + pred[j]) == pass;
        if (__all(ok)) return true;
        if (++spins > spin_limit || ((spins & 63u) == 0u && ld_agent(err) != 0u)) {
            if (spins > spin_limit) st_agent(err, 1u);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ------------------------------------------------------------------------------------------------- basicMF, the contract configuration
// k_basicmf_slots<LANES, V, G> (svdf_k_basic.hip) as a persistent grid over tiles of TS = G * 64 / LANES instances.
//
// Tile assignment is STATIC: wave w of W takes tiles w, w + W, w + 2 W, ... (a ticket counter was measured first: one word serves ~80
// returning atomics per microsecond, which at 32 instances per ticket caps a pass at 2.4 G instances/s; tickets of several tiles put
// consecutive tiles behind each other in one wave and stretch the dependency path 8-fold).  Forward progress then needs all W waves
// resident at once: the launcher keeps W at or below HALF of what the occupancy query admits, and every wait is bounded anyway.  The
// lowest unfinished tile X belongs to wave X mod W, whose earlier tiles are all finished: it is working on X and X waits for nothing.
// The next tile's header and records are requested while the current one is in flight (the assignment is known in advance).
template <int G>
struct StreamRecs {
    unsigned ur[G], ir[G], pred[2 * G];
    float label[G];
    bool valid[G];
};
template <int G, int IPS>
__device__ __forceinline__ void stream_load_recs(const DevParams &P, const BasicSchedule &S, const StreamPlan &T, const uint2 hdr, int gslot, StreamRecs<G> &R) {
#pragma unroll
    for (int g = 0; g < G; g++) {
        const int j = g * IPS + gslot;
        R.valid[g] = j < (int)hdr.y;
        const unsigned s = hdr.x + (R.valid[g] ? (unsigned)j : 0u);
        R.ur[g] = P.user_off + S.user[s];
        R.ir[g] = P.item_off + S.item[s];
        R.label[g] = S.label[s];
        R.pred[2 * g] = R.valid[g] ? T.pred[0][s] : 0xFFFFFFFFu;
        R.pred[2 * g + 1] = R.valid[g] ? T.pred[1][s] : 0xFFFFFFFFu;
    }
}
template <int LANES, int V, int G, int MODE>
__global__ __launch_bounds__(64) void k_basicmf_stream(const DevParams P, const BasicSchedule S, const StreamPlan T, unsigned pass) {
    constexpr int TT = 16 / LANES;
    constexpr int IPS = 64 / LANES;
    const int lane = threadIdx.x & 63;
    const int m = (lane & 15) / TT;
    const int gslot = (lane >> 4) * TT + (lane & (TT - 1));
    const unsigned pitch_b = (unsigned)P.pitch * 4u;
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc(P.W, 0, (int)T.w_bytes, 0x00020000);
    const float dec_u1 = snap_to_one(1.0f - P.lr * P.wd_user), dec_i1 = snap_to_one(1.0f - P.lr * P.wd_item);
    const unsigned W = gridDim.x;
    unsigned tile = blockIdx.x;
    if (tile >= T.ntiles) return;
    StreamRecs<G> R;
    stream_load_recs<G, IPS>(P, S, T, T.tile_hdr[tile], gslot, R);
    for (;;) {
        const unsigned next = tile + W;
        const bool has_next = next < T.ntiles;
        const uint2 hdr_n = T.tile_hdr[has_next ? next : tile];
        if (!(MODE & 8) && !stream_wait<2 * G>(R.pred, T.done, pass, T.err, T.spin_limit)) return;
        float bu[G], bi[G];
        float4 p[G][V], q[G][V];
#pragma unroll
        for (int g = 0; g < G; g++) {
#pragma unroll
            for (int v = 0; v < V; v++) { p[g][v] = f4zero(); q[g][v] = f4zero(); }
            bu[g] = 0.0f; bi[g] = 0.0f;
            if (R.valid[g]) {
#pragma unroll
                for (int v = 0; v < V; v++) p[g][v] = ld_row_wt<(MODE & 1) ? 0 : ST_AUX>(rW, R.ur[g] * pitch_b + (unsigned)(m + v * LANES) * 16u);
#pragma unroll
                for (int v = 0; v < V; v++) q[g][v] = ld_row_wt<(MODE & 1) ? 0 : ST_AUX>(rW, R.ir[g] * pitch_b + (unsigned)(m + v * LANES) * 16u);
                bu[g] = (MODE & 4) ? P.bias[R.ur[g]] : ldf_agent(P.bias + R.ur[g]);
                bi[g] = (MODE & 4) ? P.bias[R.ir[g]] : ldf_agent(P.bias + R.ir[g]);
            }
        }
        StreamRecs<G> Rn;   // the next tile's records travel behind this tile's rows
        stream_load_recs<G, IPS>(P, S, T, hdr_n, gslot, Rn);
#pragma unroll
        for (int g = 0; g < G; g++) {
            // the arithmetic of k_basicmf_slots (= basicmf_wave<K / 4, ., true, true, true>), V chunks per lane
            double bs = 0.0;
            bs += (double)(1.0f * bu[g]); bs += 0.0;
            bs += 0.0;
            bs += (double)(1.0f * bi[g]);
            double sum = (double)P.base_score + bs;
            float4 tu[V], ti[V];
#pragma unroll
            for (int v = 0; v < V; v++) { tu[v] = f4zero(); ti[v] = f4zero(); axpy4(tu[v], p[g][v], 1.0f); axpy4(ti[v], q[g][v], 1.0f); }
            sum += (double)dot_slots<LANES, V>(tu, ti, m, lane);
            const float predv = (float)sum;
            const float err = (R.label[g] - predv) * 1.0f;
            const float su = P.lr * err * 1.0f;
            const float si = P.lr * err * 1.0f;
            float nbu = bu[g] + su, nbi = bi[g] + si;
            nbu = nbu * (1.0f - P.lr * P.wd_user_bias);
            nbi = nbi * (1.0f - P.lr * P.wd_item_bias);
#pragma unroll
            for (int v = 0; v < V; v++) {
                float4 wu = p[g][v], wi = q[g][v];
                axpy4(wu, ti[v], su);
                axpy4(wi, tu[v], si);
                wu.x = wu.x * dec_u1; wu.y = wu.y * dec_u1; wu.z = wu.z * dec_u1; wu.w = wu.w * dec_u1;
                wi.x = wi.x * dec_i1; wi.y = wi.y * dec_i1; wi.z = wi.z * dec_i1; wi.w = wi.w * dec_i1;
                p[g][v] = wu; q[g][v] = wi;
            }
            if (R.valid[g]) {
#pragma unroll
                for (int v = 0; v < V; v++) st_row_wt<(MODE & 2) ? 0 : ST_AUX>(rW, R.ur[g] * pitch_b + (unsigned)(m + v * LANES) * 16u, p[g][v]);
#pragma unroll
                for (int v = 0; v < V; v++) st_row_wt<(MODE & 2) ? 0 : ST_AUX>(rW, R.ir[g] * pitch_b + (unsigned)(m + v * LANES) * 16u, q[g][v]);
                if (m == 0) {   // one lane per instance: an sc1 word store is one fabric write each
                    if (MODE & 4) { P.bias[R.ur[g]] = nbu; P.bias[R.ir[g]] = nbi; }
                    else { stf_agent(P.bias + R.ur[g], nbu); stf_agent(P.bias + R.ir[g], nbi); }
                }
            }
        }
        if (!(MODE & 16)) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every store of this wave has been written through before the stamp
            if (lane == 0) st_agent(T.done + tile, pass);
        }
        if (!has_next) return;
        tile = next;
        R = Rn;
    }
}

int stream_basic_tile_size(const DevParams &P) { return P.k == 64 ? 32 : 0; }
bool stream_basic_applies(const DevParams &P, const BasicSchedule &S) {
    return P.basic_i8 && S.uval == nullptr && P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 &&
           P.u_rng.n == 0 && P.i_rng.n == 0 && P.k == 64;
}
// the most waves a launch may use: half of what the occupancy query admits (static tile assignment needs every wave resident; the query
// over-reports by up to one block per CU in some SGPR ranges, /opt/skills/guides/MI355X_MICROARCH.md "Residency")
int stream_basic_max_waves(int num_cu) {
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_basicmf_stream<8, 2, 4, 0>, 64, 0) != hipSuccess || per_cu < 2) return num_cu;
    return num_cu * (per_cu / 2);
}
void launch_basicmf_stream(const DevParams &P, const BasicSchedule &S, const StreamPlan &T, unsigned pass, int waves, hipStream_t st) {
    if (T.ntiles == 0) return;
    switch (T.debug_mode) {   // experiment: which of the write-through accesses costs what (any non-zero mode is NOT coherent across XCDs)
    case 1: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 1>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 2: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 2>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 3: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 3>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 4: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 4>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 7: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 7>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 8: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 8>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 16: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 16>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 24: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 24>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    case 31: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 31>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    default: hipLaunchKernelGGL((k_basicmf_stream<8, 2, 4, 0>), dim3(waves), dim3(64), 0, st, P, S, T, pass); break;
    }
}

}  // namespace svdf
