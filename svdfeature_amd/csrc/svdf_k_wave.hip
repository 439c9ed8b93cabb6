// svdf_k_wave.hip -- one-wave-per-user SVD++ kernel (k_svdpp_wave) and its launcher
// (part of the gfx950 kernel set described at the top of svdf_device.h)
#include "svdf_device.h"

namespace svdf {

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
// HW > 1: HW waves per user (one workgroup).  A user's rows are a strict recurrence and stay with wave 0, but the two feedback phases
// are not: prepare_ufeedback GATHERS nfb independent rows (only their accumulation is ordered) and update_ufeedback SCATTERS the same
// delta into nfb independent rows.  One wave keeps 16 rows in flight (0.18 us per id: 36 of the 56 us a 100-row, 100-id user takes,
// and a level holds ~6 users on a 256-CU chip); with helpers, every wave gathers its share of the rows into LDS (linear layout), wave
// 0 accumulates them from there in the reference's order, and the scatter is applied by every wave to its share straight from the
// LDS copies (distinct ids, nothing else touches the rows inside the unit), no second fetch.  Users whose list does not fit the LDS
// budget (lds_rows) and split users' END blocks take the global path.
template <int NR, bool FAST, bool FULL, bool UV, bool RX, int HW = 1>
__device__ __forceinline__ void svdpp_unit_wave(const DevParams &P, const DevCSR &D, const DevUnit &u, const unsigned *fb_index,
                                                const float *fb_value, int lane, int wv = 0, float *lds = nullptr, int lds_rows = 0,
                                                bool have_x = false, int x_e0 = 0, unsigned x_user = 0) {
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
    // ---- helper waves (HW > 1): block-uniform decisions, LDS = rows[lds_rows][64 NR] | bias[lds_rows] | delta[64 NR] | db
    constexpr bool HELP = HW > 1;
    const bool help = HELP && nfb > 0 && nfb <= lds_rows;
    const bool rows_in_lds = help && (u.flags & UNIT_START);
    float *l_rows = lds, *l_bias = lds + (size_t)lds_rows * 64 * NR, *l_delta = l_bias + lds_rows;
    unsigned my_id = 0;
    float my_v = 0.0f;
    if (help) {   // this wave's share of the list: entries wv, wv + HW, ...; lane l keeps entry wv + HW l
        const int j = min(wv + HW * lane, nfb - 1);
        my_id = fidx[j];
        my_v = fval[j];
    }
    if (rows_in_lds) {
        constexpr int GC = 16;   // rows a wave requests before it parks any of them in LDS: HW x GC gathers in flight per user
        for (int t0 = 0; wv + HW * t0 < nfb; t0 += GC) {
            ChainRow<NR> w[GC];
            float b[GC];
#pragma unroll
            for (int c = 0; c < GC; c++) {
                const int j = min(wv + HW * (t0 + c), nfb - 1);   // (past the end: the last row again, not stored)
                const unsigned row = P.fb_off + fidx[j];
                w[c] = lin_load<NR>(P.W, row, pitch, lane, kio);
                b[c] = ub ? P.bias[row] : 0.0f;
            }
#pragma unroll
            for (int c = 0; c < GC; c++) {
                const int j = wv + HW * (t0 + c);
                if (j < nfb) {
#pragma unroll
                    for (int q = 0; q < NR; q++) l_rows[((size_t)j * 64 + lane) * NR + q] = w[c].r[q];
                    if (lane == 0) l_bias[j] = b[c];
                }
            }
        }
        __syncthreads();
    }
    if (!HELP || wv == 0) {
    if (u.flags & UNIT_LOAD) {
        tmp_fb = chain_load<NR>(st, 0, pitch, lane, kio);
        old_fb = chain_load<NR>(st, 1, pitch, lane, kio);
        norm = st[2 * pitch]; tmp_bias = st[2 * pitch + 1]; old_bias = st[2 * pitch + 2];
    }
    if (u.flags & UNIT_START) {   // prepare_ufeedback (:523-538)
        norm = 0.0f; tmp_fb = chain_zero<NR>(); tmp_bias = 0.0f;
        if (rows_in_lds) {   // the rows are in LDS: accumulate them in list order (same operations as the global path below)
            FbBlock blk = fb_block(fidx, fval, 0, nfb, lane);
            constexpr int AC = 8;   // LDS rows read ahead of their (ordered) accumulation; 64 % AC == 0
            for (int j0 = 0; j0 < nfb; j0 += AC) {
                if (j0 > 0 && (j0 & 63) == 0) blk = fb_block(fidx, fval, j0, nfb, lane);
                ChainRow<NR> w[AC];
                float b[AC];
#pragma unroll
                for (int c = 0; c < AC; c++) {
                    const int j = min(j0 + c, nfb - 1);
#pragma unroll
                    for (int q = 0; q < NR; q++) w[c].r[q] = l_rows[((size_t)j * 64 + lane) * NR + q];
                    b[c] = l_bias[j];
                }
#pragma unroll
                for (int c = 0; c < AC; c++) {
                    if (j0 + c < nfb) {
                        const float v = fb_val(blk, (j0 + c) & 63);
                        chain_axpy(tmp_fb, w[c], v);
                        norm = norm + v * v;
                        if (ub) tmp_bias = tmp_bias + b[c] * v;
                    }
                }
            }
            tmp_fb = lin_to_chain<NR>(tmp_fb, lane);
        } else if (nfb > 0) {
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
        // rows are (0,1,1): entries of row j start at e0 + 2j; e0 and the user id come with the launch record when there is one
        // (two dependent loads less in front of the user's row)
        const int e0 = have_x ? x_e0 : uniform_load(D.row_ptr + 3 * (long)u.row_begin);
        const unsigned urow = P.user_off + (have_x ? x_user : uniform_load(D.feat_index + e0));
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
        if (help) {   // every wave scatters its share (after the barrier below): hand the delta over through LDS
#pragma unroll
            for (int q = 0; q < NR; q++) l_delta[lane * NR + q] = dl.r[q];
            if (lane == 0) l_delta[64 * NR] = db;
        } else {
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
        }   // !help
    }
    if (u.flags & UNIT_SAVE) {
        chain_store<NR>(st, 0, pitch, lane, kio, tmp_fb);
        chain_store<NR>(st, 1, pitch, lane, kio, old_fb);
        if (lane == 0) { st[2 * pitch] = norm; st[2 * pitch + 1] = tmp_bias; st[2 * pitch + 2] = old_bias; }
    }
    }   // wave 0
    if (HELP) {
        if (help && (u.flags & UNIT_END)) {   // update_ufeedback (:539-554), every wave its share of the rows
            __syncthreads();
            ChainRow<NR> dl;
#pragma unroll
            for (int q = 0; q < NR; q++) dl.r[q] = l_delta[lane * NR + q];
            const float db = l_delta[64 * NR];
            constexpr int SC = 8;
            for (int t0 = 0; wv + HW * t0 < nfb; t0 += SC) {
                ChainRow<NR> w[SC];
                float b[SC];
#pragma unroll
                for (int c = 0; c < SC; c++) {
                    const int j = min(wv + HW * (t0 + c), nfb - 1);
                    if (rows_in_lds) {
#pragma unroll
                        for (int q = 0; q < NR; q++) w[c].r[q] = l_rows[((size_t)j * 64 + lane) * NR + q];
                        b[c] = l_bias[j];
                    } else {   // a split user's END block: the rows were gathered by an earlier unit
                        const unsigned row = P.fb_off + fidx[j];
                        w[c] = lin_load<NR>(P.W, row, pitch, lane, kio);
                        b[c] = ub ? P.bias[row] : 0.0f;
                    }
                }
#pragma unroll
                for (int c = 0; c < SC; c++) {
                    const int j = wv + HW * (t0 + c);
                    if (j < nfb) {
                        const unsigned row = P.fb_off + (unsigned)__builtin_amdgcn_readlane((int)my_id, t0 + c);
                        const float v = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(my_v), t0 + c));
                        chain_axpy(w[c], dl, v);
                        lin_store<NR>(P.W, row, pitch, lane, kio, w[c]);
                        if (ub) P.bias[row] = b[c] + db * v;
                    }
                }
            }
        }
        __syncthreads();   // the LDS rows / delta are free for the workgroup's next user
    }
}

// Kernel 4a: the simple units of one conflict-free batch, one wave per user (HW = 1) or one workgroup of HW waves per user
template <int NR, bool FAST, bool FULL, bool UV, bool RX, int HW = 1>
__global__ __launch_bounds__(64 * HW) void k_svdpp_wave(const DevParams P, const DevCSR D, const DevUnit *units, const unsigned *fb_index,
                                                        const float *fb_value, const int *order, const DevUnitX *xunits, long begin, long end, int lds_rows) {
    extern __shared__ float svdpp_lds[];
    const int lane = threadIdx.x & 63;
    const int wv = threadIdx.x >> 6;
    for (long s = begin + blockIdx.x; s < end; s += gridDim.x) {
        const DevUnit *up = xunits ? &xunits[s].u : units + __builtin_amdgcn_readfirstlane(order ? order[s] : (int)s);
        DevUnit u;
        u.fb_begin = __builtin_amdgcn_readfirstlane(up->fb_begin); u.fb_end = __builtin_amdgcn_readfirstlane(up->fb_end);
        u.row_begin = __builtin_amdgcn_readfirstlane(up->row_begin); u.row_end = __builtin_amdgcn_readfirstlane(up->row_end);
        u.flags = __builtin_amdgcn_readfirstlane(up->flags);
        const int x_e0 = xunits ? __builtin_amdgcn_readfirstlane(xunits[s].e0) : 0;
        const unsigned x_user = xunits ? (unsigned)__builtin_amdgcn_readfirstlane((int)xunits[s].user) : 0u;
        svdpp_unit_wave<NR, FAST, FULL, UV, RX, HW>(P, D, u, fb_index, fb_value, lane, wv, svdpp_lds, lds_rows, xunits != nullptr, x_e0, x_user);
    }
}

void launch_svdpp_wave(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                       const int *order, const DevUnitX *xunits, long begin, long end, hipStream_t st) {
    if (end <= begin) return;
    long grid = end - begin;           // one 64-thread workgroup (= one wave) per user: a batch rarely holds more users than CUs
    if (grid > 16384) grid = 16384;
    const int nr = (P.k + 63) / 64;
    const bool fast = P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 &&
                      P.u_rng.n == 0 && P.i_rng.n == 0;
#define SVDF_WAVE_LAUNCH(NR_, FAST_, FULL_, UV_, RX_) \
    hipLaunchKernelGGL((k_svdpp_wave<NR_, FAST_, FULL_, UV_, RX_>), dim3((int)grid), dim3(64), 0, st, P, D, units, fb_index, fb_value, order, xunits, begin, end, 0)
    // helper waves (svdpp_helpers knob: 8 waves per user by default -- 301.6 ms per pass of the 40 K-user set against 316.8 with 4,
    // 311.0 with 16 and 368.7 with one wave per user) for the configuration every BASELINE run uses; LDS budget 64 KB
    if (P.svdpp_helpers > 1 && fast && P.k == 64 * nr && D.unit_values && !(P.relax_item_from == 0u || P.relax_feedback != 0) && nr <= 4) {
        const int lds_floats = 16384;
        int rows = (lds_floats - 64 * nr - 4) / (64 * nr + 1);
        const int hw = P.svdpp_helpers >= 16 ? 16 : (P.svdpp_helpers >= 8 ? 8 : 4);
        if (rows > 64 * hw) rows = 64 * hw;   // a wave keeps its share of the ids in one register: 64 entries per wave
        const size_t lds_bytes = ((size_t)rows * (64 * nr + 1) + 64 * nr + 4) * sizeof(float);
#define SVDF_WAVE_HELP(NR_, HW_) hipLaunchKernelGGL((k_svdpp_wave<NR_, true, true, true, false, HW_>), dim3((int)grid), dim3(64 * HW_), lds_bytes, st, P, D, units, fb_index, fb_value, order, xunits, begin, end, rows)
#define SVDF_WAVE_HELP_NR(NR_) \
        if (P.svdpp_helpers >= 16) SVDF_WAVE_HELP(NR_, 16); else if (P.svdpp_helpers >= 8) SVDF_WAVE_HELP(NR_, 8); else SVDF_WAVE_HELP(NR_, 4)
        switch (nr) {
        case 1: SVDF_WAVE_HELP_NR(1); break;
        case 2: SVDF_WAVE_HELP_NR(2); break;
        case 3: SVDF_WAVE_HELP_NR(3); break;
        default: SVDF_WAVE_HELP_NR(4); break;
        }
#undef SVDF_WAVE_HELP_NR
#undef SVDF_WAVE_HELP
        return;
    }
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

// ------------------------------------------------------------------------------------------------- window-minibatch step, one WAVE per user unit
// The window step for user units (svdf_k_wunit.hip) is bound by the LATENCY of its longest unit: a window of the
// BASELINE configs[3] SVD++ data holds ~1 500 users (about one per SIMD), every one a strict recurrence of 100 rows between a gather and a
// scatter of 100 feedback rows.  k_wunit_fast gives a unit 16 lanes (float4 x 2 per lane: ~400 instructions per row step on a wave that
// issues one every >= 4 cycles); this kernel gives it the whole wave in the chain layout of k_svdpp_wave above -- one element per lane and
// register, each SSE accumulation chain in its own DPP row: ~140 instructions per row -- with the window step's semantics: the shared rows
// are only READ (window-start values, so nothing fetched ahead can go stale), what the reference would have changed on them goes to the
// unit's contribution slots.  With every unit of the window in flight at once the memory system is loaded (~3 TB/s) and its latency is a
// multiple of the idle one, so more is kept in flight than in the exact kernel: item rows PFW = 16 ahead, feedback rows in linear layout two
// batches of 16 (one being accumulated / scattered, one requested), the 64-record blocks one block ahead.  Same statements in the same order
// as k_wunit_walk ==> the same bits (tests/test_gpu_wunit.py, tests/fuzz_wunit.py).  Fixed row layout without global entries (estride 1),
// unit user values, full rows (k = 64 NR); contribution rows fp32 or bfloat16.
template <int NR, bool BF16> __device__ __forceinline__ void contrib_chain_store(float *base, size_t slot, int pitch, int lane, const ChainRow<NR> &x) {
    if constexpr (!BF16) chain_store<NR>(base, slot, pitch, lane, -1, x);
    else {
        unsigned short *row = reinterpret_cast<unsigned short *>(base) + slot * (size_t)pitch;
        const int e0 = 4 * NR * (lane & 15) + (lane >> 4);
#pragma unroll
        for (int q = 0; q < NR; q++) row[e0 + 4 * q] = (unsigned short)bf16_rne(x.r[q]);
    }
}
template <int NR, bool BF16> __device__ __forceinline__ void contrib_lin_store(float *base, size_t slot, int pitch, int lane, const ChainRow<NR> &x) {
    if constexpr (!BF16) lin_store<NR>(base, slot, pitch, lane, -1, x);
    else {
        unsigned short *row = reinterpret_cast<unsigned short *>(base) + slot * (size_t)pitch + NR * lane;
        if constexpr (NR == 2) *reinterpret_cast<unsigned *>(row) = bf16_rne(x.r[0]) | (bf16_rne(x.r[1]) << 16);
        else if constexpr (NR == 4) *reinterpret_cast<uint2 *>(row) = make_uint2(bf16_rne(x.r[0]) | (bf16_rne(x.r[1]) << 16), bf16_rne(x.r[2]) | (bf16_rne(x.r[3]) << 16));
        else {
#pragma unroll
            for (int c = 0; c < NR; c++) row[c] = (unsigned short)bf16_rne(x.r[c]);
        }
    }
}
static_assert(sizeof(WinUnit) == 32 && sizeof(WinSeg) == 16 && sizeof(WinEnt) == 16 && offsetof(WinUnit, first) == 16,
              "k_wunit_wave reads unit and segment records as 16-byte words");
// a 64-entry block of a feedback list, one entry per lane (entries past the list repeat its last one), with the entry's feedback-bias word
struct WaveFbBlock { WinEnt e; float b; };
__device__ __forceinline__ WaveFbBlock wave_fb_block(const DevParams &P, const WUnitSchedule &S, int fb_begin, int first, int fb_last, int lane, bool) {
    WaveFbBlock x;
    x.e = S.fbent[fb_begin + min(first + lane, fb_last)];
    x.b = P.bias[P.fb_off + x.e.idx];   // (also without user bias: a single contribution applied in place adds its +0 to the word like the sum kernel)
    return x;
}
template <int NR, int FBW> struct WaveFbBatch { ChainRow<NR> w[FBW]; };
template <int NR, int FBW>   // request the feedback rows of entries off .. off+FBW-1 of a block
__device__ __forceinline__ void wave_fb_issue(const DevParams &P, const WinEnt &fe, int off, int lane, WaveFbBatch<NR, FBW> &o) {
#pragma unroll
    for (int c = 0; c < FBW; c++) o.w[c] = lin_load<NR>(P.W, P.fb_off + pick(fe.idx, off + c), P.pitch, lane, -1);
}
// a 64-record block of a segment's rows, one row per lane: the record (item id, value, contribution slot), the label and the item's bias word
struct WaveRowBlock { WinEnt e; float label, bi; };
__device__ __forceinline__ WaveRowBlock wave_row_block(const DevParams &P, const WUnitSchedule &S, int row_begin, int first, int row_last, int lane) {
    WaveRowBlock x;
    const int j = row_begin + min(first + lane, row_last);
    x.e = S.ent[j];
    x.label = S.label[j];
    x.bi = P.bias[P.item_off + x.e.idx];
    return x;
}
// FAST: the configuration of every BASELINE run (k_svdpp_wave's predicate: linear link, L2 decay, user bias on, no per-range decay, no clamp)
// compiled without the per-row switches.  Vector-memory instructions are the scarce resource of a wave here (at most 64 in flight, and all
// ~1 500 waves of a window run their phases at the same time): bias words come with the 64-entry blocks (one load per block instead of
// one per row), bias contributions leave as one store per group, and what is fetched ahead are row requests only.
template <int NR, int PFW, bool BF16, bool FAST>
__global__ __launch_bounds__(64, (NR <= 2 ? 2 : 1)) void k_wunit_wave(   // two waves per SIMD (<= 256 registers) at k <= 128: a window has ~1.5 units per SIMD
    const DevParams P, const WUnitSchedule S) {
    constexpr int FBW = 16;
    static_assert(64 % PFW == 0 && 64 % (2 * FBW) == 0, "a group must not straddle two 64-record blocks");
    const int lane = threadIdx.x & 63;
    const long uidx = blockIdx.x;
    if (uidx >= S.nunits) return;
    const int pitch = P.pitch, k = 64 * NR, kio = -1;
    const bool ub = FAST ? true : P.no_user_bias == 0;
    const float lr = P.lr, lr2 = P.lr * P.scale_lr_ufeedback;
    const int4 u0 = *reinterpret_cast<const int4 *>(&S.units[uidx]);            // user, seg_begin, seg_count, rows
    const int4 u1 = *(reinterpret_cast<const int4 *>(&S.units[uidx]) + 1);      // the first segment
    const unsigned user = (unsigned)__builtin_amdgcn_readfirstlane(u0.x);
    const int seg_begin = __builtin_amdgcn_readfirstlane(u0.y), seg_count = __builtin_amdgcn_readfirstlane(u0.z);
    const unsigned urow = P.user_off + user;
    ChainRow<NR> p = chain_load<NR>(P.W, urow, pitch, lane, kio);
    float bu = ub ? P.bias[urow] : 0.0f;
    const float wd_u = FAST ? P.wd_user : get_wd(P.u_rng, user, P.wd_user);
    const float dec_ub = 1.0f - lr * P.wd_user_bias, dec_ib = 1.0f - lr * P.wd_item_bias;
    const float dec_u = 1.0f - lr * wd_u, dec_i = 1.0f - lr * P.wd_item;       // FAST: chain_reg's L2 form with the constant rates
    for (int sg = 0; sg < seg_count; sg++) {
        int4 sv = u1;
        if (sg > 0) sv = *reinterpret_cast<const int4 *>(&S.segs[seg_begin + sg]);
        const int fb_begin = __builtin_amdgcn_readfirstlane(sv.x), nfb = __builtin_amdgcn_readfirstlane(sv.y);
        const int row_begin = __builtin_amdgcn_readfirstlane(sv.z), nrow = __builtin_amdgcn_readfirstlane(sv.w);
        const int fb_last = max(nfb - 1, 0), row_last = max(nrow - 1, 0);
        // ---- prepare_ufeedback (:523-538) in LINEAR layout: batches of 16 rows, accumulated in list order, the next batch requested
        ChainRow<NR> tl = chain_zero<NR>();
        float norm = 0.0f, tmp_bias = 0.0f;
        auto gather = [&](const WaveFbBatch<NR, FBW> &x, const WaveFbBlock &fb, int j) {   // entries j .. j+FBW-1
#pragma unroll
            for (int c = 0; c < FBW; c++) {
                if (j + c < nfb) {
                    const float v = pick(fb.e.val, (j & 63) + c);
                    chain_axpy(tl, x.w[c], v);
                    norm = norm + v * v;
                    if (ub) tmp_bias = tmp_bias + pick(fb.b, (j & 63) + c) * v;
                }
            }
        };
        if (nfb > 0) {
            WaveFbBlock fb = wave_fb_block(P, S, fb_begin, 0, fb_last, lane, ub), fb_next = wave_fb_block(P, S, fb_begin, 64, fb_last, lane, ub);
            WaveFbBatch<NR, FBW> A, B;
            wave_fb_issue<NR, FBW>(P, fb.e, 0, lane, A);
            for (int j = 0; j < nfb; j += 2 * FBW) {   // A holds batch j; batch j + FBW lies in the same block
                if (j + FBW < nfb) wave_fb_issue<NR, FBW>(P, fb.e, (j + FBW) & 63, lane, B);
                gather(A, fb, j);
                WaveFbBlock fn = fb;
                if (((j + 2 * FBW) & 63) == 0) { fn = fb_next; fb_next = wave_fb_block(P, S, fb_begin, j + 2 * FBW + 64, fb_last, lane, ub); }
                if (j + 2 * FBW < nfb) wave_fb_issue<NR, FBW>(P, fn.e, (j + 2 * FBW) & 63, lane, A);
                if (j + FBW < nfb) gather(B, fb, j + FBW);
                fb = fn;
            }
        }
        ChainRow<NR> tmp_fb = lin_to_chain<NR>(tl, lane);
        const ChainRow<NR> old_fb = tmp_fb;
        const float old_bias = tmp_bias;
        // ---- the rows: records 64 at a time (one per lane, the next block requested a block ahead), item rows PFW ahead
        WaveRowBlock rb = wave_row_block(P, S, row_begin, 0, row_last, lane), rb_next = wave_row_block(P, S, row_begin, 64, row_last, lane);
        ChainRow<NR> cur[PFW], nxt[PFW];
#pragma unroll
        for (int c = 0; c < PFW; c++) cur[c] = chain_load<NR>(P.W, P.item_off + pick(rb.e.idx, c), pitch, lane, kio);
        for (int j0 = 0; j0 < nrow; j0 += PFW) {
            const int off_cur = j0 & 63, off_pre = (j0 + PFW) & 63;
            WaveRowBlock rb_pre = rb;
            if (off_pre == 0) { rb_pre = rb_next; rb_next = wave_row_block(P, S, row_begin, j0 + PFW + 64, row_last, lane); }   // the next group starts a new block
            if (j0 + PFW < nrow) {
#pragma unroll
                for (int c = 0; c < PFW; c++) nxt[c] = chain_load<NR>(P.W, P.item_off + pick(rb_pre.e.idx, off_pre + c), pitch, lane, kio);
            }
            float cbv = 0.0f;   // lane off_cur + c: the bias contribution of row j0 + c
#pragma unroll
            for (int c = 0; c < PFW; c++) {
                if (j0 + c < nrow) {
                    const ChainRow<NR> &xq = cur[c];
                    const float label = pick(rb.label, off_cur + c), iv = pick(rb.e.val, off_cur + c), bi = pick(rb.bi, off_cur + c);
                    const int slot = pick(rb.e.slot, off_cur + c);
                    double bs = 0.0;
                    if (ub) { bs += (double)(1.0f * bu); bs += (double)tmp_bias; }
                    bs += 0.0;
                    bs += (double)(iv * bi);
                    double sum = (double)P.base_score + bs;
                    ChainRow<NR> tu = tmp_fb, ti = chain_zero<NR>();
                    chain_axpy(tu, p, 1.0f);
                    chain_axpy(ti, xq, iv);
                    sum += (double)chain_dot(tu, ti, lane, k);
                    const float pred = FAST ? (float)sum : map_active((float)sum, P.active_type);
                    const float err = (FAST ? label - pred : cal_grad(label, pred, P.active_type)) * 1.0f;
                    const float su = lr * err * 1.0f, si = lr * err * iv;
                    ChainRow<NR> wu = p;
                    chain_axpy(wu, ti, su);
                    float nbu = bu + su;
                    ChainRow<NR> w = xq;
                    chain_axpy(w, tu, si);
                    float nbi = bi + si;
                    if (FAST) chain_scale(w, dec_i);
                    else chain_reg(P, w, get_wd(P.i_rng, pick(rb.e.idx, off_cur + c), P.wd_item), true, lane, k);
                    nbi = nbi * dec_ib;
#pragma unroll
                    for (int q = 0; q < NR; q++) w.r[q] = w.r[q] - xq.r[q];
                    if (slot < 0) {   // the row's only contribution of this window: applied here (apply_single, svdf_device.h)
                        ChainRow<NR> a;
#pragma unroll
                        for (int q = 0; q < NR; q++) a.r[q] = apply_single(xq.r[q], w.r[q], BF16);
                        chain_store<NR>(P.W, P.item_off + pick(rb.e.idx, off_cur + c), pitch, lane, kio, a);
                    } else {
                        contrib_chain_store<NR, BF16>(S.contrib, (size_t)slot, pitch, lane, w);
                    }
                    cbv = (lane == off_cur + c) ? nbi - bi : cbv;
                    chain_axpy(tmp_fb, ti, lr2 * err * norm);          // update_svdpp (:512-520)
                    chain_scale(tmp_fb, 1.0f - lr2 * P.wd_ufeedback);
                    if (ub) {
                        tmp_bias = tmp_bias + lr2 * err * norm;
                        tmp_bias = tmp_bias * (1.0f - lr2 * P.wd_ufeedback_bias);
                    }
                    if (FAST) chain_scale(wu, dec_u);
                    else chain_reg(P, wu, wd_u, false, lane, k);
                    nbu = nbu * dec_ub;
                    p = wu;
                    if (ub) bu = nbu;
                }
            }
            if (lane >= off_cur && lane < off_cur + PFW && j0 + (lane - off_cur) < nrow) {   // one store for the group
                if (rb.e.slot < 0) P.bias[P.item_off + rb.e.idx] = apply_single(rb.bi, cbv, false);
                else S.cbias[rb.e.slot] = cbv;
            }
#pragma unroll
            for (int c = 0; c < PFW; c++) cur[c] = nxt[c];
            rb = rb_pre;
        }
        // ---- update_ufeedback (:539-554) against the window-start rows: contributions (w + d val) - w, linear layout, same pipeline
        if (nfb > 0) {
            ChainRow<NR> d = tmp_fb;
#pragma unroll
            for (int q = 0; q < NR; q++) d.r[q] = d.r[q] - old_fb.r[q];
            float db = tmp_bias - old_bias;
            const float inv = 1.0f / norm;
            chain_scale(d, inv);
            db = db * inv;
            const ChainRow<NR> dl = chain_to_lin<NR>(d, lane);
            if (S.fbrec) {   // deferred scatter (round 5): the segment's delta goes out once; k_wunit_sum forms (w + d val) - w against the rows it updates
                lin_store<NR>(S.dvec, (size_t)(seg_begin + sg), pitch, lane, -1, dl);
                if (lane == 0) S.dbias[seg_begin + sg] = db;
                continue;
            }
            auto scatter = [&](const WaveFbBatch<NR, FBW> &x, const WaveFbBlock &fb, int j) {
#pragma unroll
                for (int c = 0; c < FBW; c++) {
                    if (j + c < nfb) {
                        const float v = pick(fb.e.val, (j & 63) + c);
                        const int slot = pick(fb.e.slot, (j & 63) + c);
                        ChainRow<NR> w2 = x.w[c];
                        chain_axpy(w2, dl, v);
#pragma unroll
                        for (int q = 0; q < NR; q++) w2.r[q] = w2.r[q] - x.w[c].r[q];
                        if (slot < 0) {
                            ChainRow<NR> a;
#pragma unroll
                            for (int q = 0; q < NR; q++) a.r[q] = apply_single(x.w[c].r[q], w2.r[q], BF16);
                            lin_store<NR>(P.W, P.fb_off + pick(fb.e.idx, (j & 63) + c), pitch, lane, -1, a);
                        } else {
                            contrib_lin_store<NR, BF16>(S.contrib, (size_t)slot, pitch, lane, w2);
                        }
                    }
                }
            };
            auto scatter_bias = [&](const WaveFbBlock &fb, int first) {   // the block's bias contributions, one entry per lane: one store per block
                float cb = 0.0f;
                if (ub) { const float b2 = fb.b + db * fb.e.val; cb = b2 - fb.b; }
                if (first + lane < nfb) {
                    if (fb.e.slot < 0) P.bias[P.fb_off + fb.e.idx] = apply_single(fb.b, cb, false);
                    else S.cbias[fb.e.slot] = cb;
                }
            };
            WaveFbBlock fb = wave_fb_block(P, S, fb_begin, 0, fb_last, lane, ub), fb_next = wave_fb_block(P, S, fb_begin, 64, fb_last, lane, ub);
            WaveFbBatch<NR, FBW> A, B;
            wave_fb_issue<NR, FBW>(P, fb.e, 0, lane, A);
            scatter_bias(fb, 0);
            for (int j = 0; j < nfb; j += 2 * FBW) {
                if (j + FBW < nfb) wave_fb_issue<NR, FBW>(P, fb.e, (j + FBW) & 63, lane, B);
                scatter(A, fb, j);
                WaveFbBlock fn = fb;
                if (((j + 2 * FBW) & 63) == 0) {
                    fn = fb_next; fb_next = wave_fb_block(P, S, fb_begin, j + 2 * FBW + 64, fb_last, lane, ub);
                    if (j + 2 * FBW < nfb) scatter_bias(fn, j + 2 * FBW);
                }
                if (j + 2 * FBW < nfb) wave_fb_issue<NR, FBW>(P, fn.e, (j + 2 * FBW) & 63, lane, A);
                if (j + FBW < nfb) scatter(B, fb, j + FBW);
                fb = fn;
            }
        }
    }
    chain_store<NR>(P.W, urow, pitch, lane, kio, p);
    if (ub && lane == 0) P.bias[urow] = bu;
}
bool wunit_wave_applies(const DevParams &P, const WUnitSchedule &S, bool feedback) {
    return feedback && S.rptr == nullptr && S.estride == 1 && S.uval == nullptr && P.k % 64 == 0 && P.k <= 256;
}
template <int NR, int PFW> static void launch_wunit_wave_nr(const DevParams &P, const WUnitSchedule &S, hipStream_t st) {
    const bool fast = P.active_type == ACT_LINEAR && P.reg_method == 0 && P.no_user_bias == 0 && P.user_nonnegative == 0 && P.u_rng.n == 0 && P.i_rng.n == 0;
    const dim3 grid((unsigned)S.nunits), block(64);
    if (S.contrib_bf16) {
        if (fast) hipLaunchKernelGGL((k_wunit_wave<NR, PFW, true, true>), grid, block, 0, st, P, S);
        else hipLaunchKernelGGL((k_wunit_wave<NR, PFW, true, false>), grid, block, 0, st, P, S);
    } else {
        if (fast) hipLaunchKernelGGL((k_wunit_wave<NR, PFW, false, true>), grid, block, 0, st, P, S);
        else hipLaunchKernelGGL((k_wunit_wave<NR, PFW, false, false>), grid, block, 0, st, P, S);
    }
}
void launch_wunit_wave(const DevParams &P, const WUnitSchedule &S, hipStream_t st) {
    if (S.nunits <= 0) return;
    switch (P.k / 64) {
    case 1: launch_wunit_wave_nr<1, 8>(P, S, st); break;
    case 2: launch_wunit_wave_nr<2, 8>(P, S, st); break;
    case 3: launch_wunit_wave_nr<3, 8>(P, S, st); break;
    default: launch_wunit_wave_nr<4, 8>(P, S, st); break;
    }
}


// ------------------------------------------------------------------------------------------------- user-run units of rank pairs (round 6)
// One wave per unit (PairUnit): the user's row (and bias) in registers, the unit's pairs walked in file order -- update_inner (apex_svd_base.h:456-462) on
// (user:1, {lower item: vlo, higher item: -vlo}, label 1) pair after pair: the arithmetic of k_window_users<., ., 2> / k_fewrow_slots in the chain layout.
// The two item rows of the pairs PUF ahead are in flight while a pair is computed: the unit's item ids are pairwise distinct (the builder cuts a unit where
// an id repeats) and no other unit of the level touches them, so a row fetched ahead cannot go stale.  PRED: scores only (out[pair], nothing written).
constexpr int PUF = 4;
// PLAIN: the demo configuration (sigmoid rank loss, L2 decay, no per-range decay): the switches over links and regularisers are compiled out of the chain.
template <int NR, bool FULL, bool PRED, bool PLAIN>
__device__ __forceinline__ void pair_unit_walk(const DevParams &P, const PairUnitSchedule &S, long idx, int lane, float *out) {
    PairUnit u;
    u.user = uniform_load(&S.units[idx].user); u.begin = uniform_load(&S.units[idx].begin); u.count = uniform_load(&S.units[idx].count); u.pad = 0;
    const int pitch = P.pitch;
    const int k = FULL ? 64 * NR : P.k;
    const int kio = FULL ? -1 : P.k;
    const bool ub = P.no_user_bias == 0;
    const unsigned urow = P.user_off + u.user;
    ChainRow<NR> p = chain_load<NR>(P.W, urow, pitch, lane, kio);
    float bu = ub ? P.bias[urow] : 0.0f;
    const float wd_u = PLAIN ? P.wd_user : get_wd(P.u_rng, u.user, P.wd_user);
    const float lr = P.lr;
    const float dec_u = 1.0f - lr * wd_u, dec_i = 1.0f - lr * P.wd_item;   // (PLAIN: chain_reg's method 0 with the row-invariant scalars hoisted)
    struct Ahead { ChainRow<NR> ql, qh; float bl, bh, vl; unsigned rl, rh; };
    Ahead cur[PUF], nxt[PUF];
    auto fetch = [&](int j0, Ahead *a) {
#pragma unroll
        for (int c = 0; c < PUF; c++) {
            const long t = (long)u.begin + min(j0 + c, u.count - 1);   // (past the end: the last pair again, fetched, never used)
            a[c].rl = P.item_off + uniform_load(S.lo + t);
            a[c].rh = P.item_off + uniform_load(S.hi + t);
            a[c].vl = uniform_load(S.vlo + t);
        }
#pragma unroll
        for (int c = 0; c < PUF; c++) {
            a[c].ql = chain_load<NR>(P.W, a[c].rl, pitch, lane, kio);
            a[c].qh = chain_load<NR>(P.W, a[c].rh, pitch, lane, kio);
            a[c].bl = P.bias[a[c].rl];
            a[c].bh = P.bias[a[c].rh];
        }
    };
    fetch(0, cur);
    for (int j0 = 0; j0 < u.count; j0 += PUF) {
        if (j0 + PUF < u.count) fetch(j0 + PUF, nxt);
#pragma unroll
        for (int c = 0; c < PUF; c++) {
            if (j0 + c < u.count) {
                const Ahead &x = cur[c];
                const float vl = x.vl, vh = -x.vl;
                double bs = 0.0;                                   // calc_bias (:313-353); "+ 0.0": the svdpp / plugin hooks
                if (ub) { bs += (double)(1.0f * bu); bs += 0.0; }
                bs += 0.0;
                bs += (double)(vl * x.bl);
                bs += (double)(vh * x.bh);
                double sum = (double)P.base_score + bs;
                ChainRow<NR> tu = chain_zero<NR>(), ti = chain_zero<NR>();   // prepare_tmp (:354-381)
                chain_axpy(tu, p, 1.0f);
                chain_axpy(ti, x.ql, vl);
                chain_axpy(ti, x.qh, vh);
                sum += (double)chain_dot(tu, ti, lane, k);
                const float pred = PLAIN ? (float)sum : map_active((float)sum, P.active_type);
                if (PRED) {
                    if (lane == 0) out[(long)u.begin + j0 + c] = pred;
                } else {
                    const float err = (PLAIN ? 1.0f - 1.0f / (1.0f + glibc_expf(-pred)) : cal_grad(1.0f, pred, P.active_type)) * 1.0f;
                    const float su = lr * err * 1.0f;              // update_no_decay (:383-427) + regularize (:286-311)
                    ChainRow<NR> wu = p;
                    chain_axpy(wu, ti, su);
                    float nbu = bu + su;
                    if (PLAIN) chain_scale(wu, dec_u); else chain_reg(P, wu, wd_u, false, lane, k);
                    nbu = nbu * (1.0f - lr * P.wd_user_bias);
                    const float sl = lr * err * vl, sh = lr * err * vh;
                    ChainRow<NR> wl = x.ql, wh = x.qh;
                    chain_axpy(wl, tu, sl);
                    chain_axpy(wh, tu, sh);
                    float nbl = x.bl + sl, nbh = x.bh + sh;
                    if (PLAIN) { chain_scale(wl, dec_i); chain_scale(wh, dec_i); }
                    else {
                        chain_reg(P, wl, get_wd(P.i_rng, x.rl - P.item_off, P.wd_item), true, lane, k);
                        chain_reg(P, wh, get_wd(P.i_rng, x.rh - P.item_off, P.wd_item), true, lane, k);
                    }
                    nbl = nbl * (1.0f - lr * P.wd_item_bias);
                    nbh = nbh * (1.0f - lr * P.wd_item_bias);
                    chain_store<NR>(P.W, x.rl, pitch, lane, kio, wl);
                    chain_store<NR>(P.W, x.rh, pitch, lane, kio, wh);
                    if (lane == 0) { P.bias[x.rl] = nbl; P.bias[x.rh] = nbh; }
                    p = wu;
                    if (ub) bu = nbu;
                }
            }
        }
#pragma unroll
        for (int c = 0; c < PUF; c++) cur[c] = nxt[c];
    }
    if (!PRED) {
        chain_store<NR>(P.W, urow, pitch, lane, kio, p);
        if (ub && lane == 0) P.bias[urow] = bu;
    }
}
template <int NR, bool FULL, bool PRED, bool PLAIN>
__global__ __launch_bounds__(64) void k_pair_units(const DevParams P, const PairUnitSchedule S, long begin, long end, float *out) {
    const long idx = begin + blockIdx.x;
    if (idx < end) pair_unit_walk<NR, FULL, PRED, PLAIN>(P, S, idx, (int)threadIdx.x, out);
}
bool pair_units_applies(const DevParams &P) { return P.k >= 1 && P.k <= 256 && P.reg_method <= 3; }
void launch_pair_units(const DevParams &P, const PairUnitSchedule &S, long begin, long end, float *out, hipStream_t st) {
    if (end <= begin) return;
    const int nr = (P.k + 63) / 64;
    const bool full = P.k == 64 * nr;
    const dim3 grid((unsigned)(end - begin)), block(64);
    const bool plain = P.active_type == ACT_SIGMOID_RANK && P.reg_method == 0 && P.u_rng.n == 0 && P.i_rng.n == 0 && P.user_nonnegative == 0;
#define PU_LAUNCH(NR_, FULL_) { if (out) hipLaunchKernelGGL((k_pair_units<NR_, FULL_, true, false>), grid, block, 0, st, P, S, begin, end, out); \
                                else if (plain) hipLaunchKernelGGL((k_pair_units<NR_, FULL_, false, true>), grid, block, 0, st, P, S, begin, end, out); \
                                else hipLaunchKernelGGL((k_pair_units<NR_, FULL_, false, false>), grid, block, 0, st, P, S, begin, end, out); }
#define PU_BY_FULL(NR_) { if (full) PU_LAUNCH(NR_, true) else PU_LAUNCH(NR_, false) }
    if (nr == 1) PU_BY_FULL(1) else if (nr == 2) PU_BY_FULL(2) else if (nr == 3) PU_BY_FULL(3) else PU_BY_FULL(4)
#undef PU_BY_FULL
#undef PU_LAUNCH
}

}  // namespace svdf
