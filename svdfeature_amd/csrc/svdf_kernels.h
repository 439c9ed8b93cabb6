// svdf_kernels.h -- launch wrappers of the gfx950 kernels (svdf_k_*.hip)
#ifndef SVDF_KERNELS_H_
#define SVDF_KERNELS_H_

#include <hip/hip_runtime.h>

#include <type_traits>
#include <vector>

#include "svdf_types.h"

namespace svdf {

// lanes of a wave that own one factor row (one float4 each): next power of two >= ceil(k/4)
int lanes_per_instance(int k);
int max_supported_factor();
int max_fast_path_factor();   // widest row of the register-tiled kernels (k_basicmf, k_fused, simple SVD++ units)

// every launch below processes ONE conflict-free batch [begin,end) on stream st
void launch_basicmf(const DevParams &P, const BasicSchedule &S, long begin, long end, int groups_per_wave, int block_threads, hipStream_t st);
bool launch_basicmf_chain(const DevParams &P, const BasicSchedule &S, const long *d_level_ptr, long l0, long l1, hipStream_t st);
// few-row fused kernel: instances with <= max_nu (1|2) user ids and <= max_ni (1|2) item ids
void launch_fused(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int groups_per_wave,
                  int block_threads, hipStream_t st);
// the same step specialised for data sets without global features under L2 decay without ranges / relaxed ids (svdf_k_fewrow.hip)
bool fewrow_fast_applies(const DevParams &P, const FusedSchedule &S);
bool launch_fewrow_chain(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, const long *d_level_ptr, long l0, long l1, hipStream_t st);
bool fewrow_gslots_applies(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, bool dense_slots);
void launch_fewrow_gslots(const DevParams &P, const FusedSchedule &S, long begin, long end, hipStream_t st);
void launch_fewrow_fast(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long begin, long end, int block_threads, hipStream_t st);
void launch_predict_fused(const DevParams &P, const FusedSchedule &S, int max_nu, int max_ni, long n, float *out, hipStream_t st);
// counter_base: the reference's sample_counter at the first instance of D (row r runs with counter_base + r); only the
// lazy decay modes read it
void launch_general(const DevParams &P, const DevCSR &D, const int *order, long begin, long end, unsigned counter_base, hipStream_t st);
void launch_svdpp(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                  const int *order, long begin, long end, unsigned counter_base, hipStream_t st);
// units flagged UNIT_SIMPLE (num_factor <= 256): one wave per user, see k_svdpp_wave
void launch_svdpp_wave(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                       const int *order, const DevUnitX *xunits, long begin, long end, hipStream_t st);
// extend_type 2 (multi-level implicit feedback): units = block ranges of blks[]; predict_out != nullptr scores instead of training
void launch_imfb(const DevParams &P, const DevCSR &D, const DevUnit *units, const DevBlk *blks, const unsigned *fb_index, const float *fb_value,
                 const int *order, long begin, long end, unsigned counter_base, float *predict_out, hipStream_t st);
// read-only scoring
void launch_predict(const DevParams &P, const DevCSR &D, long n, float *out, hipStream_t st);
void launch_predict_basic(const DevParams &P, const BasicSchedule &S, long n, float *out, hipStream_t st);
void launch_svdpp_predict(const DevParams &P, const DevCSR &D, const DevUnit *units, const unsigned *fb_index, const float *fb_value,
                          long nunit, float *out, hipStream_t st);
// multi-GPU item-side delta over n floats
void launch_delta_pack(const DeltaRanges &R, const float *snap, void *dst, int half, hipStream_t st);
void launch_delta_unpack(const DeltaRanges &R, float *snap, const void *src, int half, int refresh, hipStream_t st);
// window-minibatch step (svdf_k_window.hip): user walk with the item side read-only, per-item sum of the contributions into the wire
// buffer (item range [lo, hi) + nglobal zeros), and replicated ranges += all-reduced wire buffer
bool window_slots_applies(const DevParams &P, const WindowSchedule &S);
void launch_window_apply(const DevParams &P, const WindowSchedule &S, long num_item, float *w_item, float *i_bias, hipStream_t st);   // a window WITH hot items: ordered sub-steps of the hot ones beside the in-place sums of the others, one launch
void launch_window_users(const DevParams &P, const WindowSchedule &S, int slots, int groups_per_wave, hipStream_t st);
void launch_window_items(const WindowSchedule &S, int pitch, int k, long lo, long hi, long nglobal, void *dst, int half, hipStream_t st, long nslots = -1);
void launch_delta_addto(const DeltaRanges &R, const void *src, int half, hipStream_t st);
void launch_window_items_local(const WindowSchedule &S, int pitch, int k, long lo, long hi, float *w_item, float *i_bias, hipStream_t st, long nslots = -1);   // nslots: contributions in the window (sparse windows take k_window_items_sparse)
void launch_ranges_copy(const DeltaRanges &R, float *buf, int set, hipStream_t st);
// the same step for user units: user-group (SVD++) blocks, rows with global features (svdf_k_wunit.hip)
void launch_wunit_walk(const DevParams &P, const WUnitSchedule &S, bool feedback, int fast, hipStream_t st);
bool wunit_wave_applies(const DevParams &P, const WUnitSchedule &S, bool feedback);   // svdf_k_wave.hip: one wave per user unit (SVD++ shape)
void launch_wunit_wave(const DevParams &P, const WUnitSchedule &S, hipStream_t st);
void launch_wunit_sum(const DevParams &P, const WUnitSchedule &S, void *dst, int half, hipStream_t st);   // dst == nullptr: add to the model in place
// cross-process direct exchange (svdf_ipc.cpp): sequence flags in IPC-mapped device memory
void launch_ipc_signal(unsigned *const *pages, int n, int phase, int me, unsigned seq, const unsigned *err, hipStream_t st);
void launch_ipc_wait(unsigned *page, int phase, int n, unsigned seq, unsigned *err, unsigned long long spin_limit, hipStream_t st);
void launch_ipc_copy(float *dst, const float *src, long n, const unsigned *err, hipStream_t st);
void launch_window_user_column(const WinUser *urec, int nusers, unsigned *user_out, hipStream_t st);
void launch_pairs_prepare(long n, const unsigned *pos, const unsigned *neg, unsigned *lo, unsigned *hi, float *vlo, float *vhi, float *ones,
                          unsigned *flag, hipStream_t st);
void launch_delta_sum(const void *const *srcs, int n, void *dst, long total, int half, hipStream_t st);   // up to 16 buffers
// rank d's share of the direct exchange: elements [begin, end) of every buffer <- their sum over the n buffers (rank order, fp32)
void launch_delta_reduce_gather(void *const *bufs, int n, long begin, long end, int half, hipStream_t st, const unsigned *err = nullptr);
void launch_rows_strided_copy(float *dst, long dst_first, long dst_stride, const float *src, long src_first, long src_stride, long nrows, int width,
                              hipStream_t st);
void launch_delta_sub(const float *cur, const float *snap, float *delta, long n, hipStream_t st);
void launch_delta_add(float *cur, const float *snap, const float *delta, long n, hipStream_t st);
// ---- ranker (svdf_k_rank.hip): SVDFeatureRanker's prepare_ifactor / proc_user / proc_spec / proc_rank, and the evaluator's sum
void launch_rank_items(const DevParams &P, const DevCSR &D, long first, long n, float *ifactors, float *ibias, hipStream_t st);
void launch_rank_feedback(const DevParams &P, const unsigned *fidx, const float *fval, int nfb, float *fb_out, hipStream_t st);
struct RankSection { int nu, npos, nprev, nnew; };   // words staged per ranker section: uidx[nu] uval[nu] pos[npos] prev[nprev] new_idx[nnew] new_tag[nnew]
void launch_rank_user(const DevParams &P, const unsigned *stage, const RankSection &S, const float *fb_in, float *tu_out, signed char *tag, int *cnt,
                      unsigned *flag, long cap, const float *ifT, const float *ibias, float *pos_score, unsigned *zero_words, int nzero, hipStream_t st);
void launch_rank_spec(const DevParams &P, const DevCSR &D, long n, const int *spec_idx, const float *tu, float *item_score, hipStream_t st);
void launch_rank_transpose(const DevParams &P, long first, long n, long cap, const float *ifactors, float *ifT, hipStream_t st);
// what the scoring pass does besides the scores: 1 = count the positives' rank positions, 2 = emit the top_k sort keys
struct RankFused { int mode; const int *pos_item; const float *pos_score; int npos; int *greater, *ties; unsigned *keys, *vals, *flag, *hist1; };
void launch_rank_score(const DevParams &P, long n, long cap, const float *tu, const float *ifT, const float *ibias, const signed char *tag,
                       float *item_score, int fresh, const RankFused &F, hipStream_t st);
// a tile of user sections sharing one pass over the candidate matrix (no special samples): per section u the staged words at off[u] are
// uidx[nu] uval[nu] pos[npos] ban[nban]; its counters / positive scores start at entry pos0[u].  32 = the bits of a ban word.
#define RANK_TILE 32
struct RankTile { int nsec; int off[RANK_TILE], nu[RANK_TILE], npos[RANK_TILE], nban[RANK_TILE], pos0[RANK_TILE]; };
// tuT: the tile's user factors chunk-major, RANK_TILE * pitch floats (written here, read as scalars by the scoring pass)
void launch_rank_tile_open(const DevParams &P, const unsigned *stage, const RankTile &T, const float *fb_in, float *tuT, unsigned *banmask,
                           const unsigned *prev_ban, int nprev, int *cnt, unsigned *flag, long cap, const float *ifT, const float *ibias, float *pos_score,
                           unsigned *zero_words, long nzero, hipStream_t st);
// mode 0: out[u * cap + i] = score bits of the ranked candidates + the positives' counters; mode 1: out = sort keys of every candidate,
// wmin[u * rank_tile_minima(n) + wave] = per-wave minimum keys
void launch_rank_score_tile(const DevParams &P, long n, long cap, const float *tuT, const float *ifT, const float *ibias, const unsigned *banmask, unsigned *out,
                            const unsigned *stage, const RankTile &T, const float *pos_score, int *cnt, int mode, unsigned *wmin, unsigned *flag, hipStream_t st);
struct RselSecs { unsigned K1[RANK_TILE]; };
long rank_tile_minima(long n);
bool rank_tile_select_applies(long n, long cap, long K1max);
// top_k of every section of a tile from its keys and wave minima: out + u * out_stride = K1[u] keys, K1[u] candidates, flag word
void launch_rank_tile_select(long n, long cap, int nsec, const unsigned *keys, const unsigned *wmin, const RselSecs &Ks, unsigned *out, long out_stride,
                             const unsigned *flag, hipStream_t st);
void launch_rank_select_tile(long n, long cap, int nsec, const unsigned *keys, const RselSecs &Ks, unsigned *work, unsigned *ck, unsigned *cv, unsigned *out,
                             long out_stride, unsigned *flag, hipStream_t st);
void launch_rank_positions(long n, const float *score, const signed char *tag, const int *pos_item, int npos, int *greater, int *ties, hipStream_t st);
// radix selection of the K1 smallest (key, value) pairs, ascending (svdf_k_rank.hip): work = rank_select_work_words() words
// (zeroed by k_rank_user, first histogram filled by the scoring pass: RankFused::hist1 = work), ck / cv = rank_select_cap()
// words each, K1 <= rank_select_cap() / 2; out = K1 keys, K1 values, flag word (| 2 when too many keys tie at the threshold)
long rank_select_work_words();
long rank_select_cap();
void launch_rank_select(long n, const unsigned *keys, const unsigned *vals, unsigned K1, unsigned *work, unsigned *ck, unsigned *cv, unsigned *out,
                        const unsigned *flag, hipStream_t st);
// ascending radix sort of (key, value) pairs (svdf_k_sched.hip, rocPRIM); tmp is grown as needed
long device_exclusive_scan_u32(const unsigned *in, unsigned *out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st);
// runs of an item's consecutive ratings as schedule units (svdf_k_runs.hip)
void launch_runs_check(const unsigned *user, const unsigned *item, long n, unsigned nu, unsigned ni, unsigned *flag, hipStream_t st);
void launch_runs_prev(const unsigned *keys, const unsigned *pos, long n, unsigned *prev, hipStream_t st);
void launch_runs_form(const unsigned *ikeys, const unsigned *ipos, long n, unsigned num_item, const unsigned *prev, int R, unsigned *head, unsigned *head_of,
                      unsigned char *idx, hipStream_t st);
void launch_runs_fill(const unsigned *user, const unsigned *item, const float *label, long n, const unsigned *unit_at, const unsigned *head_of,
                      const unsigned char *idx, long nunit, unsigned *c_item, unsigned *c_user, float *c_label, hipStream_t st);
// user-run units of rank pairs (svdf_k_wave.hip: k_pair_units); out != nullptr: scores only
bool pair_units_applies(const DevParams &P);
void launch_pair_units(const DevParams &P, const PairUnitSchedule &S, long begin, long end, float *out, hipStream_t st);
void launch_runs_iota(unsigned *v, long n, hipStream_t st);
void launch_runs_fill_u32(unsigned *v, long n, unsigned x, hipStream_t st);
bool basicmf_runs_soa_applies(const DevParams &P);
void launch_basicmf_runs_soa(const DevParams &P, const RunSchedule &S, long begin, long end, int R, int G, int block_threads, hipStream_t st);
void device_sort_pairs_u32(unsigned *keys_in, unsigned *keys_out, unsigned *vals_in, unsigned *vals_out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st);
int sqerr_partials_grid(long n);
void launch_sqerr_partials(const float *pred, const float *label, long n, float scale, double *partials, hipStream_t st);
// ---- rank-pair sampling on the device (svdf_k_sample.hip)
struct RankSourceDev {          // a user-group buffer file resident in HBM, rows of shape (0 globals, 1 user entry, 1 item entry)
    long num_block, num_row;
    const long *block_row_ptr;  // [num_block + 1]
    const float *label, *uval, *ival;
    const unsigned *uidx, *iidx;
};
struct SamplerParams { float pos_lowerb, neg_upperb; int sample_num, sample_max; };
struct PairColumns { float *label, *uval, *v0, *v1; unsigned *uidx, *i0, *i1; };   // generated instances, file order
// the general device sampler (any row shape, both sampling methods, pointwise output): svdf_k_gsample.hip
struct RankRowsDev {            // a user-group buffer file's rows resident in HBM, reference CSR layout
    long num_block, num_row;
    const long *block_row_ptr;  // [num_block + 1]
    const float *label;         // [num_row]
    const int *row_ptr;         // [3 * num_row + 1]
    const unsigned *index;
    const float *value;
};
struct GSamplerParams { float pos_lowerb, neg_upperb, gap; int sample_num, sample_max, method, method_raw, pointwise; };
void launch_gsample_counts(const RankRowsDev &S, const GSamplerParams &sp, int *scratch, long *draws, long *pairs, hipStream_t st);
void launch_gsample_pairs(const RankRowsDev &S, const GSamplerParams &sp, const long *draw_off, const long *pair_off, const unsigned *raw, int *pos_list,
                          int *neg_list, int *pair_p, int *pair_n, hipStream_t st);
void launch_gpair_sizes(const RankRowsDev &S, const GSamplerParams &sp, long npairs, const int *pair_p, const int *pair_n, int *lens, hipStream_t st);
void launch_gpair_write(const RankRowsDev &S, const GSamplerParams &sp, long npairs, const int *pair_p, const int *pair_n, const int *optr, float *olabel,
                        unsigned *oi, float *ov, hipStream_t st);
void device_exclusive_scan_i32(const int *in, int *out, long n, void **tmp, size_t *tmp_bytes, hipStream_t st);
void host_sort_by_label(const float *label, long n, int *ids);
// std::sort of candidate ids by descending score as the reference's ranker does it (apex_svd_base.h:767), its partitions on nthreads threads
void host_parallel_sort_scores(const float *score, int *ids, long n, int nthreads);   // the restated std::sort (svdf_stdsort.h) on the host, for tests
void launch_rand_expand(const unsigned *tables, long nchunks, long C, long D, unsigned *raw, hipStream_t st);
// ---- window data sets of plain ratings / rank pairs (kind 5) regrouped on the device (svdf_k_wbuild.hip): the arrays of Engine::window_build.
// All pointers are device pointers; E = n (ratings) or 2 n (pairs) item entries.
struct WBuildIn { long n; int pairs; const unsigned *user, *item, *neg; const float *label; long num_user, num_item; };
struct WBuildBuffers {
    unsigned *k0, *k1, *v0, *v1;   // [E] sort ping-pong
    unsigned *inst;                // [n] instances in (user, file order)
    int *slot_e;                   // [E] slot of every item entry
    int *head, *mark;              // [n]
    int *run_user, *run_start, *run_begin;   // [n]
    void *tmp; size_t tmp_bytes;   // wbuild_tmp_bytes(E)
    unsigned *state;               // [8]
};
struct WBuildOut { WinUser *urec; unsigned *item, *item1; float *label, *v0, *v1; int *slot, *slot1, *iptr; };
size_t wbuild_tmp_bytes(long m);
void device_window_build(const WBuildIn &in, const WBuildBuffers &B, const WBuildOut &out, long *nact, long *item_lo, long *item_hi, hipStream_t st);
// ---- SVDModel::rand_init on the device (svdf_k_init.hip): the j-th matrix element is the j-th accepted attempt of the polar loop
struct InitSeg { long begin, count; long row0; int k; float sigma; int absf; };   // elements [begin, begin + count) -> rows row0.. of W, k per row
struct InitPlan { InitSeg seg[3]; int nseg; long total; int pitch; double margin; };
struct InitReport { long j; unsigned r1, r2; };   // an element whose double lies within `margin` of a float rounding boundary, with its two raw draws
void launch_init_expand(const unsigned *tables, long nchunks, long C, long D, unsigned *raw, hipStream_t st);
size_t init_scan_tmp_bytes(long A);
void launch_init_patch(long n, const long *idx, const float *val, float *W, hipStream_t st);
void launch_init_tile(const unsigned *raw, long A, long base, const InitPlan &plan, float *W, unsigned *flag, unsigned *off, void *tmp, size_t tmp_bytes,
                      unsigned long long *state, InitReport *reports, int report_cap, hipStream_t st);
void launch_sample_counts(const RankSourceDev &S, const SamplerParams &sp, long *draws, long *pairs, hipStream_t st);
void launch_sample_posneg(const RankSourceDev &S, const SamplerParams &sp, const long *draw_off, const long *pair_off, const unsigned *raw,
                          int *pos_list, int *neg_list, const PairColumns &out, hipStream_t st);
// device-resident columns (already in HBM, file order) -> level schedule + level-sorted copies; see Engine::schedule_device_columns
// ---- conflict-free level scheduling on the device (svdf_k_sched.hip).  Unit u touches the parameter rows off[s] + col[s][u]
// for its K slots (SLOT_ABSENT = none; ids >= limit[s] raise limit_msg[s]); no row may repeat inside a unit.  Writes the
// level-sorted unit order (ties: sort_key, then file position -- the order of the host scheduler's stable sorts) to the
// device array order_out[n] and the level boundaries to level_ptr; returns the number of levels.  Throws std::runtime_error.
#define SVDF_SCHED_MAX_SLOTS 8
struct SchedColumns {
    int K;
    long n;
    const unsigned *col[SVDF_SCHED_MAX_SLOTS];     // device pointers, file order
    unsigned off[SVDF_SCHED_MAX_SLOTS], limit[SVDF_SCHED_MAX_SLOTS];
    const char *limit_msg[SVDF_SCHED_MAX_SLOTS];
    unsigned num_res;                               // resource ids are < num_res
    const unsigned *sort_key;                       // device, per unit; nullptr = file order inside a level
    unsigned sort_key_max;
};
long device_schedule(const SchedColumns &in, int *order_out, std::vector<long> &level_ptr, long *max_level_size, hipStream_t st);
// ---- the same for USER UNITS (SVD++ blocks; Engine::schedule_units, svdf_sched.cpp): unit u touches the rows of its instances
// [row_begin, row_end) of the staged CSR (global ids at goff, user ids at user_off, item ids at item_off), its feedback list at fb_off and,
// with UNIT_LOAD / UNIT_SAVE, the state resource.  Also decides the host scan's by-products: which units take the one-wave-per-user kernel
// (simple_out, 0 / 1 per unit), which rows repeat an item inside such a unit (fresh_out, per row), whether every simple unit has unit values.
struct UnitSchedIn {
    long n, nrow;
    const DevUnit *units;          // device, file order (flags without UNIT_SIMPLE)
    const int *row_ptr;            // device, 3 * nrow + 1
    const unsigned *index;
    const float *value;
    const unsigned *fb_index;
    unsigned goff, user_off, item_off, fb_off, num_item, num_fb, state_res;   // resource ids; state_res = the last one
    int simple_ok, fast_ok;
};
struct UnitSchedOut {
    std::vector<long> level_ptr;
    long max_level_size;
    bool any_fresh, unit_values;
};
long device_schedule_units(const UnitSchedIn &in, int *order_out, unsigned char *simple_out, unsigned char *fresh_out, UnitSchedOut &out, hipStream_t st);
void device_gather_u32(const unsigned *src, const int *order, unsigned *dst, long n, hipStream_t st);
void device_gather_f32(const float *src, const int *order, float *dst, long n, hipStream_t st);
void device_scatter_f32(const float *src, const int *order, float *dst, long n, hipStream_t st);   // dst[order[s]] = src[s]
// test probe: out[j] = device expf of in[j], or (in == nullptr) of the float with bit pattern first + j*step
int device_expf(const float *in, unsigned first, unsigned step, float *out, long n);

}  // namespace svdf
#endif
