// svdf_engine.h -- host side of the MI355X apex_svd engine (one instance == one ISVDTrainer).
//
// Mirrors the call protocol of the reference's base solver (solvers/base-solver/apex_svd_base.h)
// behind the C ABI of include/svdfeature_amd.h.  The host does: config parsing, model file I/O,
// rand_init (libc rand(), bit-identical start), staging of borrowed instances, and the
// CONFLICT-FREE BATCH SCHEDULER that turns the reference's strictly sequential SGD into
// dependency-respecting parallel launches.  All arithmetic on parameters
// happens in the HIP kernels (svdf_k_*.hip); there is no CPU compute fallback.
#ifndef SVDF_ENGINE_H_
#define SVDF_ENGINE_H_

#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdio>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <thread>
#include <string>
#include <deque>
#include <functional>
#include <future>
#include <vector>

#include "svdf_types.h"

namespace svdf {

struct Error : std::runtime_error {
    explicit Error(const std::string &m) : std::runtime_error(m) {}
};
[[noreturn]] void fail(const std::string &msg);
// true while the calling thread runs the multi-GPU code of an amd:gpus handle (svdf_engine.cpp)
bool in_multi_scope();
void pin_malloc_threshold();   // svdf_engine.cpp: glibc's dynamic mmap threshold off, once per process, from the ranker only (see there)
struct MultiScope { MultiScope(); ~MultiScope(); MultiScope(const MultiScope &) = delete; MultiScope &operator=(const MultiScope &) = delete; };

// --------------------------------------------------------------------------- device memory
template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t cap = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    ~DevBuf() { release(); }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr;
        cap = 0;
    }
    void reserve(size_t n) {           // grow-only, contents not preserved
        if (n <= cap && p) return;
        release();
        const size_t want = n ? n : 1;
        hipError_t e = hipMalloc((void **)&p, want * sizeof(T));
        if (e != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e) + " at hipMalloc (DevBuf::reserve)");
        cap = want;
    }
    void upload(const T *src, size_t n, hipStream_t st) {
        reserve(n);
        if (!n) return;
        hipError_t e = hipMemcpyAsync(p, src, n * sizeof(T), hipMemcpyHostToDevice, st);
        if (e != hipSuccess) fail(std::string("HIP error: ") + hipGetErrorString(e) + " at hipMemcpyAsync (DevBuf::upload)");
    }
};

// --------------------------------------------------------------------------- scheduler (pure host)
// Assigns every unit (instance or SVD++ user) the earliest batch in which none of its parameter
// rows is touched by an earlier unit of the same or a later batch:
//   level(t) = 1 + max(last[r] for r in resources(t)),  last[r] = level(t)
// Units of one level share no resource, units that share a resource keep their file order, so
// executing levels in increasing order reproduces the sequential result exactly.
struct LevelTracker {
    std::vector<int> last;   // per resource: level of the most recent unit touching it
    int base = 0;            // levels <= base belong to work already enqueued
    void resize(size_t nres) { if (last.size() < nres) last.resize(nres, 0); }
};
struct Schedule {
    std::vector<int> order;        // unit ids sorted by level, stable
    std::vector<long> level_ptr;   // level l occupies order[level_ptr[l] .. level_ptr[l+1])
    std::vector<long> level_mid;   // user-unit schedules only: [level_ptr[l], level_mid[l]) are the fast-path (UNIT_SIMPLE) units
    long max_level_size = 0;
    size_t num_levels() const { return level_ptr.empty() ? 0 : level_ptr.size() - 1; }
};
// levels[t] must already hold each unit's level (> tracker.base); builds order/level_ptr.
void build_schedule(const std::vector<int> &levels, int base, Schedule &out);

struct SideTable {   // SparseFeatureArray<float> (apex-utils/apex_utils.h:140-196)
    std::vector<unsigned> row_ptr{0};
    std::vector<unsigned> index;
    std::vector<float> value;
    unsigned num_row() const { return (unsigned)row_ptr.size() - 1; }
    void load(const char *fname);
};
struct ParamSet {    // ParameterSet (apex_svd_base.h:33-75)
    std::vector<float> wd;
    std::vector<unsigned> bound;
    std::string prefix_a, prefix_b;
    void set_param(const char *name, const char *val);
};

struct HostCSR {     // staged SVDFeatureCSR rows (apex_svd_data.h:109-127)
    std::vector<float> row_label;
    std::vector<int> row_ptr{0};
    std::vector<unsigned> feat_index;
    std::vector<float> feat_value;
    long num_row() const { return (long)row_label.size(); }
    void clear() { row_label.clear(); row_ptr.assign(1, 0); feat_index.clear(); feat_value.clear(); }
};

// level-sorted SoA form of few-row instances (FusedSchedule) on the host and in HBM
struct FusedHost {
    std::vector<float> label, uval[2], ival[2], gval;
    std::vector<unsigned> uidx[2], iidx[2], gidx;
    std::vector<int> gptr;
    std::vector<unsigned> gsi[4];   // inline global slots, filled when every instance has <= 4 distinct global ids
    std::vector<float> gsv[4];
    int max_nu = 1, max_ni = 1;
    bool has_g = false, inline_g = false;
};
struct FusedDev {
    DevBuf<float> label, uval[2], ival[2], gval, gsv[4];
    DevBuf<unsigned> uidx[2], iidx[2], gidx, gsi[4];
    DevBuf<int> gptr;
    int max_nu = 1, max_ni = 1;
    bool has_g = false, inline_g = false;
    bool dense_slots = false;   // every instance has exactly one user id and one item id (k_fewrow_gslots reads slot 0 without an absent test)
    void upload(const FusedHost &h, hipStream_t st);
    FusedSchedule view() const;
};

// user-group (SVD++) stream in HBM: CSR rows + feedback lists + unit records + batch order
struct UnitDev {
    DevBuf<float> label, value, fbval;
    DevBuf<int> ptr, order;
    DevBuf<unsigned> index, fbidx;
    DevBuf<DevUnit> units;
    DevBuf<DevUnitX> xunits;       // the same units in launch order with their first row entry and user id (k_svdpp_wave)
    DevBuf<DevBlk> blks;           // extend_type 2: the blocks the units range over
    DevBuf<unsigned char> fresh;   // DevCSR::row_fresh (allocated only when some row needs it)
    bool has_fresh = false, unit_values = false;
    DevCSR csr() const { return DevCSR{label.p, ptr.p, index.p, value.p, unit_values ? 1 : 0, has_fresh ? fresh.p : nullptr}; }
};

// One instance of a user block as PairwiseRankGenerator sees it (SVDFeatureCSR::Elem, apex_svd_data.h:38-107): the
// global, user and item entries are contiguous in that order.
struct RankRow {
    float label;
    int ng, nu, ni;
    const unsigned *index;
    const float *value;
};

// libc rand() as a random-access stream (svdf_randstream.cpp): the generator's last 31 values, oldest first
struct WUnitHost;   // one window's host-built arrays of the user-unit step (svdf_wunit.cpp)
struct LibcRand { uint32_t x[31]; char *handle = nullptr; };
bool libc_rand_capture(LibcRand &s);
void libc_rand_restore(const LibcRand &s);
void libc_jump_poly(uint64_t J, uint32_t a[31]);
void libc_rand_chunk_states(const LibcRand &s0, long nchunks, long C, std::vector<uint32_t> &states);
void libc_rand_peek(long n, int *out);
void libc_rand_skip(long n);

// PairwiseRankGenerator (apex_svd_data.cpp:812-1025) restated for whole passes: the rows of a user block are replaced by
// rank pairs drawn with libc rand() in the reference's call order, so that a seeded run sees the same pairs.
class PairSampler {
public:
    void set_param(const char *name, const char *val);   // apex_svd_data.cpp:971-981
    void init();                                          // apex_svd_data.cpp:982-988 (once per handle)
    // appends the rows generated from one block; row_ptr holds absolute value offsets and starts as {0}
    void sample_block(const std::vector<RankRow> &rows, std::vector<float> &label, std::vector<int64_t> &row_ptr,
                      std::vector<unsigned> &index, std::vector<float> &value);
    // what the device sampler (svdf_k_sample.hip) needs to know
    int method() const { return method_; }
    int pointwise() const { return pointwise_; }
    int sample_num() const { return sample_num_; }
    int sample_max() const { return sample_max_; }
    float pos_lowerb() const { return pos_lowerb_; }
    float neg_upperb() const { return neg_upperb_; }
    int seed_bytime() const { return seed_bytime_; }
    float gap() const { return gap_; }
private:
    int sample_num_ = -1, sample_max_ = 0x7fffffff, method_ = 0, pointwise_ = 0, seed_bytime_ = 0;
    float gap_ = 0.0001f, pos_lowerb_ = 0.8f, neg_upperb_ = 1e-6f;
    bool init_done_ = false;
    std::vector<RankRow> pos_, neg_;
};

struct RankPrefetch;
struct UserGroupArrays;
struct MultiState;                                   // svdf_multi.cpp: the other ranks of an "amd:gpus = N" handle
struct MultiDeleter { void operator()(MultiState *m) const; };
struct IpcState;                                     // svdf_ipc.cpp: IPC-mapped wire buffers / flag pages of the one-process-per-GPU ranks
struct IpcDeleter { void operator()(IpcState *s) const; };
struct RcclState;                                    // svdf_rccl.cpp: the rank's own RCCL communicator, side stream and hand-over buffers
struct RcclDeleter { void operator()(RcclState *s) const; };
void rccl_unique_id(unsigned char *out128);
// a user-group buffer file kept in HBM for the device sampler (svdf_k_sample.hip); built once per file, reused every pass
struct RankSource {
    std::string path;
    long file_size = -1, file_mtime = -1, file_ino = -1;   // cache key: size, mtime in ns, inode
    bool eligible = false;
    long num_block = 0, num_row = 0;
    DevBuf<long> block_row_ptr, draws, pairs, draw_off, pair_off;
    DevBuf<float> label, uval, ival;
    DevBuf<unsigned> uidx, iidx, raw, tables;
    DevBuf<int> pos_list, neg_list;
    // the general device sampler (svdf_k_gsample.hip): the file's rows in the reference's CSR layout, the host copy of everything a
    // generated pass keeps from the file (tags, feedback lists), and the pass's scratch / output
    bool general = false;
    std::vector<int> h_tag;
    std::vector<int64_t> h_fb_ptr;
    std::vector<unsigned> h_fb_index;
    std::vector<float> h_fb_value;
    DevBuf<int> g_row_ptr, pair_p, pair_n, lens, optr;
    DevBuf<unsigned> g_index, out_index;
    DevBuf<float> g_value, out_label, out_value;
    void *scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    ~RankSource() { if (scan_tmp) (void)hipFree(scan_tmp); }
};
class Engine;
class Ranker;

// HBM-resident scheduled training set
struct Dataset {
    Engine *owner = nullptr;
    long num_row = 0;
    int kind = 0;                 // 0 basicMF fused kernel, 1 general sparse kernel, 2 few-row fused kernel, 3 SVD++ units, 4 multi-level units, 9 ratings with hot rows (svdf_pivot.cpp),
                                  // 5 window-minibatch (user-grouped instances of one exchange window, svdf_k_window.hip)
    // kind 5: user records in launch order, user-grouped columns (item / label / uval / ival above), contribution slots, item segments
    DevBuf<WinUser> win_urec;
    DevBuf<int> win_slot, win_iptr, win_slot1;
    DevBuf<unsigned> win_item1;   // rank pairs: the second (higher-id) item entry, its slot and sign; entry 0 uses item / win_slot / ival
    DevBuf<float> win_ival1;
    long win_item_lo = 0, win_item_hi = -1;   // kind 5: lowest / highest item id with an instance in the window (-1: none)
    DevBuf<long> d_level_ptr;     // kind 2: the level boundaries in HBM, uploaded when a run of narrow levels is first chained (k_fewrow_slots_chain)
    bool d_level_ptr_ok = false;
    // kind 10: runs of an item's consecutive ratings (svdf_runs.cpp): level-sorted columns of the runs; user / item / label above stay in FILE order
    // kind 11: user-run units of rank pairs (svdf_punit.cpp): pu_units in launch order (sched.level_ptr over units), pair columns in file order
    DevBuf<PairUnit> pu_units;
    DevBuf<unsigned> pu_lo, pu_hi;
    DevBuf<float> pu_vlo, pu_one;
    DevBuf<unsigned> rn_item, rn_user[8];
    DevBuf<float> rn_label[8];
    int rn_len = 0;
    // kind 9: hot rows walked as units (svdf_pivot.cpp): sched.level_ptr = the cold ratings' level boundaries (user / item / label above),
    // pv_unit_ptr = the units' (unitdev.xunits in launch order, their rows in unitdev)
    std::vector<long> pv_unit_ptr;
    bool pv_item_pivot = false;
    long pv_cold = 0, pv_hot_rows = 0;
    int64_t chained_levels = 0;   // levels the current launch sequence of this data set walks inside chained launches
    bool win_hot = false;         // kind 5 inside a one-GPU sequence: some item has more than window_hot_sub slots in this window (ordered sub-steps: k_window_apply)
    long win_slots = 0;           // contribution slots of the window = item entries (kind 7: + feedback entries)
    // kind 7: window-minibatch data set of user units (svdf_k_wunit.hip): user-group blocks / rows with global features
    DevBuf<WinUnit> wu_units;
    DevBuf<WinSeg> wu_segs;
    DevBuf<int> wu_rptr, wu_tptr, wu_gptr;
    DevBuf<WinEnt> wu_ent, wu_fbent;
    DevBuf<WinTouched> wu_touched; // one-GPU window sequences: targets with slots (in-place sums)
    long wu_ntouched = -1;         // -1: no list
    DevBuf<WinFbRec> wu_fbrec;    // deferred feedback scatter: slot-ordered (segment, value) records; empty = the walk writes contribution rows
    long wu_nseg = 0;
    bool wu_defer_fb = false;
    long wu_gslots = 0;           // contribution words of the global biases = global entries
    int wu_estride = 0;           // > 0: every row has wu_estride - 1 global entries and one item entry (no row pointer array)
    bool wu_feedback = false;     // the units carry implicit-feedback lists (user-group trainer)
    // kind 8: one GPU, `amd:step = minibatch`: the pass as a sequence of windows (kind 5 or kind 7 children), each trained and applied in place
    std::vector<Dataset *> wchild;
    // kind 6: a data set of an amd:gpus = N handle (svdf_multi.cpp): mchild[rank][window] lives in that rank's HBM
    std::vector<std::vector<Dataset *>> mchild;
    bool m_minibatch = false;     // the children are window-minibatch data sets (kind 5), else level-scheduled ones
    FusedDev fused;               // kind 2
    UnitDev unitdev;              // kind 3: user-group (SVD++) units
    Schedule sched;               // level_ptr always; order on the host only when built there or asked for (host_order)
    DevBuf<int> order_dev;        // device-built schedules: the unit order stays in HBM until a prediction needs it un-permuted
    // kind 0: level-sorted compact records
    DevBuf<unsigned> user, item;
    DevBuf<float> label, uval, ival;
    bool unit_values = true;
    // kind 1: CSR stream + order
    DevBuf<float> row_label, feat_value;
    DevBuf<int> row_ptr, order;
    DevBuf<unsigned> feat_index;
    long algorithmic_bytes = 0;
    long num_units = 0, num_simple_units = 0;
    uint64_t sched_signature = 0;  // Engine::schedule_signature() at build time
    // captured launch sequence of one pass (Engine::train_dataset)
    hipGraphExec_t graph_exec = nullptr;
    uint64_t graph_version = 0;
    hipStream_t graph_stream = nullptr;
    Dataset() = default;
    ~Dataset();
    Dataset(const Dataset &) = delete;
    Dataset &operator=(const Dataset &) = delete;
};

class Engine {
  public:
    Engine(TypeParam mtype, int device);
    ~Engine();

    // ISVDTrainer surface
    void set_param(const char *name, const char *val);
    void init_model();
    void load_model(FILE *fi);
    void save_model(FILE *fo);
    void save_model_begin(FILE *fo);   // the file written beside the next pass (svdf_model.cpp)
    void save_model_end();
    void init_trainer();
    void set_round(int nround);
    void finish_round();
    void update_csr(float label, int ng, int nu, int ni, const unsigned *index, const float *value);
    float predict_csr(float label, int ng, int nu, int ni, const unsigned *index, const float *value);
    void update_csr_batch(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value);
    void predict_csr_batch(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out);
    void update_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                      const int *row_ptr, const unsigned *feat_index, const float *feat_value);
    void predict_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                       const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out);

    // resident datasets
    Dataset *dataset_from_csr(long num_row, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    Dataset *dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    Dataset *dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg);
    Dataset *dataset_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                 const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                 const float *feat_value);
    // the reference's binary buffer files, read natively (svdf_buffer.cpp): CSR buffer or user-group buffer
    Dataset *dataset_from_buffer_file(const char *path, int user_group_format);
    // input_type = 2: a user-group buffer file through the rank-pair sampler, one pass per call
    Dataset *dataset_from_rank_buffer_file(const char *path);
    long rank_sample_buffer_file(const char *in_path, const char *out_path);   // host only; returns the number of rows written
    // draws the NEXT pass on a background thread; the next dataset_from_rank_buffer_file / rank_sample_buffer_file call
    // for the same file takes it.  Nothing else may call rand() until then (the reference's loop does not, svd_feature.cpp:272-283)
    void rank_prefetch(const char *path);
    void rank_prefetch_drop();
    // window-minibatch data set of one exchange window (N > 1 ranks: user side exact, item side one minibatch step per window)
    Dataset *dataset_window_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    Dataset *dataset_window_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg);
    void window_delta_pack(Dataset *ds, void *device_dst, int half, int64_t *count);   // per-item sum of the window's contributions -> wire buffer
    void window_delta_apply(const void *device_src, int half);                           // replicated ranges += wire buffer
    void window_delta_apply_local(Dataset *ds);                 // stratified schedule: the active item block += the window's per-item sums, in place
    // cross-process direct exchange through IPC-mapped buffers (svdf_ipc.cpp)
    void ipc_setup(int rank, int world, int64_t wire_bytes, int64_t block_floats, unsigned char *handles_out);
    void ipc_connect(const unsigned char *all_handles);
    void ipc_window_pack(Dataset *ds, int half);
    void ipc_window_reduce(int half);
    void ipc_window_apply(int half);
    void ipc_block_send(int dst, int slot);
    void ipc_block_recv(int src, int slot, unsigned seq);
    int ipc_status() const;
    void ipc_set_spin_limit(long polls);       // knob "ipc_spin_limit": polls before a flag wait gives up (tests; default ~ several seconds)
    long ipc_spin_limit_ = 0;
    void ipc_fail_if_dead(const char *where);   // raised by svdf_synchronize: a timeout in the last window must not pass silently
    void ipc_close();
    // the same exchanges issued from C++ straight into RCCL (svdf_rccl.cpp)
    void stratum_step(Dataset *const *ds, int n, int block, int nblocks, float *device_out);
    void item_block_set_at(int block, int nblocks, const float *device_src);
    void rccl_init(const unsigned char *id128, int rank, int world);
    void rccl_window_allreduce(Dataset *ds, int half);
    void rccl_block_handoff(int dst, int src, int slot, int in_block, int nblocks);
    void rccl_block_arrive(int slot);
    int64_t rccl_counter(int what) const;
    void rccl_close();
    // the same step for user units (svdf_k_wunit.hip): rows with global features / several item entries, user-group (SVD++) blocks
    Dataset *dataset_window_from_csr(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    Dataset *dataset_window_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                        const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                        const float *feat_value);
    void item_block_copy(float *device_buf, int set, int64_t *count);   // the active partition of the replicated ranges <-> a packed fp32 buffer
    void train_dataset(Dataset *ds);
    void predict_dataset(Dataset *ds, float *out);
    void eval_dataset(Dataset *ds, float scale, double *sum_sq, int64_t *count);

    // multi-GPU item-side delta
    // the per-rank exchange entry points (one process per GPU) make no sense on a handle that exchanges by itself
    void per_rank_api() const {
        if (multi_ && !in_multi_scope())
            fail("svdfeature_amd: svdf_item_delta_* / svdf_window_delta_* / svdf_item_block_* are the per-rank entry points of the one-process-per-GPU "
                 "scheme; an amd:gpus > 1 handle exchanges by itself (train through svdf_update* / svdf_train_dataset)");
    }
    void item_delta_begin();
    void *item_delta_buffer(int64_t *count);
    void item_delta_apply();
    void item_delta_copy(float *device_dst, const float *device_src);
    void item_delta_into(float *device_dst, int64_t *count);
    void item_delta_apply_from(const float *device_src);
    void item_delta_select(int part, int nparts);                              // active exchange partition (item id ranges)
    void item_delta_pack(void *device_dst, int half, int64_t *count);          // one launch, fp32 or fp16 wire format
    void item_delta_unpack(const void *device_src, int half, int refresh_snapshot);
    void set_stream(hipStream_t s);

    // introspection
    int64_t get_view(int which, float *out, int64_t capacity);
    int64_t set_view(int which, const float *in, int64_t count);
    void view_shape(int which, int *rows, int *cols);
    hipStream_t stream() const { return stream_; }
    std::string path_for(const Dataset *ds) const;   // the schedule form + kernel family a resident data set takes (svdf_dataset.cpp)
    void note_dataset(Dataset *ds);                  // once per data set from the C entry points: the default-step guard (counters 26 .. 28)
    void synchronize();
    int64_t counter(int what) const;
    int set_knob(const char *name, long value);
    void flush();

  private:
    // ---- configuration
    TypeParam mtype_;
    ModelParam mp_;
    TrainParam tp_;
    ParamSet u_param_, i_param_, g_param_;
    std::string name_feat_user_ = "NULL", name_feat_item_ = "NULL";
    SideTable feat_user_, feat_item_;
    int round_counter_ = 0;
    bool space_allocated_ = false, trainer_ready_ = false;
    // ---- geometry (SVDModel::alloc_space, apex_svd_model.h:511-556)
    int pitch_ = 0;
    long n_uiset_ = 0;
    unsigned user_off_ = 0, item_off_ = 0, fb_off_ = 0;
    int num_fb_rows() const { return mp_.common_feedback_space == 0 ? mp_.num_ufeedback : mp_.num_user; }
    bool user_group() const { return mtype_.format_type == 1; }
    void compute_geometry();
    // ---- host model (only between init/load and upload, or transiently for save/get_view)
    std::vector<float> hW_, hbias_, hg_;
    bool host_model_valid_ = false;
    void alloc_host_model();
    void rand_init();
    bool rand_init_device();
    void upload_model();
    void download_model();
    void read_model(FILE *fi);
    void read_model_to_device(FILE *fi);
    void file_to_dev(FILE *fi, float *ddst, long rows, long cols, long pitch);
    void write_model(FILE *fo);
    // ---- device
    int device_ = -1;
    bool host_only_ = false;
    hipStream_t stream_ = nullptr;
    bool owns_stream_ = true;
    DevBuf<float> dW_, dbias_, dg_, dstate_;
    bool device_model_ = false;
    DevBuf<unsigned> d_ubound_, d_ibound_, d_gbound_, d_fu_ptr_, d_fu_idx_, d_fi_ptr_, d_fi_idx_;
    DevBuf<float> d_uwd_, d_iwd_, d_gwd_, d_fu_val_, d_fi_val_;
    DevParams dev_params_;
    bool params_dirty_ = true;
    const DevParams &params();
    void need_device(const char *what);
    // ---- staging + scheduling
    HostCSR staged_;
    struct HostUnit { int fb_begin, fb_end, row_begin, row_end, flags; };
    std::vector<HostUnit> staged_units_;
    std::vector<unsigned> staged_fb_index_;
    std::vector<float> staged_fb_value_;
    // extend_type 2 (SVDPPMultiIMFB, solvers/multi-imfb/apex_multi_imfb.h): blocks + units over block ranges, depth = the
    // reference's `top` after everything staged or issued so far
    bool imfb() const { return mtype_.extend_type == 2; }
    std::vector<DevBlk> staged_blks_;
    std::vector<DevUnit> staged_iunits_;
    int imfb_depth_ = 0;
    bool iunit_open_ = false;
    unsigned imfb_disable_ = 0;
    bool imfb_deep_ = false;   // the data nested deeper than IMFB_DEPTH levels at some point: launches use the IMFB_DEPTH_MAX build of k_imfb
    void update_block_imfb(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                           const int *row_ptr, const unsigned *feat_index, const float *feat_value);
    void schedule_iunits(int base, Schedule &sched);
    void upload_iunits(UnitDev &dst, const Schedule &sched);
    void flush_iunits();
    void drop_staged_units();
    // extend_type 15 (SVDBiLinearTrainer, solvers/bilinear/apex_svd_bilinear.h): BParam + W_bi travel with the model file
    bool bilinear() const { return mtype_.extend_type == 15; }
    struct BiParam { int num_bi_feedback, start_ufeedback, reserved[32]; } bi_param_;
    std::vector<float> hbi_;
    bool bi_allocated_ = false;
    int reg_bi_feedback_ = 0;
    bool unit_open_ = false;          // a START block was staged and its END has not arrived
    bool unit_open_on_device_ = false;  // ... and its first part was already flushed (state saved on device)
    long stage_window_ = 1 << 21;   // rows per staged window: 1-4 M measure alike in tools/update_call_cost (125-153 M inst/s), smaller windows end a round sooner
    bool use_graph_ = false;   // measured: no gain, dependent short kernels are bound on the GPU side
    int graph_min_levels_ = 2;
    uint64_t launch_version_ = 1;
    std::vector<unsigned char> staged_fresh_;   // per staged row, see DevCSR::row_fresh
    bool any_fresh_ = false, simple_unit_values_ = false;
    // relaxed mode for shared ids (extension keys "amd:relax_global", "amd:relax_user_from", "amd:relax_item_from")
    bool relax_global_ = false, relax_feedback_ = false;
    int g_stride_ = 1;                  // device layout of g_bias, see DevParams::g_stride
    int wanted_g_stride() const { return (relax_global_ && mp_.num_global <= (1 << 24)) ? 32 : 1; }
    void upload_globals(int stride);
    void download_globals(float *dst);
    unsigned relax_user_from_ = 0xFFFFFFFFu, relax_item_from_ = 0xFFFFFFFFu;
    PairSampler pair_sampler_;
    RankPrefetch *rank_prefetch_ = nullptr;
    std::vector<Dataset *> datasets_;     // live datasets of this trainer (Dataset::owner back-pointers)
    void adopt(Dataset *ds);
    // device-side scheduling of column-shaped data sets (svdf_k_sched.hip): ucols / fcols = (host column in file order, destination
    // buffer in level order); res_col[s] = index into ucols of slot s's id column
    struct UCol { const unsigned *src; DevBuf<unsigned> *dst; };
    struct FCol { const float *src; DevBuf<float> *dst; };
    void schedule_columns_on_device(Dataset *ds, long n, int K, const int *res_col, const unsigned *off, const unsigned *limit,
                                    const char *const *msg, int sort_col, unsigned sort_max, const std::vector<UCol> &ucols, const std::vector<FCol> &fcols);
    const int *host_order(Dataset *ds);
    bool device_sched_ = true;            // knob "device_schedule"
    long device_sched_min_ = 1 << 16;     // staged windows smaller than this stay on the host scheduler (knob "device_schedule_min")
    Dataset w_dataset_;                   // schedule scratch of device-scheduled staging windows (never adopted)
    uint64_t schedule_signature() const;
    void disown(Dataset *ds);
    void rank_pass(const char *path, UserGroupArrays &g);
    Dataset *rank_pass_device(const char *path);   // nullptr when the file or the sampler settings need the host path
    bool rank_source_load(const char *path);       // (re)loads rank_source_ for this file; false when it cannot be opened / stat'ed
    bool rank_pass_device_general(const char *path, UserGroupArrays &g);   // any row shape / method: the pass drawn in HBM, blocks back in g
    std::unique_ptr<RankSource> rank_source_;
    bool device_rank_ = true;                       // knob "device_rank"
    bool device_load_ = true;                       // knob "device_load": load_model streams the matrices file -> pinned chunks -> HBM (0 = through a host copy of the model)
    bool device_window_ = true;                     // knob "device_window": kind-5 window data sets regrouped on the device (svdf_k_wbuild.hip)
    bool device_init_ = true;                       // knob "device_init": rand_init on the device (svdf_k_init.hip)
    int device_init_margin_log2_ = 46;              // knob "device_init_margin_log2": values closer than 2^-this (relative) to a float rounding boundary go to the host libm
    int64_t n_init_reports_ = 0, n_init_draws_ = 0, n_chained_levels_ = 0;
    long chain_width_ = 128;                        // knob "chain_width": levels of at most this many instances are walked in runs inside ONE launch by one workgroup (0 = off)
    bool fewrow_fast_ = true;                       // knob "fewrow_fast": specialised few-row kernel (svdf_k_fewrow.hip)
    // columns that are already in HBM (file order) -> level schedule + level-sorted copies
    Dataset *dataset_fewrow_on_device(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    struct DUCol { const unsigned *src; DevBuf<unsigned> *dst; };
    struct DFCol { const float *src; DevBuf<float> *dst; };
    void schedule_device_columns(Dataset *ds, long n, int K, const unsigned *const *res_col, const unsigned *off, const unsigned *limit,
                                 const char *const *msg, const unsigned *sort_key, unsigned sort_max, const std::vector<DUCol> &ucols,
                                 const std::vector<DFCol> &fcols);
    bool rows_without_feedback_ = true;   // knob: block datasets without any feedback id are scheduled row by row
    bool rows_as_instances_ = false;      // set while such a dataset is being built
    bool relaxed() const { return relax_global_ || relax_feedback_ || relax_user_from_ != 0xFFFFFFFFu || relax_item_from_ != 0xFFFFFFFFu; }
    // lazy decay modes (apex_svd_base.h:95-97,157-170): the reference's sample_counter and per-id ref words
    unsigned sample_counter_ = 0;
    DevBuf<unsigned> d_ref_ui_, d_ref_global_;
    bool lazy_decay() const { return tp_.reg_method >= 4 || tp_.reg_global >= 4; }
    int groups_per_wave_ = 0, block_threads_ = 0, store_mode_ = 0, load_mode_ = 2, basic_i8_ = 1, svdpp_helpers_ = 8, svdpp_xunits_ = 1, small_blocks_ = 1, fewrow_i16_ = 1, sort_batches_ = 1, xcd_remap_ = 1, hot_reduce_ = 1;   // 0 = tuned per factor width
    LevelTracker tracker_;
    void stage_rows(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value);
    void stage_rows_into(HostCSR &dst, int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value);
    void check_row(int ng, int nu, int ni, const unsigned *index);
    bool basic_fast_path_allowed() const;
    bool fused_allowed() const;
    bool fused_allowed_for_rows() const;   // the same for the rows of a feedback-free user-group pass
    template <typename PtrT> bool fused_shape_ok(long n, const PtrT *row_ptr, const unsigned *idx, FusedHost &out);
    template <typename PtrT> void fill_fused(long n, const float *row_label, const PtrT *row_ptr, const unsigned *idx, const float *val,
                                             const int *order, FusedHost &out);
    FusedHost w_fused_host_;
    FusedDev w_fused_;
    bool use_fused_ = true, use_simple_units_ = true;
    size_t num_resources() const { return (size_t)n_uiset_ + (size_t)mp_.num_global; }
    // per-unit level assignment
    int level_of_row(const unsigned *ig, int ng, const unsigned *iu, int nu, const unsigned *ii, int ni, int lvl0);
    void touch_row(const unsigned *ig, int ng, const unsigned *iu, int nu, const unsigned *ii, int ni, int lvl);
    void flush_csr(HostCSR &src);
    void flush_units();
    // background flush of full windows (random-order trainers only)
    void submit_window();
    void wait_worker();
    void worker_main();
    std::thread worker_;
    std::mutex mu_;
    std::condition_variable cv_;
    HostCSR job_;
    bool worker_busy_ = false, worker_stop_ = false, async_flush_ = true;
    std::string worker_error_;
    // levels + DevUnit records for the staged units (marks UNIT_SIMPLE); returns the schedule
    void schedule_units(int base, Schedule &sched, std::vector<DevUnit> &du);
    void upload_unit_arrays(UnitDev &dst);
    void upload_units(UnitDev &dst, const Schedule &sched, const std::vector<DevUnit> &du, bool scheduled_on_device = false);
    bool schedule_units_on_device(UnitDev &dst, Schedule &sched, std::vector<DevUnit> &du);
    int64_t unit_sched_us_ = 0;
    bool unit_sched_on_device_ = false;   // resident user-group data sets (svdf_k_sched.hip)
    std::vector<int64_t> stamp_;   // scratch for per-unit distinctness checks: stamp_epoch_ + unit index of the last toucher
    int64_t stamp_epoch_ = 0;
    // reusable device staging buffers
    DevBuf<float> w_label_, w_value_, w_uval_, w_ival_, w_fbval_, w_out_, w_pred_;
    DevBuf<int> w_ptr_, w_order_;
    DevBuf<unsigned> w_index_, w_user_, w_item_, w_fbidx_;
    DevBuf<DevUnit> w_units_;
    UnitDev w_unitdev_;
    // ---- item delta
    DevBuf<float> d_snap_, d_delta_;
    // scratch of the device window builder (svdf_k_wbuild.hip), grow-only: the window's columns as handed over + sort / scan buffers
    DevBuf<unsigned> wb_user_, wb_item_, wb_neg_, wb_k0_, wb_k1_, wb_v0_, wb_v1_, wb_inst_, wb_state_;
    DevBuf<float> wb_label_;
    DevBuf<int> wb_slot_e_, wb_head_, wb_mark_, wb_run_user_, wb_run_start_, wb_run_begin_;
    DevBuf<char> wb_tmp_;
    bool window_build_device(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg);
    void window_build_resident(Dataset *ds, long n, const unsigned *d_user, const unsigned *d_item, const float *d_label, const unsigned *d_neg);
    void window_build_header(Dataset *ds, long n, bool pairs);
    bool device_window_ready() const { return !host_only_ && device_window_; }
    DevBuf<float> d_dvec_, d_dbias_;      // deferred feedback scatter: one scaled delta row + bias delta per segment of the largest window
    DevBuf<float> d_contrib_, d_cbias_;   // window-minibatch scratch: one contribution row + bias word per instance of the largest window
    std::unique_ptr<IpcState, IpcDeleter> ipc_;
    std::unique_ptr<RcclState, RcclDeleter> rccl_;
    void rccl_check(const char *what);
    void ipc_check(const char *what);
    DevBuf<float> d_gcontrib_;            // ... and one word per global entry (user-unit windows)
    // user-unit windows (svdf_wunit.cpp)
    WUnitSchedule wunit_view(const Dataset *ds) const;
    void wunit_check_config(const char *what) const;
    bool wunit_config_ok() const;       // the same conditions as a predicate (svdf_multi.cpp picks the step per data set)
    void wunit_build(Dataset *ds, const void *segs, size_t nseg, const std::vector<int64_t> &seg_rows, bool by_row_order, long num_src_row,
                     const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value,
                     const unsigned *fb_index, const float *fb_value);
    void wunit_build_host(WUnitHost &H, bool inplace, const void *segs, size_t nseg, const std::vector<int64_t> &seg_rows, bool by_row_order, long num_src_row,
                          const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value,
                          const unsigned *fb_index, const float *fb_value) const;
    void wunit_adopt(Dataset *ds, const WUnitHost &H);
    void wunit_host_from_csr(WUnitHost &H, bool inplace, long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) const;
    void wunit_host_from_blocks(WUnitHost &H, bool inplace, long b0, long b1, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value) const;
    int wseq_build_threads_ = 32;         // knob "wseq_build_threads": host threads building the windows of a one-GPU window sequence (user units)
    void wunit_fill_from_csr(Dataset *ds, long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    void wunit_fill_from_blocks(Dataset *ds, long b0, long b1, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    void wunit_train(Dataset *ds);
    void wunit_sum(Dataset *ds, void *dst, int half);
    // one GPU, `amd:step = minibatch` (opt-in; not the reference's semantics): resident data sets become window sequences (kind 8)
    bool step_minibatch_set_ = false;
    bool contrib_bf16_ = false;           // "amd:contrib = bf16": contribution rows of the window-minibatch step in bfloat16 (opt-in)
    int fewrow_gslots_ = 1;               // knob "fewrow_gslots": 0 = k_fused for few-row data sets with inline global slots (A/B)
    int wunit_defer_fb_ = 1;              // knob "wunit_defer_fb": feedback-row contributions are formed by k_wunit_sum from the segments' deltas (1) or written as rows by the walk (0: A/B; same bits)
    int wunit_inplace_ = 1;               // knob "wunit_inplace": one-GPU window sequences apply a row's only contribution of a window in place (no slot); 0 = every contribution through a slot (A/B)
    bool wunit_inplace_build_ = false;    // set while wseq_from_csr / _from_blocks build their windows
    int wunit_fast_ = 2;                  // knob "wunit_fast": 0 = the general lane-group kernel for every shape, 1 = + the slot kernel, 2 = + one wave per unit (A/B and tests)
    int wseq_per_target_fb_ = 16;         // knob "window_per_target_fb": the same for feedback rows (instance-sized updates pushed by whole blocks)
    int wseq_per_target_max_ = 128;       // knob "window_per_target_max": the MOST updates any shared row may meet per window (binds on skewed data only)
    // ordered sub-steps for hot items of a one-GPU window sequence of plain ratings (svdf_k_window.hip: k_window_apply; round 6)
    int wseq_hot_sub_ = 128;              // knob "window_hot_sub": an item with more slots than this in a window is applied in sub-steps of this many (0 = off: the round-5 rule, no row more than window_per_target_max per window)
    int wseq_hot_max_ = 2048;             // knob "window_hot_max": the most updates a hot row may meet per window (how stale the USERS' view of it gets); 3 seeds of Zipf(0.7) at the configs[1] size: 1 024 max |dRMSE| 4.2e-5 / 66 ms per pass, 2 048 6.6e-5 / 55 ms, 3 072 7.2e-5 / 52 ms (profiles/r06_hot_lane_calibration.txt)
    DevBuf<float> d_clabel_;
    bool wseq_hot_ok() const;             // the configuration has the hot lane (unit ratings, fp32 contribution rows, one GPU)
    long wseq_windows_hot(long n, const std::vector<long> &item_count) const;
    double wseq_max_ratio() const { return (double)wseq_per_target_ / (double)wseq_per_target_max_; }
    int wseq_per_target_ = 24;            // knob "window_per_target": updates a shared row meets per window when amd:window is not given
    bool single_minibatch() const { return step_minibatch_set_ && gpus_ == 1 && !multi_ && !is_peer_; }
    // runs of an item's consecutive ratings as the units of the contract workload's schedule (svdf_runs.cpp / svdf_k_runs.hip)
    int runs_exec_ = 1;                   // knob "runs_exec"
    int runs_len_ = 4;                    // knob "runs_len": ratings per run at most (2 .. 7)
    int runs_sets_ = 1, runs_block_ = 64; // knobs "runs_sets" / "runs_block": row sets per wave, threads per workgroup of k_basicmf_runs_soa
    long runs_min_rows_ = 1 << 20;        // knob "runs_min_rows": smaller data sets keep the plain schedule (levels too narrow for runs to matter)
    int64_t n_runs_passes_ = 0;
    bool runs_config_ok() const;
    Dataset *runs_dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    void runs_train(Dataset *ds);
    // exact passes over data with hot rows: runs of a hot row's ratings as walker units (svdf_pivot.cpp)
    int pivot_exec_ = 1;                  // knob "pivot_exec"
    int pivot_run_ = 256, pivot_run_long_ = 256;   // knobs "pivot_run" / "pivot_run_long": ratings per unit at most, among cold levels / beyond them (a longer tail cap measured slower)
    int pivot_min_ = 2048;                // knob "pivot_min": a row with at least this many ratings in the data set is hot
    int64_t n_pivot_passes_ = 0;
    // user-run units of rank pairs (svdf_punit.cpp; knob "pair_units", default on; "pair_unit_cap": pairs per unit at most)
    int pair_units_ = 1, pair_unit_cap_ = 16;
    int64_t n_punit_passes_ = 0;
    bool punit_config_ok() const;
    Dataset *punit_dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg);
    bool punit_build(long n, const unsigned *user, const unsigned *lo, const unsigned *hi, std::vector<PairUnit> &sorted, std::vector<long> &level_ptr) const;
    bool punit_flush(HostCSR &src);        // a staged window of user-grouped rank pairs (the per-instance route of the reference CLI)
    DevBuf<PairUnit> w_pu_units_;
    PairUnitSchedule punit_view(const Dataset *ds) const;
    void punit_train(Dataset *ds);
    void punit_predict(Dataset *ds, float *d_out);
    bool pivot_config_ok() const;
    Dataset *pivot_dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    void pivot_train(Dataset *ds);
    // one GPU, `amd:step = auto` (opt-in): every resident data set is level-scheduled first (that is cheap on the device); when the
    // dependency depth of exact sequential semantics -- levels x the latency of one unit -- exceeds twice what the pass would take at the
    // streaming rate, the data set is rebuilt as a window sequence (the contract of `amd:step = minibatch`), else the exact levels stay.
    bool step_auto_set_ = false, auto_building_ = false;
    struct AutoDecision { int decided = 0; long levels = 0, windows = 0; double dag_ms = 0.0, stream_ms = 0.0; } auto_last_;
    bool auto_step_active() const { return step_auto_set_ && !auto_building_ && gpus_ == 1 && !multi_ && !is_peer_ && !host_only_; }
    Dataset *auto_step(Dataset *exact, bool window_ok, const std::function<Dataset *()> &build_window);
    static constexpr long AUTO_PROBE_ROWS = 2000000, AUTO_PROBE_MIN = 8000000;
    AutoDecision auto_probe_;
    std::string auto_rank_path_;          // rank-pair input: the candidate file whose passes the decision below was taken for
    int auto_rank_decision_ = 0;
    bool auto_probe_deep(Dataset *probe, long n_full);
    // ONE place that names the path a resident data set takes (kernel family, schedule form) and, in the DEFAULT (exact) step, runs the
    // estimator of `amd:step = auto` on it: when the data's dependency depth is predicted to cost more than 10 x the streaming model, one
    // stderr line says so and names `amd:step = auto` (svdf_dataset.cpp; counters 26 .. 28).  Called once per data set by the C entry points.
    int64_t n_guard_warnings_ = 0;
    AutoDecision guard_last_;
    long wseq_windows(long n, const std::vector<double> &updates_per_target) const;
    Dataset *wseq_from_csr(long n, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    Dataset *wseq_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                              const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    Dataset *wseq_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    Dataset *wseq_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg);
    void wseq_train(Dataset *ds);
    WindowSchedule window_view(const Dataset *ds) const;
    Dataset *window_trained_ = nullptr;   // the window data set whose contributions the scratch holds
    int window_slots_ = 1, window_groups_ = 0;   // knobs "window_slots", "window_groups"
    int delta_part_ = 0, delta_nparts_ = 1;
    DevBuf<double> d_partials_;
    struct Range { float *base; long n; };
    std::vector<Range> shared_ranges();
    // ---- N GPUs behind this handle (svdf_multi.cpp)
    int gpus_ = 1;
    bool is_peer_ = false, delta_half_ = true, window_set_ = false;
    std::vector<std::pair<std::string, std::string>> param_log_;   // every set_param so far, replayed on the other ranks
    std::unique_ptr<MultiState, MultiDeleter> multi_;
    Engine *rank_engine(int d);
    void multi_setup();
    void multi_copy_model_to_peers();
    void multi_flush(HostCSR &src);
    void multi_prepare();
    void multi_window(const std::function<void(int, Engine *)> &train, bool minibatch, Dataset *const *mb);
    bool multi_minibatch_allowed() const;
    Dataset *multi_dataset_from_triples(long n, const unsigned *user, const unsigned *item, const float *label);
    Dataset *multi_dataset_from_pairs(long n, const unsigned *user, const unsigned *pos, const unsigned *neg);
    Dataset *multi_dataset_from_csr(long num_row, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index, const float *feat_value);
    Dataset *multi_dataset_from_blocks(long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index, const float *fb_value,
                                       const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                       const float *feat_value);
    void multi_train_dataset(Dataset *ds);
    long multi_windows_for(long n, const std::vector<long> &item_count, bool minibatch) const;
    void multi_synchronize();
    int multi_predict_rank_ = 0;         // owner of the user-group block scored last (predict_block on an amd:gpus handle)
    int multi_exchange_mode_ = 0;        // "amd:exchange": 0 p2p (peer loads / stores between the ranks of this process), 1 rccl
    bool multi_step_levels_ = false;     // "amd:step = levels": exact conflict-free levels per rank instead of the window-minibatch step
    std::unique_ptr<Dataset> w_window_;  // the staged path's window-minibatch data set of this rank, rebuilt in place every window
    void window_build(Dataset *ds, long n, const unsigned *user, const unsigned *item, const float *label, const unsigned *neg = nullptr);
    void multi_predict(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out);
    void multi_gather_user_rows();
    int64_t multi_counter(int what) const;
    void item_delta_begin_local();
    void predict_csr_batch_local(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out);
    // ---- counters
    int64_t n_instances_ = 0, n_launches_ = 0, n_batches_ = 0, n_flushes_ = 0;
    static constexpr size_t PRED_PIN_WORDS = 1 << 16;   // predict of a few rows: pinned, device-mapped staging (rows in, predictions out)
    unsigned *pred_pin_ = nullptr;
    // save_model: pinned double buffer of the device -> file pipeline, one for the synchronous path and one for the writer thread of the asynchronous one
    static constexpr size_t SAVE_PIN_FLOATS = (size_t)8 << 20;
    struct SavePipe { float *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; hipStream_t st = nullptr; };
    SavePipe save_pipe_;
    struct SaveAsync { DevBuf<float> W, bias; std::vector<float> g; hipEvent_t ready = nullptr; hipStream_t st = nullptr; SavePipe pipe; std::thread th;
                       std::string error; bool active = false; } save_async_;
    void save_pipe_init(SavePipe &sp, hipStream_t st);
    static void save_pipe_free(SavePipe &sp);
    void dev_to_file(SavePipe &sp, FILE *fo, const float *dsrc, long rows, long cols, long pitch);
    void write_model_from_device(SavePipe &sp, FILE *fo, const float *W, const float *bias, const float *g);
    void write_model_from_device(FILE *fo);
    int64_t ns_flush_ = 0, ns_model_ = 0;   // host-side time accounting (SVDF_PROFILE=1 prints it)
    int64_t n_device_rank_passes_ = 0;
    int64_t n_kind_[3] = {0, 0, 0};   // launches of k_basicmf / k_general / k_fused
    DeltaRanges delta_ranges();
    friend struct Dataset;
    friend class Ranker;
};

// ISVDRanker (apex_svd.h:160-197) / SVDFeatureRanker (apex_svd_base.h:597-813) on the device, see svdf_ranker.cpp
class Ranker {
  public:
    Ranker(TypeParam mtype, int device);
    ~Ranker();
    void set_param(const char *name, const char *val);
    void load_model(FILE *fi);
    void init_ranker(int num_item_set);
    // process(vector<int>&, Elem) / process(vector<int>&, SVDPlusBlock): results appended to out (up to cap), count returned
    long process(float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out, long cap);
    // a whole CSR input of the rank task in one call (the loop of svd_feature_infer.cpp:347-375), sections pipelined
    long process_rows(int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, int *out, long cap);
    long process_block(int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label, const int *row_ptr,
                       const unsigned *feat_index, const float *feat_value, int *out, long cap);
    int64_t counter(int what) const {
        return what == 0 ? n_sections_ : (what == 1 ? n_host_sorts_ : (what == 2 ? (int64_t)pos_item_.size() : (what == 3 ? n_tiles_ : (what == 4 ? tie_copy_ns_ : (what == 5 ? tie_wait_ns_ : (what == 6 ? event_wait_ns_ : (what == 7 ? flush_ns_ : -1)))))));
    }
  private:
    std::unique_ptr<Engine> eng_;   // owns the model in HBM, the side tables and the kernel parameter block
    int top_k_ = 0, num_item_set_ = 0, num_item_processed_ = 0;
    long items_on_device_ = 0;
    bool init_end_ = false, user_open_ = false;
    HostCSR items_, spec_;
    std::vector<int> spec_idx_, pos_item_;
    std::vector<int> tagged_, dev_tagged_;   // candidates tagged in the open section / in the section the device's tag array reflects
    long n_banned_ = 0;
    // sections in flight (svdf_ranker.cpp): per slot the pinned staging (one upload per section), its device copy, the
    // section's counters / positives' scores / NaN flag / item_score, the pinned readback and the event behind it
    static constexpr int RANK_SLOTS = 8;
    struct RankSlot {
        unsigned *pin = nullptr, *back = nullptr;
        size_t pin_words = 0, back_words = 0;
        DevBuf<unsigned> d_stage, d_flag;
        DevBuf<int> d_cnt;
        DevBuf<float> d_ps, d_score;
        hipEvent_t ev = nullptr;
    };
    struct RankPending { int slot = 0, npos = 0, take = 0; long n = 0; std::vector<int> pos_item, banned;
                         // a tile of sections sharing one scoring pass (svdf_k_rank.hip: k_rank_score_tile): per section its positives / bans
                         int nsec = 0; std::vector<std::vector<int>> tile_pos, tile_ban; std::vector<int> tile_take; long out_stride = 0; };
    struct TileSec { std::vector<unsigned> user_idx; std::vector<float> user_val; std::vector<int> pos_item, banned; int take = 0; };
    std::vector<TileSec> tile_;      // sections staged for the next tile (process_rows, positions mode, no special samples)
    long tile_n_ = 0;                // the candidate count they were staged against
    std::vector<int> tile_prev_ban_; // candidates whose ban bits the previous tile set (cleared by the next tile's opening kernel)
    DevBuf<unsigned> d_banmask_;
    DevBuf<float> d_tu_tile_;
    void flush_tile();
    bool tile_enabled_ = true;
    int64_t n_tiles_ = 0;
    RankSlot slots_[RANK_SLOTS];
    struct RankChunk { std::vector<int> vals; std::future<std::vector<int>> fut; bool pending = false; };   // one section's results
    std::deque<RankPending> pending_;
    std::deque<RankChunk> chunks_;
    void flush_chunks();
    int next_slot_ = 0;
    bool deferred_ = false;
    int *out_ptr_ = nullptr;
    long out_cap_ = 0, out_n_ = 0;
    void slot_reserve_pin(RankSlot &S, size_t words);
    void slot_reserve_back(RankSlot &S, size_t words);
    void enqueue();
    void resolve();
    void drain_quietly();
    std::vector<signed char> tag_;
    std::vector<unsigned> user_idx_;
    std::vector<float> user_val_;
    DevBuf<float> d_ifactors_, d_ift_, d_ibias_, d_tu_, d_fb_, w_label_, w_value_, w_uval_, w_fbval_, s_label_, s_value_;
    DevBuf<int> w_ptr_, s_ptr_, s_idx_;
    DevBuf<unsigned> w_index_, w_fbidx_, s_index_, d_keys_, d_vals_, d_sel_;
    void *sort_tmp_ = nullptr;
    size_t sort_tmp_bytes_ = 0;
    DevBuf<signed char> d_tag_;
    int64_t n_sections_ = 0, n_host_sorts_ = 0;
    std::unique_ptr<struct TieWork> tie_scores(const float *d_src, long n);
    hipStream_t tie_stream_ = nullptr;
    float *tie_pin_ = nullptr;
    size_t tie_pin_floats_ = 0;
    int64_t event_wait_ns_ = 0, flush_ns_ = 0;   // waiting for a slot's event; staging + enqueueing tiles (includes resolving the oldest slot)
    int64_t tie_copy_ns_ = 0, tie_wait_ns_ = 0;   // tied sections: copying their scores out (a stream sync each); waiting for their sorts at the end
    void stage(HostCSR &dst, int ng, int nu, int ni, const unsigned *index, const float *value);
    void check_item_side(int ng, int nu, int ni, const unsigned *index);
    long rank(int *out, long cap);
};

}  // namespace svdf
#endif
