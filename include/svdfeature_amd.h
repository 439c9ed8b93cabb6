/*
 * svdfeature_amd.h -- C ABI of the MI355X-native apex_svd SGD engine (libsvdfeature_amd.so).
 *
 * This is the drop-in boundary for ONE path of Gnnng/SVDFeature: the SGD training/prediction step
 * behind `class apex_svd::ISVDTrainer` (reference apex_svd.h:33-107), as implemented by the
 * reference's base solver (solvers/base-solver/apex_svd_base.h: SVDFeature :79-479, SVDPPFeature
 * :484-592).  The reference has no FFI layer of its own -- solvers are swapped at link time by
 * providing `apex_svd::create_svd_trainer` (apex_svd.h:212, solvers/base-solver/Makefile:17-23) --
 * so every entry point below is one ISVDTrainer virtual flattened to C: plain pointers and
 * sizes, no C++ or torch types.  `integration/apex_svd_amd.cpp` is the C++ class deriving
 * from the reference's own ISVDTrainer that forwards to these functions (INTEGRATION.md shows the
 * reference-side link line).
 *
 * Error behaviour follows the reference (apex-utils/apex_utils.h:47-58: message on stderr, then
 * exit(-1)) unless svdf_set_error_mode(1) was called, in which case a failing call returns a
 * negative status / NULL and svdf_last_error() holds the message.  Index-bound violations
 * ("user feature index exceed bound", apex_svd_base.h:320,327,343,360,530) are detected when
 * instances are staged, before anything reaches the GPU.
 *
 * Threading: like the reference a handle is driven from one thread at a time.
 * Borrowed pointers (instances, blocks) are copied before the call returns.
 */
#ifndef SVDFEATURE_AMD_H_
#define SVDFEATURE_AMD_H_

#include <stddef.h>
#include <stdint.h>
#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svdf_trainer svdf_trainer; /* opaque: one ISVDTrainer instance */
typedef struct svdf_dataset svdf_dataset; /* opaque: a scheduled, HBM-resident training set */
typedef struct svdf_ranker svdf_ranker;   /* opaque: one ISVDRanker instance */

/* ---- library ---- */
const char *svdf_version(void);
/* 0 (default): reference behaviour, errors print and exit(-1).  1: errors return status codes. */
void svdf_set_error_mode(int mode);
const char *svdf_last_error(void);
/* number of visible HIP devices (0 when there is no GPU; the library still loads). */
int svdf_device_count(void);

/* ---- lifecycle: apex_svd::create_svd_trainer(SVDTypeParam) (apex_svd.h:212; the four bytes are
 * SVDTypeParam, apex_svd_model.h:242-261) and the virtual destructor (apex_svd.h:106).
 * device < 0 selects the current HIP device. */
svdf_trainer *svdf_create(uint8_t format_type, uint8_t active_type, uint8_t extend_type, uint8_t variant_type,
                          int device);
void svdf_destroy(svdf_trainer *t);

/* ISVDTrainer::set_param (apex_svd.h:42; keys: apex_svd_base.h:126-136, apex_svd_model.h:350-368
 * and :456-476, ParameterSet prefixes up:/ip:/uip:/gp: apex_svd_base.h:48-67).  Unknown keys are
 * ignored, model-shape keys are ignored once the model is allocated. */
int svdf_set_param(svdf_trainer *t, const char *name, const char *val);
/* apex_random::seed (apex-tensor/apex_random.h:42-44) -> srand; process-global like the reference. */
void svdf_seed(unsigned seed);
/* ISVDTrainer::init_model (apex_svd.h:58; apex_svd_base.h:146-149): alloc + SVDModel::rand_init (apex_svd_model.h:665-705) over the
 * libc rand() stream -- the reference's draws and values bit for bit, libc's generator left where the reference's calls would have left
 * it.  On a device handle the matrices are sampled IN HBM (svdf_k_init.hip): 70 M normals in 4.5 ms instead of 1.9 s. */
int svdf_init_model(svdf_trainer *t);
/* ISVDTrainer::load_model / save_model (apex_svd.h:47,52): byte-compatible with
 * SVDModel::load_from_file / save_to_file (apex_svd_model.h:570-660).  The caller owns the FILE*
 * and reads/writes the leading 4-byte SVDTypeParam itself (svd_feature.cpp:165-190). */
int svdf_load_model(svdf_trainer *t, FILE *fi);
int svdf_save_model(svdf_trainer *t, FILE *fo);
/* EXTENSION for loops that own their files (integration/svdf_train_bulk.c): the model file written BESIDE the next pass.  _begin snapshots the model
 * in HBM (ordered behind everything enqueued so far) and returns; a writer thread streams the snapshot into `fo` (the same bytes svdf_save_model
 * would have written at that moment); training may go on.  _end joins the writer (the caller then closes the file).  One save in flight per
 * handle; amd:gpus handles, host-only handles and the bilinear solver write synchronously inside _begin.  The reference's protocol
 * (ISVDTrainer::save_model returns with the file complete, svd_feature.cpp:184-191) is svdf_save_model above. */
int svdf_save_model_begin(svdf_trainer *t, FILE *fo);
int svdf_save_model_end(svdf_trainer *t);
/* ISVDTrainer::init_trainer (apex_svd.h:64; apex_svd_base.h:151-173, 499-503) */
int svdf_init_trainer(svdf_trainer *t);
/* ISVDTrainer::set_round / finish_round (apex_svd.h:72,77).  finish_round flushes staged work. */
int svdf_set_round(svdf_trainer *t, int nround);
int svdf_finish_round(svdf_trainer *t);

/* ---- random-order input: ISVDTrainer::update(const SVDFeatureCSR::Elem&) / predict(const Elem&)
 * (apex_svd.h:83,89).  index/value hold the global, user and item sections back to back exactly
 * like Elem::set_space (apex_svd_data.h:71-78).  update stages the instance; the sequential
 * result is materialised at the next flush point (finish_round, predict, save_model, set_round,
 * destroy, or when the staging window fills). */
int svdf_update_csr(svdf_trainer *t, float label, int num_global, int num_ufactor, int num_ifactor,
                    const unsigned *index, const float *value);
float svdf_predict_csr(svdf_trainer *t, float label, int num_global, int num_ufactor, int num_ifactor,
                       const unsigned *index, const float *value);
/* bulk forms over one SVDFeatureCSR block (apex_svd_data.h:109-127 member layout): equivalent to
 * calling update/predict on rows 0..num_row-1 in order. */
int svdf_update_csr_batch(svdf_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                          const unsigned *feat_index, const float *feat_value);
int svdf_predict_csr_batch(svdf_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value, float *out);

/* ---- user-grouped input: ISVDTrainer::update(const SVDPlusBlock&) /
 * predict(std::vector<float>&, const SVDPlusBlock&) (apex_svd.h:97,104); fields of SVDPlusBlock
 * (apex_svd_data.h:376-393) passed flat.  out must hold num_row floats. */
int svdf_update_block(svdf_trainer *t, int num_ufeedback, int extend_tag,
                      const unsigned *index_ufeedback, const float *value_ufeedback,
                      int num_row, const float *row_label, const int *row_ptr,
                      const unsigned *feat_index, const float *feat_value);
int svdf_predict_block(svdf_trainer *t, int num_ufeedback, int extend_tag,
                       const unsigned *index_ufeedback, const float *value_ufeedback,
                       int num_row, const float *row_label, const int *row_ptr,
                       const unsigned *feat_index, const float *feat_value, float *out);

/* ---- HBM-resident training sets (removes the per-instance virtual call).
 * A dataset is the instance stream of one pass of svd_feature.cpp:220-248's inner loop, scheduled
 * once and kept in HBM; svdf_train_dataset(t, ds) == update(x) for every x in file order.
 * svdf_dataset_from_triples is the three-column form of SVDBasicLoader (apex_svd_data.cpp:32-67):
 * no global feature, one user id and one item id with value 1. */
svdf_dataset *svdf_dataset_from_csr(svdf_trainer *t, long num_row, const float *row_label, const int64_t *row_ptr,
                                    const unsigned *feat_index, const float *feat_value);
svdf_dataset *svdf_dataset_from_triples(svdf_trainer *t, long n, const unsigned *user, const unsigned *item,
                                        const float *label);
/* rank pairs (user, positive item, negative item) in three columns: the instance PairwiseRankGenerator emits for two rows
 * with one unit item entry each (apex_svd_data.cpp:828-860 merge by index with the negative's sign flipped, :905-911
 * label 1): no global entry, user:1, {min(pos,neg): +-1, max(pos,neg): -+1}.  svdf_train_dataset == update(x) for
 * every such instance in order.  pos_item[r] != neg_item[r]. */
svdf_dataset *svdf_dataset_from_pairs(svdf_trainer *t, long n, const unsigned *user, const unsigned *pos_item,
                                      const unsigned *neg_item);
/* user-group form: equivalent to update(SVDPlusBlock b) for b = 0..num_block-1 in order (the content of a
 * user-group buffer file, apex_svd_data.cpp:558-595).  Block b has extend_tag[b], feedback entries
 * fb_index/fb_value[fb_ptr[b] .. fb_ptr[b+1]) and rows block_row_ptr[b] .. block_row_ptr[b+1] of the CSR
 * arrays (row_ptr has 3*num_row+1 entries over all rows).  Every START must be closed by its END. */
svdf_dataset *svdf_dataset_from_blocks(svdf_trainer *t, long num_block, const int *extend_tag, const int64_t *fb_ptr,
                                       const unsigned *fb_index, const float *fb_value, const int64_t *block_row_ptr,
                                       const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                       const float *feat_value);
/* The same, straight from the reference's binary buffer files: user_group_format 0 = the CSR buffer written by
 * tools/make_feature_buffer (SVDFeatureCSRFactory::create_buffer, apex_svd_data.cpp:118-195; block layout
 * apex_svd_data.h:200-230), 1 = the user-group buffer written by tools/make_ugroup_buffer (SVDPlusBlock
 * save/load, apex_svd_data.h:419-450; apex_svd_data.cpp:558-595).  Replaces the loader thread + one virtual
 * update() per instance of svd_feature.cpp:220-248 for callers that train whole passes. */
svdf_dataset *svdf_dataset_from_buffer_file(svdf_trainer *t, const char *path, int user_group_format);
/* input_type = 2 of the reference (input_type::BINARY_BUFFER_RANK, apex_svd_data.h:516, apex_svd_data.cpp:1330-1332): the
 * user-group buffer file seen through PairwiseRankGenerator (apex_svd_data.cpp:812-1025).  Every block's rows are
 * replaced by rank pairs (positive entries merged with the sign-flipped negative's, label 1), drawn with libc rand()
 * in the generator's call order, so a process seeded like the reference's (srand(seed), then svdf_init_model) trains
 * on the same pairs.  One call = one pass of the iterator: the reference re-draws the pairs every round, so build a
 * new dataset per round.  Sampler keys are taken from svdf_set_param like the reference's iterator takes them from the
 * config: pos_sample_lowerb, neg_sample_upperb, rank_sample_num, rank_sample_max, rank_sample_method (0, 1),
 * rank_sample_gap, rank_sample_pointwise, seed_sampler_bytime.  Needs format_type = 1. */
svdf_dataset *svdf_dataset_from_rank_buffer_file(svdf_trainer *t, const char *path);
/* Draws the NEXT pass on a background host thread (the sampler is host work on the mapped file and libc rand(); the
 * device can train the current pass meanwhile).  The next svdf_dataset_from_rank_buffer_file / svdf_rank_sample_buffer_file
 * call for the same path takes the prefetched pass; the pairs are the ones that call would have drawn itself as long as
 * nothing else calls rand() in between (the reference's round loop does not, svd_feature.cpp:272-283).  0 on success. */
int svdf_rank_prefetch_buffer_file(svdf_trainer *t, const char *path);
/* The same pass written to out_path as a user-group buffer file instead of being uploaded (host only, works on a
 * handle created with device = -2).  Returns the number of generated rows, -1 on error. */
int64_t svdf_rank_sample_buffer_file(svdf_trainer *t, const char *in_path, const char *out_path);
void svdf_dataset_destroy(svdf_dataset *ds);                     /* also valid after svdf_destroy of its trainer */
int svdf_train_dataset(svdf_trainer *t, svdf_dataset *ds);       /* one pass, asynchronous on the trainer's stream */
int svdf_predict_dataset(svdf_trainer *t, svdf_dataset *ds, float *out); /* out[num_row], file order */
/* dataset facts: 0 num_row, 1 number of conflict-free batches, 2 largest batch, 3 kernel kind
 * (0 = basicMF fused kernel, 1 = general sparse kernel, 2 = few-row fused kernel, 3 = SVD++ user units),
 * 4 algorithmic bytes per pass, 5 number of user units, 6 units on the register-resident
 * fast path, 7 a digest of the host-resident schedule (level boundaries, fast-path split, unit order) */
int64_t svdf_dataset_info(const svdf_dataset *ds, int what);

/* ---- multi-GPU support: item-side parameters are replicated, each rank trains its
 * user shard, and the item-side deltas of a window are summed across ranks by the caller's
 * collective (RCCL all-reduce on the returned device buffer).
 *   svdf_item_delta_begin   snapshot item-side parameters (W_item, i_bias [, g_bias])
 *   svdf_item_delta_buffer  delta = current - snapshot, returned as ONE contiguous fp32 device
 *                           buffer of *count floats (caller all-reduces it in place)
 *   svdf_item_delta_apply   current = snapshot + (all-reduced) delta */
int svdf_item_delta_begin(svdf_trainer *t);
void *svdf_item_delta_buffer(svdf_trainer *t, int64_t *count);
int svdf_item_delta_apply(svdf_trainer *t);
/* copy the packed delta to / from a caller-owned DEVICE buffer (e.g. a torch tensor the collective
 * runs on); both enqueue on the trainer's stream and return after it has drained. */
int svdf_item_delta_export(svdf_trainer *t, float *device_dst);
int svdf_item_delta_import(svdf_trainer *t, const float *device_src);
/* stream-ordered forms without host synchronisation: write (current - snapshot) straight into a caller-owned
 * device buffer / set current = snapshot + caller's buffer.  Both only enqueue on the trainer's stream; use
 * svdf_set_stream so the collective and these kernels share one stream order. */
int svdf_item_delta_into(svdf_trainer *t, float *device_dst, int64_t *count);
int svdf_item_delta_apply_from(svdf_trainer *t, const float *device_src);
/* the same in ONE launch over all replicated ranges and in the wire format of the collective: half = 0 packs fp32,
 * half = 1 packs IEEE fp16 (round to nearest even; parameters and arithmetic stay fp32).  device_dst = NULL only
 * returns *count (elements).  unpack sets current = snapshot + delta and, with refresh_snapshot != 0, also
 * snapshot = current, so the next window can pack again without svdf_item_delta_begin's copy. */
int svdf_item_delta_pack(svdf_trainer *t, void *device_dst, int half, int64_t *count);
/* The exchange of a window can be cut into nparts pieces by ITEM ID RANGE (part p = items [num_item*p/nparts, num_item*(p+1)/nparts);
 * ranges that are not indexed by item id travel with part 0): pack / unpack then cover the selected piece only, so the
 * all-reduce of one piece can run while the rank trains the instances whose items lie in another piece (no added staleness:
 * a piece's rows are not touched between its pack and its unpack).  svdf_item_delta_begin always snapshots everything. */
int svdf_item_delta_select(svdf_trainer *t, int part, int nparts);
int svdf_item_delta_unpack(svdf_trainer *t, const void *device_src, int half, int refresh_snapshot);

/* ---- window-minibatch step of the N-rank path (svdf_k_window.hip).  The reference has no counterpart: it
 * is single-process.  What is re-arranged is what ONE instance contributes in SVDFeature::update_no_decay +
 * regularize (solvers/base-solver/apex_svd_base.h:383-427, :286-311): the user-side change is applied at once (users are
 * private to a rank: exact sequential SGD), the item-side change is collected per window and applied after the all-reduce.
 *   ds = svdf_dataset_window_from_triples(t, ...)   one exchange window of this rank's shard (user % N == rank), grouped by user
 *   svdf_train_dataset(t, ds)                       every instance = the reference's update_inner on (current user side,
 *                                                   window-start item side); W_item / i_bias are NOT written
 *   svdf_window_delta_pack(t, ds, dst, half, &n)    dst = per item, the sum in file order of what its instances would have changed
 *                                                   (same packed layout / item-range partition as svdf_item_delta_pack; NULL dst = size)
 *   all-reduce(dst) over the ranks
 *   svdf_window_delta_apply(t, dst, half)           replicated ranges += dst
 * Deterministic (no float atomics); equals oracle/svdf_oracle.c: svdo_update_csr_batch_stale bit for bit with fp32 deltas. */
svdf_dataset *svdf_dataset_window_from_triples(svdf_trainer *t, long n, const unsigned *user, const unsigned *item, const float *label);
/* the same for rank pairs (user, positive item, negative item), the instances of svdf_dataset_from_pairs (BASELINE configs[4]) */
svdf_dataset *svdf_dataset_window_from_pairs(svdf_trainer *t, long n, const unsigned *user, const unsigned *pos_item, const unsigned *neg_item);
/* The same step for USER UNITS (svdf_k_wunit.hip): rows with global features and / or several item entries
 * (random-order trainers: _from_csr, the arrays of svdf_dataset_from_csr) and user-group (SVD++) blocks (_from_blocks, the arrays of
 * svdf_dataset_from_blocks; every START closed by its END inside the window).  A user's unit is exact on its private state -- its W_user
 * row and bias and SVDPPFeature's tmp_ufeedback / old_ufeedback (solvers/base-solver/apex_svd_base.h:486-488, 506-520) --; the shared rows
 * -- W_item / i_bias, W_ufeedback / ufeedback_bias (prepare_ufeedback / update_ufeedback :523-554), g_bias (:188-210, :313-353) -- are
 * read as of the window start and their change is summed per row in file order: svdf_window_delta_pack writes the sums in the packed
 * layout [W_ufeedback | W_item | ufeedback_bias | i_bias | g_bias] (one piece: svdf_item_delta_select(t, 0, 1)),
 * svdf_window_delta_apply_local adds them to the model in place.  Every row needs exactly one user entry; ids are distinct inside a
 * row / a feedback list.  Equals oracle/svdf_oracle.c: svdo_update_csr_batch_stale / svdo_update_block_stale bit for bit. */
svdf_dataset *svdf_dataset_window_from_csr(svdf_trainer *t, long num_row, const float *row_label, const int64_t *row_ptr,
                                           const unsigned *feat_index, const float *feat_value);
svdf_dataset *svdf_dataset_window_from_blocks(svdf_trainer *t, long num_block, const int *extend_tag, const int64_t *fb_ptr,
                                              const unsigned *fb_index, const float *fb_value, const int64_t *block_row_ptr,
                                              const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                              const float *feat_value);
int svdf_window_delta_pack(svdf_trainer *t, svdf_dataset *ds, void *device_dst, int half, int64_t *count);
int svdf_window_delta_apply(svdf_trainer *t, const void *device_src, int half);
/* STRATIFIED schedule: item block b = the partition chosen with svdf_item_delta_select(t, b, N).  While a rank
 * trains a stratum (its users x one item block) it owns that block exclusively, so the window's per-item sums are added to the model in
 * place (no wire buffer, no sum over ranks); afterwards the block -- its W_item rows and i_bias words (global biases travel with block 0),
 * fp32, the packed layout of svdf_item_delta_pack -- is handed to the next rank: _get copies it out, _set copies a received block in.
 * device_dst = NULL only returns *count (floats). */
int svdf_window_delta_apply_local(svdf_trainer *t, svdf_dataset *ds);
int svdf_item_block_get(svdf_trainer *t, float *device_dst, int64_t *count);
int svdf_item_block_set(svdf_trainer *t, const float *device_src);
/* one stratum step in one call: every window data set trained (svdf_train_dataset) and summed in place into item block `block` of `nblocks`
 * (svdf_window_delta_apply_local), then -- device_out != NULL -- the block copied out for its hand-over; svdf_item_block_set_at puts an arrived
 * block in place.  The same launches as the calls they fuse (the host thread of a rank has ~40 us per step at 8 ranks). */
int svdf_stratum_step(svdf_trainer *t, svdf_dataset *const *windows, int num_windows, int block, int nblocks, float *device_out);
int svdf_item_block_set_at(svdf_trainer *t, int block, int nblocks, const float *device_src);
/* ---- cross-PROCESS direct exchange (svdf_ipc.cpp): one process per GPU, but the exchange of a window runs through
 * IPC-mapped device buffers -- every rank's wire buffer and flag page mapped into every process (over xGMI on distinct devices) -- with the
 * peer-pointer reduce-scatter + all-gather kernel of the amd:gpus handle, ordered across processes by sequence flags in device memory
 * (no collective library, no host rendezvous inside a pass).  The reference has no counterpart (single process).
 *   svdf_ipc_setup(t, rank, world, wire_bytes, block_floats, handles)   allocate + export; handles = 128 bytes to all-gather
 *   svdf_ipc_connect(t, all_handles)                                    world x 128 bytes in rank order
 *   per window: svdf_train_dataset(ds); svdf_ipc_window_pack(t, ds, half); svdf_ipc_window_reduce(t, half); svdf_ipc_window_apply(t, half)
 *   stratified hand-over: svdf_ipc_block_send(t, dst, slot) stores the active item block (svdf_item_delta_select) into rank dst's inbox;
 *   svdf_ipc_block_recv(t, src, slot, seq) waits for the seq-th block of rank src, puts it in place and acknowledges the slot.
 * A wait that hits its spin limit raises an error word instead of hanging the queue: svdf_ipc_status(t) != 0, later calls fail. */
int svdf_ipc_setup(svdf_trainer *t, int rank, int world, int64_t wire_bytes, int64_t block_floats, unsigned char *handles_out);
int svdf_ipc_connect(svdf_trainer *t, const unsigned char *all_handles);
int svdf_ipc_window_pack(svdf_trainer *t, svdf_dataset *ds, int half);
int svdf_ipc_window_reduce(svdf_trainer *t, int half);
int svdf_ipc_window_apply(svdf_trainer *t, int half);
int svdf_ipc_block_send(svdf_trainer *t, int dst_rank, int slot);
int svdf_ipc_block_recv(svdf_trainer *t, int src_rank, int slot, unsigned seq);
int svdf_ipc_status(svdf_trainer *t);
int svdf_ipc_close(svdf_trainer *t);

/* ---- the same exchanges issued from C++ straight into RCCL (svdf_rccl.cpp): one process per GPU, the rank's own communicator
 * (ncclCommInitRank on the trainer's device; librccl.so resolved at run time -- inside a torch process the library torch loaded), so that a
 * pass needs no Python call and no torch.distributed work object per collective.  Replaces, like the rest of section 6, the round loop of ONE
 * process (svd_feature.cpp:220-248) on N ranks.
 *   svdf_rccl_unique_id(out[128])                      rank 0; the bytes reach the other ranks through the caller's process group / store
 *   svdf_rccl_init(t, id, rank, world)
 *   all-reduce step, per window: svdf_train_dataset(ds); svdf_rccl_window_allreduce(t, ds, half)   (per-item sums -> ncclAllReduce in place -> add)
 *   stratified hand-over: svdf_rccl_block_handoff(t, dst, src, slot, in_block, nblocks) sends the ACTIVE item block (svdf_item_delta_select) to rank
 *   dst while block in_block of nblocks arrives from rank src into inbox slot `slot` (0 / 1), on a side stream ordered by events;
 *   svdf_rccl_block_arrive(t, slot) makes the trainer's stream wait for that transfer and puts the block in place (active partition = its block).
 *   svdf_rccl_counter: 0 hand-overs issued, 1 all-reduces issued. */
int svdf_rccl_unique_id(unsigned char *out128);
int svdf_rccl_init(svdf_trainer *t, const unsigned char *id128, int rank, int world);
int svdf_rccl_window_allreduce(svdf_trainer *t, svdf_dataset *ds, int half);
int svdf_rccl_block_handoff(svdf_trainer *t, int dst_rank, int src_rank, int slot, int in_block, int nblocks);
int svdf_rccl_block_arrive(svdf_trainer *t, int slot);
int64_t svdf_rccl_counter(svdf_trainer *t, int what);
int svdf_rccl_close(svdf_trainer *t);

/* test probe of the device rank sampler's sort (svdf_stdsort.h: libstdc++'s std::sort restated for host and device, because
 * PairwiseRankGenerator::sample_cmp, apex_svd_data.cpp:920-944, picks rows by POSITION after an unstable std::sort): ids 0..n-1
 * sorted by label with the restated code (restated[]) and with the C++ library's own std::sort (library[]). */
int svdf_debug_sort_labels(long n, const float *label, int *restated, int *library);
/* test probe of the ranker's tie-breaking sort (SVDFeatureRanker sorts its entry vector with std::sort, apex_svd_base.h:767; sections whose
 * scores tie are finished by exactly that sort, its partitions spread over `threads` host threads): ids 0..n-1 by descending score with
 * the threaded restatement (parallel[]) and with the library's std::sort over the reference's Entry struct (library[]). */
int svdf_debug_sort_scores(long n, const float *score, int threads, int *parallel, int *library);

/* ---- introspection used by tests, bench.py and the harness ---- */
/* raw copies of parameter views: 0 u_bias 1 W_user 2 i_bias 3 W_item 4 g_bias 5 ufeedback_bias
 * 6 W_ufeedback; rows are returned unpadded.  Returns number of floats or -1. */
int64_t svdf_get_view(svdf_trainer *t, int which, float *out, int64_t capacity);
int svdf_view_shape(svdf_trainer *t, int which, int *rows, int *cols);
/* overwrite a view from rows*cols unpadded floats (multi-GPU: every rank owns a slice of the user rows, the slices are
 * gathered before a model is saved).  Returns the number of floats or -1. */
int64_t svdf_set_view(svdf_trainer *t, int which, const float *in, int64_t count);
/* the HIP stream (hipStream_t) all of this trainer's work is enqueued on; timing code records
 * HIP events on it. */
void *svdf_stream(svdf_trainer *t);
/* adopt a caller-owned HIP stream (e.g. the stream a torch process group orders its collectives against);
 * pending work on the old stream is drained first.  The trainer never destroys an adopted stream. */
int svdf_set_stream(svdf_trainer *t, void *hip_stream);
int svdf_synchronize(svdf_trainer *t);
/* counters: 0 instances trained, 1 kernels launched, 2 conflict-free batches executed,
 * 3 staged-window flushes, 4/5/6 launches of the basicMF / general / few-row fused kernel, 7 rank passes sampled on the device,
 * 8..12 amd:gpus handles (exchanges, RCCL, distinct devices, window steps, exchange path), 13 / 14 the last svdf_init_model on the
 * device: values the host libm decided (next to a float rounding boundary) / rand() draws consumed (0 = the host loop ran), 15 conflict-free
 * levels executed inside chained launches, 16 .. 20 the decision of `amd:step = auto` for the data set built last (16: 0 none, 1 exact levels
 * kept, 2 window step chosen, 3 exact kept because the window step does not cover the configuration / the rows; 17 conflict-free levels;
 * 18 / 19 dag bound and stream model in microseconds; 20 windows), 21 unused,
 * 22 / 23 passes over hot-row units / runs, 24 microseconds the schedule of the last user-unit data set took, 25 whether the device built it,
 * 26 data sets whose DEFAULT (exact) step drew the depth warning (a stderr line naming `amd:step = auto`: the level schedule predicts the pass
 * more than 10 x slower than the streaming model), 27 / 28 the last noted data set's dag bound / stream model in microseconds,
 * 29 passes over rank pairs walked as user-run units */
int64_t svdf_counter(svdf_trainer *t, int what);
/* Tuning knobs (not part of the reference surface).  None changes a result bit except the five marked (*), which move the windows of the
 * OPT-IN window step only.  Every knob, its default, what other values select (round 6: knobs no test or tool sets were deleted).
 *   staging / launches
 *     stage_window        2^21   instances staged by svdf_update_* before an automatic flush (also set by the config key amd:window)
 *     async_flush         1      full staged windows are scheduled on a background thread
 *     use_graph           0      1 = a resident data set's pass is replayed as a captured hipGraph
 *     groups_per_wave     0      lane-group sets per wave of the contract / few-row kernels (0 = tuned per width; 1..6, 8)
 *     block_threads       0      workgroup size (0 = tuned per width; 64, 128, 256)
 *     xcd_remap           1      blockIdx -> tile mapping that keeps neighbouring tiles on one XCD's L2 (0 = plain)
 *     load_mode           2      row gathers: 0 plain, 1 nontemporal, 2 = by row size
 *     store_mode          0      row stores: 0 plain, 1 nontemporal, 2 write-through
 *     sort_batches        1      order inside a conflict-free level: 0 file order, 1 by item, 2 by user
 *     chain_width         128    levels of at most this many instances are walked inside ONE launch by one workgroup (0 = a launch per level)
 *   kernel routing (0 = the more general kernel; same bits)
 *     use_fused 1, fewrow_i16 1, fewrow_gslots 1, basic_i8 1, use_simple_units 1, svdpp_helpers 8 (waves per SVD++ user: 1, 4, 8, 16),
 *     rows_without_feedback 1 (0 = whole users stay sequential units even when no block carries a feedback id)
 *   schedule forms of plain ratings
 *     runs_exec           1      runs of an item's consecutive ratings as units (svdf_k_runs.hip); runs_len 4 (2..7), runs_sets 1, runs_block 64,
 *                                runs_min_rows 2^20 (smaller data sets keep one instance per lane group)
 *     pivot_exec          1      hot rows walked as units (svdf_pivot.cpp); pivot_min 2048 (ratings that make a row hot), pivot_run 256 (per unit)
 *   schedule forms of rank pairs
 *     pair_units          1      user-grouped pair streams (the generator's own order: a user's pairs back to back) walked as user-run units
 *                                (svdf_punit.cpp: k_pair_units, the user's row in registers); pair_unit_cap 16 (pairs per unit at most)
 *   one-off builders on the device (0 = the host builder; same arrays / model)
 *     device_schedule 1 (device_schedule_min 2^16: smaller staged windows stay on the host), device_rank 1, device_init 1
 *     (device_init_margin_log2 46), device_window 1, device_load 1
 *   the opt-in window step (amd:step = minibatch / auto; N-rank handles)
 *     wunit_fast 2, wunit_inplace 1, wunit_defer_fb 1, window_slots 1, window_groups 0      kernel forms, same bits
 *     window_per_target (*) 24, window_per_target_fb (*) 16     updates a shared row / feedback row meets per window on average
 *     window_per_target_max (*) 128                             ... and at most (N-rank steps, rank pairs, user units; plain ratings on one GPU with window_hot_sub = 0)
 *     window_hot_sub (*) 128, window_hot_max (*) 2048           one-GPU sequences of plain ratings (round 6): an item with more than window_hot_sub slots in a window
 *                                                               moves in ordered sub-steps of that many (k_window_apply; 0 = off) and meets at most window_hot_max
 *                                                               updates per window -- the hottest item no longer sets the number of windows
 *     ipc_spin_limit             polls before a flag wait of the IPC exchange gives up
 * Returns 0 if the knob exists, -1 otherwise.  The relaxed mode is switched by CONFIG keys through svdf_set_param ("amd:relax_global",
 * "amd:relax_user_from", "amd:relax_item_from", "amd:relax_feedback"), not by knobs: it changes results. */
int svdf_set_knob(svdf_trainer *t, const char *name, long value);

/* ---- evaluation: RMSEEvaluator of svd_feature_infer.cpp:38-56,243-277 over a resident data set.  Predictions
 * stay in HBM; *sum_sq_err = sum over instances of ((pred - label) * scale_score)^2 (difference and scaling in fp32, square and
 * sum in fp64 like add_eval), *count = instances; RMSE = sqrt(sum / count) (print_stat).  The sum is a fixed fp64 tree on the
 * device + long double over the partial sums, the reference's is a sequential long double sum: equal to ~1e-13 relative.
 * Works on the data sets of an "amd:gpus" handle and on window data sets too (every piece is scored on the rank that holds it);
 * svdf_predict_dataset does not: their rows are regrouped, there is no file order to report predictions in. */
int svdf_eval_dataset(svdf_trainer *t, svdf_dataset *ds, float scale_score, double *sum_sq_err, int64_t *count);

/* ---- ranking: class apex_svd::ISVDRanker (apex_svd.h:160-197) as implemented by SVDFeatureRanker
 * (solvers/base-solver/apex_svd_base.h:597-813), obtained from create_svd_ranker(SVDTypeParam) (apex_svd.h:222).
 *   svdf_ranker_set_param   ISVDRanker::set_param: feature_user, feature_item, top_k (:656-660)
 *   svdf_ranker_load_model  ISVDRanker::load_model (:662-664); the caller has consumed the 4-byte SVDTypeParam
 *   svdf_ranker_init        ISVDRanker::init_ranker(num_item_set) (:666-685)
 *   svdf_ranker_process_*   ISVDRanker::process(std::vector<int>&, Elem / SVDPlusBlock) (:797-812): the tag travels in the
 *                           label field (svdranker_tag, apex_svd.h:115-152: 0 ITEM, 2 USER, 1 POS, -1 BAN, 3 SPEC, 4 PROCESS).
 *                           Results of the line (top_k item indices, or the rank positions of the positive samples) are
 *                           written to out[0..capacity); the return value is the number of results (may exceed capacity:
 *                           call again with a larger buffer is NOT possible, size it num_item_set), -1 on error.
 * Scores are computed on the GPU in the reference's fp32 order; the results are the reference's, index for index. */
svdf_ranker *svdf_ranker_create(uint8_t format_type, uint8_t active_type, uint8_t extend_type, uint8_t variant_type, int device);
void svdf_ranker_destroy(svdf_ranker *r);
int svdf_ranker_set_param(svdf_ranker *r, const char *name, const char *val);
int svdf_ranker_load_model(svdf_ranker *r, FILE *fi);
int svdf_ranker_init(svdf_ranker *r, int num_item_set);
int64_t svdf_ranker_process_csr(svdf_ranker *r, float label, int num_global, int num_ufactor, int num_ifactor, const unsigned *index,
                                const float *value, int *out, int64_t capacity);
/* Bulk form: the rank task's loop over a whole CSR input (svd_feature_infer.cpp:347-375: every line of the iterator goes through
 * process(), results appended) in ONE call.  Same results in the same order as num_row calls of svdf_ranker_process_csr; user
 * sections are pipelined on the device (up to 8 in flight), so a section costs its enqueue, not a launch + sync round trip. */
int64_t svdf_ranker_process_rows(svdf_ranker *r, int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index,
                                 const float *feat_value, int *out, int64_t capacity);
int64_t svdf_ranker_process_block(svdf_ranker *r, int num_ufeedback, int extend_tag, const unsigned *index_ufeedback,
                                  const float *value_ufeedback, int num_row, const float *row_label, const int *row_ptr,
                                  const unsigned *feat_index, const float *feat_value, int *out, int64_t capacity);
/* counters: 0 user sections ranked, 1 sections finished by the host sort because scores tied at a requested position,
 * 2 positive samples of the open user section (what a PROCESS line would report now when top_k = 0: sizes the output) */
int64_t svdf_ranker_counter(svdf_ranker *r, int what);

/* ---- libc rand() as a random-access stream (the reference draws rank pairs with rand(), apex-tensor/apex_random.h:42-67;
 * SURVEY 8f2).  svdf_rand_peek writes the next n rand() results WITHOUT consuming them, svdf_rand_skip advances the generator
 * by n draws without calling rand() n times (jump-ahead of glibc's additive-feedback table).  Host functions, no GPU needed;
 * the device sampler uses the same machinery.  -1 when libc's generator is not in its default 31-word mode. */
int svdf_rand_peek(long n, int *out);
int svdf_rand_skip(long n);

/* ---- probe of the device-side expf used by the sigmoid links (active_type::map_active / cal_grad call libm's expf,
 * apex_svd_model.h:112-156): out[j] = expf evaluated ON THE GPU for in[j], or, with in == NULL, for the float whose bit
 * pattern is first_bits + j*step_bits.  Tests compare it with the host libm bit for bit.  Needs a GPU. */
int svdf_device_expf(const float *in, unsigned first_bits, unsigned step_bits, float *out, long n);

/* ---- host-side conflict-free batch scheduler, exposed so it can be tested without a GPU.
 * Unit r touches resources res[res_ptr[r] .. res_ptr[r+1]) out of num_res.  Writes the batch-sorted
 * (stable) unit order and level_ptr[0..nlevels]; returns the number of batches, -2 if level_cap
 * (capacity of level_ptr_out) is too small. */
int svdf_schedule_resources(long n, const int64_t *res_ptr, const unsigned *res, long num_res, int *order_out,
                            int64_t *level_ptr_out, long level_cap);

#ifdef __cplusplus
}
#endif
#endif /* SVDFEATURE_AMD_H_ */
