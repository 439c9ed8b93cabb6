// svdf_capi.cpp -- extern "C" surface of include/svdfeature_amd.h over svdf::Engine.
#include <svdfeature_amd.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "svdf_engine.h"
#include "svdf_kernels.h"

namespace svdf {
void set_error_mode(int m);
const char *last_error();
void note_error(const std::string &m);
}  // namespace svdf

struct svdf_trainer { svdf::Engine *e; };
struct svdf_dataset { svdf::Dataset *d; };
struct svdf_ranker { svdf::Ranker *r; };

// In error mode 0 svdf::fail() has already printed and exited (reference behaviour); in mode 1 it
// throws and the wrapper turns that into a status code.
#define SVDF_GUARD(retval, ...)                                 \
    try { __VA_ARGS__; }                                        \
    catch (const std::exception &ex) { svdf::note_error(ex.what()); return retval; } \
    catch (...) { svdf::note_error("unknown error"); return retval; }

extern "C" {

const char *svdf_version(void) { return "svdfeature_amd 0.1 (gfx950)"; }
void svdf_set_error_mode(int mode) { svdf::set_error_mode(mode); }
const char *svdf_last_error(void) { return svdf::last_error(); }
int svdf_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

svdf_trainer *svdf_create(uint8_t format_type, uint8_t active_type, uint8_t extend_type, uint8_t variant_type, int device) {
    SVDF_GUARD(nullptr, {
        svdf::TypeParam tp{format_type, active_type, extend_type, variant_type};
        svdf_trainer *t = new svdf_trainer();
        t->e = nullptr;
        try { t->e = new svdf::Engine(tp, device); } catch (...) { delete t; throw; }
        return t;
    })
}
void svdf_destroy(svdf_trainer *t) {
    if (!t) return;
    try { delete t->e; } catch (...) {}
    delete t;
}
int svdf_set_param(svdf_trainer *t, const char *name, const char *val) { SVDF_GUARD(-1, { t->e->set_param(name, val); return 0; }) }
void svdf_seed(unsigned seed) { srand(seed); }
int svdf_init_model(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->init_model(); return 0; }) }
int svdf_load_model(svdf_trainer *t, FILE *fi) { SVDF_GUARD(-1, { t->e->load_model(fi); return 0; }) }
int svdf_save_model(svdf_trainer *t, FILE *fo) { SVDF_GUARD(-1, { t->e->save_model(fo); return 0; }) }
int svdf_save_model_begin(svdf_trainer *t, FILE *fo) { SVDF_GUARD(-1, { t->e->save_model_begin(fo); return 0; }) }
int svdf_save_model_end(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->save_model_end(); return 0; }) }
int svdf_init_trainer(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->init_trainer(); return 0; }) }
int svdf_set_round(svdf_trainer *t, int nround) { SVDF_GUARD(-1, { t->e->set_round(nround); return 0; }) }
int svdf_finish_round(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->finish_round(); return 0; }) }

int svdf_update_csr(svdf_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    SVDF_GUARD(-1, { t->e->update_csr(label, ng, nu, ni, index, value); return 0; })
}
float svdf_predict_csr(svdf_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    SVDF_GUARD(0.0f, { return t->e->predict_csr(label, ng, nu, ni, index, value); })
}
int svdf_update_csr_batch(svdf_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                          const unsigned *feat_index, const float *feat_value) {
    SVDF_GUARD(-1, { t->e->update_csr_batch(num_row, row_label, row_ptr, feat_index, feat_value); return 0; })
}
int svdf_predict_csr_batch(svdf_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value, float *out) {
    SVDF_GUARD(-1, { t->e->predict_csr_batch(num_row, row_label, row_ptr, feat_index, feat_value, out); return 0; })
}
int svdf_update_block(svdf_trainer *t, int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row,
                      const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    SVDF_GUARD(-1, { t->e->update_block(nfb, tag, ifb, vfb, num_row, row_label, row_ptr, feat_index, feat_value); return 0; })
}
int svdf_predict_block(svdf_trainer *t, int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row,
                       const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, float *out) {
    SVDF_GUARD(-1, { t->e->predict_block(nfb, tag, ifb, vfb, num_row, row_label, row_ptr, feat_index, feat_value, out); return 0; })
}

svdf_dataset *svdf_dataset_from_csr(svdf_trainer *t, long num_row, const float *row_label, const int64_t *row_ptr,
                                    const unsigned *feat_index, const float *feat_value) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_csr(num_row, row_label, row_ptr, feat_index, feat_value);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_from_triples(svdf_trainer *t, long n, const unsigned *user, const unsigned *item, const float *label) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_triples(n, user, item, label);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_window_from_triples(svdf_trainer *t, long n, const unsigned *user, const unsigned *item, const float *label) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_window_from_triples(n, user, item, label);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_window_from_pairs(svdf_trainer *t, long n, const unsigned *user, const unsigned *pos_item, const unsigned *neg_item) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_window_from_pairs(n, user, pos_item, neg_item);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_window_from_csr(svdf_trainer *t, long num_row, const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                           const float *feat_value) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_window_from_csr(num_row, row_label, row_ptr, feat_index, feat_value);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_window_from_blocks(svdf_trainer *t, long num_block, const int *extend_tag, const int64_t *fb_ptr, const unsigned *fb_index,
                                              const float *fb_value, const int64_t *block_row_ptr, const float *row_label, const int64_t *row_ptr,
                                              const unsigned *feat_index, const float *feat_value) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_window_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr, feat_index, feat_value);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
int svdf_ipc_setup(svdf_trainer *t, int rank, int world, int64_t wire_bytes, int64_t block_floats, unsigned char *handles_out) { SVDF_GUARD(-1, { t->e->ipc_setup(rank, world, wire_bytes, block_floats, handles_out); return 0; }) }
int svdf_ipc_connect(svdf_trainer *t, const unsigned char *all_handles) { SVDF_GUARD(-1, { t->e->ipc_connect(all_handles); return 0; }) }
int svdf_ipc_window_pack(svdf_trainer *t, svdf_dataset *ds, int half) { SVDF_GUARD(-1, { t->e->ipc_window_pack(ds ? ds->d : nullptr, half); return 0; }) }
int svdf_ipc_window_reduce(svdf_trainer *t, int half) { SVDF_GUARD(-1, { t->e->ipc_window_reduce(half); return 0; }) }
int svdf_ipc_window_apply(svdf_trainer *t, int half) { SVDF_GUARD(-1, { t->e->ipc_window_apply(half); return 0; }) }
int svdf_ipc_block_send(svdf_trainer *t, int dst_rank, int slot) { SVDF_GUARD(-1, { t->e->ipc_block_send(dst_rank, slot); return 0; }) }
int svdf_ipc_block_recv(svdf_trainer *t, int src_rank, int slot, unsigned seq) { SVDF_GUARD(-1, { t->e->ipc_block_recv(src_rank, slot, seq); return 0; }) }
int svdf_ipc_status(svdf_trainer *t) { SVDF_GUARD(-1, { return t->e->ipc_status(); }) }
int svdf_ipc_close(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->ipc_close(); return 0; }) }
int svdf_rccl_unique_id(unsigned char *out128) { SVDF_GUARD(-1, { svdf::rccl_unique_id(out128); return 0; }) }
int svdf_rccl_init(svdf_trainer *t, const unsigned char *id128, int rank, int world) { SVDF_GUARD(-1, { t->e->rccl_init(id128, rank, world); return 0; }) }
int svdf_rccl_window_allreduce(svdf_trainer *t, svdf_dataset *ds, int half) { SVDF_GUARD(-1, { t->e->rccl_window_allreduce(ds ? ds->d : nullptr, half); return 0; }) }
int svdf_rccl_block_handoff(svdf_trainer *t, int dst_rank, int src_rank, int slot, int in_block, int nblocks) { SVDF_GUARD(-1, { t->e->rccl_block_handoff(dst_rank, src_rank, slot, in_block, nblocks); return 0; }) }
int svdf_rccl_block_arrive(svdf_trainer *t, int slot) { SVDF_GUARD(-1, { t->e->rccl_block_arrive(slot); return 0; }) }
int64_t svdf_rccl_counter(svdf_trainer *t, int what) { SVDF_GUARD(-1, { return t->e->rccl_counter(what); }) }
int svdf_rccl_close(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->rccl_close(); return 0; }) }
int svdf_window_delta_pack(svdf_trainer *t, svdf_dataset *ds, void *dst, int half, int64_t *count) {
    SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->window_delta_pack(ds ? ds->d : nullptr, dst, half, count); return 0; })
}
int svdf_window_delta_apply_local(svdf_trainer *t, svdf_dataset *ds) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->window_delta_apply_local(ds ? ds->d : nullptr); return 0; }) }
int svdf_item_block_get(svdf_trainer *t, float *dst, int64_t *count) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_block_copy(dst, 0, count); return 0; }) }
int svdf_stratum_step(svdf_trainer *t, svdf_dataset *const *windows, int num_windows, int block, int nblocks, float *device_out) {
    SVDF_GUARD(-1, {
        t->e->per_rank_api();
        std::vector<svdf::Dataset *> d((size_t)(num_windows > 0 ? num_windows : 0));
        for (int w = 0; w < num_windows; w++) d[(size_t)w] = windows[w] ? windows[w]->d : nullptr;
        t->e->stratum_step(d.data(), num_windows, block, nblocks, device_out);
        return 0;
    })
}
int svdf_item_block_set_at(svdf_trainer *t, int block, int nblocks, const float *src) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_block_set_at(block, nblocks, src); return 0; }) }
int svdf_item_block_set(svdf_trainer *t, const float *src) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_block_copy(const_cast<float *>(src), 1, nullptr); return 0; }) }
int svdf_window_delta_apply(svdf_trainer *t, const void *src, int half) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->window_delta_apply(src, half); return 0; }) }
int svdf_debug_sort_labels(long n, const float *label, int *restated, int *library) {
    SVDF_GUARD(-1, {
        svdf::host_sort_by_label(label, n, restated);
        std::vector<int> ids((size_t)std::max<long>(n, 0));
        for (long j = 0; j < n; j++) ids[(size_t)j] = (int)j;
        std::sort(ids.begin(), ids.end(), [label](int a, int b) { return label[a] < label[b]; });
        for (long j = 0; j < n; j++) library[j] = ids[(size_t)j];
        return 0;
    })
}
int svdf_debug_sort_scores(long n, const float *score, int threads, int *parallel, int *library) {
    SVDF_GUARD(-1, {
        for (long j = 0; j < n; j++) parallel[j] = (int)j;
        svdf::host_parallel_sort_scores(score, parallel, n, threads);
        struct Entry { int iid; float score; bool operator<(const Entry &p) const { return score > p.score; } };   // apex_svd_base.h:617-624
        std::vector<Entry> e((size_t)std::max<long>(n, 0));
        for (long j = 0; j < n; j++) e[(size_t)j] = Entry{(int)j, score[j]};
        std::sort(e.begin(), e.end());
        for (long j = 0; j < n; j++) library[j] = e[(size_t)j].iid;
        return 0;
    })
}
svdf_dataset *svdf_dataset_from_pairs(svdf_trainer *t, long n, const unsigned *user, const unsigned *pos_item, const unsigned *neg_item) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_pairs(n, user, pos_item, neg_item);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_from_blocks(svdf_trainer *t, long num_block, const int *extend_tag, const int64_t *fb_ptr,
                                       const unsigned *fb_index, const float *fb_value, const int64_t *block_row_ptr,
                                       const float *row_label, const int64_t *row_ptr, const unsigned *feat_index,
                                       const float *feat_value) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_blocks(num_block, extend_tag, fb_ptr, fb_index, fb_value, block_row_ptr, row_label, row_ptr,
                                                     feat_index, feat_value);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_from_buffer_file(svdf_trainer *t, const char *path, int user_group_format) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_buffer_file(path, user_group_format);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
svdf_dataset *svdf_dataset_from_rank_buffer_file(svdf_trainer *t, const char *path) {
    SVDF_GUARD(nullptr, {
        svdf::Dataset *d = t->e->dataset_from_rank_buffer_file(path);
        t->e->note_dataset(d);
        svdf_dataset *h = new svdf_dataset();
        h->d = d;
        return h;
    })
}
int svdf_rank_prefetch_buffer_file(svdf_trainer *t, const char *path) { SVDF_GUARD(-1, { t->e->rank_prefetch(path); return 0; }) }
int64_t svdf_rank_sample_buffer_file(svdf_trainer *t, const char *in_path, const char *out_path) {
    SVDF_GUARD(-1, { return (int64_t)t->e->rank_sample_buffer_file(in_path, out_path); })
}
void svdf_dataset_destroy(svdf_dataset *ds) {
    if (!ds) return;
    try { if (ds->d && ds->d->owner) ds->d->owner->synchronize(); } catch (...) {}
    delete ds->d;
    delete ds;
}
int svdf_train_dataset(svdf_trainer *t, svdf_dataset *ds) { SVDF_GUARD(-1, { t->e->train_dataset(ds->d); return 0; }) }
int svdf_predict_dataset(svdf_trainer *t, svdf_dataset *ds, float *out) { SVDF_GUARD(-1, { t->e->predict_dataset(ds->d, out); return 0; }) }
int64_t svdf_dataset_info(const svdf_dataset *ds, int what) {
    if (!ds || !ds->d) return -1;
    switch (what) {
    case 0: return ds->d->num_row;
    case 1: return ds->d->kind == 8 ? (int64_t)ds->d->wchild.size() : (int64_t)ds->d->sched.num_levels();   // kind 8: windows per pass
    case 2: return ds->d->sched.max_level_size;
    case 3: return ds->d->kind;
    case 4: return ds->d->algorithmic_bytes;
    case 5: return ds->d->num_units;
    case 6: return ds->d->num_simple_units;
    case 7: {   // FNV-1a over the host-resident schedule (level_ptr, level_mid, order): equal schedules <=> equal digests (tests)
        uint64_t h = 1469598103934665603ull;
        auto mix = [&h](uint64_t v) { for (int b = 0; b < 8; b++) { h ^= (v >> (8 * b)) & 0xFFu; h *= 1099511628211ull; } };
        for (long v : ds->d->sched.level_ptr) mix((uint64_t)v);
        for (long v : ds->d->sched.level_mid) mix((uint64_t)v);
        for (int v : ds->d->sched.order) mix((uint64_t)(unsigned)v);
        return (int64_t)(h >> 1);
    }
    default: return -1;
    }
}

int svdf_item_delta_begin(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_begin(); return 0; }) }
void *svdf_item_delta_buffer(svdf_trainer *t, int64_t *count) { SVDF_GUARD(nullptr, { t->e->per_rank_api(); return t->e->item_delta_buffer(count); }) }
int svdf_item_delta_apply(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_apply(); return 0; }) }
int svdf_item_delta_export(svdf_trainer *t, float *dst) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_copy(dst, nullptr); return 0; }) }
int svdf_item_delta_import(svdf_trainer *t, const float *src) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_copy(nullptr, src); return 0; }) }

int svdf_item_delta_into(svdf_trainer *t, float *dst, int64_t *count) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_into(dst, count); return 0; }) }
int svdf_item_delta_pack(svdf_trainer *t, void *dst, int half, int64_t *count) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_pack(dst, half, count); return 0; }) }
int svdf_item_delta_unpack(svdf_trainer *t, const void *src, int half, int refresh_snapshot) {
    SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_unpack(src, half, refresh_snapshot); return 0; })
}
int svdf_item_delta_select(svdf_trainer *t, int part, int nparts) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_select(part, nparts); return 0; }) }
int svdf_item_delta_apply_from(svdf_trainer *t, const float *src) { SVDF_GUARD(-1, { t->e->per_rank_api(); t->e->item_delta_apply_from(src); return 0; }) }
int svdf_set_stream(svdf_trainer *t, void *hip_stream) { SVDF_GUARD(-1, { t->e->set_stream((hipStream_t)hip_stream); return 0; }) }

int64_t svdf_get_view(svdf_trainer *t, int which, float *out, int64_t capacity) { SVDF_GUARD(-1, { return t->e->get_view(which, out, capacity); }) }
int64_t svdf_set_view(svdf_trainer *t, int which, const float *in, int64_t count) { SVDF_GUARD(-1, { return t->e->set_view(which, in, count); }) }
int svdf_view_shape(svdf_trainer *t, int which, int *rows, int *cols) { SVDF_GUARD(-1, { t->e->view_shape(which, rows, cols); return 0; }) }
void *svdf_stream(svdf_trainer *t) { return (void *)t->e->stream(); }
int svdf_synchronize(svdf_trainer *t) { SVDF_GUARD(-1, { t->e->synchronize(); return 0; }) }
int64_t svdf_counter(svdf_trainer *t, int what) { return t->e->counter(what); }
int svdf_set_knob(svdf_trainer *t, const char *name, long value) { SVDF_GUARD(-1, { return t->e->set_knob(name, value); }) }

int svdf_eval_dataset(svdf_trainer *t, svdf_dataset *ds, float scale_score, double *sum_sq_err, int64_t *count) {
    SVDF_GUARD(-1, { t->e->eval_dataset(ds->d, scale_score, sum_sq_err, count); return 0; })
}

/* ---- ISVDRanker ---- */
svdf_ranker *svdf_ranker_create(uint8_t format_type, uint8_t active_type, uint8_t extend_type, uint8_t variant_type, int device) {
    SVDF_GUARD(nullptr, {
        svdf::TypeParam tp{format_type, active_type, extend_type, variant_type};
        svdf_ranker *h = new svdf_ranker();
        h->r = nullptr;
        try { h->r = new svdf::Ranker(tp, device); } catch (...) { delete h; throw; }
        return h;
    })
}
void svdf_ranker_destroy(svdf_ranker *r) {
    if (!r) return;
    try { delete r->r; } catch (...) {}
    delete r;
}
int svdf_ranker_set_param(svdf_ranker *r, const char *name, const char *val) { SVDF_GUARD(-1, { r->r->set_param(name, val); return 0; }) }
int svdf_ranker_load_model(svdf_ranker *r, FILE *fi) { SVDF_GUARD(-1, { r->r->load_model(fi); return 0; }) }
int svdf_ranker_init(svdf_ranker *r, int num_item_set) { SVDF_GUARD(-1, { r->r->init_ranker(num_item_set); return 0; }) }
int64_t svdf_ranker_process_csr(svdf_ranker *r, float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out,
                                int64_t capacity) {
    SVDF_GUARD(-1, { return (int64_t)r->r->process(label, ng, nu, ni, index, value, out, (long)capacity); })
}
int64_t svdf_ranker_process_rows(svdf_ranker *r, int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index,
                                 const float *feat_value, int *out, int64_t capacity) {
    SVDF_GUARD(-1, { return (int64_t)r->r->process_rows(num_row, row_label, row_ptr, feat_index, feat_value, out, (long)capacity); })
}
int64_t svdf_ranker_process_block(svdf_ranker *r, int nfb, int tag, const unsigned *ifb, const float *vfb, int num_row, const float *row_label,
                                  const int *row_ptr, const unsigned *feat_index, const float *feat_value, int *out, int64_t capacity) {
    SVDF_GUARD(-1, { return (int64_t)r->r->process_block(nfb, tag, ifb, vfb, num_row, row_label, row_ptr, feat_index, feat_value, out, (long)capacity); })
}
int64_t svdf_ranker_counter(svdf_ranker *r, int what) { return r->r->counter(what); }

int svdf_rand_peek(long n, int *out) { SVDF_GUARD(-1, { svdf::libc_rand_peek(n, out); return 0; }) }
int svdf_rand_skip(long n) { SVDF_GUARD(-1, { svdf::libc_rand_skip(n); return 0; }) }

int svdf_device_expf(const float *in, unsigned first_bits, unsigned step_bits, float *out, long n) {
    SVDF_GUARD(-1, { return svdf::device_expf(in, first_bits, step_bits, out, n); })
}

// host-side scheduler exposed for CPU tests (tests/test_scheduler.py): levels for a CSR stream over
// `num_res` resources where instance r touches resources res[res_ptr[r]..res_ptr[r+1]).
int svdf_schedule_resources(long n, const int64_t *res_ptr, const unsigned *res, long num_res, int *order_out,
                            int64_t *level_ptr_out, long level_cap) {
    SVDF_GUARD(-1, {
        std::vector<int> last((size_t)num_res, 0), levels((size_t)n);
        for (long r = 0; r < n; r++) {
            int l = 0;
            for (int64_t j = res_ptr[r]; j < res_ptr[r + 1]; j++) l = std::max(l, last[res[j]]);
            l += 1;
            for (int64_t j = res_ptr[r]; j < res_ptr[r + 1]; j++) last[res[j]] = l;
            levels[(size_t)r] = l;
        }
        svdf::Schedule s;
        svdf::build_schedule(levels, 0, s);
        if ((long)s.num_levels() + 1 > level_cap) return -2;
        for (long r = 0; r < n; r++) order_out[r] = s.order[(size_t)r];
        for (size_t l = 0; l <= s.num_levels(); l++) level_ptr_out[l] = s.level_ptr[l];
        return (int)s.num_levels();
    })
}

}  // extern "C"
