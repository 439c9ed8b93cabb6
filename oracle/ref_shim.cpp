/*
 * ref_shim.cpp -- C wrapper (svdf_oracle.h API) around the REFERENCE's own solver classes.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  This file contains no reference code: it #includes the
 * reference's public header from /root/reference (apex_svd.h) and is linked, by oracle/Makefile,
 * against the reference's own translation units compiled where they lie
 * (solvers/base-solver/apex_svd_base.cpp = the factory, apex_svd_data.cpp = data iterators).
 * The result, oracle/_ref/libsvdf_ref.so, is git-ignored and is used only to validate the C
 * restatement (svdf_oracle.c), to generate tests/golden/ and as bench.py's cpu_baseline
 * (kind "reference").
 *
 * Objects are obtained exactly the way svd_feature.cpp:198-216 obtains them:
 * apex_svd::create_svd_trainer(SVDTypeParam) and then only ISVDTrainer virtual calls.
 */
#define _GNU_SOURCE 1
#include "apex_svd.h" /* -I/root/reference */
#include "apex-tensor/apex_random.h"

#include <unistd.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "svdf_oracle.h"

using namespace apex_svd;

struct svdo_trainer {
    SVDTypeParam mtype;
    ISVDTrainer *tr;
    int stale_bf16 = 0;
};

extern "C" {

int svdo_kind(void) { return 2; }
void svdo_libm_expf(const float *in, unsigned first, unsigned step, float *out, long n) {
    for (long j = 0; j < n; j++) {
        float x;
        if (in) x = in[j];
        else { unsigned u = first + (unsigned)j * step; memcpy(&x, &u, 4); }
        out[j] = expf(x);
    }
}


svdo_trainer *svdo_create(int format_type, int active_type, int extend_type, int variant_type) {
    svdo_trainer *t = new svdo_trainer();
    t->mtype.format_type = (uint8_t)format_type;
    t->mtype.active_type = (uint8_t)active_type;
    t->mtype.extend_type = (uint8_t)extend_type;
    t->mtype.variant_type = (uint8_t)variant_type;
    t->tr = create_svd_trainer(t->mtype);
    return t;
}
void svdo_destroy(svdo_trainer *t) {
    if (!t) return;
    delete t->tr;
    delete t;
}
void svdo_set_param(svdo_trainer *t, const char *name, const char *val) { t->tr->set_param(name, val); }
void svdo_seed(unsigned seed) { apex_random::seed(seed); }
void svdo_init_model(svdo_trainer *t) { t->tr->init_model(); }
void svdo_init_trainer(svdo_trainer *t) { t->tr->init_trainer(); }
void svdo_set_round(svdo_trainer *t, int nround) { t->tr->set_round(nround); }
void svdo_finish_round(svdo_trainer *t) { t->tr->finish_round(); }

int svdo_save_model_path(svdo_trainer *t, const char *path, int with_type_header) {
    FILE *fo = fopen(path, "wb");
    if (!fo) return -1;
    if (with_type_header) fwrite(&t->mtype, sizeof(SVDTypeParam), 1, fo);
    t->tr->save_model(fo);
    fclose(fo);
    return 0;
}
int svdo_load_model_path(svdo_trainer *t, const char *path, int with_type_header) {
    FILE *fi = fopen(path, "rb");
    if (!fi) return -1;
    if (with_type_header) {
        SVDTypeParam mt;
        if (fread(&mt, sizeof(SVDTypeParam), 1, fi) != 1) { fclose(fi); return -1; }
    }
    /* this fork's load_from_file drops u_bias.txt etc. into the CWD (apex_svd_model.h:586-621);
     * run it inside a scratch directory so the caller's CWD stays clean */
    char cwd[4096];
    char tmpl[] = "/tmp/svdf_ref_XXXXXX";
    char *scratch = mkdtemp(tmpl);
    bool moved = scratch && getcwd(cwd, sizeof(cwd)) && chdir(scratch) == 0;
    t->tr->load_model(fi);
    if (moved) {
        const char *junk[] = {"u_bias.txt", "i_bias.txt", "w_user.txt", "w_item.txt"};
        for (int i = 0; i < 4; i++) unlink(junk[i]);
        if (chdir(cwd) != 0) { /* nothing sensible to do */ }
        rmdir(scratch);
    }
    fclose(fi);
    return 0;
}

static SVDFeatureCSR::Elem make_elem(float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    SVDFeatureCSR::Elem e;
    e.label = label;
    e.num_global = ng; e.num_ufactor = nu; e.num_ifactor = ni;
    e.set_space(const_cast<unsigned *>(index), const_cast<float *>(value));
    return e;
}
static SVDFeatureCSR make_csr(int num_row, const float *row_label, const int *row_ptr,
                              const unsigned *feat_index, const float *feat_value) {
    SVDFeatureCSR m;
    m.num_row = num_row;
    m.num_val = row_ptr[3 * num_row] - row_ptr[0];
    m.row_label = const_cast<float *>(row_label);
    m.row_ptr = const_cast<int *>(row_ptr);
    m.feat_index = const_cast<unsigned *>(feat_index);
    m.feat_value = const_cast<float *>(feat_value);
    return m;
}

void svdo_update_csr(svdo_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    t->tr->update(make_elem(label, ng, nu, ni, index, value));
}
float svdo_predict_csr(svdo_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    return t->tr->predict(make_elem(label, ng, nu, ni, index, value));
}
void svdo_update_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value) {
    SVDFeatureCSR m = make_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    for (int r = 0; r < num_row; r++) t->tr->update(m[r]);
}
void svdo_predict_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                            const unsigned *feat_index, const float *feat_value, float *out) {
    SVDFeatureCSR m = make_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    for (int r = 0; r < num_row; r++) out[r] = t->tr->predict(m[r]);
}
static SVDPlusBlock make_block(int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                               int num_row, const float *row_label, const int *row_ptr,
                               const unsigned *feat_index, const float *feat_value) {
    SVDPlusBlock b;
    b.num_ufeedback = nfb;
    b.extend_tag = extend_tag;
    b.index_ufeedback = const_cast<unsigned *>(idx_fb);
    b.value_ufeedback = const_cast<float *>(val_fb);
    b.data = make_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    return b;
}
void svdo_update_block(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                       int num_row, const float *row_label, const int *row_ptr,
                       const unsigned *feat_index, const float *feat_value) {
    t->tr->update(make_block(nfb, extend_tag, idx_fb, val_fb, num_row, row_label, row_ptr, feat_index, feat_value));
}
void svdo_predict_block(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                        int num_row, const float *row_label, const int *row_ptr,
                        const unsigned *feat_index, const float *feat_value, float *out) {
    std::vector<float> p;
    t->tr->predict(p, make_block(nfb, extend_tag, idx_fb, val_fb, num_row, row_label, row_ptr, feat_index, feat_value));
    for (size_t i = 0; i < p.size(); i++) out[i] = p[i];
}

/* views are recovered from the bytes ISVDTrainer::save_model writes (layout: SURVEY.md section 5) */
static bool dump_model(svdo_trainer *t, std::vector<char> &buf) {
    char *mem = NULL;
    size_t len = 0;
    FILE *fo = open_memstream(&mem, &len);
    if (!fo) return false;
    t->tr->save_model(fo);
    fclose(fo);
    buf.assign(mem, mem + len);
    free(mem);
    return true;
}
struct view_pos { long off; int rows, cols; };
static bool locate(svdo_trainer *t, const std::vector<char> &buf, int which, view_pos &vp) {
    const int *hdr = reinterpret_cast<const int *>(buf.data());
    const int common_latent = hdr[12], common_fb = hdr[14];
    if (common_latent != 0) return false; /* not needed by the tests */
    long off = 1056;
    for (int v = 0; v <= 6; v++) {
        if (v >= 5 && !(t->mtype.format_type == 1 && common_fb == 0)) return false;
        if (off >= (long)buf.size()) return false;
        const int *h = reinterpret_cast<const int *>(buf.data() + off);
        int rows, cols;
        long start;
        if (v == 1 || v == 3 || v == 6) { cols = h[0]; rows = h[1]; start = off + 8; }
        else { cols = 1; rows = h[0]; start = off + 4; }
        if (v == which) { vp.off = start; vp.rows = rows; vp.cols = cols; return true; }
        off = start + 4L * rows * cols;
    }
    return false;
}
void svdo_view_shape(svdo_trainer *t, int which, int *rows, int *cols) {
    std::vector<char> buf;
    view_pos vp;
    *rows = -1; *cols = 0;
    if (dump_model(t, buf) && locate(t, buf, which, vp)) { *rows = vp.rows; *cols = vp.cols; }
}
long svdo_get_view(svdo_trainer *t, int which, float *out, long capacity) {
    std::vector<char> buf;
    view_pos vp;
    if (!dump_model(t, buf) || !locate(t, buf, which, vp)) return -1;
    long n = (long)vp.rows * vp.cols;
    if (n > capacity) return -1;
    memcpy(out, buf.data() + vp.off, sizeof(float) * (size_t)n);
    return n;
}

long svdo_set_view(svdo_trainer *, int, const float *, long) { return -1; }

/* the window-minibatch checker step (svdf_oracle.h) on the REFERENCE's own classes: every row is one ISVDTrainer::update on a
 * trainer whose replicated side has been put back to the window-start values through the reference's own save_model /
 * load_model (model bytes patched in memory).  O(model size) per row: for small pinning tests only. */
static void load_from_bytes(svdo_trainer *t, std::vector<char> &buf) {
    FILE *fi = fmemopen(buf.data(), buf.size(), "rb");
    char cwd[4096];
    char tmpl[] = "/tmp/svdf_ref_XXXXXX";
    char *scratch = mkdtemp(tmpl);   /* load_from_file drops text dumps into the CWD (apex_svd_model.h:586-621) */
    bool moved = scratch && getcwd(cwd, sizeof(cwd)) && chdir(scratch) == 0;
    t->tr->load_model(fi);
    if (moved) {
        const char *junk[] = {"u_bias.txt", "i_bias.txt", "w_user.txt", "w_item.txt"};
        for (int i = 0; i < 4; i++) unlink(junk[i]);
        if (chdir(cwd) != 0) { }
        rmdir(scratch);
    }
    fclose(fi);
}
/* one row: ISVDTrainer::update(elem) -- or, for a user-group trainer, update(MIDDLE block holding that one row) = update_each of the
 * row (apex_svd_base.h:560-565, 568-582) -- then the replicated rows it touched are diffed against the bytes of before and put back */
static float bf16_round(float x) {
    union { float f; unsigned u; } v;
    v.f = x;
    v.u = ((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16) << 16;
    return v.f;
}
void svdo_set_stale_rounding(svdo_trainer *t, int bf16) { t->stale_bf16 = bf16 != 0; }
static int stale_row_ref(svdo_trainer *t, const SVDFeatureCSR::Elem &e, const SVDPlusBlock *as_block, float *dW_item, float *di_bias, float *dg_bias) {
    std::vector<char> before, after;
    view_pos vw, vb, vg;
    if (!dump_model(t, before)) return -1;
    if (!locate(t, before, 3, vw) || !locate(t, before, 2, vb) || !locate(t, before, 4, vg)) return -1;
    if (as_block) t->tr->update(*as_block);
    else t->tr->update(e);
    if (!dump_model(t, after)) return -1;
    const int k = vw.cols;
    for (int i = 0; i < e.num_ifactor; i++) {
        const unsigned iid = e.index_ifactor[i];
        float *a = reinterpret_cast<float *>(after.data() + vw.off) + (size_t)iid * k;
        const float *b = reinterpret_cast<const float *>(before.data() + vw.off) + (size_t)iid * k;
        for (int j = 0; j < k; j++) { float c = a[j] - b[j]; if (t->stale_bf16) c = bf16_round(c); dW_item[(size_t)iid * k + j] = dW_item[(size_t)iid * k + j] + c; a[j] = b[j]; }
        float *ab = reinterpret_cast<float *>(after.data() + vb.off) + iid;
        const float *bb = reinterpret_cast<const float *>(before.data() + vb.off) + iid;
        float cb = *ab - *bb;
        di_bias[iid] = di_bias[iid] + cb;
        *ab = *bb;
    }
    for (int i = 0; i < e.num_global; i++) {
        const unsigned gid = e.index_global[i];
        float *ag = reinterpret_cast<float *>(after.data() + vg.off) + gid;
        const float *bg = reinterpret_cast<const float *>(before.data() + vg.off) + gid;
        float c = *ag - *bg;
        dg_bias[gid] = dg_bias[gid] + c;
        *ag = *bg;
    }
    load_from_bytes(t, after);
    return 0;
}
int svdo_update_csr_batch_stale(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                                const unsigned *feat_index, const float *feat_value, float *dW_item, float *di_bias, float *dg_bias) {
    if (t->mtype.format_type != 0 || t->mtype.extend_type != 0) return -1;
    SVDFeatureCSR m = make_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    for (int r = 0; r < num_row; r++)
        if (stale_row_ref(t, m[r], NULL, dW_item, di_bias, dg_bias) != 0) return -1;
    return 0;
}
/* The block step of svdf_oracle.h on the reference's own SVDPPFeature.  update(block) with tag T is, by the reference's own code
 * (:568-582), the sequence  [T opens: prepare_ufeedback + backup]  update_each(rows)  [T closes: update_ufeedback]; the same sequence is
 * issued here as update(START block without rows), update(MIDDLE block with ONE row) per row, update(END block without rows), so that the
 * replicated side can be diffed and put back (save_model / load_model, which leave the trainer's tmp_ufeedback state alone) in between. */
int svdo_update_block_stale(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                            int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value,
                            float *dW_item, float *di_bias, float *dg_bias, float *dW_fb, float *dfb_bias) {
    if (t->mtype.format_type != 1 || t->mtype.extend_type != 0) return -1;
    const int none[1] = {0};
    if (extend_tag == svdpp_tag::DEFAULT || extend_tag == svdpp_tag::START_TAG)
        t->tr->update(make_block(nfb, svdpp_tag::START_TAG, idx_fb, val_fb, 0, row_label, none, feat_index, feat_value));
    SVDFeatureCSR m = make_csr(num_row, row_label, row_ptr, feat_index, feat_value);
    for (int r = 0; r < num_row; r++) {
        int one_ptr[4] = {row_ptr[3 * r] - row_ptr[3 * r], row_ptr[3 * r + 1] - row_ptr[3 * r], row_ptr[3 * r + 2] - row_ptr[3 * r], row_ptr[3 * r + 3] - row_ptr[3 * r]};
        SVDPlusBlock b = make_block(nfb, svdpp_tag::MIDDLE_TAG, idx_fb, val_fb, 1, row_label + r, one_ptr, feat_index + row_ptr[3 * r], feat_value + row_ptr[3 * r]);
        if (stale_row_ref(t, m[r], &b, dW_item, di_bias, dg_bias) != 0) return -1;
    }
    if (extend_tag == svdpp_tag::DEFAULT || extend_tag == svdpp_tag::END_TAG) {
        std::vector<char> before, after;
        view_pos vw, vb;
        if (!dump_model(t, before) || !locate(t, before, 6, vw) || !locate(t, before, 5, vb)) return -1;
        t->tr->update(make_block(nfb, svdpp_tag::END_TAG, idx_fb, val_fb, 0, row_label, none, feat_index, feat_value));
        if (!dump_model(t, after)) return -1;
        const int k = vw.cols;
        for (int i = 0; i < nfb; i++) {
            const unsigned fid = idx_fb[i];
            float *a = reinterpret_cast<float *>(after.data() + vw.off) + (size_t)fid * k;
            const float *b = reinterpret_cast<const float *>(before.data() + vw.off) + (size_t)fid * k;
            for (int j = 0; j < k; j++) { float c = a[j] - b[j]; if (t->stale_bf16) c = bf16_round(c); dW_fb[(size_t)fid * k + j] = dW_fb[(size_t)fid * k + j] + c; a[j] = b[j]; }
            float *ab = reinterpret_cast<float *>(after.data() + vb.off) + fid;
            const float *bb = reinterpret_cast<const float *>(before.data() + vb.off) + fid;
            float cb = *ab - *bb;
            dfb_bias[fid] = dfb_bias[fid] + cb;
            *ab = *bb;
        }
        load_from_bytes(t, after);
    }
    return 0;
}


/* ---- the reference's own ranker, obtained like svd_feature_infer.cpp obtains it: create_svd_ranker(SVDTypeParam) ---- */
struct svdo_ranker { SVDTypeParam mtype; ISVDRanker *rk; };
svdo_ranker *svdo_ranker_create(int format_type, int active_type, int extend_type, int variant_type) {
    svdo_ranker *r = new svdo_ranker();
    r->mtype.format_type = (uint8_t)format_type; r->mtype.active_type = (uint8_t)active_type;
    r->mtype.extend_type = (uint8_t)extend_type; r->mtype.variant_type = (uint8_t)variant_type;
    r->rk = create_svd_ranker(r->mtype);
    return r;
}
void svdo_ranker_destroy(svdo_ranker *r) { if (r) { delete r->rk; delete r; } }
void svdo_ranker_set_param(svdo_ranker *r, const char *name, const char *val) { r->rk->set_param(name, val); }
int svdo_ranker_load_model_path(svdo_ranker *r, const char *path, int with_type_header) {
    FILE *fi = fopen(path, "rb");
    if (!fi) return -1;
    if (with_type_header) { SVDTypeParam mt; if (fread(&mt, sizeof(SVDTypeParam), 1, fi) != 1) { fclose(fi); return -1; } }
    char cwd[4096];
    char tmpl[] = "/tmp/svdf_ref_XXXXXX";
    char *scratch = mkdtemp(tmpl);   /* load_from_file drops text dumps into the CWD (apex_svd_model.h:586-621) */
    bool moved = scratch && getcwd(cwd, sizeof(cwd)) && chdir(scratch) == 0;
    r->rk->load_model(fi);
    if (moved) {
        const char *junk[] = {"u_bias.txt", "i_bias.txt", "w_user.txt", "w_item.txt"};
        for (int i = 0; i < 4; i++) unlink(junk[i]);
        if (chdir(cwd) != 0) { }
        rmdir(scratch);
    }
    fclose(fi);
    return 0;
}
void svdo_ranker_init(svdo_ranker *r, int num_item_set) { r->rk->init_ranker(num_item_set); }
long svdo_ranker_process_csr(svdo_ranker *r, float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out, long cap) {
    std::vector<int> res;
    r->rk->process(res, make_elem(label, ng, nu, ni, index, value));
    for (size_t i = 0; i < res.size() && (long)i < cap; i++) out[i] = res[i];
    return (long)res.size();
}
long svdo_ranker_process_block(svdo_ranker *r, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb, int num_row,
                               const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, int *out, long cap) {
    std::vector<int> res;
    r->rk->process(res, make_block(nfb, extend_tag, idx_fb, val_fb, num_row, row_label, row_ptr, feat_index, feat_value));
    for (size_t i = 0; i < res.size() && (long)i < cap; i++) out[i] = res[i];
    return (long)res.size();
}
double svdo_sum_sq_err(const float *pred, const float *label, long n, float scale) {   /* svd_feature_infer.cpp:43-47 */
    long double sum = 0.0f;
    for (long i = 0; i < n; i++) { double diff = (pred[i] - label[i]) * scale; sum += diff * diff; }
    return (double)sum;
}

} /* extern "C" */
