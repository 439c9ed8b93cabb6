/*
 * svdf_oracle.c -- plain-C, single-threaded restatement of the reference's apex_svd SGD hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Loaded only by tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg (kind "port").  The product path never links or calls this file.
 *
 * Parity status: PINNED.  tests/test_oracle.py compares this file, byte for byte on
 * model files and bit for bit on predictions, against the compiled reference
 * (oracle/_ref/libsvdf_ref.so, built by oracle/Makefile from /root/reference) and against the
 * golden vectors under tests/golden/ that were generated from that compiled reference.
 *
 * Every function cites the reference file:line it follows (paths relative to /root/reference).
 * Arithmetic rules that matter for bit parity:
 *   - fp32 everywhere, no FMA contraction (compile with -ffp-contract=off), products and sums
 *     are separate roundings (apex-tensor/apex_tensor_sse.h:261-272: _mm_mul_ps then _mm_add_ps)
 *   - tensor-times-scalar expressions carry the scalar as a double and cast it to float at
 *     evaluation (apex-tensor/apex_exp_template.h:485-502, apex_tensor_func_decl_common.h:239-245)
 *   - multiply-by-scalar is skipped when |s-1| <= 1e-6 (apex_tensor_sse.h:231-242)
 *   - the dot product keeps 4 lane accumulators (lane l sums j = l mod 4 in index order),
 *     reduces (l0+l2)+(l1+l3) and then adds the k%4 tail (apex_tensor_sse.h:88-97,289-317)
 *   - bias sum and final score accumulate in double (solvers/base-solver/apex_svd_base.h:317,446)
 */
#define _GNU_SOURCE
#include "svdf_oracle.h"

#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ---- apex_svd_model.h:373-435 SVDModelParam: 1056-byte on-disk header ---- */
typedef struct {
    int num_user, num_item, num_factor, num_global;
    float u_init_sigma, i_init_sigma, base_score;
    int no_user_bias, num_ufeedback;
    float ufeedback_init_sigma;
    int num_randinit_ufactor, num_randinit_ifactor;
    int common_latent_space, user_nonnegative, common_feedback_space, extend_flag, item_nonnegative;
    int reserved[247];
} model_param;

/* ---- apex_svd_model.h:291-344 SVDTrainParam ---- */
typedef struct {
    float learning_rate;
    int decay_learning_rate;
    float decay_rate, min_learning_rate;
    float wd_user, wd_item, wd_user_bias, wd_item_bias;
    int reg_method;
    float wd_global;
    int reg_global;
    unsigned num_regfree_global;
    float scale_lr_ufeedback, wd_ufeedback_user, wd_ufeedback, wd_ufeedback_bias;
} train_param;

/* ---- solvers/base-solver/apex_svd_base.h:33-75 ParameterSet ---- */
typedef struct {
    float *wd; unsigned *bound; int nwd, nbound;
    const char *prefix_a, *prefix_b;
} param_set;

/* ---- apex-utils/apex_utils.h:140-196 SparseFeatureArray<float> ---- */
typedef struct { unsigned index; float value; } sf_entry;
typedef struct { unsigned num_row; unsigned *row_ptr; sf_entry *data; size_t ndata; } sparse_feat;

#define IMFB_MAX 64
struct svdo_trainer {
    uint8_t mtype[4]; /* format_type, active_type, extend_type, variant_type */
    model_param mp;
    train_param tp;
    int space_allocated, init_end, round_counter;
    /* W_uiset / ui_bias with views (apex_svd_model.h:511-556) */
    int pitch;          /* floats per row: ceil(4k/16)*16 bytes (apex_tensor_sse.h:26-27) */
    int n_uiset;
    float *ui_bias, *W_uiset, *g_bias;
    float *u_bias, *W_user, *i_bias, *W_item, *ufb_bias, *W_ufb;
    /* scratch (apex_svd_base.h:85, 486-488) */
    float *tmp_u, *tmp_i, *tmp_fb, *old_fb;
    float norm_fb, tmp_fb_bias, old_fb_bias;
    int stale_bf16;     /* checker steps only: row contributions rounded to bfloat16 before they are summed (svdo_set_stale_rounding) */
    sparse_feat feat_user, feat_item;
    char name_feat_user[256], name_feat_item[256];
    unsigned sample_counter;
    unsigned *ref_user, *ref_item, *ref_global;
    param_set u_param, i_param, g_param;
    /* extend_type 2: SVDPPMultiIMFB (solvers/multi-imfb/apex_multi_imfb.h:33-44) -- a stack of implicit-feedback levels */
    struct { int num_ufeedback; float norm, tmp_bias, old_bias; float *tmp, *old; } imfb[IMFB_MAX];
    int imfb_top, imfb_alloc;
    unsigned char imfb_disable[IMFB_MAX];
    /* extend_type 15: SVDBiLinearTrainer (solvers/bilinear/apex_svd_bilinear.h:30-77): BParam + W_bi ride along in the model file */
    struct { int num_bi_feedback, start_ufeedback, reserved[32]; } bparam;
    float *W_bi; int bi_allocated, reg_bi_feedback;
};

static void die(const char *msg) { /* apex-utils/apex_utils.h:47-50 */
    fprintf(stderr, "%s\n", msg);
    exit(-1);
}
static void assert_true(int ok, const char *msg) { if (!ok) die(msg); }

int svdo_kind(void) { return 1; }
void svdo_libm_expf(const float *in, unsigned first, unsigned step, float *out, long n) {
    for (long j = 0; j < n; j++) {
        float x;
        if (in) x = in[j];
        else { unsigned u = first + (unsigned)j * step; memcpy(&x, &u, 4); }
        out[j] = expf(x);
    }
}


/* ================= tensor micro-ops ================= */

/* apex_tensor_sse.h:231-242 ScalarOptimizer<ST,Mul>: multiply skipped when |s-1|<=1e-6 */
static int scalar_is_one(float s) { return !(fabs((double)fabsf(s - 1.0f)) > 1e-6); }

/* K1: dst += src*s  (apex_tensor_sse.h:261-272 with Store<AddTo>, :161-166) */
static void axpy(float *dst, const float *src, float s, int n) {
    if (scalar_is_one(s)) { for (int j = 0; j < n; j++) dst[j] = dst[j] + src[j]; return; }
    for (int j = 0; j < n; j++) { float m = src[j] * s; dst[j] = dst[j] + m; }
}
/* K2: dst *= s  (same routine with Store<SaveTo>) */
static void scale(float *dst, float s, int n) {
    if (scalar_is_one(s)) return;
    for (int j = 0; j < n; j++) dst[j] = dst[j] * s;
}
/* K3: apex_tensor_sse.h:289-317 sdot + :88-97 sum_all */
static float sdot(const float *a, const float *b, int n) {
    int len = (n >> 2) << 2;
    float l0 = 0.0f, l1 = 0.0f, l2 = 0.0f, l3 = 0.0f;
    for (int j = 0; j < len; j += 4) {
        float m0 = a[j] * b[j], m1 = a[j + 1] * b[j + 1], m2 = a[j + 2] * b[j + 2], m3 = a[j + 3] * b[j + 3];
        l0 = l0 + m0; l1 = l1 + m1; l2 = l2 + m2; l3 = l3 + m3;
    }
    float h02 = l0 + l2, h13 = l1 + l3; /* movehl add, then shuffle add_ss */
    float sum = h02 + h13;
    for (int j = len; j < n; j++) { float m = a[j] * b[j]; sum = sum + m; }
    return sum;
}
/* K6: apex_tensor_cpu_inline_common.h:168-175 */
static void regularize_l1(float *w, float eps, int n) {
    for (int j = 0; j < n; j++) {
        if (w[j] > eps) w[j] -= eps;
        else if (w[j] < -eps) w[j] += eps;
        else w[j] = 0.0f;
    }
}
/* apex_svd_base.h:175-180 */
static void reg_l1_scalar(float *w, float wd) {
    if (*w > wd) *w -= wd;
    else if (*w < -wd) *w += wd;
    else *w = 0.0f;
}
/* apex_svd_base.h:181-186 */
static void project(float *w, float B, int n) {
    float sum = sdot(w, w, n);
    if (sum > B) scale(w, sqrtf(B / sum), n);
}

/* ================= loss / link (apex_svd_model.h:90-156, 220-237) ================= */
static float smooth_hinge_grad(float z) {
    if (z > 1.0f) return 0.0f;
    if (z < 0.0f) return 1.0f;
    return 1.0f - z;
}
static float map_active(float sum, int type) {
    switch (type) {
    case 0: return sum;
    case 1: case 2: return 1.0f / (1.0f + expf(-sum));
    case 3: case 5: case 6: case 7: return sum;
    default: die("unkown active type"); return 0.0f;
    }
}
static float cal_grad(float r, float pred, int type) {
    switch (type) {
    case 0: return r - pred;
    case 1: return (r - pred) * pred * (1 - pred);
    case 2: return r - pred;
    case 7: case 3: return r - 1.0f / (1.0f + expf(-pred));
    case 5:
        if (r > 0.5f) return smooth_hinge_grad(pred - 0.5f);
        else return -smooth_hinge_grad(0.5f - pred);
    case 6:
        if (r > 0.5f) { if (pred > 1.0f) return 0.0f; else return r - pred; }
        else { if (pred < 0.0f) return 0.0f; else return r - pred; }
    default: die("unkown active type"); return 0.0f;
    }
}
static float calc_base_score(float base_score, int type) {
    switch (type) {
    case 0: case 6: case 5: return base_score;
    case 1: case 2: case 3: case 7:
        assert_true(base_score > 0.0f && base_score < 1.0f, "sigmoid range constrain");
        return -logf(1.0f / base_score - 1.0f);
    default: die("unkown active type"); return 0.0f;
    }
}

/* ================= PRNG (apex-tensor/apex_random.h:42-77) ================= */
void svdo_seed(unsigned seed) { srand(seed); }
static double next_double2(void) { return ((double)rand() + 1.0) / ((double)RAND_MAX + 2.0); }
static double sample_normal(void) {
    double x, y, s;
    do {
        x = 2 * next_double2() - 1.0;
        y = 2 * next_double2() - 1.0;
        s = x * x + y * y;
    } while (s >= 1.0 || s == 0.0);
    return x * sqrt(-2.0 * log(s) / s);
}
/* apex_tensor_cpu_inline_common.h:249-253 on a row-major sub-matrix */
static void sample_gaussian(float *w, int rows, int cols, int pitch, float sd) {
    for (int y = 0; y < rows; y++)
        for (int x = 0; x < cols; x++)
            w[(size_t)y * pitch + x] = (float)sample_normal() * sd;
}

/* ================= ParameterSet (apex_svd_base.h:33-75) ================= */
static void pset_set_param(param_set *p, const char *name, const char *val) {
    size_t la = strlen(p->prefix_a), lb = strlen(p->prefix_b);
    if (!strncmp(name, p->prefix_a, la)) name += la;
    else if (!strncmp(name, p->prefix_b, lb)) name += lb;
    else return;
    if (!strcmp("bound", name)) {
        unsigned bd = (unsigned)atoi(val);
        assert_true(bd > 0, "can't give 0 as bound");
        assert_true(p->nbound == 0 || p->bound[p->nbound - 1] < bd, "bound must be given in order");
        assert_true(p->nbound + 1 == p->nwd, "must specifiy wd in each range");
        p->bound = (unsigned *)realloc(p->bound, sizeof(unsigned) * (p->nbound + 1));
        p->bound[p->nbound++] = bd - 1;
    }
    if (!strcmp("wd", name)) {
        assert_true(p->nwd == p->nbound, "setting must be exactly");
        p->wd = (float *)realloc(p->wd, sizeof(float) * (p->nwd + 1));
        p->wd[p->nwd++] = (float)atof(val);
    }
}
static float pset_get_wd(const param_set *p, unsigned gid, float wd_default) {
    if (p->nbound == 0) return wd_default;
    int lo = 0, hi = p->nbound; /* std::lower_bound */
    while (lo < hi) { int mid = (lo + hi) / 2; if (p->bound[mid] < gid) lo = mid + 1; else hi = mid; }
    assert_true(lo < p->nbound, "bound set err");
    return p->wd[lo];
}

/* ================= SparseFeatureArray (apex-utils/apex_utils.h:172-195) ================= */
static void sf_load(sparse_feat *f, const char *fname) {
    free(f->row_ptr); free(f->data);
    memset(f, 0, sizeof(*f));
    size_t cap_r = 16, cap_d = 16;
    f->row_ptr = (unsigned *)malloc(sizeof(unsigned) * cap_r);
    f->data = (sf_entry *)malloc(sizeof(sf_entry) * cap_d);
    f->row_ptr[0] = 0;
    FILE *fi = fopen(fname, "r");
    if (!fi) { fprintf(stderr, "can not open file \"%s\"\n", fname); exit(-1); }
    int n;
    while (fscanf(fi, "%d", &n) == 1) {
        if (f->num_row + 2 > cap_r) { cap_r *= 2; f->row_ptr = (unsigned *)realloc(f->row_ptr, sizeof(unsigned) * cap_r); }
        f->row_ptr[f->num_row + 1] = f->row_ptr[f->num_row] + (unsigned)n;
        f->num_row++;
        for (int i = 0; i < n; i++) {
            sf_entry e;
            assert_true(fscanf(fi, "%u:%f", &e.index, &e.value) == 2, "load sparse feature");
            if (f->ndata + 1 > cap_d) { cap_d *= 2; f->data = (sf_entry *)realloc(f->data, sizeof(sf_entry) * cap_d); }
            f->data[f->ndata++] = e;
        }
    }
    fclose(fi);
}
static int sf_get(const sparse_feat *f, unsigned idx, const sf_entry **out) {
    if (idx < f->num_row) { *out = f->data + f->row_ptr[idx]; return (int)(f->row_ptr[idx + 1] - f->row_ptr[idx]); }
    *out = NULL;
    return 0;
}

/* ================= param parsing ================= */
/* apex_svd_model.h:350-368 */
static void tp_set_param(train_param *p, const char *name, const char *val) {
    if (!strcmp("learning_rate", name)) p->learning_rate = (float)atof(val);
    if (!strcmp("wd_user", name)) p->wd_user = (float)atof(val);
    if (!strcmp("wd_item", name)) p->wd_item = (float)atof(val);
    if (!strcmp("wd_uiset", name)) p->wd_user = p->wd_item = (float)atof(val);
    if (!strcmp("wd_user_bias", name)) p->wd_user_bias = (float)atof(val);
    if (!strcmp("wd_item_bias", name)) p->wd_item_bias = (float)atof(val);
    if (!strcmp("wd_uiset_bias", name)) p->wd_user_bias = p->wd_item_bias = (float)atof(val);
    if (!strcmp("wd_global", name)) p->wd_global = (float)atof(val);
    if (!strcmp("reg_method", name)) p->reg_method = atoi(val);
    if (!strcmp("reg_global", name)) p->reg_global = atoi(val);
    if (!strcmp("num_regfree_global", name)) p->num_regfree_global = (unsigned)atoi(val);
    if (!strcmp("decay_learning_rate", name)) p->decay_learning_rate = atoi(val);
    if (!strcmp("min_learning_rate", name)) p->min_learning_rate = (float)atof(val);
    if (!strcmp("decay_rate", name)) p->decay_rate = (float)atof(val);
    if (!strcmp("scale_lr_ufeedback", name)) p->scale_lr_ufeedback = (float)atof(val);
    if (!strcmp("wd_ufeedback", name)) p->wd_ufeedback = (float)atof(val);
    if (!strcmp("wd_ufeedback_bias", name)) p->wd_ufeedback_bias = (float)atof(val);
}
/* apex_svd_model.h:456-476 */
static void mp_set_param(model_param *p, const char *name, const char *val) {
    if (!strcmp("num_user", name)) p->num_user = atoi(val);
    if (!strcmp("num_item", name)) p->num_item = atoi(val);
    if (!strcmp("num_uiset", name)) p->num_user = p->num_item = atoi(val);
    if (!strcmp("num_global", name)) p->num_global = atoi(val);
    if (!strcmp("num_factor", name)) p->num_factor = atoi(val);
    if (!strcmp("u_init_sigma", name)) p->u_init_sigma = (float)atof(val);
    if (!strcmp("i_init_sigma", name)) p->i_init_sigma = (float)atof(val);
    if (!strcmp("ui_init_sigma", name)) p->u_init_sigma = p->i_init_sigma = (float)atof(val);
    if (!strcmp("base_score", name)) p->base_score = (float)atof(val);
    if (!strcmp("no_user_bias", name)) p->no_user_bias = atoi(val);
    if (!strcmp("num_ufeedback", name)) p->num_ufeedback = atoi(val);
    if (!strcmp("num_randinit_ufactor", name)) p->num_randinit_ufactor = atoi(val);
    if (!strcmp("num_randinit_ifactor", name)) p->num_randinit_ifactor = atoi(val);
    if (!strcmp("num_randinit_uifactor", name)) p->num_randinit_ifactor = p->num_randinit_ufactor = atoi(val);
    if (!strcmp("ufeedback_init_sigma", name)) p->ufeedback_init_sigma = (float)atof(val);
    if (!strcmp("common_latent_space", name)) p->common_latent_space = atoi(val);
    if (!strcmp("common_feedback_space", name)) p->common_feedback_space = atoi(val);
    if (!strcmp("user_nonnegative", name)) p->user_nonnegative = atoi(val);
    if (!strcmp("item_nonnegative", name)) p->item_nonnegative = atoi(val);
}

/* ================= lifecycle ================= */
svdo_trainer *svdo_create(int format_type, int active_type, int extend_type, int variant_type) {
    svdo_trainer *t = (svdo_trainer *)calloc(1, sizeof(*t));
    t->mtype[0] = (uint8_t)format_type; t->mtype[1] = (uint8_t)active_type;
    t->mtype[2] = (uint8_t)extend_type; t->mtype[3] = (uint8_t)variant_type;
    /* SVDModelParam() apex_svd_model.h:436-450 */
    t->mp.u_init_sigma = t->mp.i_init_sigma = 0.01f;
    t->mp.base_score = 0.5f;
    /* SVDTrainParam() apex_svd_model.h:334-344 */
    t->tp.learning_rate = 0.01f;
    t->tp.decay_rate = 1.0f;
    t->tp.scale_lr_ufeedback = 1.0f;
    /* SVDFeature() apex_svd_base.h:102-109 */
    t->u_param.prefix_a = "up:"; t->u_param.prefix_b = "uip:";
    t->i_param.prefix_a = "ip:"; t->i_param.prefix_b = "uip:";
    t->g_param.prefix_a = "gp:"; t->g_param.prefix_b = "gp:";
    strcpy(t->name_feat_user, "NULL");
    strcpy(t->name_feat_item, "NULL");
    return t;
}

static void free_model(svdo_trainer *t) {
    if (!t->space_allocated) return;
    free(t->ui_bias); free(t->W_uiset); free(t->g_bias);
    t->ui_bias = t->W_uiset = t->g_bias = NULL;
    t->space_allocated = 0;
}

void svdo_destroy(svdo_trainer *t) {
    if (!t) return;
    free_model(t);
    free(t->tmp_u); free(t->tmp_i); free(t->tmp_fb); free(t->old_fb);
    for (int i = 0; i < t->imfb_alloc; i++) { free(t->imfb[i].tmp); free(t->imfb[i].old); }
    free(t->W_bi);
    free(t->ref_user); if (t->ref_item != t->ref_user) free(t->ref_item); free(t->ref_global);
    free(t->feat_user.row_ptr); free(t->feat_user.data);
    free(t->feat_item.row_ptr); free(t->feat_item.data);
    free(t->u_param.wd); free(t->u_param.bound);
    free(t->i_param.wd); free(t->i_param.bound);
    free(t->g_param.wd); free(t->g_param.bound);
    free(t);
}

/* apex_svd_base.h:126-136 */
void svdo_set_param(svdo_trainer *t, const char *name, const char *val) {
    if (t->mtype[2] == 2 && !strcmp(name, "ufeedback_disable_level")) { /* apex_multi_imfb.h:58-67 */
        int level = atoi(val);
        assert_true(level >= 0 && level < IMFB_MAX, "oracle: ufeedback_disable_level beyond IMFB_MAX");
        t->imfb_disable[level] = 1;
    }
    if (t->mtype[2] == 15) { /* apex_svd_bilinear.h:187-193 */
        if (!strcmp(name, "reg_bi_feedback")) t->reg_bi_feedback = atoi(val);
        if (t->bi_allocated == 0) {
            if (!strcmp(name, "num_bi_feedback")) t->bparam.num_bi_feedback = atoi(val);
            if (!strcmp(name, "start_ufeedback")) t->bparam.start_ufeedback = atoi(val);
        }
    }
    if (!strcmp(name, "feature_user")) strcpy(t->name_feat_user, val);
    if (!strcmp(name, "feature_item")) strcpy(t->name_feat_item, val);
    tp_set_param(&t->tp, name, val);
    pset_set_param(&t->u_param, name, val);
    pset_set_param(&t->i_param, name, val);
    pset_set_param(&t->g_param, name, val);
    if (t->space_allocated == 0) mp_set_param(&t->mp, name, val);
}

/* apex_svd_model.h:511-556 SVDModel::alloc_space */
static void alloc_space(svdo_trainer *t) {
    model_param *p = &t->mp;
    const int user_group = (t->mtype[0] == 1);
    const int ustart = (p->common_feedback_space == 0 && user_group) ? p->num_ufeedback : 0;
    int n;
    if (p->common_latent_space == 0) n = ustart + p->num_user + p->num_item;
    else {
        assert_true(p->num_user == p->num_item, "num_user and num_item must be the same to use common latent space");
        assert_true(p->common_feedback_space != 0, "common latent space must enforce common feedback space");
        n = p->num_item;
    }
    t->n_uiset = n;
    t->pitch = ((p->num_factor * 4 + 15) >> 4) << 2; /* floats */
    t->ui_bias = (float *)calloc((size_t)n + 4, sizeof(float));
    t->W_uiset = (float *)calloc((size_t)n * t->pitch + 4, sizeof(float));
    t->g_bias = (float *)calloc((size_t)p->num_global + 4, sizeof(float));
    if (p->common_latent_space == 0) {
        t->u_bias = t->ui_bias + ustart;
        t->W_user = t->W_uiset + (size_t)ustart * t->pitch;
        t->i_bias = t->ui_bias + ustart + p->num_user;
        t->W_item = t->W_uiset + (size_t)(ustart + p->num_user) * t->pitch;
    } else {
        t->W_user = t->W_uiset + (size_t)ustart * t->pitch;
        t->u_bias = t->ui_bias + ustart;
        t->W_item = t->W_user;
        t->i_bias = t->u_bias;
    }
    t->ufb_bias = NULL; t->W_ufb = NULL;
    if (user_group) {
        if (p->common_feedback_space == 0) { t->ufb_bias = t->ui_bias; t->W_ufb = t->W_uiset; }
        else { t->ufb_bias = t->u_bias; t->W_ufb = t->W_user; }
    }
    t->space_allocated = 1;
}

/* apex_svd_model.h:665-705 SVDModel::rand_init */
static void rand_init(svdo_trainer *t) {
    model_param *p = &t->mp;
    /* ui_bias = 0, g_bias = 0 : calloc */
    memset(t->ui_bias, 0, sizeof(float) * (size_t)t->n_uiset);
    memset(t->g_bias, 0, sizeof(float) * (size_t)p->num_global);
    p->base_score = calc_base_score(p->base_score, t->mtype[1]);
    {
        int rows = p->num_randinit_ufactor != 0 ? p->num_randinit_ufactor : p->num_user;
        sample_gaussian(t->W_user, rows, p->num_factor, t->pitch, p->u_init_sigma);
        if (p->user_nonnegative)
            for (int y = 0; y < p->num_user; y++)
                for (int x = 0; x < p->num_factor; x++) {
                    float *w = &t->W_user[(size_t)y * t->pitch + x];
                    *w = fabsf(*w);
                }
    }
    if (p->common_latent_space == 0) {
        int rows = p->num_randinit_ifactor != 0 ? p->num_randinit_ifactor : p->num_item;
        sample_gaussian(t->W_item, rows, p->num_factor, t->pitch, p->i_init_sigma);
        if (p->item_nonnegative)
            for (int y = 0; y < rows; y++)
                for (int x = 0; x < p->num_factor; x++) {
                    float *w = &t->W_item[(size_t)y * t->pitch + x];
                    *w = fabsf(*w);
                }
    }
    if (t->mtype[0] == 1) {
        /* note: draws are consumed even when ufeedback_init_sigma == 0 (apex_svd_model.h:702-704) */
        int rows = p->common_feedback_space == 0 ? p->num_ufeedback : p->num_user;
        sample_gaussian(t->W_ufb, rows, p->num_factor, t->pitch, p->ufeedback_init_sigma);
    }
}

static void bi_alloc(svdo_trainer *t) { /* BModel::alloc_space apex_svd_bilinear.h:49-54: W_bi[num_item][num_bi_feedback] = 0 */
    free(t->W_bi);
    t->W_bi = (float *)calloc((size_t)t->mp.num_item * (size_t)(t->bparam.num_bi_feedback > 0 ? t->bparam.num_bi_feedback : 0) + 1, sizeof(float));
    t->bi_allocated = 1;
}
void svdo_init_model(svdo_trainer *t) { /* apex_svd_base.h:146-149 */
    alloc_space(t);
    rand_init(t);
    if (t->mtype[2] == 15) bi_alloc(t); /* apex_svd_bilinear.h:202-205 */
}

/* apex_svd_base.h:151-173 (+ :499-503 for the user-group trainer) */
void svdo_init_trainer(svdo_trainer *t) {
    if (strcmp(t->name_feat_user, "NULL")) sf_load(&t->feat_user, t->name_feat_user);
    if (strcmp(t->name_feat_item, "NULL")) sf_load(&t->feat_item, t->name_feat_item);
    size_t nb = sizeof(float) * (size_t)(t->pitch + 4);
    t->tmp_u = (float *)calloc(1, nb); t->tmp_i = (float *)calloc(1, nb);
    t->tmp_fb = (float *)calloc(1, nb); t->old_fb = (float *)calloc(1, nb);
    t->sample_counter = 0;
    if (t->tp.reg_global >= 4) t->ref_global = (unsigned *)calloc((size_t)t->mp.num_global + 1, sizeof(unsigned));
    if (t->tp.reg_method >= 4) {
        t->ref_user = (unsigned *)calloc((size_t)t->mp.num_user + 1, sizeof(unsigned));
        if (t->mp.common_latent_space == 0) t->ref_item = (unsigned *)calloc((size_t)t->mp.num_item + 1, sizeof(unsigned));
        else t->ref_item = t->ref_user;
    }
    t->init_end = 1;
}

void svdo_set_round(svdo_trainer *t, int nround) { /* apex_svd_base.h:470-478 */
    if (t->tp.decay_learning_rate != 0) {
        assert_true(t->round_counter <= nround, "round counter restriction");
        while (t->round_counter < nround) {
            t->tp.learning_rate *= t->tp.decay_rate;
            t->round_counter++;
        }
    }
}
void svdo_finish_round(svdo_trainer *t) { (void)t; } /* apex_svd.h:78 default no-op */

/* ================= model file (apex_svd_model.h:570-660, tensor serialisation
 * apex_tensor_cpu_inline_common.h:72-87: int header x_max[,y_max] then unpadded rows) ======== */
static void save_1d(FILE *fo, const float *v, int n) {
    fwrite(&n, sizeof(int), 1, fo);
    fwrite(v, sizeof(float), (size_t)n, fo);
}
static void save_2d(FILE *fo, const float *w, int rows, int cols, int pitch) {
    int hdr[2] = { cols, rows };
    fwrite(hdr, sizeof(int), 2, fo);
    for (int y = 0; y < rows; y++) fwrite(w + (size_t)y * pitch, sizeof(float), (size_t)cols, fo);
}
static void load_1d(FILE *fi, float *v, int n) {
    int x;
    assert_true(fread(&x, sizeof(int), 1, fi) > 0, "tensor::load_from_file");
    assert_true(x == n, "tensor shape mismatch");
    if (n > 0) assert_true(fread(v, sizeof(float), (size_t)n, fi) > 0, "tensor::load_from_file");
}
static void load_2d(FILE *fi, float *w, int rows, int cols, int pitch) {
    int hdr[2];
    assert_true(fread(hdr, sizeof(int), 2, fi) > 0, "tensor::load_from_file");
    assert_true(hdr[0] == cols && hdr[1] == rows, "tensor shape mismatch");
    for (int y = 0; y < rows; y++)
        if (cols > 0) assert_true(fread(w + (size_t)y * pitch, sizeof(float), (size_t)cols, fi) > 0, "tensor::load_from_file");
}
static void save_model(svdo_trainer *t, FILE *fo) {
    model_param *p = &t->mp;
    fwrite(p, sizeof(model_param), 1, fo);
    if (p->common_latent_space == 0) {
        save_1d(fo, t->u_bias, p->num_user);
        save_2d(fo, t->W_user, p->num_user, p->num_factor, t->pitch);
        save_1d(fo, t->i_bias, p->num_item);
        save_2d(fo, t->W_item, p->num_item, p->num_factor, t->pitch);
    } else {
        save_1d(fo, t->ui_bias, t->n_uiset);
        save_2d(fo, t->W_uiset, t->n_uiset, p->num_factor, t->pitch);
    }
    save_1d(fo, t->g_bias, p->num_global);
    if (t->mtype[0] == 1 && p->common_feedback_space == 0) {
        save_1d(fo, t->ufb_bias, p->num_ufeedback);
        save_2d(fo, t->W_ufb, p->num_ufeedback, p->num_factor, t->pitch);
    }
}
static void load_model(svdo_trainer *t, FILE *fi) {
    if (fread(&t->mp, sizeof(model_param), 1, fi) == 0) die("error loading CF SVD model");
    if (t->space_allocated) free_model(t);
    alloc_space(t);
    model_param *p = &t->mp;
    if (p->common_latent_space == 0) {
        load_1d(fi, t->u_bias, p->num_user);
        load_2d(fi, t->W_user, p->num_user, p->num_factor, t->pitch);
        load_1d(fi, t->i_bias, p->num_item);
        load_2d(fi, t->W_item, p->num_item, p->num_factor, t->pitch);
    } else {
        load_1d(fi, t->ui_bias, t->n_uiset);
        load_2d(fi, t->W_uiset, t->n_uiset, p->num_factor, t->pitch);
    }
    load_1d(fi, t->g_bias, p->num_global);
    if (t->mtype[0] == 1 && p->common_feedback_space == 0) {
        load_1d(fi, t->ufb_bias, p->num_ufeedback);
        load_2d(fi, t->W_ufb, p->num_ufeedback, p->num_factor, t->pitch);
    }
}
/* BModel::save_to_file / load_from_file (apex_svd_bilinear.h:60-68): BParam, then W_bi as a 2D tensor.  Training never
 * changes W_bi: SVDPPFeature::update calls ITS OWN non-virtual prepare_ufeedback (apex_svd_base.h:523,571), so the derived
 * class's version that would fill up_index is never reached and get_bias_plugin / update_bias_plugin loop over nothing. */
static void bi_save(svdo_trainer *t, FILE *fo) {
    fwrite(&t->bparam, sizeof(t->bparam), 1, fo);
    save_2d(fo, t->W_bi, t->mp.num_item, t->bparam.num_bi_feedback, t->bparam.num_bi_feedback);
}
static void bi_load(svdo_trainer *t, FILE *fi) {
    assert_true(fread(&t->bparam, sizeof(t->bparam), 1, fi) > 0, "load from file");
    if (t->bi_allocated == 0) bi_alloc(t);
    load_2d(fi, t->W_bi, t->mp.num_item, t->bparam.num_bi_feedback, t->bparam.num_bi_feedback);
}
int svdo_save_model_path(svdo_trainer *t, const char *path, int with_type_header) {
    FILE *fo = fopen(path, "wb");
    if (!fo) return -1;
    if (with_type_header) fwrite(t->mtype, 1, 4, fo);
    save_model(t, fo);
    if (t->mtype[2] == 15) bi_save(t, fo);
    fclose(fo);
    return 0;
}
int svdo_load_model_path(svdo_trainer *t, const char *path, int with_type_header) {
    FILE *fi = fopen(path, "rb");
    if (!fi) return -1;
    if (with_type_header) assert_true(fread(t->mtype, 1, 4, fi) == 4, "loading model");
    load_model(t, fi);
    if (t->mtype[2] == 15) bi_load(t, fi);
    fclose(fi);
    return 0;
}

/* ================= the hot path ================= */
typedef struct {
    float label;
    int ng, nu, ni;
    const unsigned *ig, *iu, *ii;
    const float *vg, *vu, *vi;
} elem; /* SVDFeatureCSR::Elem apex_svd_data.h:38-107 */

static elem make_elem(float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    elem e; /* Elem::set_space apex_svd_data.h:71-78 */
    e.label = label; e.ng = ng; e.nu = nu; e.ni = ni;
    e.ig = index; e.iu = index + ng; e.ii = index + ng + nu;
    e.vg = value; e.vu = value + ng; e.vi = value + ng + nu;
    return e;
}

/* apex_svd_base.h:188-210 */
static void reg_global(svdo_trainer *t, unsigned gid) {
    float lambda = t->tp.learning_rate * pset_get_wd(&t->g_param, gid, t->tp.wd_global);
    if (gid >= t->tp.num_regfree_global) {
        switch (t->tp.reg_global) {
        case 0: t->g_bias[gid] *= (1.0f - lambda); break;
        case 1: reg_l1_scalar(&t->g_bias[gid], lambda); break;
        case 4: {
            float k = (float)(t->ref_global[gid] - t->sample_counter);
            t->g_bias[gid] *= expf(logf(1.0f - lambda) * k);
            t->ref_global[gid] = t->sample_counter;
            break;
        }
        case 5: {
            float k = (float)(t->ref_global[gid] - t->sample_counter);
            reg_l1_scalar(&t->g_bias[gid], lambda * k);
            t->ref_global[gid] = t->sample_counter;
            break;
        }
        default: die("unknown global decay method");
        }
    }
}
/* apex_svd_base.h:211-250 */
static void reg_user(svdo_trainer *t, unsigned uid) {
    const int k = t->mp.num_factor;
    float *w = t->W_user + (size_t)uid * t->pitch;
    float wd = pset_get_wd(&t->u_param, uid, t->tp.wd_user);
    float lambda = t->tp.learning_rate * wd;
    switch (t->tp.reg_method) {
    case 0: scale(w, (float)(double)(1.0f - lambda), k); break;
    case 3: case 1: regularize_l1(w, lambda, k); break;
    case 2: project(w, wd, k); break;
    case 4: {
        float kk = (float)(t->ref_user[uid] - t->sample_counter);
        scale(w, (float)(double)expf(logf(1.0f - lambda) * kk), k);
        t->ref_user[uid] = t->sample_counter;
        break;
    }
    case 5: {
        float kk = (float)(t->ref_user[uid] - t->sample_counter);
        regularize_l1(w, lambda * kk, k);
        t->ref_user[uid] = t->sample_counter;
        break;
    }
    default: die("unknown reg_method");
    }
    if (t->mp.user_nonnegative) /* K7 apex_tensor_cpu_inline_common.h:177-181 */
        for (int j = 0; j < k; j++) if (w[j] <= 0.0f) w[j] = 0.0f;
    if (t->mp.no_user_bias == 0)
        t->u_bias[uid] *= (1.0f - t->tp.learning_rate * t->tp.wd_user_bias);
}
/* apex_svd_base.h:251-283 */
static void reg_item(svdo_trainer *t, unsigned iid) {
    const int k = t->mp.num_factor;
    float *w = t->W_item + (size_t)iid * t->pitch;
    float wd = pset_get_wd(&t->i_param, iid, t->tp.wd_item);
    float lambda = t->tp.learning_rate * wd;
    switch (t->tp.reg_method) {
    case 3: case 0: scale(w, (float)(double)(1.0f - lambda), k); break;
    case 1: regularize_l1(w, lambda, k); break;
    case 2: project(w, wd, k); break;
    case 4: {
        float kk = (float)(t->ref_item[iid] - t->sample_counter);
        scale(w, (float)(double)expf(logf(1.0f - lambda) * kk), k);
        t->ref_item[iid] = t->sample_counter;
        break;
    }
    case 5: {
        float kk = (float)(t->ref_item[iid] - t->sample_counter);
        regularize_l1(w, lambda * kk, k);
        t->ref_item[iid] = t->sample_counter;
        break;
    }
    default: die("unknown reg_method");
    }
    t->i_bias[iid] *= (1.0f - t->tp.learning_rate * t->tp.wd_item_bias);
}
/* apex_svd_base.h:286-311 */
static void regularize(svdo_trainer *t, const elem *f, int is_after_update) {
    const sf_entry *vec;
    if ((is_after_update && t->tp.reg_global < 4) || (!is_after_update && t->tp.reg_global >= 4))
        for (int i = 0; i < f->ng; i++) reg_global(t, f->ig[i]);
    if ((is_after_update && t->tp.reg_method < 4) || (!is_after_update && t->tp.reg_method >= 4)) {
        for (int i = 0; i < f->nu; i++) {
            reg_user(t, f->iu[i]);
            int n = sf_get(&t->feat_user, f->iu[i], &vec);
            for (int j = 0; j < n; j++) reg_user(t, vec[j].index);
        }
        for (int i = 0; i < f->ni; i++) {
            reg_item(t, f->ii[i]);
            int n = sf_get(&t->feat_item, f->ii[i], &vec);
            for (int j = 0; j < n; j++) reg_item(t, vec[j].index);
        }
    }
}
static int is_user_group(const svdo_trainer *t) { return t->mtype[0] == 1; }

/* apex_svd_base.h:313-353 */
static double calc_bias(svdo_trainer *t, const elem *f) {
    const sf_entry *vec;
    double sum = 0.0f;
    for (int i = 0; i < f->ng; i++) {
        unsigned gid = f->ig[i];
        assert_true(gid < (unsigned)t->mp.num_global, "global feature index exceed setting");
        sum += f->vg[i] * t->g_bias[gid];
    }
    if (t->mp.no_user_bias == 0) {
        for (int i = 0; i < f->nu; i++) {
            unsigned uid = f->iu[i];
            assert_true(uid < (unsigned)t->mp.num_user, "user feature index exceed bound");
            sum += f->vu[i] * t->u_bias[uid];
            int n = sf_get(&t->feat_user, uid, &vec);
            for (int j = 0; j < n; j++) sum += t->u_bias[vec[j].index] * vec[j].value;
        }
        if (t->mtype[2] == 2) { /* apex_multi_imfb.h:80-86 */
            float s2 = 0.0f;
            for (int i = 0; i < t->imfb_top; i++) s2 += t->imfb[i].tmp_bias;
            sum += s2;
        } else
        sum += is_user_group(t) ? t->tmp_fb_bias : 0.0f; /* get_bias_svdpp :433-435,509-511 */
    }
    sum += 0.0f; /* get_bias_plugin :436-438 */
    for (int i = 0; i < f->ni; i++) {
        unsigned iid = f->ii[i];
        float ival = f->vi[i];
        assert_true(iid < (unsigned)t->mp.num_item, "item feature index exceed bound");
        sum += ival * t->i_bias[iid];
        int n = sf_get(&t->feat_item, iid, &vec);
        for (int j = 0; j < n; j++) sum += t->i_bias[vec[j].index] * vec[j].value * ival;
    }
    return sum;
}
/* apex_svd_base.h:354-381 */
static void prepare_tmp(svdo_trainer *t, const elem *f) {
    const int k = t->mp.num_factor;
    const sf_entry *vec;
    if (t->mtype[2] == 2) { /* apex_multi_imfb.h:70-79 */
        if (t->imfb_top == 0) for (int j = 0; j < k; j++) t->tmp_u[j] = 0.0f;
        else memcpy(t->tmp_u, t->imfb[0].tmp, sizeof(float) * (size_t)k);
        for (int i = 1; i < t->imfb_top; i++) for (int j = 0; j < k; j++) t->tmp_u[j] = t->tmp_u[j] + t->imfb[i].tmp[j];
    } else
    if (is_user_group(t)) memcpy(t->tmp_u, t->tmp_fb, sizeof(float) * (size_t)k); /* :506-508 */
    else for (int j = 0; j < k; j++) t->tmp_u[j] = 0.0f;                          /* :430-432 */
    for (int j = 0; j < k; j++) t->tmp_i[j] = 0.0f;
    for (int i = 0; i < f->nu; i++) {
        unsigned uid = f->iu[i];
        assert_true(uid < (unsigned)t->mp.num_user, "user feature index exceed bound");
        axpy(t->tmp_u, t->W_user + (size_t)uid * t->pitch, (float)(double)f->vu[i], k);
        int n = sf_get(&t->feat_user, uid, &vec);
        for (int j = 0; j < n; j++)
            axpy(t->tmp_u, t->W_user + (size_t)vec[j].index * t->pitch, (float)(double)vec[j].value, k);
    }
    for (int i = 0; i < f->ni; i++) {
        unsigned iid = f->ii[i];
        float ival = f->vi[i];
        axpy(t->tmp_i, t->W_item + (size_t)iid * t->pitch, (float)(double)ival, k);
        int n = sf_get(&t->feat_item, iid, &vec);
        for (int j = 0; j < n; j++) /* scalar product formed in double: apex_exp_template.h:500-502 */
            axpy(t->tmp_i, t->W_item + (size_t)vec[j].index * t->pitch, (float)((double)vec[j].value * (double)ival), k);
    }
}
/* apex_svd_base.h:445-454 */
static float pred(svdo_trainer *t, const elem *f) {
    double sum = t->mp.base_score + calc_bias(t, f);
    prepare_tmp(t, f);
    sum += sdot(t->tmp_u, t->tmp_i, t->mp.num_factor);
    return map_active((float)sum, t->mtype[1]);
}
/* apex_svd_base.h:512-520 */
static void update_svdpp(svdo_trainer *t, float err) {
    const int k = t->mp.num_factor;
    float lr = t->tp.learning_rate * t->tp.scale_lr_ufeedback;
    if (t->mtype[2] == 2) { /* apex_multi_imfb.h:87-98 */
        for (int i = 0; i < t->imfb_top; i++) {
            if (t->imfb_disable[i] || t->imfb[i].num_ufeedback == 0) continue;
            axpy(t->imfb[i].tmp, t->tmp_i, (float)(double)(lr * err * t->imfb[i].norm), k);
            scale(t->imfb[i].tmp, (float)(double)(1.0f - lr * t->tp.wd_ufeedback), k);
            if (t->mp.no_user_bias == 0) {
                t->imfb[i].tmp_bias += lr * err * t->imfb[i].norm;
                t->imfb[i].tmp_bias *= (1.0f - lr * t->tp.wd_ufeedback_bias);
            }
        }
        return;
    }
    axpy(t->tmp_fb, t->tmp_i, (float)(double)(lr * err * t->norm_fb), k);
    scale(t->tmp_fb, (float)(double)(1.0f - lr * t->tp.wd_ufeedback), k);
    if (t->mp.no_user_bias == 0) {
        t->tmp_fb_bias += lr * err * t->norm_fb;
        t->tmp_fb_bias *= (1.0f - lr * t->tp.wd_ufeedback_bias);
    }
}
/* apex_svd_base.h:383-427 */
static void update_no_decay(svdo_trainer *t, float err, const elem *f) {
    const int k = t->mp.num_factor;
    const float lr = t->tp.learning_rate;
    const sf_entry *vec;
    for (int i = 0; i < f->ng; i++) t->g_bias[f->ig[i]] += lr * err * f->vg[i];
    for (int i = 0; i < f->nu; i++) {
        unsigned uid = f->iu[i];
        float sc = lr * err * f->vu[i];
        axpy(t->W_user + (size_t)uid * t->pitch, t->tmp_i, sc, k);
        if (t->mp.no_user_bias == 0) t->u_bias[uid] += sc;
        int n = sf_get(&t->feat_user, uid, &vec);
        for (int j = 0; j < n; j++) {
            float s2 = lr * err * vec[j].value;
            axpy(t->W_user + (size_t)vec[j].index * t->pitch, t->tmp_i, s2, k);
            if (t->mp.no_user_bias == 0) t->u_bias[vec[j].index] += s2;
        }
    }
    for (int i = 0; i < f->ni; i++) {
        unsigned iid = f->ii[i];
        float ival = f->vi[i];
        float sc = lr * err * ival;
        axpy(t->W_item + (size_t)iid * t->pitch, t->tmp_u, sc, k);
        t->i_bias[iid] += sc;
        int n = sf_get(&t->feat_item, iid, &vec);
        for (int j = 0; j < n; j++) {
            float s2 = lr * err * vec[j].value * ival;
            t->i_bias[vec[j].index] += s2;
            axpy(t->W_item + (size_t)vec[j].index * t->pitch, t->tmp_u, s2, k);
        }
    }
    if (is_user_group(t) || t->mtype[2] == 2) update_svdpp(t, err);
    /* update_bias_plugin :439-440 no-op; bilinear (apex_svd_bilinear.h:148-162): nothing but reg_feedback(lr, iid) per item entry */
    if (t->mtype[2] == 15 && f->ni > 0)
        assert_true(t->reg_bi_feedback >= 0 && t->reg_bi_feedback <= 5, "unknown bi feedback decay method");
}
/* apex_svd_base.h:456-462 */
static void update_inner(svdo_trainer *t, const elem *f) {
    regularize(t, f, 0);
    float err = cal_grad(f->label, pred(t, f), t->mtype[1]) * 1.0f;
    update_no_decay(t, err, f);
    t->sample_counter++;
    regularize(t, f, 1);
}

void svdo_update_csr(svdo_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    elem e = make_elem(label, ng, nu, ni, index, value);
    update_inner(t, &e);
}
float svdo_predict_csr(svdo_trainer *t, float label, int ng, int nu, int ni, const unsigned *index, const float *value) {
    elem e = make_elem(label, ng, nu, ni, index, value);
    return pred(t, &e);
}
/* SVDFeatureCSR::operator[] apex_svd_data.h:129-142 */
static elem csr_row(int r, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value) {
    elem e;
    e.label = row_label[r];
    e.ng = row_ptr[r * 3 + 1] - row_ptr[r * 3 + 0];
    e.nu = row_ptr[r * 3 + 2] - row_ptr[r * 3 + 1];
    e.ni = row_ptr[r * 3 + 3] - row_ptr[r * 3 + 2];
    e.ig = feat_index + row_ptr[r * 3 + 0]; e.iu = feat_index + row_ptr[r * 3 + 1]; e.ii = feat_index + row_ptr[r * 3 + 2];
    e.vg = feat_value + row_ptr[r * 3 + 0]; e.vu = feat_value + row_ptr[r * 3 + 1]; e.vi = feat_value + row_ptr[r * 3 + 2];
    return e;
}
void svdo_update_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value) {
    for (int r = 0; r < num_row; r++) {
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        update_inner(t, &e);
    }
}
/* NOT a reference function -- the "window minibatch" step of the multi-GPU design restated on the CPU so
 * that the HIP kernels (svdf_k_window.hip) can be checked bit for bit.  Every instance IS the reference's update_inner
 * (apex_svd_base.h:456-462), applied to (the current user side, the WINDOW-START replicated side): whatever the instance
 * changed on the replicated side -- W_item / i_bias rows of its item entries, g_bias of its global entries -- is put back
 * afterwards, and the change is ADDED, instance after instance in file order, to the caller's delta arrays instead
 * (dW_item: num_item x num_factor unpadded, di_bias: num_item, dg_bias: num_global).  The user side therefore sees exact
 * sequential SGD, the replicated side one minibatch step per window: replicated += sum of the deltas of all ranks.
 * Returns 0, or -1 for configurations whose per-instance step has side effects outside those rows (lazy decay, side tables on
 * the item side, user-group trainers). */
/* `amd:contrib = bf16` of the HIP engine: a row contribution is stored as bfloat16 (round to nearest even) and summed in fp32 */
static float bf16_round(float x) {
    union { float f; unsigned u; } v;
    v.f = x;
    v.u = ((v.u + 0x7FFFu + ((v.u >> 16) & 1u)) >> 16) << 16;
    return v.f;
}
void svdo_set_stale_rounding(svdo_trainer *t, int bf16) { t->stale_bf16 = bf16 != 0; }
/* one row of the checker step: update_inner on (current private side, window-start replicated side); the change of the row's item
 * rows / biases and global biases goes to the delta arrays, the rows themselves are put back */
typedef struct { float *buf; size_t cap; } stale_save;
static void stale_row(svdo_trainer *t, const elem *e, stale_save *sv, float *dW_item, float *di_bias, float *dg_bias) {
    const int k = t->mp.num_factor;
    const size_t need = (size_t)e->ni + (size_t)e->ng;
    if (need > sv->cap || sv->buf == NULL) { sv->cap = need * 2 + 64; sv->buf = (float *)realloc(sv->buf, sizeof(float) * (size_t)(k + 1) * sv->cap); }
    float *save = sv->buf;
    for (int i = 0; i < e->ni; i++) {
        assert_true(e->ii[i] < (unsigned)t->mp.num_item, "item feature index exceed bound");
        memcpy(save + (size_t)i * (k + 1), t->W_item + (size_t)e->ii[i] * t->pitch, sizeof(float) * (size_t)k);
        save[(size_t)i * (k + 1) + k] = t->i_bias[e->ii[i]];
    }
    float *gsave = save + (size_t)e->ni * (k + 1);
    for (int i = 0; i < e->ng; i++) {
        assert_true(e->ig[i] < (unsigned)t->mp.num_global, "global feature index exceed setting");
        gsave[i] = t->g_bias[e->ig[i]];
    }
    update_inner(t, e);
    for (int i = 0; i < e->ni; i++) {   /* an id listed twice: the first entry carries the whole change, the second sees none */
        float *w = t->W_item + (size_t)e->ii[i] * t->pitch, *d = dW_item + (size_t)e->ii[i] * k;
        const float *s = save + (size_t)i * (k + 1);
        for (int j = 0; j < k; j++) { float c = w[j] - s[j]; if (t->stale_bf16) c = bf16_round(c); d[j] = d[j] + c; w[j] = s[j]; }
        float cb = t->i_bias[e->ii[i]] - s[k];
        di_bias[e->ii[i]] = di_bias[e->ii[i]] + cb;
        t->i_bias[e->ii[i]] = s[k];
    }
    for (int i = 0; i < e->ng; i++) {
        float c = t->g_bias[e->ig[i]] - gsave[i];
        dg_bias[e->ig[i]] = dg_bias[e->ig[i]] + c;
        t->g_bias[e->ig[i]] = gsave[i];
    }
}
static int stale_supported(const svdo_trainer *t) {
    return !(t->tp.reg_method >= 4 || t->tp.reg_global >= 4 || t->feat_item.num_row > 0 || t->feat_user.num_row > 0 || t->mtype[2] != 0 ||
             t->mp.common_latent_space != 0);
}
int svdo_update_csr_batch_stale(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                                const unsigned *feat_index, const float *feat_value, float *dW_item, float *di_bias, float *dg_bias) {
    if (!stale_supported(t) || is_user_group(t)) return -1;
    stale_save sv = {NULL, 0};
    for (int r = 0; r < num_row; r++) {
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        stale_row(t, &e, &sv, dW_item, di_bias, dg_bias);
    }
    free(sv.buf);
    return 0;
}
/* ---- one window of the one-GPU window step with ORDERED SUB-STEPS on the item side (HIP engine round 6: svdf_k_window.hip, k_window_apply;
 * `amd:step = minibatch / auto` on one GPU).  Not the reference's semantics either -- the same checker role as svdo_update_csr_batch_stale:
 *   * user side: exactly as the stale step -- every row is update_inner (apex_svd_base.h:456-462) on (the user's current row, the item row as
 *     it was at the WINDOW START); the item row is put back after every row;
 *   * item side: item i's rows of the window, in file order, are applied in sub-steps of at most `sub` rows.  Inside a sub-step every row's
 *     item-side change is what update_inner gives on (the user's row and bias as they were just BEFORE that row's own update in the user
 *     walk, the item's row as the PREVIOUS sub-step left it); the changes are added up in file order (acc = 0 + c_1 + c_2 ...) and the sum is
 *     added to the item's row.  An item with at most `sub` rows in the window moves exactly as in svdo_update_csr_batch_stale followed by
 *     `W_item += delta`; a hot item no longer collects thousands of changes computed against one stale value (which diverges), it takes
 *     count / sub minibatch steps in order.
 * Rows must be (no global entry, one user entry, one item entry).  Returns 0, -1 for unsupported configurations / rows. */
int svdo_update_window_substeps(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                                const unsigned *feat_index, const float *feat_value, int sub) {
    if (!stale_supported(t) || is_user_group(t) || sub < 1) return -1;
    const int k = t->mp.num_factor;
    for (int r = 0; r < num_row; r++) {
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        if (e.ng != 0 || e.nu != 1 || e.ni != 1) return -1;
    }
    float *pre = (float *)malloc(sizeof(float) * (size_t)(k + 1) * (size_t)(num_row > 0 ? num_row : 1));   /* the user's row + bias before row r's update */
    float *save = (float *)malloc(sizeof(float) * (size_t)(k + 1) * 2);
    /* user walks against the window-start item side */
    for (int r = 0; r < num_row; r++) {
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        assert_true(e.ii[0] < (unsigned)t->mp.num_item, "item feature index exceed bound");
        assert_true(e.iu[0] < (unsigned)t->mp.num_user, "user feature index exceed bound");
        float *wi = t->W_item + (size_t)e.ii[0] * t->pitch, *wu = t->W_user + (size_t)e.iu[0] * t->pitch;
        memcpy(pre + (size_t)r * (k + 1), wu, sizeof(float) * (size_t)k);
        pre[(size_t)r * (k + 1) + k] = t->u_bias[e.iu[0]];
        memcpy(save, wi, sizeof(float) * (size_t)k);
        save[k] = t->i_bias[e.ii[0]];
        update_inner(t, &e);
        memcpy(wi, save, sizeof(float) * (size_t)k);
        t->i_bias[e.ii[0]] = save[k];
    }
    /* rows grouped by item, file order inside an item (counting sort) */
    int *iptr = (int *)calloc((size_t)t->mp.num_item + 1, sizeof(int));
    int *rows = (int *)malloc(sizeof(int) * (size_t)(num_row > 0 ? num_row : 1));
    for (int r = 0; r < num_row; r++) iptr[feat_index[row_ptr[3 * r + 2]] + 1]++;
    for (int i = 0; i < t->mp.num_item; i++) iptr[i + 1] += iptr[i];
    {
        int *cur = (int *)malloc(sizeof(int) * (size_t)(t->mp.num_item + 1));
        memcpy(cur, iptr, sizeof(int) * (size_t)(t->mp.num_item + 1));
        for (int r = 0; r < num_row; r++) rows[cur[feat_index[row_ptr[3 * r + 2]]]++] = r;
        free(cur);
    }
    float *acc = (float *)malloc(sizeof(float) * (size_t)(k + 1));
    float *usave = save + (k + 1);
    for (int i = 0; i < t->mp.num_item; i++) {
        float *wi = t->W_item + (size_t)i * t->pitch;
        for (int c0 = iptr[i]; c0 < iptr[i + 1]; c0 += sub) {
            const int c1 = c0 + sub < iptr[i + 1] ? c0 + sub : iptr[i + 1];
            for (int j = 0; j <= k; j++) acc[j] = 0.0f;
            for (int s = c0; s < c1; s++) {
                const int r = rows[s];
                elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
                float *wu = t->W_user + (size_t)e.iu[0] * t->pitch;
                memcpy(usave, wu, sizeof(float) * (size_t)k);                 /* the user's true row: put back below */
                usave[k] = t->u_bias[e.iu[0]];
                memcpy(wu, pre + (size_t)r * (k + 1), sizeof(float) * (size_t)k);
                t->u_bias[e.iu[0]] = pre[(size_t)r * (k + 1) + k];
                memcpy(save, wi, sizeof(float) * (size_t)k);
                save[k] = t->i_bias[i];
                update_inner(t, &e);
                for (int j = 0; j < k; j++) { float c = wi[j] - save[j]; acc[j] = acc[j] + c; wi[j] = save[j]; }
                { float cb = t->i_bias[i] - save[k]; acc[k] = acc[k] + cb; t->i_bias[i] = save[k]; }
                memcpy(wu, usave, sizeof(float) * (size_t)k);
                t->u_bias[e.iu[0]] = usave[k];
            }
            for (int j = 0; j < k; j++) wi[j] = wi[j] + acc[j];
            t->i_bias[i] = t->i_bias[i] + acc[k];
        }
    }
    free(acc); free(rows); free(iptr); free(save); free(pre);
    return 0;
}
void svdo_predict_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                            const unsigned *feat_index, const float *feat_value, float *out) {
    for (int r = 0; r < num_row; r++) {
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        out[r] = pred(t, &e);
    }
}

/* ---- SVD++ user-group path, apex_svd_base.h:523-591 ---- */
static void prepare_ufeedback(svdo_trainer *t, int nfb, const unsigned *idx, const float *val) {
    const int k = t->mp.num_factor;
    t->norm_fb = 0.0f;
    for (int j = 0; j < k; j++) t->tmp_fb[j] = 0.0f;
    t->tmp_fb_bias = 0.0f;
    for (int i = 0; i < nfb; i++) {
        unsigned fid = idx[i];
        float v = val[i];
        assert_true(fid < (unsigned)t->mp.num_ufeedback, "ufeedback id exceed bound");
        axpy(t->tmp_fb, t->W_ufb + (size_t)fid * t->pitch, (float)(double)v, k);
        t->norm_fb += v * v;
        if (t->mp.no_user_bias == 0) t->tmp_fb_bias += t->ufb_bias[fid] * v;
    }
}
static void update_ufeedback(svdo_trainer *t, int nfb, const unsigned *idx, const float *val) {
    const int k = t->mp.num_factor;
    if (nfb == 0) return;
    for (int j = 0; j < k; j++) t->tmp_fb[j] = t->tmp_fb[j] - t->old_fb[j]; /* K5 */
    t->tmp_fb_bias -= t->old_fb_bias;
    scale(t->tmp_fb, (float)(double)(1.0f / t->norm_fb), k);
    t->tmp_fb_bias *= 1.0f / t->norm_fb;
    for (int i = 0; i < nfb; i++) {
        unsigned fid = idx[i];
        float v = val[i];
        axpy(t->W_ufb + (size_t)fid * t->pitch, t->tmp_fb, (float)(double)v, k);
        if (t->mp.no_user_bias == 0) t->ufb_bias[fid] += t->tmp_fb_bias * v;
    }
}
/* ---- extend_type 2, apex_multi_imfb.h:121-195 ---- */
static void imfb_push(svdo_trainer *t, int nfb, const unsigned *idx, const float *val) { /* push_ufeedback :161-171 + prepare_ufeedback :121-137 */
    const int k = t->mp.num_factor;
    assert_true(t->imfb_top < IMFB_MAX, "oracle: more nested implicit-feedback levels than IMFB_MAX");
    if (t->imfb_top == t->imfb_alloc) {
        t->imfb[t->imfb_alloc].tmp = (float *)calloc((size_t)t->pitch + 4, sizeof(float));
        t->imfb[t->imfb_alloc].old = (float *)calloc((size_t)t->pitch + 4, sizeof(float));
        t->imfb_alloc++;
    }
    float *tmp = t->imfb[t->imfb_top].tmp;
    float norm = 0.0f, bias = 0.0f;
    for (int j = 0; j < k; j++) tmp[j] = 0.0f;
    for (int i = 0; i < nfb; i++) {
        unsigned fid = idx[i];
        float v = val[i];
        assert_true(fid < (unsigned)t->mp.num_ufeedback, "ufeedback id exceed bound");
        axpy(tmp, t->W_ufb + (size_t)fid * t->pitch, (float)(double)v, k);
        norm += v * v;
        if (t->mp.no_user_bias == 0) bias += t->ufb_bias[fid] * v;
    }
    t->imfb[t->imfb_top].norm = norm;
    t->imfb[t->imfb_top].tmp_bias = bias;
    t->imfb[t->imfb_top].old_bias = bias;
    t->imfb[t->imfb_top].num_ufeedback = nfb;
    memcpy(t->imfb[t->imfb_top].old, tmp, sizeof(float) * (size_t)k);
    t->imfb_top++;
}
static void imfb_scatter(svdo_trainer *t, int lvl, int nfb, const unsigned *idx, const float *val) { /* update_ufeedback :138-153 */
    const int k = t->mp.num_factor;
    if (nfb == 0) return;
    float *tmp = t->imfb[lvl].tmp;
    for (int j = 0; j < k; j++) tmp[j] = tmp[j] - t->imfb[lvl].old[j];
    t->imfb[lvl].tmp_bias -= t->imfb[lvl].old_bias;
    scale(tmp, (float)(double)(1.0f / t->imfb[lvl].norm), k);
    t->imfb[lvl].tmp_bias *= 1.0f / t->imfb[lvl].norm;
    for (int i = 0; i < nfb; i++) {
        unsigned fid = idx[i];
        float v = val[i];
        axpy(t->W_ufb + (size_t)fid * t->pitch, tmp, (float)(double)v, k);
        if (t->mp.no_user_bias == 0) t->ufb_bias[fid] += t->imfb[lvl].tmp_bias * v;
    }
}
void svdo_update_block(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                       int num_row, const float *row_label, const int *row_ptr,
                       const unsigned *feat_index, const float *feat_value) {
    if (t->mtype[2] == 2) { /* SVDPPMultiIMFB::update :173-192 */
        if (extend_tag == 0 || extend_tag == 1) imfb_push(t, nfb, idx_fb, val_fb);
        svdo_update_csr_batch(t, num_row, row_label, row_ptr, feat_index, feat_value);
        if (extend_tag == 0 || extend_tag == 2) {
            assert_true(t->imfb_top != 0, "start tag,end tag error in implicit feedback");
            --t->imfb_top;
            if (!t->imfb_disable[t->imfb_top]) imfb_scatter(t, t->imfb_top, nfb, idx_fb, val_fb);
        }
        return;
    }
    if (extend_tag == 0 || extend_tag == 1) { /* DEFAULT or START_TAG */
        prepare_ufeedback(t, nfb, idx_fb, val_fb);
        t->old_fb_bias = t->tmp_fb_bias;
        memcpy(t->old_fb, t->tmp_fb, sizeof(float) * (size_t)t->mp.num_factor);
    }
    svdo_update_csr_batch(t, num_row, row_label, row_ptr, feat_index, feat_value); /* update_each :560-565 */
    if (extend_tag == 0 || extend_tag == 2) update_ufeedback(t, nfb, idx_fb, val_fb); /* DEFAULT or END_TAG */
}
/* NOT a reference function -- the window-minibatch step for USER-GROUP blocks, the checker of
 * svdf_k_wunit.hip.  A block is the reference's SVDPPFeature::update (apex_svd_base.h:568-582) on (the user's private state --
 * its W_user row and bias, tmp_ufeedback / old_ufeedback -- , the WINDOW-START replicated side): prepare_ufeedback (:523-538) reads the
 * window-start W_ufeedback / ufeedback_bias; every row is update_inner with its item rows / biases and global biases put back afterwards
 * and their change added to dW_item / di_bias / dg_bias (stale_row above); update_ufeedback (:539-554) runs on the window-start
 * feedback rows, which are put back too, their change (w + d * val) - w added to dW_fb / dfb_bias in list order.  Returns 0, or -1 for
 * configurations outside the step (lazy decay, side tables, variant solvers, shared latent / feedback spaces). */
int svdo_update_block_stale(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                            int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value,
                            float *dW_item, float *di_bias, float *dg_bias, float *dW_fb, float *dfb_bias) {
    if (!stale_supported(t) || !is_user_group(t) || t->mp.common_feedback_space != 0) return -1;
    const int k = t->mp.num_factor;
    if (extend_tag == 0 || extend_tag == 1) { /* DEFAULT or START_TAG */
        prepare_ufeedback(t, nfb, idx_fb, val_fb);
        t->old_fb_bias = t->tmp_fb_bias;
        memcpy(t->old_fb, t->tmp_fb, sizeof(float) * (size_t)k);
    }
    stale_save sv = {NULL, 0};
    for (int r = 0; r < num_row; r++) { /* update_each :560-565 */
        elem e = csr_row(r, row_label, row_ptr, feat_index, feat_value);
        stale_row(t, &e, &sv, dW_item, di_bias, dg_bias);
    }
    free(sv.buf);
    if (extend_tag == 0 || extend_tag == 2) { /* DEFAULT or END_TAG */
        float *save = (float *)malloc(sizeof(float) * (size_t)(k + 1) * (size_t)(nfb > 0 ? nfb : 1));
        for (int i = 0; i < nfb; i++) {
            assert_true(idx_fb[i] < (unsigned)t->mp.num_ufeedback, "ufeedback id exceed bound");
            memcpy(save + (size_t)i * (k + 1), t->W_ufb + (size_t)idx_fb[i] * t->pitch, sizeof(float) * (size_t)k);
            save[(size_t)i * (k + 1) + k] = t->ufb_bias[idx_fb[i]];
        }
        update_ufeedback(t, nfb, idx_fb, val_fb);
        for (int i = 0; i < nfb; i++) {
            float *w = t->W_ufb + (size_t)idx_fb[i] * t->pitch, *d = dW_fb + (size_t)idx_fb[i] * k;
            const float *s_ = save + (size_t)i * (k + 1);
            for (int j = 0; j < k; j++) { float c = w[j] - s_[j]; if (t->stale_bf16) c = bf16_round(c); d[j] = d[j] + c; w[j] = s_[j]; }
            float cb = t->ufb_bias[idx_fb[i]] - s_[k];
            dfb_bias[idx_fb[i]] = dfb_bias[idx_fb[i]] + cb;
            t->ufb_bias[idx_fb[i]] = s_[k];
        }
        free(save);
    }
    return 0;
}
void svdo_predict_block(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                        int num_row, const float *row_label, const int *row_ptr,
                        const unsigned *feat_index, const float *feat_value, float *out) {
    if (t->mtype[2] == 2) { /* SVDPPMultiIMFB::predict :193-207 */
        if (extend_tag == 0 || extend_tag == 1) imfb_push(t, nfb, idx_fb, val_fb);
        svdo_predict_csr_batch(t, num_row, row_label, row_ptr, feat_index, feat_value, out);
        if (extend_tag == 0 || extend_tag == 2) {
            assert_true(t->imfb_top != 0, "start tag,end tag error in implicit feedback");
            --t->imfb_top;
        }
        return;
    }
    if (extend_tag == 0 || extend_tag == 1) prepare_ufeedback(t, nfb, idx_fb, val_fb);
    svdo_predict_csr_batch(t, num_row, row_label, row_ptr, feat_index, feat_value, out);
}

/* ================= views ================= */
void svdo_view_shape(svdo_trainer *t, int which, int *rows, int *cols) {
    const model_param *p = &t->mp;
    int nfb = p->common_feedback_space == 0 ? p->num_ufeedback : p->num_user;
    *rows = -1; *cols = 0;
    switch (which) {
    case 0: *rows = p->num_user; *cols = 1; break;
    case 1: *rows = p->num_user; *cols = p->num_factor; break;
    case 2: *rows = p->num_item; *cols = 1; break;
    case 3: *rows = p->num_item; *cols = p->num_factor; break;
    case 4: *rows = p->num_global; *cols = 1; break;
    case 5: if (t->mtype[0] == 1) { *rows = nfb; *cols = 1; } break;
    case 6: if (t->mtype[0] == 1) { *rows = nfb; *cols = p->num_factor; } break;
    default: break;
    }
}
long svdo_get_view(svdo_trainer *t, int which, float *out, long capacity) {
    int rows, cols;
    svdo_view_shape(t, which, &rows, &cols);
    if (rows < 0) return -1;
    long n = (long)rows * cols;
    if (n > capacity) return -1;
    const float *vec = NULL, *mat = NULL;
    switch (which) {
    case 0: vec = t->u_bias; break;
    case 1: mat = t->W_user; break;
    case 2: vec = t->i_bias; break;
    case 3: mat = t->W_item; break;
    case 4: vec = t->g_bias; break;
    case 5: vec = t->ufb_bias; break;
    case 6: mat = t->W_ufb; break;
    }
    if (vec) memcpy(out, vec, sizeof(float) * (size_t)n);
    else for (int y = 0; y < rows; y++) memcpy(out + (size_t)y * cols, mat + (size_t)y * t->pitch, sizeof(float) * (size_t)cols);
    return n;
}

long svdo_set_view(svdo_trainer *t, int which, const float *in, long count) {
    int rows, cols;
    svdo_view_shape(t, which, &rows, &cols);
    if (rows < 0 || (long)rows * cols != count) return -1;
    float *vec = NULL, *mat = NULL;
    switch (which) {
    case 0: vec = t->u_bias; break;
    case 1: mat = t->W_user; break;
    case 2: vec = t->i_bias; break;
    case 3: mat = t->W_item; break;
    case 4: vec = t->g_bias; break;
    case 5: vec = t->ufb_bias; break;
    case 6: mat = t->W_ufb; break;
    }
    if (vec) memcpy(vec, in, sizeof(float) * (size_t)count);
    else for (int y = 0; y < rows; y++) memcpy(mat + (size_t)y * t->pitch, in + (size_t)y * cols, sizeof(float) * (size_t)cols);
    return count;
}

/* ================= ISVDRanker: SVDFeatureRanker, solvers/base-solver/apex_svd_base.h:597-813 =================
 * The model, its file format and the side tables are the trainer's (SVDModel, SparseFeatureArray), so the ranker keeps a
 * trainer object for them.  Ordering of candidates with EQUAL scores: the reference leaves it to std::sort (:767), which is
 * not a stable sort; this restatement orders ties by candidate index.  The compiled reference (oracle/_ref) is the checker
 * for tied scores, the tests for this file keep the scores distinct. */
struct svdo_ranker {
    svdo_trainer *m;
    int top_k, num_item_set, num_item_processed, init_end;
    float *tmp_ifactors, *bias_ifactors, *item_score, *tmp_ufactor, *tmp_ifactor, *tmp_ufeedback;
    int *item_tag, *pos_item, npos;
};
svdo_ranker *svdo_ranker_create(int format_type, int active_type, int extend_type, int variant_type) {
    svdo_ranker *r = (svdo_ranker *)calloc(1, sizeof(*r));
    r->m = svdo_create(format_type, active_type, extend_type, variant_type);
    return r;
}
void svdo_ranker_destroy(svdo_ranker *r) {
    if (!r) return;
    svdo_destroy(r->m);
    free(r->tmp_ifactors); free(r->bias_ifactors); free(r->item_score); free(r->tmp_ufactor); free(r->tmp_ifactor); free(r->tmp_ufeedback);
    free(r->item_tag); free(r->pos_item);
    free(r);
}
void svdo_ranker_set_param(svdo_ranker *r, const char *name, const char *val) { /* :656-660 */
    if (!strcmp(name, "feature_user")) strcpy(r->m->name_feat_user, val);
    if (!strcmp(name, "feature_item")) strcpy(r->m->name_feat_item, val);
    if (!strcmp(name, "top_k")) r->top_k = atoi(val);
}
int svdo_ranker_load_model_path(svdo_ranker *r, const char *path, int with_type_header) { /* :662-664 */
    FILE *fi = fopen(path, "rb");
    if (!fi) return -1;
    if (with_type_header) { uint8_t mt[4]; assert_true(fread(mt, 1, 4, fi) == 4, "loading model"); }
    load_model(r->m, fi);
    fclose(fi);
    return 0;
}
void svdo_ranker_init(svdo_ranker *r, int num_item_set) { /* :666-685 */
    svdo_trainer *t = r->m;
    if (strcmp(t->name_feat_user, "NULL")) sf_load(&t->feat_user, t->name_feat_user);
    if (strcmp(t->name_feat_item, "NULL")) sf_load(&t->feat_item, t->name_feat_item);
    const size_t row = (size_t)t->pitch + 4;
    r->num_item_processed = 0;
    r->num_item_set = num_item_set;
    r->tmp_ufactor = (float *)calloc(row, sizeof(float));
    r->tmp_ifactor = (float *)calloc(row, sizeof(float));
    r->tmp_ufeedback = (float *)calloc(row, sizeof(float));
    /* user-group models: tmp_ufeedback = clone( model.W_user[0] ) (:680-682) -- a COPY of user row 0 (CloneSolver,
     * apex_tensor_func_decl_common.h:265-274), which user sections see until the first block arrives */
    if (t->mtype[0] == 1 && t->mp.num_user > 0) memcpy(r->tmp_ufeedback, t->W_user, sizeof(float) * (size_t)t->mp.num_factor);
    r->tmp_ifactors = (float *)calloc((size_t)(num_item_set > 0 ? num_item_set : 1) * (size_t)t->pitch + 4, sizeof(float));
    r->bias_ifactors = (float *)calloc((size_t)num_item_set + 1, sizeof(float));
    r->item_score = (float *)calloc((size_t)num_item_set + 1, sizeof(float));
    r->item_tag = (int *)calloc((size_t)num_item_set + 1, sizeof(int));
    r->pos_item = (int *)calloc((size_t)num_item_set + 1, sizeof(int));
    r->init_end = 1;
}
/* :676-700 */
static void rk_prepare_ifactor(svdo_ranker *r, float *ifactor, float *bias_out, const elem *f) {
    svdo_trainer *t = r->m;
    const int k = t->mp.num_factor;
    const sf_entry *vec;
    float bias = 0.0f;
    for (int j = 0; j < k; j++) ifactor[j] = 0.0f;
    for (int i = 0; i < f->ni; i++) {
        unsigned iid = f->ii[i];
        float ival = f->vi[i];
        assert_true(iid < (unsigned)t->mp.num_item, "item feature index exceed setting");
        axpy(ifactor, t->W_item + (size_t)iid * t->pitch, (float)(double)ival, k);
        bias += t->i_bias[iid] * ival;
        int n = sf_get(&t->feat_item, iid, &vec);
        for (int j = 0; j < n; j++) {
            axpy(ifactor, t->W_item + (size_t)vec[j].index * t->pitch, (float)((double)vec[j].value * (double)ival), k);
            bias += t->i_bias[vec[j].index] * vec[j].value * ival;
        }
    }
    for (int i = 0; i < f->ng; i++) {
        unsigned gid = f->ig[i];
        assert_true(gid < (unsigned)t->mp.num_global, "global feature index exceed setting");
        bias += f->vg[i] * t->g_bias[gid];
    }
    *bias_out = bias;
}
typedef struct { int iid; float score; } rk_entry;
static int rk_cmp(const void *a, const void *b) { /* Entry::operator< (:623): descending score; ties by index (see the note above) */
    const rk_entry *x = (const rk_entry *)a, *y = (const rk_entry *)b;
    if (x->score > y->score) return -1;
    if (y->score > x->score) return 1;
    return (x->iid > y->iid) - (x->iid < y->iid);
}
static long rk_proc(svdo_ranker *r, const elem *f, int *out, long cap) { /* proc :786-796 */
    svdo_trainer *t = r->m;
    const int k = t->mp.num_factor;
    const sf_entry *vec;
    const int tag = (int)f->label;
    long nout = 0;
    if (tag == 0) { /* ITEM_TAG, proc_item :702-707 */
        const int idx = r->num_item_processed++;
        assert_true(r->num_item_processed <= r->num_item_set, "item instance exceed specified item set size");
        rk_prepare_ifactor(r, r->tmp_ifactors + (size_t)idx * t->pitch, &r->bias_ifactors[idx], f);
    } else if (tag == 2) { /* USER_TAG, proc_user :709-728 */
        if (t->mtype[0] == 1) memcpy(r->tmp_ufactor, r->tmp_ufeedback, sizeof(float) * (size_t)k);
        else for (int j = 0; j < k; j++) r->tmp_ufactor[j] = 0.0f;
        for (int i = 0; i < f->nu; i++) {
            unsigned uid = f->iu[i];
            assert_true(uid < (unsigned)t->mp.num_user, "user feature index exceed bound");
            axpy(r->tmp_ufactor, t->W_user + (size_t)uid * t->pitch, (float)(double)f->vu[i], k);
            int n = sf_get(&t->feat_user, uid, &vec);
            for (int j = 0; j < n; j++) axpy(r->tmp_ufactor, t->W_user + (size_t)vec[j].index * t->pitch, (float)(double)vec[j].value, k);
        }
        r->npos = 0;
        for (int i = 0; i < r->num_item_set; i++) r->item_score[i] = 0.0f;
        for (int i = 0; i < r->num_item_processed; i++) r->item_tag[i] = 0;
    } else if (tag == 1 || tag == -1) { /* POS_SAMPLE / BAN_SAMPLE, proc_tag :729-738 */
        for (int i = 0; i < f->nu; i++) {
            const int idx = (int)f->iu[i];
            assert_true(idx < r->num_item_processed, "sample item index exceed bound");
            assert_true(r->item_tag[idx] == 0, "each pos sample item can not occur in baned sample list");
            r->item_tag[idx] = tag;
            if (tag == 1) r->pos_item[r->npos++] = idx;
        }
    } else if (tag == 3) { /* SPEC_SAMPLE, proc_spec :739-747 */
        assert_true(f->nu == 1, "must specify item index of sample in user feature field\n");
        const int idx = (int)f->iu[0];
        assert_true(idx < r->num_item_processed, "sample item index exceed bound");
        float bias;
        rk_prepare_ifactor(r, r->tmp_ifactor, &bias, f);
        r->item_score[idx] = bias + sdot(r->tmp_ufactor, r->tmp_ifactor, k);
    } else if (tag == 4) { /* PROCESS_TAG, proc_rank :748-785 */
        rk_entry *e = (rk_entry *)malloc(sizeof(rk_entry) * (size_t)(r->num_item_processed + 1));
        int ne = 0;
        for (int i = 0; i < r->num_item_processed; i++) {
            if (r->item_tag[i] == -1) continue;
            r->item_score[i] += r->bias_ifactors[i] + sdot(r->tmp_ufactor, r->tmp_ifactors + (size_t)i * t->pitch, k);
            e[ne].iid = i; e[ne].score = r->item_score[i]; ne++;
        }
        qsort(e, (size_t)ne, sizeof(rk_entry), rk_cmp);
        if (r->top_k > 0) {
            assert_true(ne >= r->top_k, "k can not exceed candidate size");
            for (int j = 0; j < r->top_k; j++) { if (nout < cap) out[nout] = e[j].iid; nout++; }
        } else {
            for (int i = 0; i < ne; i++) r->item_tag[e[i].iid] = i;
            for (int i = 0; i < r->npos; i++) { if (nout < cap) out[nout] = r->item_tag[r->pos_item[i]]; nout++; }
        }
        free(e);
    }
    return nout;
}
long svdo_ranker_process_csr(svdo_ranker *r, float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out, long cap) {
    elem e = make_elem(label, ng, nu, ni, index, value);
    return rk_proc(r, &e, out, cap);
}
long svdo_ranker_process_block(svdo_ranker *r, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb, int num_row,
                               const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value, int *out, long cap) {
    svdo_trainer *t = r->m; /* :797-812 */
    if (extend_tag == 0 || extend_tag == 1) {
        const int k = t->mp.num_factor;
        for (int j = 0; j < k; j++) r->tmp_ufeedback[j] = 0.0f;
        for (int i = 0; i < nfb; i++) {
            assert_true(idx_fb[i] < (unsigned)t->mp.num_ufeedback, "ufeedback id exceed bound");
            axpy(r->tmp_ufeedback, t->W_ufb + (size_t)idx_fb[i] * t->pitch, (float)(double)val_fb[i], k);
        }
    }
    long total = 0;
    for (int i = 0; i < num_row; i++) {
        elem e = csr_row(i, row_label, row_ptr, feat_index, feat_value);
        total += rk_proc(r, &e, out + total, cap - total);
    }
    return total;
}
double svdo_sum_sq_err(const float *pred, const float *label, long n, float scale) { /* svd_feature_infer.cpp:43-47 */
    long double sum = 0.0f;
    for (long i = 0; i < n; i++) { double diff = (pred[i] - label[i]) * scale; sum += diff * diff; }
    return (double)sum;
}
