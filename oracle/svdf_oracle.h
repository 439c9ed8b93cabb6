/*
 * svdf_oracle.h -- C API shared by the two CPU checkers of the apex_svd SGD hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Two shared libraries export exactly this API:
 *
 *   oracle/libsvdf_oracle.so     plain-C restatement of the reference algorithm (svdf_oracle.c)
 *   oracle/_ref/libsvdf_ref.so   the reference's own C++ classes (SVDFeature / SVDPPFeature),
 *                                compiled from /root/reference where they lie and wrapped by
 *                                ref_shim.cpp
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load either
 * library; the product (svdfeature_amd/) never does.
 *
 * The API mirrors the reference's ISVDTrainer virtual surface (apex_svd.h:33-107) flattened to C.
 */
#ifndef SVDF_ORACLE_H_
#define SVDF_ORACLE_H_

#include <stdio.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct svdo_trainer svdo_trainer;

/* create_svd_trainer(SVDTypeParam) (apex_svd.h:212); the four bytes of SVDTypeParam */
svdo_trainer *svdo_create(int format_type, int active_type, int extend_type, int variant_type);
void svdo_destroy(svdo_trainer *t);

/* ISVDTrainer::set_param / init_model / init_trainer / set_round / finish_round */
void svdo_set_param(svdo_trainer *t, const char *name, const char *val);
void svdo_seed(unsigned seed);          /* apex_random::seed -> srand (apex_random.h:42-44) */
void svdo_init_model(svdo_trainer *t);  /* alloc_space + rand_init, consumes libc rand() */
void svdo_init_trainer(svdo_trainer *t);
void svdo_set_round(svdo_trainer *t, int nround);
void svdo_finish_round(svdo_trainer *t);

/* model file = 4-byte SVDTypeParam (written by the caller in the reference, svd_feature.cpp:184-191)
 * followed by SVDModel::save_to_file.  with_type_header != 0 reads/writes the 4 type bytes too. */
int svdo_save_model_path(svdo_trainer *t, const char *path, int with_type_header);
int svdo_load_model_path(svdo_trainer *t, const char *path, int with_type_header);

/* ISVDTrainer::update(const SVDFeatureCSR::Elem&) / predict(const Elem&) */
void svdo_update_csr(svdo_trainer *t, float label, int num_global, int num_ufactor, int num_ifactor,
                     const unsigned *index, const float *value);
float svdo_predict_csr(svdo_trainer *t, float label, int num_global, int num_ufactor, int num_ifactor,
                       const unsigned *index, const float *value);

/* the same over a whole SVDFeatureCSR block (apex_svd_data.h:34-231), rows in order */
void svdo_update_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                           const unsigned *feat_index, const float *feat_value);
void svdo_predict_csr_batch(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                            const unsigned *feat_index, const float *feat_value, float *out);

/* NOT a reference function: the window-minibatch step of the multi-GPU design, the checker of
 * svdf_train_dataset on a window data set.  Every row is the reference's update_inner applied to (current user side,
 * window-start replicated side); the replicated side (W_item / i_bias / g_bias) is left unchanged and its would-be change is
 * added, in file order, to dW_item (num_item x num_factor, unpadded) / di_bias (num_item) / dg_bias (num_global).
 * Returns 0, -1 where the configuration is not supported. */
int svdo_update_csr_batch_stale(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                                const unsigned *feat_index, const float *feat_value, float *dW_item, float *di_bias, float *dg_bias);
/* one window of the one-GPU window step with ordered sub-steps of at most `sub` rows per item (svdf_oracle.c; HIP: k_window_apply) */
int svdo_update_window_substeps(svdo_trainer *t, int num_row, const float *row_label, const int *row_ptr,
                                const unsigned *feat_index, const float *feat_value, int sub);
/* checker steps: round every ROW contribution to bfloat16 before it is summed (the HIP engine's `amd:contrib = bf16`); bias words stay fp32 */
void svdo_set_stale_rounding(svdo_trainer *t, int bf16);
/* the same step for one user-group block (SVDPPFeature::update on the window-start replicated side; svdf_oracle.c) */
int svdo_update_block_stale(svdo_trainer *t, int nfb, int extend_tag, const unsigned *idx_fb, const float *val_fb,
                            int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value,
                            float *dW_item, float *di_bias, float *dg_bias, float *dW_fb, float *dfb_bias);

/* ISVDTrainer::update(const SVDPlusBlock&) / predict(vector<float>&, const SVDPlusBlock&) */
void svdo_update_block(svdo_trainer *t, int num_ufeedback, int extend_tag,
                       const unsigned *index_ufeedback, const float *value_ufeedback,
                       int num_row, const float *row_label, const int *row_ptr,
                       const unsigned *feat_index, const float *feat_value);
void svdo_predict_block(svdo_trainer *t, int num_ufeedback, int extend_tag,
                        const unsigned *index_ufeedback, const float *value_ufeedback,
                        int num_row, const float *row_label, const int *row_ptr,
                        const unsigned *feat_index, const float *feat_value, float *out);

/* raw views for bit-exact comparison.  which: 0 u_bias 1 W_user 2 i_bias 3 W_item 4 g_bias
 * 5 ufeedback_bias 6 W_ufeedback.  Copies rows unpadded into out (rows*cols floats);
 * returns rows*cols, or -1 when the view does not exist. */
long svdo_get_view(svdo_trainer *t, int which, float *out, long capacity);
void svdo_view_shape(svdo_trainer *t, int which, int *rows, int *cols);
/* overwrite a view from rows*cols unpadded floats (test plumbing for the multi-rank exchange tests;
 * returns -1 where unsupported, i.e. in the compiled-reference shim) */
long svdo_set_view(svdo_trainer *t, int which, const float *in, long count);
/* 1 for the plain-C restatement, 2 for the compiled reference */
int svdo_kind(void);
/* the host libm's expf (what active_type::map_active / cal_grad call, apex_svd_model.h:112-156) over an array, or with
 * in == NULL over the floats with bit patterns first + j*step: checker for the device's expf restatement */
void svdo_libm_expf(const float *in, unsigned first, unsigned step, float *out, long n);

/* ---- ISVDRanker (apex_svd.h:160-197) / SVDFeatureRanker (solvers/base-solver/apex_svd_base.h:597-813) ---- */
typedef struct svdo_ranker svdo_ranker;
svdo_ranker *svdo_ranker_create(int format_type, int active_type, int extend_type, int variant_type);
void svdo_ranker_destroy(svdo_ranker *r);
void svdo_ranker_set_param(svdo_ranker *r, const char *name, const char *val);
int svdo_ranker_load_model_path(svdo_ranker *r, const char *path, int with_type_header);
void svdo_ranker_init(svdo_ranker *r, int num_item_set);
/* process(vector<int>&, Elem / SVDPlusBlock): the results of the line go to out[0..cap); returns their number */
long svdo_ranker_process_csr(svdo_ranker *r, float label, int ng, int nu, int ni, const unsigned *index, const float *value, int *out, long cap);
long svdo_ranker_process_block(svdo_ranker *r, int num_ufeedback, int extend_tag, const unsigned *index_ufeedback, const float *value_ufeedback,
                               int num_row, const float *row_label, const int *row_ptr, const unsigned *feat_index, const float *feat_value,
                               int *out, long cap);
/* RMSEEvaluator (svd_feature_infer.cpp:38-56): sequential long double sum of ((pred - label) * scale)^2, returned as double */
double svdo_sum_sq_err(const float *pred, const float *label, long n, float scale);

#ifdef __cplusplus
}
#endif
#endif
