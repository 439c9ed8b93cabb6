// apex_svd_amd.cpp -- the reference-side binding a maintainer adds to make Gnnng/SVDFeature train on
// an MI355X: it takes the place of solvers/base-solver/apex_svd_base.cpp (the 35-line factory file,
// reference solvers/base-solver/Makefile:17-23) at LINK time.
//
//   g++ -O3 -pthread -I<reference root> -I<this repo>/include \
//       <reference root>/svd_feature.cpp <reference root>/apex_svd_data.cpp apex_svd_amd.cpp \
//       -L<this repo>/svdfeature_amd -lsvdfeature_amd -o svd_feature
//
// It contains no reference code: it includes the reference's own public header (apex_svd.h) and
// forwards every ISVDTrainer virtual (apex_svd.h:33-107) to the C ABI of include/svdfeature_amd.h.
// The reference's CLI, config parser, buffer iterators, loader thread and pairwise-rank generator
// keep running unchanged on the host; only the solver behind ISVDTrainer changes.
#define _CRT_SECURE_NO_WARNINGS
#include "apex_svd.h"   // from the reference tree

#include <svdfeature_amd.h>

#include <vector>

namespace apex_svd {

class SVDFeatureAMD : public ISVDTrainer {
  private:
    svdf_trainer *h;
    std::vector<unsigned> tmp_index;
    std::vector<float> tmp_value;

  public:
    explicit SVDFeatureAMD(const SVDTypeParam &mtype) {
        // errors keep the reference's behaviour: message on stderr, exit(-1) (svdf error mode 0)
        h = svdf_create(mtype.format_type, mtype.active_type, mtype.extend_type, mtype.variant_type, -1);
    }
    virtual ~SVDFeatureAMD() { svdf_destroy(h); }
    virtual void set_param(const char *name, const char *val) { svdf_set_param(h, name, val); }
    virtual void load_model(FILE *fi) { svdf_load_model(h, fi); }
    virtual void save_model(FILE *fo) { svdf_save_model(h, fo); }
    virtual void init_model(void) { svdf_init_model(h); }
    virtual void init_trainer(void) { svdf_init_trainer(h); }
    virtual void set_round(int nround) { svdf_set_round(h, nround); }
    virtual void finish_round(void) { svdf_finish_round(h); }

    // Elem::set_space (apex_svd_data.h:71-78) and SVDFeatureCSR::operator[] (:129-142) both lay the three
    // sections out back to back; hand them over as one array, or pack them if a caller built a
    // non-contiguous Elem by hand.
    inline bool contiguous(const SVDFeatureCSR::Elem &e) const {
        return e.index_ufactor == e.index_global + e.num_global && e.index_ifactor == e.index_ufactor + e.num_ufactor &&
               e.value_ufactor == e.value_global + e.num_global && e.value_ifactor == e.value_ufactor + e.num_ufactor;
    }
    inline void pack(const SVDFeatureCSR::Elem &e) {
        tmp_index.clear();
        tmp_value.clear();
        tmp_index.insert(tmp_index.end(), e.index_global, e.index_global + e.num_global);
        tmp_index.insert(tmp_index.end(), e.index_ufactor, e.index_ufactor + e.num_ufactor);
        tmp_index.insert(tmp_index.end(), e.index_ifactor, e.index_ifactor + e.num_ifactor);
        tmp_value.insert(tmp_value.end(), e.value_global, e.value_global + e.num_global);
        tmp_value.insert(tmp_value.end(), e.value_ufactor, e.value_ufactor + e.num_ufactor);
        tmp_value.insert(tmp_value.end(), e.value_ifactor, e.value_ifactor + e.num_ifactor);
        tmp_index.push_back(0);
        tmp_value.push_back(0.0f);
    }
    virtual void update(const SVDFeatureCSR::Elem &e) {
        if (contiguous(e)) {
            svdf_update_csr(h, e.label, e.num_global, e.num_ufactor, e.num_ifactor, e.index_global, e.value_global);
        } else {
            pack(e);
            svdf_update_csr(h, e.label, e.num_global, e.num_ufactor, e.num_ifactor, &tmp_index[0], &tmp_value[0]);
        }
    }
    virtual float predict(const SVDFeatureCSR::Elem &e) {
        if (contiguous(e)) return svdf_predict_csr(h, e.label, e.num_global, e.num_ufactor, e.num_ifactor, e.index_global, e.value_global);
        pack(e);
        return svdf_predict_csr(h, e.label, e.num_global, e.num_ufactor, e.num_ifactor, &tmp_index[0], &tmp_value[0]);
    }
    virtual void update(const SVDPlusBlock &b) {
        svdf_update_block(h, b.num_ufeedback, b.extend_tag, b.index_ufeedback, b.value_ufeedback, b.data.num_row, b.data.row_label,
                          b.data.row_ptr, b.data.feat_index, b.data.feat_value);
    }
    virtual void predict(std::vector<float> &p, const SVDPlusBlock &b) {
        p.clear();
        p.resize(static_cast<size_t>(b.data.num_row) + 1);
        svdf_predict_block(h, b.num_ufeedback, b.extend_tag, b.index_ufeedback, b.value_ufeedback, b.data.num_row, b.data.row_label,
                           b.data.row_ptr, b.data.feat_index, b.data.feat_value, &p[0]);
        p.resize(static_cast<size_t>(b.data.num_row));
    }
};

// same dispatch rule as the base-solver factory (solvers/base-solver/apex_svd_base.cpp:24-31): the format
// decides between the random-order and the user-group trainer; both live behind one handle here.
ISVDTrainer *create_svd_trainer(SVDTypeParam mtype) { return new SVDFeatureAMD(mtype); }

// ISVDRanker (apex_svd.h:160-197) over svdf_ranker_*: what svd_feature_infer.cpp's task_pred_rank drives (:347-375)
class SVDRankerAMD : public ISVDRanker {
    svdf_ranker *h;
    std::vector<unsigned> tmp_index;
    std::vector<float> tmp_value;
    std::vector<int> buf;
    void take(std::vector<int> &result, int64_t n) {
        if (n < 0) apex_utils::error(svdf_last_error());
        if (n > (int64_t)buf.size()) apex_utils::error("svdfeature_amd ranker: more results than the result buffer holds");   // never a silent truncation
        for (int64_t i = 0; i < n; i++) result.push_back(buf[(size_t)i]);
    }
  public:
    explicit SVDRankerAMD(const SVDTypeParam &mtype) : buf(1024) {
        h = svdf_ranker_create(mtype.format_type, mtype.active_type, mtype.extend_type, mtype.variant_type, -1);
    }
    virtual ~SVDRankerAMD() { svdf_ranker_destroy(h); }
    virtual void load_model(FILE *fi) { svdf_ranker_load_model(h, fi); }
    virtual void init_ranker(int num_item_set) {
        buf.resize(static_cast<size_t>(num_item_set) + 1024);
        svdf_ranker_init(h, num_item_set);
    }
    virtual void set_param(const char *name, const char *val) { svdf_ranker_set_param(h, name, val); }
    virtual void process(std::vector<int> &result, const SVDFeatureCSR::Elem &e) {
        const int nv = e.num_global + e.num_ufactor + e.num_ifactor;
        tmp_index.resize(nv + 1); tmp_value.resize(nv + 1);
        int p = 0;
        for (int i = 0; i < e.num_global; i++, p++) { tmp_index[p] = e.index_global[i]; tmp_value[p] = e.value_global[i]; }
        for (int i = 0; i < e.num_ufactor; i++, p++) { tmp_index[p] = e.index_ufactor[i]; tmp_value[p] = e.value_ufactor[i]; }
        for (int i = 0; i < e.num_ifactor; i++, p++) { tmp_index[p] = e.index_ifactor[i]; tmp_value[p] = e.value_ifactor[i]; }
        take(result, svdf_ranker_process_csr(h, e.label, e.num_global, e.num_ufactor, e.num_ifactor, &tmp_index[0], &tmp_value[0], &buf[0],
                                             (int64_t)buf.size()));
    }
    virtual void process(std::vector<int> &result, const SVDPlusBlock &b) {
        if (buf.size() < 1024 + static_cast<size_t>(b.data.num_row) * 64) buf.resize(1024 + static_cast<size_t>(b.data.num_row) * 64);
        take(result, svdf_ranker_process_block(h, b.num_ufeedback, b.extend_tag, b.index_ufeedback, b.value_ufeedback, b.data.num_row,
                                               b.data.row_label, b.data.row_ptr, b.data.feat_index, b.data.feat_value, &buf[0], (int64_t)buf.size()));
    }
};

ISVDRanker *create_svd_ranker(SVDTypeParam mtype) { return new SVDRankerAMD(mtype); }

};  // namespace apex_svd
