// Rank-pair sampling for whole passes (host part): PairwiseRankGenerator of the reference
// (apex_svd_data.cpp:812-1025) restated over row views of a mapped user-group buffer file.  Everything random goes
// through libc rand() exactly as apex-tensor/apex_random.h:42-134 does (next_double, next_uint32(n), shuffle), in the
// generator's call order, so a run seeded like the reference's draws the same pairs.
#include "svdf_engine.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>

namespace svdf {
namespace {

inline double next_double() { return (double)rand() / ((double)RAND_MAX + 1.0); }                    // apex_random.h:48-50
inline uint32_t next_uint32(uint32_t n) { return (uint32_t)floor(next_double() * n); }              // apex_random.h:65-67
void shuffle(std::vector<RankRow> &d) {                                                              // apex_random.h:119-130
    if (d.empty()) return;
    for (uint32_t i = (uint32_t)d.size() - 1; i > 0; i--) std::swap(d[i], d[next_uint32(i + 1)]);
}
inline bool by_label(const RankRow &a, const RankRow &b) { return a.label < b.label; }             // apex_svd_data.cpp:915-917

struct Out {
    std::vector<float> &label;
    std::vector<int64_t> &row_ptr;
    std::vector<unsigned> &index;
    std::vector<float> &value;
    void put(unsigned i, float v) { index.push_back(i); value.push_back(v); }
    void close_field(int n) { row_ptr.push_back(row_ptr.back() + n); }
};

// sorted merge of the positive's and the negative's entries, the negative's with the sign flipped (apex_svd_data.cpp:828-860)
int merge(Out &o, const unsigned *i1, const float *v1, int n1, const unsigned *i2, const float *v2, int n2) {
    int num = 0, i = 0, j = 0;
    while (i < n1 && j < n2) {
        if (i1[i] < i2[j]) { o.put(i1[i], v1[i]); i++; }
        else if (i2[j] < i1[i]) { o.put(i2[j], -v2[j]); j++; }
        else { o.put(i1[i], v1[i] - v2[j]); i++; j++; }
        num++;
    }
    for (; i < n1; i++, num++) o.put(i1[i], v1[i]);
    for (; j < n2; j++, num++) o.put(i2[j], -v2[j]);
    return num;
}

// the positive's user entries; zero-valued ones are dropped (apex_svd_data.cpp:869-877, 897-905)
int user_part(Out &o, const RankRow &p) {
    int kept = 0;
    const unsigned *iu = p.index + p.ng;
    const float *vu = p.value + p.ng;
    for (int i = 0; i < p.nu; i++)
        if (vu[i] > 1e-6f || vu[i] < -1e-6f) { o.put(iu[i], vu[i]); kept++; }
    return kept;
}

void emit_pointwise(Out &o, const RankRow &p, float label) {   // apex_svd_data.cpp:862-885
    for (int i = 0; i < p.ng; i++) o.put(p.index[i], p.value[i]);
    o.close_field(p.ng);
    o.close_field(user_part(o, p));
    for (int i = 0; i < p.ni; i++) o.put(p.index[p.ng + p.nu + i], p.value[p.ng + p.nu + i]);
    o.close_field(p.ni);
    o.label.push_back(label);
}

}  // namespace

void PairSampler::set_param(const char *name, const char *val) {
    if (!strcmp(name, "pos_sample_lowerb")) pos_lowerb_ = (float)atof(val);
    if (!strcmp(name, "neg_sample_upperb")) neg_upperb_ = (float)atof(val);
    if (!strcmp(name, "rank_sample_num")) sample_num_ = atoi(val);
    if (!strcmp(name, "rank_sample_max")) sample_max_ = atoi(val);
    if (!strcmp(name, "seed_sampler_bytime")) seed_bytime_ = atoi(val);
    if (!strcmp(name, "rank_sample_method")) method_ = atoi(val);
    if (!strcmp(name, "rank_sample_gap")) gap_ = (float)atof(val);
    if (!strcmp(name, "rank_sample_pointwise")) pointwise_ = atoi(val);
}

void PairSampler::init() {
    if (init_done_) return;
    init_done_ = true;
    if (seed_bytime_ != 0) srand((unsigned)time(nullptr));
    if (!(gap_ > 0.0f)) fail("must set rank_sample_gap to a value bigger than 0");
}

void PairSampler::sample_block(const std::vector<RankRow> &rows, std::vector<float> &label, std::vector<int64_t> &row_ptr,
                               std::vector<unsigned> &index, std::vector<float> &value) {
    Out o{label, row_ptr, index, value};
    auto genpair = [&](const RankRow &p, const RankRow &n) {   // apex_svd_data.cpp:887-913
        if (pointwise_ != 0) {
            emit_pointwise(o, p, 1.0f);
            emit_pointwise(o, n, 0.0f);
            return;
        }
        o.close_field(merge(o, p.index, p.value, p.ng, n.index, n.value, n.ng));
        o.close_field(user_part(o, p));
        o.close_field(merge(o, p.index + p.ng + p.nu, p.value + p.ng + p.nu, p.ni, n.index + n.ng + n.nu, n.value + n.ng + n.nu, n.ni));
        o.label.push_back(method_ / 10 == 0 ? 1.0f : p.label - n.label);
    };
    pos_.clear();
    neg_.clear();
    if (method_ == 0) {   // positives against negatives (apex_svd_data.cpp:942-962)
        for (const RankRow &e : rows) {
            if (e.label - pos_lowerb_ > -1e-6f) pos_.push_back(e);
            if (e.label - neg_upperb_ < 1e-6f) neg_.push_back(e);
        }
        if (pos_.empty() || neg_.empty()) return;
        shuffle(neg_);
        shuffle(pos_);
        size_t snum = neg_.size();
        if (sample_num_ > 0) snum = (size_t)sample_num_;
        if (snum > (unsigned)sample_max_) snum = (size_t)sample_max_;
        for (size_t i = 0; i < snum; i++) genpair(pos_[i % pos_.size()], neg_[i % neg_.size()]);
        return;
    }
    if (method_ == 1) {   // every row against a random row whose label differs by more than the gap (apex_svd_data.cpp:918-940)
        pos_ = rows;
        neg_ = rows;
        shuffle(neg_);
        std::sort(pos_.begin(), pos_.end(), by_label);
        // sorted copy: how many rows have a label below `bound` (the reference asks std::lower_bound with a row whose label it shifts)
        auto below = [&](const RankRow &like, float bound) {
            RankRow probe = like;
            probe.label = bound;
            return (size_t)(std::lower_bound(pos_.begin(), pos_.end(), probe, by_label) - pos_.begin());
        };
        for (const RankRow &anchor : neg_) {
            const float under = anchor.label - gap_, over = under + gap_ * 2;   // the same two float steps: label - gap, then + 2 gap
            const size_t n_under = below(anchor, under), n_to_over = below(anchor, over);
            const uint32_t partners = (uint32_t)(n_under + pos_.size() - n_to_over);   // rows more than a gap below, then rows a gap or more above
            if (partners == 0) continue;
            const size_t pick = next_uint32(partners);
            if (pick < n_under) genpair(anchor, pos_[pick]);
            else genpair(pos_[n_to_over + pick - n_under], anchor);
        }
        return;
    }
    fail("unkown rank sample method");   // the reference's text, apex_svd_data.cpp:1008
}

}  // namespace svdf
