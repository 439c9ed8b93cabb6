// Binary buffer files of the reference, read natively and handed to the resident-dataset path.
//
// CSR buffer  (written by SVDFeatureCSRFactory::create_buffer, apex_svd_data.cpp:118-195; read back by
//             SVDFeatureCSR::load_from_file, apex_svd_data.h:200-230):
//     int num_batch, batch_size, max_batch_num;
//     per batch: int num_row, num_val; int row_ptr[3*num_row+1]; float row_label[num_row];
//                unsigned feat_index[num_val]; float feat_value[num_val];
// user-group buffer (SVDPlusBlock::{save_to,load_from}_file, apex_svd_data.h:419-450; factory apex_svd_data.cpp:558-595):
//     int num_batch, max_num_ufeedback, max_num_row, max_num_val;
//     per block: int num_ufeedback (bit 31 set => an int extend_tag follows, otherwise DEFAULT_TAG = 0);
//                unsigned index_ufeedback[]; float value_ufeedback[]; then one CSR batch as above.
// The file is mapped, walked once for sizes and once to splice the per-batch row_ptr arrays into one 64-bit
// offset array; labels already carry scale_score (the reference applies it when the buffer is made,
// apex_svd_data.cpp:403,426,516).
#include "svdf_engine.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <thread>

namespace svdf {
namespace {

inline void check(bool ok, const char *msg) { if (!ok) fail(msg); }

struct MappedFile {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    explicit MappedFile(const char *path) {
        fd = ::open(path, O_RDONLY);
        if (fd < 0) fail(std::string("can not open file \"") + path + "\"");   // apex_utils::fopen_check text
        struct stat st;
        if (fstat(fd, &st) != 0) { ::close(fd); fail(std::string("can not stat \"") + path + "\""); }
        n = (size_t)st.st_size;
        if (n) {
            void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
            if (m == MAP_FAILED) { ::close(fd); fail(std::string("can not map \"") + path + "\""); }
            p = (const char *)m;
            madvise((void *)p, n, MADV_SEQUENTIAL);
        }
    }
    ~MappedFile() {
        if (p) munmap((void *)p, n);
        if (fd >= 0) ::close(fd);
    }
    MappedFile(const MappedFile &) = delete;
    MappedFile &operator=(const MappedFile &) = delete;
};

struct Cursor {
    const MappedFile &f;
    size_t off = 0;
    explicit Cursor(const MappedFile &file) : f(file) {}
    const char *take(size_t bytes) {
        if (bytes > f.n - off) fail("buffer file is truncated");
        const char *q = f.p + off;
        off += bytes;
        return q;
    }
    int i32() { int v; memcpy(&v, take(4), 4); return v; }
};

struct CsrBatchView {
    int num_row = 0, num_val = 0;
    const char *row_ptr = nullptr, *label = nullptr, *index = nullptr, *value = nullptr;
};

CsrBatchView read_batch(Cursor &c) {
    CsrBatchView b;
    b.num_row = c.i32();
    b.num_val = c.i32();
    if (b.num_row < 0 || b.num_val < 0) fail("buffer file is corrupt (negative batch size)");
    b.row_ptr = c.take(4 * ((size_t)3 * b.num_row + 1));
    b.label = c.take(4 * (size_t)b.num_row);
    b.index = c.take(4 * (size_t)b.num_val);
    b.value = c.take(4 * (size_t)b.num_val);
    return b;
}

struct CsrArrays {
    std::vector<float> label;
    std::vector<int64_t> row_ptr;
    std::vector<unsigned> index;
    std::vector<float> value;
    void reserve(size_t rows, size_t vals) {
        label.resize(rows); row_ptr.resize(3 * rows + 1); index.resize(vals); value.resize(vals);
        row_ptr[0] = 0;
    }
    // appends one batch at (row position r, value position v)
    void splice(const CsrBatchView &b, size_t r, size_t v) {
        if (b.num_row) memcpy(&label[r], b.label, 4 * (size_t)b.num_row);
        if (b.num_val) { memcpy(&index[v], b.index, 4 * (size_t)b.num_val); memcpy(&value[v], b.value, 4 * (size_t)b.num_val); }
        int first; memcpy(&first, b.row_ptr, 4);
        for (size_t j = 1; j <= (size_t)3 * b.num_row; j++) {
            int pj; memcpy(&pj, b.row_ptr + 4 * j, 4);
            if (pj < first || pj - first > b.num_val) fail("buffer file is corrupt (row_ptr outside its batch)");
            row_ptr[3 * r + j] = (int64_t)v + (pj - first);
        }
    }
};

}  // namespace

struct UserGroupArrays {
    std::vector<int> tag;
    std::vector<int64_t> fb_ptr, block_row_ptr;
    std::vector<unsigned> fb_index;
    std::vector<float> fb_value;
    CsrArrays rows;
};

namespace {

// Walks a user-group buffer file.  With a sampler every block's rows are replaced by the rank pairs drawn from them
// (the feedback part of the block is kept as it is, like PairwiseRankGenerator::next, apex_svd_data.cpp:999-1019).
void read_user_group(const MappedFile &file, PairSampler *sampler, UserGroupArrays &g) {
    Cursor c(file);
    const int num_block = c.i32();
    c.i32(); c.i32(); c.i32();   // max_num_ufeedback, max_num_row, max_num_val
    check(num_block >= 0, "buffer file is corrupt (negative block count)");
    g.tag.assign((size_t)num_block, 0);
    g.fb_ptr.assign((size_t)num_block + 1, 0);
    g.block_row_ptr.assign((size_t)num_block + 1, 0);
    size_t rows = 0, vals = 0, fbs = 0;
    for (int b = 0; b < num_block; b++) {
        int nfb = c.i32();
        if (nfb < 0) { nfb &= 0x7fffffff; g.tag[(size_t)b] = c.i32(); }
        c.take(8 * (size_t)nfb);
        CsrBatchView v = read_batch(c);
        fbs += (size_t)nfb; rows += (size_t)v.num_row; vals += (size_t)v.num_val;
        g.fb_ptr[(size_t)b + 1] = (int64_t)fbs;
        g.block_row_ptr[(size_t)b + 1] = (int64_t)rows;
    }
    g.fb_index.resize(fbs);
    g.fb_value.resize(fbs);
    if (!sampler) g.rows.reserve(rows, vals);
    else { g.rows.row_ptr.assign(1, 0); }
    Cursor d(file);
    d.take(16);
    size_t r = 0, v = 0;
    std::vector<RankRow> view;
    std::vector<int> rp;
    for (int b = 0; b < num_block; b++) {
        int nfb = d.i32();
        if (nfb < 0) { nfb &= 0x7fffffff; d.i32(); }
        const size_t f0 = (size_t)g.fb_ptr[(size_t)b];
        if (nfb) {
            memcpy(&g.fb_index[f0], d.take(4 * (size_t)nfb), 4 * (size_t)nfb);
            memcpy(&g.fb_value[f0], d.take(4 * (size_t)nfb), 4 * (size_t)nfb);
        }
        CsrBatchView bv = read_batch(d);
        if (!sampler) {
            g.rows.splice(bv, r, v);
            r += (size_t)bv.num_row; v += (size_t)bv.num_val;
            continue;
        }
        rp.resize((size_t)3 * bv.num_row + 1);
        memcpy(rp.data(), bv.row_ptr, 4 * rp.size());
        view.resize((size_t)bv.num_row);
        for (int i = 0; i < bv.num_row; i++) {   // SVDFeatureCSR::operator[], apex_svd_data.h:129-142
            const int p0 = rp[(size_t)3 * i], p1 = rp[(size_t)3 * i + 1], p2 = rp[(size_t)3 * i + 2], p3 = rp[(size_t)3 * i + 3];
            if (p0 < 0 || p0 > p1 || p1 > p2 || p2 > p3 || p3 > bv.num_val) fail("buffer file is corrupt (row_ptr outside its batch)");
            RankRow &e = view[(size_t)i];
            memcpy(&e.label, bv.label + 4 * (size_t)i, 4);
            e.ng = p1 - p0; e.nu = p2 - p1; e.ni = p3 - p2;
            e.index = (const unsigned *)(bv.index + 4 * (size_t)p0);   // the mapping is page aligned and every field 4-byte sized
            e.value = (const float *)(bv.value + 4 * (size_t)p0);
        }
        sampler->sample_block(view, g.rows.label, g.rows.row_ptr, g.rows.index, g.rows.value);
        g.block_row_ptr[(size_t)b + 1] = (int64_t)g.rows.label.size();
    }
}

}  // namespace

// One pass of the rank-pair sampler drawn ahead of time on a background thread (svdf_rank_prefetch_buffer_file): the
// sampler is pure host work on the mapped file and libc rand(), so it can run while the device trains the previous pass.
struct RankPrefetch {
    std::string path;
    std::thread worker;
    UserGroupArrays arrays;
    std::string error;
    bool failed = false;
};

void Engine::rank_prefetch_drop() {
    if (!rank_prefetch_) return;
    if (rank_prefetch_->worker.joinable()) rank_prefetch_->worker.join();
    delete rank_prefetch_;
    rank_prefetch_ = nullptr;
}

void Engine::rank_prefetch(const char *path) {
    check(path != nullptr, "rank_prefetch: null path");
    check(user_group(), "rank-pair input needs the user-group format (format_type = 1), svd_feature.cpp:129-133");
    check(rank_prefetch_ == nullptr, "rank_prefetch: a prefetched pass is still waiting to be used");
    pair_sampler_.init();
    rank_prefetch_ = new RankPrefetch();
    rank_prefetch_->path = path;
    RankPrefetch *pf = rank_prefetch_;
    PairSampler *sampler = &pair_sampler_;
    pf->worker = std::thread([pf, sampler]() {
        try {
            MappedFile file(pf->path.c_str());
            read_user_group(file, sampler, pf->arrays);
        } catch (const std::exception &e) {
            pf->failed = true;
            pf->error = e.what();
        }
    });
}

// the pass for `path`: the prefetched one if there is one (it must be for the same file), drawn now otherwise
void Engine::rank_pass(const char *path, UserGroupArrays &g) {
    if (rank_prefetch_) {
        if (rank_prefetch_->worker.joinable()) rank_prefetch_->worker.join();
        const bool same = rank_prefetch_->path == path, failed = rank_prefetch_->failed;
        const std::string err = rank_prefetch_->error;
        if (same && !failed) g = std::move(rank_prefetch_->arrays);
        rank_prefetch_drop();
        if (failed) fail(err);
        if (!same) fail("rank_prefetch: the prefetched pass was drawn from another file");
        return;
    }
    MappedFile file(path);
    pair_sampler_.init();
    read_user_group(file, &pair_sampler_, g);
}

Dataset *Engine::dataset_from_buffer_file(const char *path, int user_group_format) {
    check(path != nullptr, "dataset_from_buffer_file: null path");
    MappedFile file(path);
    if (!user_group_format) {
        Cursor c(file);
        const int num_batch = c.i32();
        c.i32(); c.i32();   // batch_size, max_batch_num: iterator sizing hints only
        check(num_batch >= 0, "buffer file is corrupt (negative batch count)");
        size_t rows = 0, vals = 0;
        for (int b = 0; b < num_batch; b++) { CsrBatchView v = read_batch(c); rows += (size_t)v.num_row; vals += (size_t)v.num_val; }
        CsrArrays a;
        a.reserve(rows, vals);
        Cursor d(file);
        d.take(12);
        size_t r = 0, v = 0;
        for (int b = 0; b < num_batch; b++) {
            CsrBatchView bv = read_batch(d);
            a.splice(bv, r, v);
            r += (size_t)bv.num_row; v += (size_t)bv.num_val;
        }
        return dataset_from_csr((long)rows, a.label.data(), a.row_ptr.data(), a.index.data(), a.value.data());
    }
    UserGroupArrays g;
    read_user_group(file, nullptr, g);
    return dataset_from_blocks((long)g.tag.size(), g.tag.data(), g.fb_ptr.data(), g.fb_index.data(), g.fb_value.data(),
                               g.block_row_ptr.data(), g.rows.label.data(), g.rows.row_ptr.data(), g.rows.index.data(),
                               g.rows.value.data());
}

// input_type = 2 of the reference (BINARY_BUFFER_RANK, apex_svd_data.cpp:1330-1332): the user-group buffer file seen
// through PairwiseRankGenerator.  One call = one pass of the iterator (pairs are re-drawn from libc rand() every pass).
Dataset *Engine::dataset_from_rank_buffer_file(const char *path) {
    check(path != nullptr, "dataset_from_rank_buffer_file: null path");
    check(user_group(), "rank-pair input needs the user-group format (format_type = 1), svd_feature.cpp:129-133");
    UserGroupArrays g;
    rank_pass(path, g);
    return dataset_from_blocks((long)g.tag.size(), g.tag.data(), g.fb_ptr.data(), g.fb_index.data(), g.fb_value.data(),
                               g.block_row_ptr.data(), g.rows.label.data(), g.rows.row_ptr.data(), g.rows.index.data(),
                               g.rows.value.data());
}

// The same pass written back as a user-group buffer file (host only: works without a device).
long Engine::rank_sample_buffer_file(const char *in_path, const char *out_path) {
    check(in_path != nullptr && out_path != nullptr, "rank_sample_buffer_file: null path");
    UserGroupArrays g;
    rank_pass(in_path, g);
    FILE *fo = fopen(out_path, "wb");
    if (!fo) fail(std::string("can not open file \"") + out_path + "\"");
    const size_t nb = g.tag.size();
    int head[4] = {(int)nb, 0, 0, 0};
    for (size_t b = 0; b < nb; b++) {
        const int64_t r0 = g.block_row_ptr[b], r1 = g.block_row_ptr[b + 1];
        head[1] = std::max(head[1], (int)(g.fb_ptr[b + 1] - g.fb_ptr[b]));
        head[2] = std::max(head[2], (int)(r1 - r0));
        head[3] = std::max(head[3], (int)(g.rows.row_ptr[3 * r1] - g.rows.row_ptr[3 * r0]));
    }
    fwrite(head, 4, 4, fo);
    std::vector<int> rp;
    for (size_t b = 0; b < nb; b++) {
        const int64_t f0 = g.fb_ptr[b], f1 = g.fb_ptr[b + 1], r0 = g.block_row_ptr[b], r1 = g.block_row_ptr[b + 1];
        int nfb = (int)(f1 - f0);
        if (g.tag[b] != 0) {   // SVDPlusBlock::save_to_file, apex_svd_data.h:419-431
            int marked = nfb | (int)(1u << 31);
            fwrite(&marked, 4, 1, fo);
            fwrite(&g.tag[b], 4, 1, fo);
        } else {
            fwrite(&nfb, 4, 1, fo);
        }
        fwrite(g.fb_index.data() + f0, 4, (size_t)nfb, fo);
        fwrite(g.fb_value.data() + f0, 4, (size_t)nfb, fo);
        const int64_t v0 = g.rows.row_ptr[3 * r0], v1 = g.rows.row_ptr[3 * r1];
        int nrow = (int)(r1 - r0), nval = (int)(v1 - v0);
        fwrite(&nrow, 4, 1, fo);
        fwrite(&nval, 4, 1, fo);
        rp.resize((size_t)3 * nrow + 1);
        for (int j = 0; j <= 3 * nrow; j++) rp[(size_t)j] = (int)(g.rows.row_ptr[3 * r0 + j] - v0);
        fwrite(rp.data(), 4, rp.size(), fo);
        fwrite(g.rows.label.data() + r0, 4, (size_t)nrow, fo);
        fwrite(g.rows.index.data() + v0, 4, (size_t)nval, fo);
        fwrite(g.rows.value.data() + v0, 4, (size_t)nval, fo);
    }
    const bool ok = !ferror(fo);
    if (fclose(fo) != 0 || !ok) fail(std::string("error writing \"") + out_path + "\"");
    return (long)g.block_row_ptr[nb];
}

}  // namespace svdf
