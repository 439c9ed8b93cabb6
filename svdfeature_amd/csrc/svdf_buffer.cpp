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

#include <cstring>

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
    Cursor c(file);
    const int num_block = c.i32();
    c.i32(); c.i32(); c.i32();   // max_num_ufeedback, max_num_row, max_num_val
    check(num_block >= 0, "buffer file is corrupt (negative block count)");
    std::vector<int> tag((size_t)num_block);
    std::vector<int64_t> fb_ptr((size_t)num_block + 1, 0), block_row_ptr((size_t)num_block + 1, 0);
    size_t rows = 0, vals = 0, fbs = 0;
    for (int b = 0; b < num_block; b++) {
        int nfb = c.i32();
        tag[(size_t)b] = 0;
        if (nfb < 0) { nfb &= 0x7fffffff; tag[(size_t)b] = c.i32(); }
        c.take(8 * (size_t)nfb);
        CsrBatchView v = read_batch(c);
        fbs += (size_t)nfb; rows += (size_t)v.num_row; vals += (size_t)v.num_val;
        fb_ptr[(size_t)b + 1] = (int64_t)fbs;
        block_row_ptr[(size_t)b + 1] = (int64_t)rows;
    }
    std::vector<unsigned> fb_index(fbs);
    std::vector<float> fb_value(fbs);
    CsrArrays a;
    a.reserve(rows, vals);
    Cursor d(file);
    d.take(16);
    size_t r = 0, v = 0;
    for (int b = 0; b < num_block; b++) {
        int nfb = d.i32();
        if (nfb < 0) { nfb &= 0x7fffffff; d.i32(); }
        const size_t f0 = (size_t)fb_ptr[(size_t)b];
        if (nfb) {
            memcpy(&fb_index[f0], d.take(4 * (size_t)nfb), 4 * (size_t)nfb);
            memcpy(&fb_value[f0], d.take(4 * (size_t)nfb), 4 * (size_t)nfb);
        }
        CsrBatchView bv = read_batch(d);
        a.splice(bv, r, v);
        r += (size_t)bv.num_row; v += (size_t)bv.num_val;
    }
    return dataset_from_blocks(num_block, tag.data(), fb_ptr.data(), fb_index.data(), fb_value.data(), block_row_ptr.data(),
                               a.label.data(), a.row_ptr.data(), a.index.data(), a.value.data());
}

}  // namespace svdf
