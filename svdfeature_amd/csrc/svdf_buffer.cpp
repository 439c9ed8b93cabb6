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
#include "svdf_kernels.h"
#include "svdf_internal.h"
#include <sys/stat.h>

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
    // (an amd:gpus handle shards the pass: the blocks below go through multi_dataset_from_blocks; the all-in-HBM form is one engine's)
    // `amd:step = auto`: the pass is re-drawn every round, the DEPTH of its dependency graph is a property of the file (user-grouped pairs): the
    // first pass goes through dataset_from_blocks' auto hook (level schedule, decision), the decision is kept for the later passes of this file --
    // exact: the all-in-HBM form below; window: the drawn blocks straight into the window builder.
    const bool auto_on = auto_step_active();
    const bool auto_known = auto_on && auto_rank_path_ == path && auto_rank_decision_ != 0;
    if (device_rank_ && device_sched_ && !rank_prefetch_ && !host_only_ && (!multi_ || in_multi_scope()) && (!auto_on || (auto_known && auto_rank_decision_ != 2))) {
        Dataset *ds = rank_pass_device(path);
        if (ds) return ds;
    }
    UserGroupArrays g;
    if (!(device_rank_ && !rank_prefetch_ && !host_only_ && rank_pass_device_general(path, g))) rank_pass(path, g);
    if (auto_known && auto_rank_decision_ == 2 && wunit_config_ok() &&
        wunit_blocks_ok((long)g.tag.size(), g.tag.data(), g.fb_ptr.data(), g.fb_index.data(), g.block_row_ptr.data(), g.rows.row_ptr.data(), g.rows.index.data())) {
        flush();
        return wseq_from_blocks((long)g.tag.size(), g.tag.data(), g.fb_ptr.data(), g.fb_index.data(), g.fb_value.data(), g.block_row_ptr.data(),
                                g.rows.label.data(), g.rows.row_ptr.data(), g.rows.index.data(), g.rows.value.data());
    }
    Dataset *ds = dataset_from_blocks((long)g.tag.size(), g.tag.data(), g.fb_ptr.data(), g.fb_index.data(), g.fb_value.data(),
                                      g.block_row_ptr.data(), g.rows.label.data(), g.rows.row_ptr.data(), g.rows.index.data(),
                                      g.rows.value.data());
    if (auto_on && !auto_known) { auto_rank_path_ = path; auto_rank_decision_ = auto_last_.decided; }
    return ds;
}

// The device form of the same pass: the file's rows live in HBM (uploaded once per file), every pass draws
// its pairs there with the libc rand() stream the host sampler would have consumed (svdf_randstream.cpp, svdf_k_sample.hip),
// schedules them there (svdf_k_sched.hip) and leaves libc's generator where the host sampler would have left it.  Taken when
// the pass needs no per-user state -- no block carries implicit feedback, so update(block) == update_inner(row)
// (apex_svd_base.h:512-527,539,557-561) -- every row has no global, one user entry that survives the |value| > 1e-6 filter
// (apex_svd_data.cpp:897-903) and one item entry, and the sampler runs positives against negatives without pointwise
// output.  Everything else goes through the host sampler above.  Returns nullptr to decline.
// the candidate file of the device samplers, parsed and uploaded once per file: cached on path, size, inode and mtime (ns)
bool Engine::rank_source_load(const char *path) {
    struct stat sb;
    if (stat(path, &sb) != 0) return false;   // the host path reports the error
    const long mt = (long)sb.st_mtim.tv_sec * 1000000000L + (long)sb.st_mtim.tv_nsec;
    if (rank_source_ && rank_source_->path == path && rank_source_->file_size == (long)sb.st_size && rank_source_->file_mtime == mt &&
        rank_source_->file_ino == (long)sb.st_ino) return true;
    std::unique_ptr<RankSource> src(new RankSource());
    src->path = path; src->file_size = (long)sb.st_size; src->file_mtime = mt; src->file_ino = (long)sb.st_ino;
    MappedFile file(path);
    UserGroupArrays g;
    read_user_group(file, nullptr, g);
    const long nb = (long)g.tag.size(), nr = (long)g.rows.label.size();
    const long nv = (long)g.rows.index.size();
    // ---- plain rows without feedback (svdf_k_sample.hip): one user entry that survives the |value| > 1e-6 filter, one item entry
    bool ok = g.fb_index.empty();
    std::vector<unsigned> ui((size_t)nr), ii((size_t)nr);
    std::vector<float> uv((size_t)nr), iv((size_t)nr);
    for (long r = 0; r < nr && ok; r++) {
        const int64_t *p = &g.rows.row_ptr[(size_t)3 * r];
        ok = p[1] == p[0] && p[2] == p[1] + 1 && p[3] == p[2] + 1;
        if (!ok) break;
        ui[(size_t)r] = g.rows.index[(size_t)p[1]]; uv[(size_t)r] = g.rows.value[(size_t)p[1]];
        ii[(size_t)r] = g.rows.index[(size_t)p[2]]; iv[(size_t)r] = g.rows.value[(size_t)p[2]];
        ok = uv[(size_t)r] > 1e-6f || uv[(size_t)r] < -1e-6f;
    }
    src->eligible = ok && nr < 0x7FFFFFFFL;
    src->general = nr < 0x7FFFFFFFL && nv < 0x7FFFFFFFL;
    if (src->eligible || src->general) {
        need_device("rank input");
        src->num_block = nb; src->num_row = nr;
        std::vector<long> brp((size_t)nb + 1);
        for (long b = 0; b <= nb; b++) brp[(size_t)b] = (long)g.block_row_ptr[(size_t)b];
        src->block_row_ptr.upload(brp.data(), brp.size(), stream_);
        src->label.upload(g.rows.label.data(), (size_t)nr, stream_);
        src->draws.reserve((size_t)nb + 1); src->pairs.reserve((size_t)nb + 1);
        src->draw_off.reserve((size_t)nb + 2); src->pair_off.reserve((size_t)nb + 2);
        src->pos_list.reserve((size_t)nr + 1); src->neg_list.reserve((size_t)nr + 1);
    }
    if (src->eligible) {
        src->uidx.upload(ui.data(), (size_t)nr, stream_); src->uval.upload(uv.data(), (size_t)nr, stream_);
        src->iidx.upload(ii.data(), (size_t)nr, stream_); src->ival.upload(iv.data(), (size_t)nr, stream_);
    }
    if (src->general) {
        std::vector<int> rp((size_t)3 * nr + 1);
        for (size_t j = 0; j < rp.size(); j++) rp[j] = (int)g.rows.row_ptr[j];
        src->g_row_ptr.upload(rp.data(), rp.size(), stream_);
        src->g_index.upload(g.rows.index.data(), (size_t)nv, stream_);
        src->g_value.upload(g.rows.value.data(), (size_t)nv, stream_);
        src->h_tag = g.tag; src->h_fb_ptr = g.fb_ptr; src->h_fb_index = g.fb_index; src->h_fb_value = g.fb_value;
    }
    if (src->eligible || src->general) HIPCHECK(hipStreamSynchronize(stream_));
    rank_source_ = std::move(src);
    return true;
}

// The pass drawn in HBM for ANY row shape and sampler setting (svdf_k_gsample.hip): counts -> scan -> rand() stream -> pairs -> sizes ->
// scan -> merged rows, then the generated blocks come back to the host arrays the resident-dataset path takes (the feedback lists
// and tags are the file's own).  Same draws as the host sampler, libc's generator left where it would have left it.
bool Engine::rank_pass_device_general(const char *path, UserGroupArrays &g) {
    if (!rank_source_load(path)) return false;
    RankSource &S = *rank_source_;
    if (!S.general) return false;
    need_device("rank input");
    LibcRand rs;
    if (!libc_rand_capture(rs)) return false;   // a caller-installed generator of another size: the host path just calls rand()
    pair_sampler_.init();
    if (pair_sampler_.seed_bytime()) { if (!libc_rand_capture(rs)) return false; }
    const int method = pair_sampler_.method();
    check(method == 0 || method == 1, "unkown rank sample method\n");   // apex_svd_data.cpp:1010
    const long nb = S.num_block;
    RankRowsDev D{nb, S.num_row, S.block_row_ptr.p, S.label.p, S.g_row_ptr.p, S.g_index.p, S.g_value.p};
    GSamplerParams sp{pair_sampler_.pos_lowerb(), pair_sampler_.neg_upperb(), pair_sampler_.gap(), pair_sampler_.sample_num(), pair_sampler_.sample_max(),
                      method, method, pair_sampler_.pointwise() != 0 ? 1 : 0};
    launch_gsample_counts(D, sp, S.pos_list.p, S.draws.p, S.pairs.p, stream_);
    std::vector<long> hd((size_t)nb + 1), hp((size_t)nb + 1);
    if (nb > 0) {
        HIPCHECK(hipMemcpyAsync(hd.data(), S.draws.p, (size_t)nb * sizeof(long), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipMemcpyAsync(hp.data(), S.pairs.p, (size_t)nb * sizeof(long), hipMemcpyDeviceToHost, stream_));
    }
    HIPCHECK(hipStreamSynchronize(stream_));
    long D_total = 0, P_total = 0;
    for (long b = 0; b < nb; b++) { const long d = hd[(size_t)b], q = hp[(size_t)b]; hd[(size_t)b] = D_total; hp[(size_t)b] = P_total; D_total += d; P_total += q; }
    hd[(size_t)nb] = D_total; hp[(size_t)nb] = P_total;
    const long per_pair = sp.pointwise ? 2 : 1, R_total = P_total * per_pair;
    check(R_total < 0x2AAAAAAAL, "rank input: too many generated rows in one pass");
    S.draw_off.upload(hd.data(), hd.size(), stream_);
    S.pair_off.upload(hp.data(), hp.size(), stream_);
    const long C = 2048, nchunks = (D_total + C - 1) / C;
    std::vector<uint32_t> tables;
    libc_rand_chunk_states(rs, nchunks, C, tables);
    S.tables.upload(tables.data(), tables.size(), stream_);
    S.raw.reserve((size_t)std::max<long>(D_total, 1));
    launch_rand_expand(S.tables.p, nchunks, C, D_total, S.raw.p, stream_);
    const size_t np = (size_t)std::max<long>(P_total, 1);
    S.pair_p.reserve(np); S.pair_n.reserve(np);
    launch_gsample_pairs(D, sp, S.draw_off.p, S.pair_off.p, S.raw.p, S.pos_list.p, S.neg_list.p, S.pair_p.p, S.pair_n.p, stream_);
    // libc's generator moves on by exactly the draws of this pass
    if (D_total > 0) {
        LibcRand after = rs;
        if (D_total >= 31) {
            HIPCHECK(hipMemcpyAsync(after.x, S.raw.p + (D_total - 31), 31 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
        } else {
            std::vector<uint32_t> tail((size_t)D_total);
            HIPCHECK(hipMemcpyAsync(tail.data(), S.raw.p, (size_t)D_total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
            for (long j = 0; j < 31 - D_total; j++) after.x[j] = rs.x[j + D_total];
            for (long j = 0; j < D_total; j++) after.x[31 - D_total + j] = tail[(size_t)j];
        }
        libc_rand_restore(after);
    }
    // section lengths -> row_ptr -> entries
    const long nsec = 3 * R_total;
    S.lens.reserve((size_t)nsec + 2); S.optr.reserve((size_t)nsec + 2);
    HIPCHECK(hipMemsetAsync(S.lens.p, 0, ((size_t)nsec + 2) * sizeof(int), stream_));
    launch_gpair_sizes(D, sp, P_total, S.pair_p.p, S.pair_n.p, S.lens.p, stream_);
    try { device_exclusive_scan_i32(S.lens.p, S.optr.p, nsec, &S.scan_tmp, &S.scan_tmp_bytes, stream_); }
    catch (const std::exception &e) { fail(std::string("rank input: ") + e.what()); }
    int total_vals = 0;
    HIPCHECK(hipMemcpyAsync(&total_vals, S.optr.p + nsec, sizeof(int), hipMemcpyDeviceToHost, stream_));
    HIPCHECK(hipStreamSynchronize(stream_));
    check(total_vals >= 0, "rank input: more than 2^31-1 feature entries in one pass");
    const size_t nvout = (size_t)std::max(total_vals, 1);
    S.out_label.reserve((size_t)std::max<long>(R_total, 1)); S.out_index.reserve(nvout); S.out_value.reserve(nvout);
    launch_gpair_write(D, sp, P_total, S.pair_p.p, S.pair_n.p, S.optr.p, S.out_label.p, S.out_index.p, S.out_value.p, stream_);
    HIPCHECK(hipGetLastError());
    // back to the host form of a pass
    g.tag = S.h_tag; g.fb_ptr = S.h_fb_ptr; g.fb_index = S.h_fb_index; g.fb_value = S.h_fb_value;
    g.block_row_ptr.resize((size_t)nb + 1);
    for (long b = 0; b <= nb; b++) g.block_row_ptr[(size_t)b] = (int64_t)hp[(size_t)b] * per_pair;
    g.rows.label.resize((size_t)R_total); g.rows.index.resize((size_t)total_vals); g.rows.value.resize((size_t)total_vals);
    std::vector<int> rp((size_t)nsec + 1, 0);
    if (R_total > 0) {
        HIPCHECK(hipMemcpyAsync(g.rows.label.data(), S.out_label.p, (size_t)R_total * sizeof(float), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipMemcpyAsync(rp.data(), S.optr.p, ((size_t)nsec + 1) * sizeof(int), hipMemcpyDeviceToHost, stream_));
    }
    if (total_vals > 0) {
        HIPCHECK(hipMemcpyAsync(g.rows.index.data(), S.out_index.p, (size_t)total_vals * sizeof(unsigned), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipMemcpyAsync(g.rows.value.data(), S.out_value.p, (size_t)total_vals * sizeof(float), hipMemcpyDeviceToHost, stream_));
    }
    HIPCHECK(hipStreamSynchronize(stream_));
    g.rows.row_ptr.assign(rp.begin(), rp.end());
    n_device_rank_passes_++;
    return true;
}

Dataset *Engine::rank_pass_device(const char *path) {
    if (pair_sampler_.method() != 0 || pair_sampler_.pointwise() != 0) return nullptr;
    if (!rows_without_feedback_ || !fused_allowed_for_rows()) return nullptr;
    if (!rank_source_load(path)) return nullptr;
    RankSource &S = *rank_source_;
    if (!S.eligible) return nullptr;
    check(trainer_ready_, "dataset: init_trainer has not been called");
    need_device("dataset");
    flush();
    check(!unit_open_, "dataset_from_blocks: a START block is pending in the trainer");
    LibcRand rs;
    if (!libc_rand_capture(rs)) return nullptr;   // a caller-installed generator of another size: the host path just calls rand()
    pair_sampler_.init();
    if (pair_sampler_.seed_bytime()) { if (!libc_rand_capture(rs)) return nullptr; }   // init() re-seeded (apex_svd_data.cpp:984-986)
    const long nb = S.num_block;
    RankSourceDev D{nb, S.num_row, S.block_row_ptr.p, S.label.p, S.uval.p, S.ival.p, S.uidx.p, S.iidx.p};
    SamplerParams sp{pair_sampler_.pos_lowerb(), pair_sampler_.neg_upperb(), pair_sampler_.sample_num(), pair_sampler_.sample_max()};
    launch_sample_counts(D, sp, S.draws.p, S.pairs.p, stream_);
    std::vector<long> hd((size_t)nb + 1), hp((size_t)nb + 1);
    if (nb > 0) {
        HIPCHECK(hipMemcpyAsync(hd.data(), S.draws.p, (size_t)nb * sizeof(long), hipMemcpyDeviceToHost, stream_));
        HIPCHECK(hipMemcpyAsync(hp.data(), S.pairs.p, (size_t)nb * sizeof(long), hipMemcpyDeviceToHost, stream_));
    }
    HIPCHECK(hipStreamSynchronize(stream_));
    long D_total = 0, P_total = 0;
    for (long b = 0; b < nb; b++) { const long d = hd[(size_t)b], p = hp[(size_t)b]; hd[(size_t)b] = D_total; hp[(size_t)b] = P_total; D_total += d; P_total += p; }
    hd[(size_t)nb] = D_total; hp[(size_t)nb] = P_total;
    check(P_total < 0x7FFFFFFFL, "rank input: more than 2^31-1 pairs in one pass");
    S.draw_off.upload(hd.data(), hd.size(), stream_);
    S.pair_off.upload(hp.data(), hp.size(), stream_);
    // the rand() stream of this pass, expanded in chunks from jump-ahead tables
    const long C = 2048, nchunks = (D_total + C - 1) / C;
    std::vector<uint32_t> tables;
    libc_rand_chunk_states(rs, nchunks, C, tables);
    S.tables.upload(tables.data(), tables.size(), stream_);
    S.raw.reserve((size_t)std::max<long>(D_total, 1));
    launch_rand_expand(S.tables.p, nchunks, C, D_total, S.raw.p, stream_);
    // the generated instances, file order
    std::unique_ptr<Dataset> ds(new Dataset());
    adopt(ds.get()); ds->kind = 2; ds->num_row = P_total;
    FusedDev &f = ds->fused;
    f.max_nu = 1; f.max_ni = 2; f.has_g = false; f.inline_g = false;
    DevBuf<float> rl, ruv, rv0, rv1;
    DevBuf<unsigned> rui, ri0, ri1;
    const size_t np = (size_t)std::max<long>(P_total, 1);
    rl.reserve(np); ruv.reserve(np); rv0.reserve(np); rv1.reserve(np); rui.reserve(np); ri0.reserve(np); ri1.reserve(np);
    PairColumns out{rl.p, ruv.p, rv0.p, rv1.p, rui.p, ri0.p, ri1.p};
    launch_sample_posneg(D, sp, S.draw_off.p, S.pair_off.p, S.raw.p, S.pos_list.p, S.neg_list.p, out, stream_);
    // libc's generator moves on by exactly the draws of this pass
    if (D_total > 0) {
        LibcRand after = rs;
        if (D_total >= 31) {
            HIPCHECK(hipMemcpyAsync(after.x, S.raw.p + (D_total - 31), 31 * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
        } else {
            std::vector<uint32_t> tail((size_t)D_total);
            HIPCHECK(hipMemcpyAsync(tail.data(), S.raw.p, (size_t)D_total * sizeof(uint32_t), hipMemcpyDeviceToHost, stream_));
            HIPCHECK(hipStreamSynchronize(stream_));
            for (long j = 0; j < 31 - D_total; j++) after.x[j] = rs.x[j + D_total];
            for (long j = 0; j < D_total; j++) after.x[31 - D_total + j] = tail[(size_t)j];
        }
        libc_rand_restore(after);
    }
    HIPCHECK(hipGetLastError());
    if (P_total > 0) {
        const unsigned *res[3] = {rui.p, ri0.p, ri1.p};
        const unsigned off[3] = {0u, (unsigned)mp_.num_user, (unsigned)mp_.num_user};
        const unsigned limit[3] = {(unsigned)mp_.num_user, (unsigned)mp_.num_item, (unsigned)mp_.num_item};
        const char *msg[3] = {"user feature index exceed bound", "item feature index exceed bound", "item feature index exceed bound"};
        const unsigned *key = sort_batches_ == 1 ? ri0.p : (sort_batches_ == 2 ? rui.p : nullptr);
        schedule_device_columns(ds.get(), P_total, 3, res, off, limit, msg, key, sort_batches_ == 1 ? limit[1] : limit[0],
                                {DUCol{rui.p, &f.uidx[0]}, DUCol{ri0.p, &f.iidx[0]}, DUCol{ri1.p, &f.iidx[1]}},
                                {DFCol{rl.p, &f.label}, DFCol{ruv.p, &f.uval[0]}, DFCol{rv0.p, &f.ival[0]}, DFCol{rv1.p, &f.ival[1]}});
    } else {
        HIPCHECK(hipStreamSynchronize(stream_));
        ds->sched.level_ptr.assign(1, 0);
    }
    const long nbias = (mp_.no_user_bias ? 0 : 1) + 2;
    ds->algorithmic_bytes = P_total * (8L * mp_.num_factor * 3 + 8 * nbias + 16 + 8 * 3);
    n_device_rank_passes_++;
    return ds.release();
}

// The same pass written back as a user-group buffer file (host only: works without a device).
long Engine::rank_sample_buffer_file(const char *in_path, const char *out_path) {
    check(in_path != nullptr && out_path != nullptr, "rank_sample_buffer_file: null path");
    UserGroupArrays g;
    if (!(device_rank_ && !rank_prefetch_ && !host_only_ && device_ >= 0 && trainer_ready_ && rank_pass_device_general(in_path, g))) rank_pass(in_path, g);
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
