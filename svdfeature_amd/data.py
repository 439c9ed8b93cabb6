"""Host-side containers and file formats of the hot path's inputs (numpy, harness side).

These mirror, field for field, the reference's input structs so that the same arrays can be
handed to the C-ABI (include/svdfeature_amd.h), to the CPU checkers and to files the
reference's own tools read/write:

* ``CSRData``    = ``SVDFeatureCSR``  (apex_svd_data.h:34-231): row_label[n], row_ptr[3n+1],
                   feat_index[nv], feat_value[nv]; per row the order is global, user, item.
* ``PlusBlock``  = ``SVDPlusBlock``   (apex_svd_data.h:376-466): one user's feedback set + CSRData.
* CSR buffer file  (apex_svd_data.cpp:118-195): Param{num_batch,batch_size,max_batch_num} then
  per block num_row,num_val,row_ptr,row_label,feat_index,feat_value.
* user-group buffer file (apex_svd_data.cpp:558-595, apex_svd_data.h:419-450).
* text feature files (apex_svd_data.cpp:70-112, 316-540).
"""
from dataclasses import dataclass, field

import numpy as np

TAG_DEFAULT, TAG_START, TAG_END, TAG_MIDDLE = 0, 1, 2, 3  # svdpp_tag, apex_svd_data.h:353-371


@dataclass
class CSRData:
    row_label: np.ndarray
    row_ptr: np.ndarray
    feat_index: np.ndarray
    feat_value: np.ndarray

    def __post_init__(self):
        self.row_label = np.ascontiguousarray(self.row_label, dtype=np.float32)
        self.row_ptr = np.ascontiguousarray(self.row_ptr, dtype=np.int32)
        self.feat_index = np.ascontiguousarray(self.feat_index, dtype=np.uint32)
        self.feat_value = np.ascontiguousarray(self.feat_value, dtype=np.float32)
        assert self.row_ptr.size == 3 * self.row_label.size + 1

    @property
    def num_row(self):
        return int(self.row_label.size)

    @property
    def num_val(self):
        return int(self.row_ptr[-1] - self.row_ptr[0])

    @staticmethod
    def empty():
        return CSRData(np.zeros(0, np.float32), np.zeros(1, np.int32), np.zeros(0, np.uint32), np.zeros(0, np.float32))

    @staticmethod
    def from_triples(user, item, label):
        """basicMF instances: no global feature, one user id and one item id, values 1."""
        n = len(label)
        row_ptr = np.empty(3 * n + 1, dtype=np.int32)
        base = 2 * np.arange(n, dtype=np.int64)
        row_ptr[0:3 * n:3] = base
        row_ptr[1:3 * n:3] = base
        row_ptr[2:3 * n:3] = base + 1
        row_ptr[3 * n] = 2 * n
        idx = np.empty(2 * n, dtype=np.uint32)
        idx[0::2] = user
        idx[1::2] = item
        return CSRData(np.asarray(label, np.float32), row_ptr, idx, np.ones(2 * n, np.float32))

    @staticmethod
    def from_rows(rows):
        """rows: iterable of (label, [(gid,val)...], [(uid,val)...], [(iid,val)...])."""
        labels, ptr, idx, val = [], [0], [], []
        for label, g, u, i in rows:
            labels.append(label)
            for sec in (g, u, i):
                for a, b in sec:
                    idx.append(a)
                    val.append(b)
                ptr.append(len(idx))
        return CSRData(np.array(labels, np.float32), np.array(ptr, np.int32),
                       np.array(idx, np.uint32), np.array(val, np.float32))

    def row(self, r):
        p = self.row_ptr[3 * r:3 * r + 4]
        return (float(self.row_label[r]), int(p[1] - p[0]), int(p[2] - p[1]), int(p[3] - p[2]),
                self.feat_index[p[0]:p[3]], self.feat_value[p[0]:p[3]])

    def slice_rows(self, start, stop):
        """SVDFeatureCSR::slice_rows (apex_svd_data.h:170-179), rebased to offset 0."""
        stop = min(stop, self.num_row)
        p = self.row_ptr[3 * start:3 * stop + 1]
        return CSRData(self.row_label[start:stop], p - p[0], self.feat_index[p[0]:p[-1]], self.feat_value[p[0]:p[-1]])

    def select_rows(self, keep):
        """the rows with keep[r] set, order kept, rebased to offset 0 (vectorised: millions of rows)"""
        rows = np.nonzero(np.asarray(keep, bool))[0]
        p = self.row_ptr.astype(np.int64)
        start = p[0:-1:3][rows]
        lens = p[3::3][rows] - start
        off = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        src = np.repeat(start - off[:-1], lens) + np.arange(off[-1], dtype=np.int64)
        rp = np.empty(3 * len(rows) + 1, np.int64)
        for j in range(3):
            rp[j:3 * len(rows):3] = p[3 * rows + j] - start + off[:-1]
        rp[-1] = off[-1]
        return CSRData(self.row_label[rows], rp, self.feat_index[src], self.feat_value[src])

    def row_user(self):
        """the user id of every row (rows with exactly one user entry)"""
        p = self.row_ptr.astype(np.int64)
        assert np.all(p[2::3] - p[1::3][:len(p[2::3])] == 1), "rows with exactly one user entry"
        return self.feat_index[p[1:-1:3]]

    @staticmethod
    def concat(parts):
        parts = [p for p in parts if p.num_row]
        if not parts:
            return CSRData.empty()
        ptrs, off = [np.zeros(1, np.int32)], 0
        for p in parts:
            ptrs.append(p.row_ptr[1:] - p.row_ptr[0] + off)
            off += p.num_val
        return CSRData(np.concatenate([p.row_label for p in parts]), np.concatenate(ptrs),
                       np.concatenate([p.feat_index[p.row_ptr[0]:p.row_ptr[-1]] for p in parts]),
                       np.concatenate([p.feat_value[p.row_ptr[0]:p.row_ptr[-1]] for p in parts]))


@dataclass
class PlusBlock:
    index_ufeedback: np.ndarray
    value_ufeedback: np.ndarray
    data: CSRData
    extend_tag: int = TAG_DEFAULT
    extra_info: int = 0

    def __post_init__(self):
        self.index_ufeedback = np.ascontiguousarray(self.index_ufeedback, dtype=np.uint32)
        self.value_ufeedback = np.ascontiguousarray(self.value_ufeedback, dtype=np.float32)

    @property
    def num_ufeedback(self):
        return int(self.index_ufeedback.size)


@dataclass
class BlockArrays:
    """One pass of a user-group buffer as FLAT arrays -- exactly the argument list of svdf_dataset_from_blocks: block b has
    extend_tag[b], feedback entries fb_index/fb_value[fb_ptr[b]:fb_ptr[b+1]] and the rows block_row_ptr[b]:block_row_ptr[b+1]
    of the CSR arrays (row_ptr: int64, 3*num_row+1 entries).  Millions of blocks without a Python object per block."""
    extend_tag: np.ndarray
    fb_ptr: np.ndarray
    fb_index: np.ndarray
    fb_value: np.ndarray
    block_row_ptr: np.ndarray
    row_label: np.ndarray
    row_ptr: np.ndarray
    feat_index: np.ndarray
    feat_value: np.ndarray

    def __post_init__(self):
        self.extend_tag = np.ascontiguousarray(self.extend_tag, np.int32)
        self.fb_ptr = np.ascontiguousarray(self.fb_ptr, np.int64)
        self.fb_index = np.ascontiguousarray(self.fb_index, np.uint32)
        self.fb_value = np.ascontiguousarray(self.fb_value, np.float32)
        self.block_row_ptr = np.ascontiguousarray(self.block_row_ptr, np.int64)
        self.row_label = np.ascontiguousarray(self.row_label, np.float32)
        self.row_ptr = np.ascontiguousarray(self.row_ptr, np.int64)
        self.feat_index = np.ascontiguousarray(self.feat_index, np.uint32)
        self.feat_value = np.ascontiguousarray(self.feat_value, np.float32)
        assert self.fb_ptr.size == self.extend_tag.size + 1 and self.block_row_ptr.size == self.extend_tag.size + 1
        assert self.row_ptr.size == 3 * self.row_label.size + 1

    @property
    def num_block(self):
        return int(self.extend_tag.size)

    @property
    def num_row(self):
        return int(self.row_label.size)

    @staticmethod
    def from_blocks(blocks):
        nb = len(blocks)
        fb_ptr, brp = np.zeros(nb + 1, np.int64), np.zeros(nb + 1, np.int64)
        for j, b in enumerate(blocks):
            fb_ptr[j + 1] = fb_ptr[j] + b.num_ufeedback
            brp[j + 1] = brp[j] + b.data.num_row
        cat = CSRData.concat([b.data for b in blocks])
        fbi = np.concatenate([b.index_ufeedback for b in blocks]) if blocks else np.zeros(0, np.uint32)
        fbv = np.concatenate([b.value_ufeedback for b in blocks]) if blocks else np.zeros(0, np.float32)
        return BlockArrays(np.array([b.extend_tag for b in blocks], np.int32), fb_ptr, fbi, fbv, brp, cat.row_label,
                           cat.row_ptr.astype(np.int64), cat.feat_index, cat.feat_value)

    def to_blocks(self):
        out = []
        for b in range(self.num_block):
            r0, r1 = int(self.block_row_ptr[b]), int(self.block_row_ptr[b + 1])
            p = self.row_ptr[3 * r0:3 * r1 + 1]
            d = CSRData(self.row_label[r0:r1], (p - p[0]).astype(np.int32), self.feat_index[p[0]:p[-1]], self.feat_value[p[0]:p[-1]])
            f0, f1 = int(self.fb_ptr[b]), int(self.fb_ptr[b + 1])
            out.append(PlusBlock(self.fb_index[f0:f1], self.fb_value[f0:f1], d, int(self.extend_tag[b])))
        return out

    def block_user(self):
        """The user a block belongs to: the first user entry of its first row; MIDDLE / END blocks (and blocks without a
        user entry) inherit from the block before them, so a START..END span stays together."""
        nb = self.num_block
        user = np.full(nb, 0xFFFFFFFF, np.uint32)
        has_row = self.block_row_ptr[1:] > self.block_row_ptr[:-1]
        r0 = self.block_row_ptr[:-1][has_row]
        pu, pi = self.row_ptr[3 * r0 + 1], self.row_ptr[3 * r0 + 2]
        ok = pi > pu
        idx = np.flatnonzero(has_row)[ok]
        user[idx] = self.feat_index[pu[ok]]
        own = (user != 0xFFFFFFFF) & ((self.extend_tag == TAG_DEFAULT) | (self.extend_tag == TAG_START))
        src = np.where(own, np.arange(nb), -1)
        np.maximum.accumulate(src, out=src)
        filled = np.where(src >= 0, user[np.maximum(src, 0)], user)
        return np.where(own, user, filled).astype(np.uint32)

    def span_closed_before(self):
        """closed[b] (b = 0..num_block): no START..END span is open between block b-1 and block b (a pass may be cut there)."""
        opens = (self.extend_tag == TAG_START) | (self.extend_tag == TAG_MIDDLE)
        return np.concatenate([[True], ~opens])

    def select(self, keep):
        """The blocks with keep[b] true, order preserved."""
        keep = np.asarray(keep, bool)
        nrow_b = np.diff(self.block_row_ptr)
        nfb_b = np.diff(self.fb_ptr)
        row_keep = np.repeat(keep, nrow_b)
        sec = np.diff(self.row_ptr).reshape(-1, 3)
        ent_keep = np.repeat(row_keep, sec.sum(axis=1)) if sec.size else np.zeros(0, bool)
        fb_keep = np.repeat(keep, nfb_b)
        base = int(self.row_ptr[0])
        new_ptr = np.concatenate([[0], np.cumsum(sec[row_keep].reshape(-1))]).astype(np.int64)
        fi, fv = self.feat_index[base:base + ent_keep.size][ent_keep], self.feat_value[base:base + ent_keep.size][ent_keep]
        f0 = int(self.fb_ptr[0])
        return BlockArrays(self.extend_tag[keep], np.concatenate([[0], np.cumsum(nfb_b[keep])]),
                           self.fb_index[f0:f0 + fb_keep.size][fb_keep], self.fb_value[f0:f0 + fb_keep.size][fb_keep],
                           np.concatenate([[0], np.cumsum(nrow_b[keep])]), self.row_label[row_keep], new_ptr, fi, fv)

    def slice(self, b0, b1):
        keep = np.zeros(self.num_block, bool)
        keep[b0:b1] = True
        return self.select(keep)

    def rows(self):
        """All rows as one CSRData (int32 offsets)."""
        p = self.row_ptr - self.row_ptr[0]
        return CSRData(self.row_label, p.astype(np.int32), self.feat_index[self.row_ptr[0]:self.row_ptr[-1]],
                       self.feat_value[self.row_ptr[0]:self.row_ptr[-1]])


def pairs_as_csr(user, pos, neg):
    """(user, positive item, negative item) -> the rank-pair instances of apex_svd_data.cpp:828-860, 905-911: label 1,
    user:1, the two item entries in index order with the negative's sign flipped."""
    user, pos, neg = (np.asarray(x, np.uint32) for x in (user, pos, neg))
    n = len(user)
    pf = pos < neg
    idx = np.empty(3 * n, np.uint32)
    val = np.empty(3 * n, np.float32)
    idx[0::3] = user
    val[0::3] = 1.0
    idx[1::3] = np.where(pf, pos, neg)
    val[1::3] = np.where(pf, 1.0, -1.0)
    idx[2::3] = np.where(pf, neg, pos)
    val[2::3] = np.where(pf, -1.0, 1.0)
    ptr = np.empty(3 * n + 1, np.int32)
    base = 3 * np.arange(n, dtype=np.int64)
    ptr[0:3 * n:3] = base
    ptr[1:3 * n:3] = base
    ptr[2:3 * n:3] = base + 1
    ptr[3 * n] = 3 * n
    return CSRData(np.ones(n, np.float32), ptr, idx, val)


# ---------------------------------------------------------------- text formats
def read_text_features(path, scale_score=1.0, sort_sections=False):
    """``label ng nu ni idx:val ...`` lines (SVDFeatureCSRLoader, apex_svd_data.cpp:70-112).
    sort_sections=True reproduces the per-section index sort of the user-group loader
    (apex_svd_data.cpp:343-352)."""
    rows = []
    with open(path) as f:
        toks = f.read().split()
    p = 0
    while p < len(toks):
        label = np.float32(toks[p]) / np.float32(scale_score)
        ng, nu, ni = int(toks[p + 1]), int(toks[p + 2]), int(toks[p + 3])
        p += 4
        secs = []
        for n in (ng, nu, ni):
            sec = []
            for t in toks[p:p + n]:
                a, b = t.split(":")
                sec.append((int(a), float(b)))
            if sort_sections:
                sec.sort(key=lambda e: e[0])
            secs.append(sec)
            p += n
        rows.append((label, *secs))
    return CSRData.from_rows(rows)


def read_feedback_file(path):
    """``nline nfb idx:val ...`` per user (apex_svd_data.cpp:472-481)."""
    out = []
    with open(path) as f:
        toks = f.read().split()
    p = 0
    while p < len(toks):
        nline, nfb = int(toks[p]), int(toks[p + 1])
        p += 2
        idx, val = [], []
        for t in toks[p:p + nfb]:
            a, b = t.split(":")
            idx.append(int(a))
            val.append(float(b))
        p += nfb
        out.append((nline, np.array(idx, np.uint32), np.array(val, np.float32)))
    return out


def make_user_blocks(data, feedback, block_max_line=10000):
    """Group consecutive rows into SVDPlusBlocks the way SVDPlusBlockLoader::next does when a
    feedback file is given (apex_svd_data.cpp:466-530): nline rows per user, split into
    START/MIDDLE/END pieces of similar size when nline > block_max_line."""
    blocks, r = [], 0
    for nline, idx, val in feedback:
        remain, first = nline, True
        while True:
            tag = TAG_MIDDLE
            if first:
                tag &= TAG_START
                first = False
            num_line = remain
            if remain > block_max_line:
                pc = (remain + block_max_line - 1) // block_max_line
                num_line = (remain + pc - 1) // pc
            else:
                tag &= TAG_END
            remain -= num_line
            if tag == TAG_MIDDLE:
                bi, bv = np.zeros(0, np.uint32), np.zeros(0, np.float32)
            else:
                bi, bv = idx, val
            blocks.append(PlusBlock(bi, bv, data.slice_rows(r, r + num_line), tag))
            r += num_line
            if remain == 0:
                break
    assert r == data.num_row, "feedback file does not cover the data"
    return blocks


# ---------------------------------------------------------------- binary buffers
def write_csr_buffer(path, data, batch_size=1000):
    nb, max_val, chunks = 0, 0, []
    for s in range(0, data.num_row, batch_size):
        b = data.slice_rows(s, s + batch_size)
        chunks.append(b)
        max_val = max(max_val, b.num_val)
        nb += 1
    with open(path, "wb") as fo:
        np.array([nb, batch_size, max_val], np.int32).tofile(fo)
        for b in chunks:
            _write_csr_block(fo, b)


def _write_csr_block(fo, b):
    np.array([b.num_row, b.num_val], np.int32).tofile(fo)
    (b.row_ptr - b.row_ptr[0]).astype(np.int32).tofile(fo)
    b.row_label.tofile(fo)
    b.feat_index[b.row_ptr[0]:b.row_ptr[-1]].tofile(fo)
    b.feat_value[b.row_ptr[0]:b.row_ptr[-1]].tofile(fo)


def _read_csr_block(buf, off):
    num_row, num_val = np.frombuffer(buf, np.int32, 2, off)
    off += 8
    row_ptr = np.frombuffer(buf, np.int32, 3 * num_row + 1, off)
    off += 4 * (3 * num_row + 1)
    label = np.frombuffer(buf, np.float32, num_row, off)
    off += 4 * num_row
    idx = np.frombuffer(buf, np.uint32, num_val, off)
    off += 4 * num_val
    val = np.frombuffer(buf, np.float32, num_val, off)
    off += 4 * num_val
    return CSRData(label, row_ptr, idx, val), off


def read_csr_buffer(path, as_blocks=False):
    with open(path, "rb") as f:
        buf = f.read()
    nb = int(np.frombuffer(buf, np.int32, 3, 0)[0])
    off, blocks = 12, []
    for _ in range(nb):
        b, off = _read_csr_block(buf, off)
        blocks.append(b)
    return blocks if as_blocks else CSRData.concat(blocks)


def write_ugroup_buffer(path, blocks):
    with open(path, "wb") as fo:
        hdr = np.array([len(blocks), max([b.num_ufeedback for b in blocks] + [0]),
                        max([b.data.num_row for b in blocks] + [0]),
                        max([b.data.num_val for b in blocks] + [0])], np.int32)
        hdr.tofile(fo)
        for b in blocks:
            if b.extend_tag != TAG_DEFAULT:
                np.array([b.num_ufeedback | (1 << 31), b.extend_tag], np.uint32).tofile(fo)
            else:
                np.array([b.num_ufeedback], np.int32).tofile(fo)
            b.index_ufeedback.tofile(fo)
            b.value_ufeedback.tofile(fo)
            _write_csr_block(fo, b.data)


def read_ugroup_buffer(path):
    with open(path, "rb") as f:
        buf = f.read()
    nb = int(np.frombuffer(buf, np.int32, 4, 0)[0])
    off, blocks = 16, []
    for _ in range(nb):
        nfb = int(np.frombuffer(buf, np.int32, 1, off)[0])
        off += 4
        tag = TAG_DEFAULT
        if nfb < 0:
            nfb &= 0x7FFFFFFF
            tag = int(np.frombuffer(buf, np.int32, 1, off)[0])
            off += 4
        idx = np.frombuffer(buf, np.uint32, nfb, off)
        off += 4 * nfb
        val = np.frombuffer(buf, np.float32, nfb, off)
        off += 4 * nfb
        d, off = _read_csr_block(buf, off)
        blocks.append(PlusBlock(idx, val, d, tag))
    return blocks


# ---------------------------------------------------------------- config files
def read_config(path):
    """``name = value`` pairs, ``#`` comments, optional double quotes (apex-utils/apex_config.h:31-124).
    Returns an ordered list of (name, value) like ConfigSaver does."""
    out = []
    with open(path) as f:
        for line in f:
            line = line.split("#", 1)[0].strip()
            if "=" not in line:
                continue
            k, v = line.split("=", 1)
            out.append((k.strip(), v.strip().strip('"')))
    return out
