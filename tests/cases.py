"""Seeded synthetic workloads shared by the CPU (oracle) and GPU parity tests.

Every case is (config pairs, training data, test data[, side-feature tables]) built from a
numpy Generator with a fixed seed, so the oracle, the compiled reference and the HIP path all
see identical bytes.
"""
import os

import numpy as np

from svdfeature_amd.data import CSRData, PlusBlock, TAG_DEFAULT, TAG_END, TAG_MIDDLE, TAG_START

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

BASICMF_CONF = [  # demo/basicMF/basicMF.conf:4-23
    ("base_score", "3"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"),
    ("num_item", "1682"), ("num_user", "943"), ("num_global", "0"), ("num_factor", "64"), ("active_type", "0"),
]


def conf_with(base, **kw):
    out = [(k, v) for k, v in base if k not in kw]
    out += [(k, str(v)) for k, v in kw.items()]
    return out


def ml100k():
    """ML-100K ua.base (shuffled order of demo/basicMF/ua.base.basicfeature) and ua.test as
    0-based (user, item, rating) arrays; fixture written by tests/golden/make_golden.py."""
    z = np.load(os.path.join(GOLDEN, "ml100k_ua.npz"))
    base = CSRData.from_triples(z["base_u"], z["base_i"], z["base_r"])
    test = CSRData.from_triples(z["test_u"], z["test_i"], z["test_r"])
    return base, test


def planted_triples(n, num_user, num_item, seed, rank=8, noise=0.3, zipf=False):
    """(u, i, r) with r in 1..5 from a planted low-rank model."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, num_user, n, dtype=np.int64)
    if zipf:
        w = 1.0 / np.arange(1, num_item + 1) ** 0.8
        i = rng.choice(num_item, size=n, p=w / w.sum())
    else:
        i = rng.integers(0, num_item, n, dtype=np.int64)
    pu = rng.standard_normal((num_user, rank)).astype(np.float32)
    qi = rng.standard_normal((num_item, rank)).astype(np.float32)
    score = 3.0 + 0.6 * np.einsum("nk,nk->n", pu[u], qi[i]) / np.sqrt(rank) + noise * rng.standard_normal(n)
    r = np.clip(np.rint(score), 1, 5).astype(np.float32)
    return u.astype(np.uint32), i.astype(np.uint32), r


def sparse_feature_rows(n, num_user, num_item, num_global, seed, max_g=3, max_u=3, max_i=3,
                        binary_label=False, allow_dup=True):
    """Ragged instances: 0..max_g globals, 1..max_u user ids, 1..max_i item ids with real-valued
    weights; some rows have empty sections and (allow_dup) a repeated id inside one section."""
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        ng = int(rng.integers(0, max_g + 1)) if num_global else 0
        nu = int(rng.integers(0 if r % 17 == 5 else 1, max_u + 1))
        ni = int(rng.integers(0 if r % 19 == 7 else 1, max_i + 1))
        g = [(int(rng.integers(0, num_global)), float(np.float32(rng.uniform(-1, 1)))) for _ in range(ng)]
        u = [(int(rng.integers(0, num_user)), float(np.float32(rng.choice([1.0, 0.5, 0.25, rng.uniform(0.1, 1.5)])))) for _ in range(nu)]
        i = [(int(rng.integers(0, num_item)), float(np.float32(rng.choice([1.0, -1.0, 0.5, rng.uniform(-1, 1)])))) for _ in range(ni)]
        if allow_dup and r % 23 == 3 and nu >= 2:
            u[1] = (u[0][0], u[1][1])
        if allow_dup and r % 29 == 4 and ni >= 2:
            i[1] = (i[0][0], i[1][1])
        label = float(rng.integers(0, 2)) if binary_label else float(rng.integers(1, 6))
        rows.append((label, g, u, i))
    return CSRData.from_rows(rows)


def write_side_table(path, num_rows, num_ids, seed, max_children=2):
    """feature_user / feature_item text table (apex-utils/apex_utils.h:172-195): per id
    ``n idx:val ...``.  Children point into the same id space."""
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        for _ in range(num_rows):
            n = int(rng.integers(0, max_children + 1))
            ent = " ".join("%d:%g" % (int(rng.integers(0, num_ids)), float(np.float32(rng.uniform(0.1, 1.0)))) for _ in range(n))
            f.write(("%d %s" % (n, ent)).strip() + "\n")


def user_blocks(num_user_blocks, num_user, num_item, num_ufeedback, seed, max_rows=6, max_fb=5, split_every=0,
                binary_label=False):
    """User-grouped SVD++ blocks: each block = one user's rows + that user's feedback set with
    value n^-1/2 (demo/implicitFeedback/mkimplicitfeedbackfeature.py:46-55).  split_every>0
    splits every split_every'th user into START/MIDDLE/END pieces; some users have no feedback."""
    rng = np.random.default_rng(seed)
    blocks = []
    users = rng.permutation(num_user)[:num_user_blocks]
    for b, uid in enumerate(users):
        nrow = int(rng.integers(1, max_rows + 1))
        nfb = 0 if b % 7 == 3 else int(rng.integers(1, max_fb + 1))
        fb_idx = np.sort(rng.choice(num_ufeedback, size=nfb, replace=False)).astype(np.uint32)
        fb_val = np.full(nfb, 1.0 / np.sqrt(max(nfb, 1)), np.float32)
        rows = []
        for _ in range(nrow):
            label = float(rng.integers(0, 2)) if binary_label else float(rng.integers(1, 6))
            rows.append((label, [], [(int(uid), 1.0)], [(int(rng.integers(0, num_item)), 1.0)]))
        data = CSRData.from_rows(rows)
        if split_every and b % split_every == 1 and nrow >= 3:
            cut1, cut2 = 1, nrow - 1
            e = np.zeros(0, np.uint32), np.zeros(0, np.float32)
            blocks.append(PlusBlock(fb_idx, fb_val, data.slice_rows(0, cut1), TAG_START))
            blocks.append(PlusBlock(e[0], e[1], data.slice_rows(cut1, cut2), TAG_MIDDLE))
            blocks.append(PlusBlock(fb_idx, fb_val, data.slice_rows(cut2, nrow), TAG_END))
        else:
            blocks.append(PlusBlock(fb_idx, fb_val, data, TAG_DEFAULT))
    return blocks


def rmse(pred, label):
    d = pred.astype(np.float64) - label.astype(np.float64)
    return float(np.sqrt(np.mean(d * d)))


# ---- rank-pair input (input_type = 2): user blocks whose rows are the candidates PairwiseRankGenerator pairs up
RANK_SAMPLER_CASES = [   # (name, graded labels, sampler keys)
    ("posneg_default", False, {}),
    ("posneg_num_max", False, {"rank_sample_num": "5", "rank_sample_max": "4"}),
    ("posneg_pointwise", False, {"rank_sample_pointwise": "1"}),
    ("cmp_default_gap", True, {"rank_sample_method": "1"}),
    ("cmp_wide_gap", True, {"rank_sample_method": "1", "rank_sample_gap": "1.5"}),
    ("posneg_thresholds", True, {"pos_sample_lowerb": "4", "neg_sample_upperb": "2"}),
]
RANK_SAMPLER_SEED = 10   # svd_feature.cpp:293
RANK_SAMPLER_ROUNDS = 2


def rank_blocks(nblocks, num_user, num_item, num_global, seed, graded=False, max_rows=8, side_user=True, max_fb=3):
    """One block per user visit: 0..max_rows candidate rows (label 0/1, or 1..5 when graded) with sorted global and
    item entries (the generator merges sorted lists, apex_svd_data.cpp:828-860), sometimes a second user entry whose
    value is 0 / 1e-7 (dropped, :897-903) or 0.5 (kept), and 0..3 feedback ids.  Some blocks are empty."""
    rng = np.random.default_rng(seed)
    blocks = []
    for b in range(nblocks):
        uid = int(rng.integers(0, num_user - 3))
        nrow = int(rng.integers(0 if b % 9 == 4 else 1, max_rows + 1))
        nfb = int(rng.integers(0, max_fb + 1))
        fbi = np.sort(rng.choice(num_item, size=nfb, replace=False)).astype(np.uint32)
        fbv = np.full(nfb, 1.0 / np.sqrt(max(nfb, 1)), np.float32)
        rows = []
        for _ in range(nrow):
            label = float(rng.integers(1, 6)) if graded else float(rng.integers(0, 2))
            ng = int(rng.integers(0, 3)) if num_global else 0
            g = [(int(x), float(np.float32(rng.uniform(-1, 1)))) for x in np.sort(rng.choice(num_global, size=ng, replace=False))] if ng else []
            u = [(uid, 1.0)]
            if side_user and rng.integers(0, 3) == 0:
                u.append((int(num_user - 1 - rng.integers(0, 3)), float(rng.choice([0.0, 0.5, 1e-7]))))
            ni = int(rng.integers(1, 4))
            it = [(int(x), float(np.float32(rng.choice([1.0, 0.5, rng.uniform(0.1, 1)]))))
                  for x in np.sort(rng.choice(num_item, size=ni, replace=False))]
            rows.append((label, g, u, it))
        data = CSRData.from_rows(rows) if rows else CSRData.empty()
        blocks.append(PlusBlock(fbi, fbv, data, TAG_DEFAULT))
    return blocks


def blocks_digest(blocks):
    """md5 over everything a list of user blocks carries, field by field in block order."""
    import hashlib
    h = hashlib.md5()
    for b in blocks:
        h.update(np.int32(b.extend_tag).tobytes())
        h.update(np.ascontiguousarray(b.index_ufeedback, np.uint32).tobytes())
        h.update(np.ascontiguousarray(b.value_ufeedback, np.float32).tobytes())
        h.update(np.ascontiguousarray(b.data.row_ptr, np.int64).tobytes())
        h.update(np.ascontiguousarray(b.data.row_label, np.float32).tobytes())
        h.update(np.ascontiguousarray(b.data.feat_index, np.uint32).tobytes())
        h.update(np.ascontiguousarray(b.data.feat_value, np.float32).tobytes())
    return h.hexdigest()


RANK_E2E_CONF = [   # demo/pairwiseRank/pairwiseRank.conf shape, shrunk
    ("learning_rate", "0.01"), ("wd_item", "0.004"), ("wd_user", "0.004"), ("num_item", "50"), ("num_user", "60"),
    ("num_global", "8"), ("wd_global", "0.001"), ("num_factor", "8"), ("active_type", "3"), ("format_type", "1"),
    ("num_ufeedback", "50"), ("wd_ufeedback", "0.004"), ("ufeedback_init_sigma", "0.01"), ("no_user_bias", "1"), ("input_type", "2"),
]
RANK_E2E_ROUNDS = 3


# ---- rank pairs (BASELINE configs[4]): (user, positive item, negative item) from a planted preference model
PAIR_CONF = [  # demo/pairwiseRank/pairwiseRank.conf: sigmoid rank loss, no user bias
    ("base_score", "0.5"), ("learning_rate", "0.005"), ("wd_item", "0.004"), ("wd_user", "0.004"),
    ("num_global", "0"), ("active_type", "3"), ("no_user_bias", "1"),
]


def planted_pairs(n, num_user, num_item, seed, rank=8):
    """n pairs in random order; the positive item is the one the planted low-rank model scores higher for the user
    (with noise), pos != neg."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, num_user, n, dtype=np.int64)
    a = rng.integers(0, num_item, n, dtype=np.int64)
    b = (a + 1 + rng.integers(0, num_item - 1, n, dtype=np.int64)) % num_item
    pu = rng.standard_normal((num_user, rank)).astype(np.float32)
    qi = rng.standard_normal((num_item, rank)).astype(np.float32)
    sa_ = np.einsum("nk,nk->n", pu[u], qi[a]) + 0.5 * rng.standard_normal(n)
    sb_ = np.einsum("nk,nk->n", pu[u], qi[b])
    first = sa_ > sb_
    pos, neg = np.where(first, a, b), np.where(first, b, a)
    return u.astype(np.uint32), pos.astype(np.uint32), neg.astype(np.uint32)


def pair_accuracy(score_pos_minus_neg):
    """share of held-out pairs ranked the right way round (score difference > 0)"""
    return float(np.mean(np.asarray(score_pos_minus_neg) > 0))


def nested_blocks(num_spans, num_user, num_item, num_ufeedback, seed, max_depth=3, max_rows=4, max_fb=5):
    """User-group blocks with NESTED START..END spans, the input of the multi-level implicit-feedback solver (extend_type 2,
    solvers/multi-imfb/apex_multi_imfb.h:173-192): START pushes a feedback level, END pops and scatters it, DEFAULT blocks
    may sit inside an open span (push + pop at once), MIDDLE blocks carry rows only.  Top-level spans belong to one user."""
    rng = np.random.default_rng(seed)
    blocks = []
    e = np.zeros(0, np.uint32), np.zeros(0, np.float32)

    def fb():
        n = int(rng.integers(0, max_fb + 1))
        idx = np.sort(rng.choice(num_ufeedback, size=n, replace=False)).astype(np.uint32)
        return idx, np.full(n, 1.0 / np.sqrt(max(n, 1)), np.float32)

    def rows(uid):
        n = int(rng.integers(0, max_rows + 1))
        rs = [(float(rng.integers(1, 6)), [], [(int(uid), 1.0)], [(int(rng.integers(0, num_item)), 1.0)]) for _ in range(n)]
        return CSRData.from_rows(rs) if rs else CSRData.empty()

    def span(uid, depth):
        kind = int(rng.integers(0, 3))
        if kind == 0 or depth >= max_depth:
            f = fb()
            blocks.append(PlusBlock(f[0], f[1], rows(uid), TAG_DEFAULT))
            return
        f = fb()
        blocks.append(PlusBlock(f[0], f[1], rows(uid), TAG_START))
        for _ in range(int(rng.integers(0, 3))):
            if rng.integers(0, 2):
                span(uid, depth + 1)
            else:
                blocks.append(PlusBlock(e[0], e[1], rows(uid), TAG_MIDDLE))
        blocks.append(PlusBlock(f[0], f[1], rows(uid), TAG_END))

    for _ in range(num_spans):
        span(int(rng.integers(0, num_user)), 0)
    return blocks


# ---- ranker input (svdranker_tag, apex_svd.h:115-152): the tag travels in the label field
def ranker_stream(num_cand, num_sections, num_user, num_item, num_global, seed, extra_item_feats=True, spec=True):
    """(item lines, [user sections]) for ISVDRanker::process: ITEM lines (tag 0) define the candidates -- one or two item
    features and sometimes global features each --, every section is USER (2), POS (1), BAN (-1), some SPEC (3) and PROCESS (4)."""
    rng = np.random.default_rng(seed)
    items = []
    for c in range(num_cand):
        it = [(int(c % num_item), 1.0)]
        if extra_item_feats and rng.integers(0, 3) == 0:
            it.append((int(rng.integers(0, num_item)), float(np.float32(rng.uniform(0.2, 1.0)))))
            it.sort(key=lambda e: e[0])
            if it[0][0] == it[1][0]:
                it = it[:1]
        g = [(int(rng.integers(0, num_global)), float(np.float32(rng.uniform(-1, 1))))] if (num_global and rng.integers(0, 4) == 0) else []
        items.append((0.0, g, [], it))
    sections = []
    for s in range(num_sections):
        rows = []
        u = [(int(rng.integers(0, num_user)), 1.0)]
        if rng.integers(0, 3) == 0:
            u.append((int(rng.integers(0, num_user)), float(np.float32(rng.uniform(0.1, 1.0)))))
        rows.append((2.0, [], u, []))
        chosen = rng.choice(num_cand, size=min(num_cand, 6), replace=False)
        npos = int(rng.integers(1, 4))
        rows.append((1.0, [], [(int(x), 1.0) for x in chosen[:npos]], []))
        if rng.integers(0, 2):
            rows.append((-1.0, [], [(int(x), 1.0) for x in chosen[npos:npos + 2]], []))
        if spec:
            for x in chosen[4:6]:
                g = [(int(rng.integers(0, num_global)), float(np.float32(rng.uniform(-1, 1))))] if num_global else []
                it = [(int(rng.integers(0, num_item)), float(np.float32(rng.uniform(0.1, 0.5))))] if rng.integers(0, 2) else []
                rows.append((3.0, g, [(int(x), 1.0)], it))
        rows.append((4.0, [], [], []))
        sections.append(CSRData.from_rows(rows))
    return CSRData.from_rows(items), sections
