"""Synthetic workloads of bench.py (BASELINE.json configs[1] / [3] / [4]): planted low-rank ratings, rank pairs, user-grouped SVD++
blocks, neighbourhood rows with global features.  No network: every number of the bench line is on data drawn here, seeds in the arguments.
The streams of one process are cached (the N > 1 secondaries train the SAME data as the main line)."""
import os

import numpy as np

class Planted:
    """Planted low-rank preference model: ratings 1..5 = clip(round(3 + b_u + b_i + <p_u,q_i>/2 + noise))."""

    def __init__(self, num_user, num_item, rng, rank=4):
        self.rank = rank
        self.pu = rng.standard_normal((num_user, rank)).astype(np.float32)
        self.qi = rng.standard_normal((num_item, rank)).astype(np.float32)
        self.bu = (0.3 * rng.standard_normal(num_user)).astype(np.float32)
        self.bi = (0.3 * rng.standard_normal(num_item)).astype(np.float32)

    def score(self, u, i, chunk=10_000_000):
        out = np.empty(len(u), np.float32)
        for s in range(0, len(u), chunk):
            uu, ii = u[s:s + chunk], i[s:s + chunk]
            out[s:s + chunk] = 3.0 + self.bu[uu] + self.bi[ii] + 0.5 * np.einsum("nk,nk->n", self.pu[uu], self.qi[ii]) / np.sqrt(self.rank)
        return out

    def rate(self, u, i, rng, noise=0.35, chunk=10_000_000):
        r = np.empty(len(u), np.float32)
        for s in range(0, len(u), chunk):
            e = min(len(u), s + chunk)
            sc = self.score(u[s:e], i[s:e]) + noise * rng.standard_normal(e - s).astype(np.float32)
            r[s:e] = np.clip(np.rint(sc), 1, 5)
        return r


_DATA_CACHE = {}   # the synthetic streams of this process: the N > 1 secondaries train the SAME data as the main line


def cached(fn, *args):
    key = (fn.__name__,) + args
    if key not in _DATA_CACHE:
        _DATA_CACHE[key] = fn(*args)
    return _DATA_CACHE[key]


def synth_triples(n, num_user, num_item, seed=12345, rank=4, noise=0.35, chunk=10_000_000):
    """(user, item, rating): u, i uniform; rating in 1..5 from a planted low-rank model + noise so that
    RMSE is meaningful."""
    cache = os.environ.get("SVDF_BENCH_DATA_CACHE")   # the --pmc children read the parent's stream instead of drawing it again
    path = os.path.join(cache, "triples_%d_%d_%d_%d.npz" % (n, num_user, num_item, seed)) if cache else None
    if path and os.path.exists(path):
        z = np.load(path)
        return z["u"], z["i"], z["r"]
    rng = np.random.default_rng(seed)
    u = rng.integers(0, num_user, n, dtype=np.uint32)
    i = rng.integers(0, num_item, n, dtype=np.uint32)
    pl = Planted(num_user, num_item, rng, rank)
    r = np.empty(n, np.float32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        score = pl.score(u[s:e], i[s:e]) + noise * rng.standard_normal(e - s).astype(np.float32)
        r[s:e] = np.clip(np.rint(score), 1, 5)
    if path and os.environ.get("SVDF_BENCH_DATA_CACHE_WRITE") == "1":
        np.savez(path, u=u, i=i, r=r)
    return u, i, r


def synth_pairs(n, num_user, num_item, seed=777, chunk=10_000_000):
    """BASELINE configs[4] / SURVEY 8d2 C5: n (user, positive item, negative item) rank pairs in uniform random order;
    the positive item is the one the planted model (+ noise) scores higher, pos != neg."""
    rng = np.random.default_rng(seed)
    u = rng.integers(0, num_user, n, dtype=np.uint32)
    a = rng.integers(0, num_item, n, dtype=np.uint32)
    b = rng.integers(0, num_item - 1, n, dtype=np.uint32)
    b = ((a.astype(np.int64) + 1 + b) % num_item).astype(np.uint32)
    pl = Planted(num_user, num_item, rng)
    pos, neg = np.empty(n, np.uint32), np.empty(n, np.uint32)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        sa_ = pl.score(u[s:e], a[s:e]) + 0.35 * rng.standard_normal(e - s).astype(np.float32)
        first = sa_ > pl.score(u[s:e], b[s:e])
        pos[s:e] = np.where(first, a[s:e], b[s:e])
        neg[s:e] = np.where(first, b[s:e], a[s:e])
    return u, pos, neg


def synth_user_blocks(num_blocks, per_user, num_user, num_item, seed=4242):
    """BASELINE configs[3] implicitFeedback: user-grouped blocks, each user's rows (per_user ratings of uniformly drawn items)
    + that user's feedback set = the items it rated, value n_u^-1/2 (demo/implicitFeedback/mkimplicitfeedbackfeature.py:46-55),
    users in random order.  Returns (BlockArrays train, BlockArrays held-out: the same users' feedback + 2 fresh rows)."""
    from svdfeature_amd import BlockArrays
    rng = np.random.default_rng(seed)
    users = rng.permutation(num_user)[:num_blocks].astype(np.uint32)
    n = num_blocks * per_user
    u = np.repeat(users, per_user)
    i = rng.integers(0, num_item, n, dtype=np.uint32)
    pl = Planted(num_user, num_item, rng)
    r = pl.rate(u, i, rng)
    # feedback set of a block = its distinct items, sorted: a row-wise sort of the (block, per_user) item matrix, duplicates masked out (the same
    # arrays np.unique(block * num_item + item) gives, without a 100 M-key sort at the configs[3] size of round 5: 1 M users)
    srt = np.sort(i.reshape(num_blocks, per_user), axis=1)
    keep = np.ones(srt.shape, dtype=bool)
    keep[:, 1:] = srt[:, 1:] != srt[:, :-1]
    fb_idx = srt[keep].astype(np.uint32)
    fb_cnt = keep.sum(axis=1).astype(np.int64)
    del srt, keep
    fb_ptr = np.concatenate([[0], np.cumsum(fb_cnt)]).astype(np.int64)
    fb_val = (1.0 / np.sqrt(np.repeat(fb_cnt, fb_cnt))).astype(np.float32)

    def rows(uu, ii, rr, per):
        m = len(rr)
        ptr = np.empty(3 * m + 1, np.int64)
        base = 2 * np.arange(m, dtype=np.int64)
        ptr[0:3 * m:3] = base; ptr[1:3 * m:3] = base; ptr[2:3 * m:3] = base + 1; ptr[3 * m] = 2 * m
        idx = np.empty(2 * m, np.uint32); idx[0::2] = uu; idx[1::2] = ii
        return BlockArrays(np.zeros(num_blocks, np.int32), fb_ptr, fb_idx, fb_val, per * np.arange(num_blocks + 1, dtype=np.int64),
                           rr, ptr, idx, np.ones(2 * m, np.float32))
    train = rows(u, i, r, per_user)
    tu = np.repeat(users, 2)
    ti = rng.integers(0, num_item, len(tu), dtype=np.uint32)
    test = rows(tu, ti, pl.rate(tu, ti, rng), 2)
    return train, test


def synth_neighbourhood(n, num_user, num_item, num_global, ng, seed=99):
    """BASELINE configs[3] neighborhoodModel shape: (user, item, rating) + ng global features per instance drawn from
    num_global ids with values U(0,1) (demo/neighborhoodModel: k-NN style global weights), distinct ids inside an instance."""
    from svdfeature_amd import CSRData
    rng = np.random.default_rng(seed)
    u = rng.integers(0, num_user, n, dtype=np.uint32)
    i = rng.integers(0, num_item, n, dtype=np.uint32)
    pl = Planted(num_user, num_item, rng)
    r = pl.rate(u, i, rng)
    g = rng.integers(0, num_global - ng, (n, ng), dtype=np.uint32)
    g.sort(axis=1)
    g += np.arange(ng, dtype=np.uint32)[None, :]     # strictly increasing -> distinct
    per = ng + 2
    ptr = np.empty(3 * n + 1, np.int64)
    base = per * np.arange(n, dtype=np.int64)
    ptr[0:3 * n:3] = base; ptr[1:3 * n:3] = base + ng; ptr[2:3 * n:3] = base + ng + 1; ptr[3 * n] = per * n
    idx = np.empty((n, per), np.uint32); idx[:, :ng] = g; idx[:, ng] = u; idx[:, ng + 1] = i
    val = np.ones((n, per), np.float32); val[:, :ng] = rng.uniform(0, 1, (n, ng))
    return CSRData(r, ptr.astype(np.int32), idx.ravel(), val.ravel())
