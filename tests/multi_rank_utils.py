"""Oracle-backed adaptor + single-process simulator for the user-sharded window-synchronous SGD of
svdfeature_amd/multi_gpu.py (test infrastructure)."""
import numpy as np

import cases
from oracle import oracle
from svdfeature_amd import CSRData
from svdfeature_amd.multi_gpu import defer_tails, shard_windows


def make_oracle(conf, seed=10):
    t = oracle.OracleTrainer("port", 0, 0)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


class OracleShard:
    """multi_gpu adaptor protocol on top of the C oracle; deltas travel as torch CPU tensors (gloo)."""

    def __init__(self, trainer, torch=None):
        self.t, self.torch = trainer, torch
        self.snap = None

    def make_windows(self, shards):
        return [CSRData.from_triples(u, i, r) for (u, i, r) in shards]

    def train(self, d):
        self.t.update_batch(d)

    def _shared(self):
        return np.concatenate([self.t.view("W_item").ravel(), self.t.view("i_bias").ravel()])

    def delta_begin(self):
        self.snap = self._shared()

    def delta_get(self):
        d = self._shared() - self.snap
        return self.torch.from_numpy(d) if self.torch is not None else d

    def delta_set(self, d):
        d = d.numpy() if hasattr(d, "numpy") else d
        new = self.snap + d
        w = self.t.view("W_item")
        self.t.set_view("W_item", new[:w.size])
        self.t.set_view("i_bias", new[w.size:])


def simulate(conf, u, i, r, world, windows, passes, seed=10, defer=0.0):
    """All ranks in one process, all-reduce replaced by an explicit sum in rank order.  defer > 0: the window
    lists go through multi_gpu.defer_tails like bench.py's (needs num_user / num_item in conf)."""
    ranks = [OracleShard(make_oracle(conf, seed)) for _ in range(world)]
    shards = [shard_windows(u, i, r, rk, world, windows) for rk in range(world)]
    if defer > 0 and world > 1:
        c = dict(conf)
        shards = [defer_tails(sh, int(c["num_user"]), int(c["num_item"]), defer) for sh in shards]
    wins = [a.make_windows(sh) for a, sh in zip(ranks, shards)]
    for _ in range(passes):
        for w in range(windows):
            if world == 1:
                ranks[0].train(wins[0][w])
                continue
            for rk, a in enumerate(ranks):
                a.delta_begin()
                a.train(wins[rk][w])
            total = None
            for a in ranks:
                d = a.delta_get()
                total = d.copy() if total is None else total + d
            for a in ranks:
                a.delta_set(total)
    return ranks


def merged_predict(ranks, world, tu, ti, tr):
    """Predictions for test triples using each user's owning rank."""
    out = np.zeros(len(tr), np.float32)
    for rk, a in enumerate(ranks):
        m = (tu % world) == rk
        if m.any():
            out[m] = a.t.predict_batch(CSRData.from_triples(tu[m], ti[m], tr[m]))
    return out
