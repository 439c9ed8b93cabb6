"""Oracle-backed adaptor + single-process simulator for the user-sharded window-synchronous SGD of
svdfeature_amd/multi_gpu.py (test infrastructure)."""
import numpy as np

import cases
from oracle import oracle
from svdfeature_amd import BlockArrays, CSRData, pairs_as_csr
from svdfeature_amd.multi_gpu import Pairs, defer_tails, shard_block_windows, shard_csr_windows, shard_pair_windows, shard_windows


CONTRIB_BF16 = False   # tests of `amd:contrib = bf16` set this around simulate*(): the oracles round row contributions to bfloat16


def make_oracle(conf, seed=10, fmt=0, active=0):
    t = oracle.OracleTrainer("port", fmt, active)
    if CONTRIB_BF16:
        t.set_stale_rounding(True)
    t.seed(seed)
    for k, v in conf:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    return t


class OracleShard:
    """multi_gpu adaptor protocol on top of the C oracle; deltas travel as torch CPU tensors (gloo)."""

    def __init__(self, trainer, torch=None, parts=1, minibatch=False):
        self.t, self.torch = trainer, torch
        self.snap = None
        self.parts = parts
        self.apply_refreshes_snapshot = parts > 1   # piece-wise exchange keeps one running snapshot per pass
        # window-minibatch mode (multi_gpu.ShardedTrainer(minibatch=True)): train() leaves the replicated side untouched and
        # collects its change (oracle.update_batch_stale); delta_get returns it; delta_set ADDS the all-reduced sum
        self.minibatch = minibatch
        self.mb_delta = None

    def make_windows(self, shards):
        out = []
        for sh in shards:
            if isinstance(sh, list):   # item-range pieces of one window
                out.append([CSRData.from_triples(*piece) for piece in sh])
                continue
            if isinstance(sh, BlockArrays):
                out.append(sh.to_blocks())
            elif isinstance(sh, CSRData):
                out.append(sh)
            elif isinstance(sh, Pairs):
                out.append(pairs_as_csr(sh.user, sh.pos, sh.neg))
            else:
                out.append(CSRData.from_triples(*sh))
        return out

    def _mb_flat(self):
        """the window's deltas in the packed order of the replicated ranges (SHARED)"""
        return np.concatenate([self.mb_delta[name].ravel() for name, _ in self._views()])

    def train(self, d):
        if self.minibatch:
            if isinstance(d, list):   # user-group blocks: svdo_update_block_stale, block after block
                delta = self.t.stale_delta_zero()
                for b in d:
                    delta = self.t.update_block_stale(b, delta)
                self.mb_delta = dict(zip(("W_item", "i_bias", "g_bias", "W_ufeedback", "ufeedback_bias"), delta))
            else:
                self.mb_delta = dict(zip(("W_item", "i_bias", "g_bias"), self.t.update_batch_stale(d)))
        elif isinstance(d, list):
            for b in d:
                self.t.update_block(b)
        else:
            self.t.update_batch(d)

    SHARED = ("W_ufeedback", "W_item", "ufeedback_bias", "i_bias", "g_bias")   # every replicated range (svdf_engine.cpp: shared_ranges)

    def _views(self):
        out = []
        for name in self.SHARED:
            v = self.t.view(name)
            if v is not None and v.size:
                out.append((name, v))
        return out

    def _shared(self):
        return np.concatenate([v.ravel() for _, v in self._views()])

    # ---- stratified schedule (multi_gpu.StratifiedTrainer)
    def _write(self, new):
        off = 0
        for name, v in self._views():
            self.t.set_view(name, new[off:off + v.size])
            off += v.size

    def apply_local(self, d, block, nblocks):
        delta = self._mb_flat()
        pos = self._piece(block, nblocks)
        cur = self._shared()
        cur[pos] = cur[pos] + delta[pos]
        self._write(cur)

    def block_get(self, block, nblocks):
        v = np.ascontiguousarray(self._shared()[self._piece(block, nblocks)])
        return self.torch.from_numpy(v) if self.torch is not None else v

    def block_like(self, block, nblocks):
        v = np.zeros(len(self._piece(block, nblocks)), np.float32)
        return self.torch.from_numpy(v) if self.torch is not None else v

    def block_set(self, block, nblocks, tensor):
        v = tensor.numpy() if hasattr(tensor, "numpy") else tensor
        cur = self._shared()
        cur[self._piece(block, nblocks)] = v
        self._write(cur)

    def handoff_start(self, dist, out, dst, inc, src):
        return dist.batch_isend_irecv([dist.P2POp(dist.isend, out, dst), dist.P2POp(dist.irecv, inc, src)])

    def handoff_wait(self, reqs):
        for q in reqs:
            q.wait()

    def broadcast(self, dist, buf, src):
        dist.broadcast(buf, src)

    def delta_begin(self):
        if self.minibatch:
            return   # the replicated side does not move inside a window: it is its own snapshot
        self.snap = self._shared()

    def _piece(self, part, parts=None):
        """flat positions of item-range piece `part` in the packed layout (svdf_item_delta_select): the item rows of W_item
        and i_bias in [num_item*part/parts, num_item*(part+1)/parts); everything else travels with piece 0"""
        parts = parts or self.parts
        pos, off = [], 0
        for name, v in self._views():
            if name in ("W_item", "i_bias"):
                ni = v.shape[0]
                lo, hi = ni * part // parts, ni * (part + 1) // parts
                width = v.size // ni
                pos.append(np.arange(off + lo * width, off + hi * width))
            elif part == 0:
                pos.append(np.arange(off, off + v.size))
            off += v.size
        return np.concatenate(pos)

    def delta_get(self, part=None):
        if self.minibatch:
            d = self._mb_flat()   # the SHARED order: [W_ufeedback] W_item [ufeedback_bias] i_bias [g_bias]
            if part is not None:
                d = np.ascontiguousarray(d[self._piece(part)])
            return self.torch.from_numpy(d) if self.torch is not None else d
        d = self._shared() - self.snap
        if part is not None:
            d = np.ascontiguousarray(d[self._piece(part)])
        return self.torch.from_numpy(d) if self.torch is not None else d

    def delta_set(self, d, part=None):
        d = d.numpy() if hasattr(d, "numpy") else d
        cur = self._shared()
        if self.minibatch:
            if part is not None:
                new = cur
                pos = self._piece(part)
                new[pos] = cur[pos] + d
            else:
                new = cur + d
            off = 0
            for name, v in self._views():
                self.t.set_view(name, new[off:off + v.size])
                off += v.size
            return
        if part is not None:
            pos = self._piece(part)
            new = cur
            new[pos] = self.snap[pos] + d
            self.snap[pos] = new[pos]      # the piece's snapshot moves along (apply_refreshes_snapshot semantics)
        else:
            new = self.snap + d
        off = 0
        for name, v in self._views():
            self.t.set_view(name, new[off:off + v.size])
            off += v.size


def simulate_parts(conf, u, i, r, world, windows, passes, parts, num_item, seed=10, minibatch=False):
    """The piece-wise exchange (ShardedTrainer(parts=p)) run synchronously in one process: per window every rank trains
    piece 0 then exchanges it, trains piece 1 then exchanges it, ...  The overlapped schedule computes the same values."""
    from svdfeature_amd.multi_gpu import shard_windows_parts
    ranks = [OracleShard(make_oracle(conf, seed), parts=parts, minibatch=minibatch) for _ in range(world)]
    wins = [a.make_windows(shard_windows_parts(u, i, r, rk, world, windows, num_item, parts)) for rk, a in enumerate(ranks)]
    for _ in range(passes):
        for w in range(windows):
            if w == 0:
                for a in ranks:
                    a.delta_begin()
            for part in range(parts):
                for rk, a in enumerate(ranks):
                    a.train(wins[rk][w][part])
                total = None
                for a in ranks:
                    d = a.delta_get(part)
                    total = d.copy() if total is None else total + d
                for a in ranks:
                    a.delta_set(total, part)
    return ranks


def simulate(conf, u, i, r, world, windows, passes, seed=10, defer=0.0, fmt=0, active=0, minibatch=False):
    """All ranks in one process, all-reduce replaced by an explicit sum in rank order.  defer > 0: the window
    lists go through multi_gpu.defer_tails like bench.py's (needs num_user / num_item in conf).
    u: user column of triples (u, i, r), of rank pairs (u = Pairs) or a BlockArrays (user-group pass; fmt = 1)."""
    ranks = [OracleShard(make_oracle(conf, seed, fmt, active), minibatch=minibatch) for _ in range(world)]
    if isinstance(u, BlockArrays):
        shards = [shard_block_windows(u, rk, world, windows) for rk in range(world)]
    elif isinstance(u, CSRData):
        shards = [shard_csr_windows(u, rk, world, windows) for rk in range(world)]
    elif isinstance(u, Pairs):
        shards = [shard_pair_windows(u.user, u.pos, u.neg, rk, world, windows) for rk in range(world)]
    else:
        shards = [shard_windows(u, i, r, rk, world, windows) for rk in range(world)]
    if defer > 0 and world > 1:
        c = dict(conf)
        shards = [defer_tails(sh, int(c["num_user"]), int(c["num_item"]), defer) for sh in shards]
    wins = [a.make_windows(sh) for a, sh in zip(ranks, shards)]
    for _ in range(passes):
        for w in range(windows):
            if world == 1 and not minibatch:
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


def simulate_stratified(conf, u, i, r, world, chunks, passes, num_item, per_item=32.0, seed=10, blocks_per_rank=1):
    """multi_gpu.StratifiedTrainer with all ranks in one process: block hand-overs are array copies (a block trained by rank r + 1 in
    step t reaches rank r before its step t + P).  Returns the rank adaptors, the item side completed everywhere (gather_blocks)."""
    from svdfeature_amd.multi_gpu import stratified_plan
    P = blocks_per_rank
    B = world * P
    ranks = [OracleShard(make_oracle(conf, seed), minibatch=True) for _ in range(world)]
    plans = [[[a.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, rk, world, chunks, num_item, per_item, P)] for rk, a in enumerate(ranks)]
    for _ in range(passes):
        for c in range(chunks):
            for t in range(B):
                for rk, a in enumerate(ranks):
                    b = (rk * P + t) % B
                    for w in plans[rk][c][t]:
                        a.train(w)
                        a.apply_local(w, b, B)
                if world > 1:   # copies may land at once: the receiver does not touch the block before step t + P
                    outs = [a.block_get((rk * P + t) % B, B).copy() for rk, a in enumerate(ranks)]
                    for rk, a in enumerate(ranks):
                        a.block_set(((rk + 1) * P + t) % B, B, outs[(rk + 1) % world])
    for b in range(B):
        blk = ranks[b // P].block_get(b, B).copy()
        for rk, a in enumerate(ranks):
            if rk != b // P:
                a.block_set(b, B, blk)
    return ranks
