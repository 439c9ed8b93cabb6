"""User-sharded multi-GPU training of the apex_svd SGD path.

The reference is single-process; this exchange step is new design, not a translation:

* instances are partitioned BY USER (rank = user % world), so W_user / u_bias rows are touched by
  exactly one rank and never exchanged;
* the same holds for rank pairs (Pairs: user, positive item, negative item -- BASELINE configs[4]) and for user-group data
  (data.BlockArrays: a block belongs to its user's rank; SVD++ state lives inside a user's blocks, so it never crosses
  ranks) -- shard_pair_windows / shard_block_windows;
* item-side parameters (W_item, i_bias, g_bias [, W_ufeedback, ufeedback_bias]) are replicated; a pass over the data
  is cut into `windows` windows; inside a window every rank runs its own exact, conflict-free
  sequential SGD on its shard, then the item-side DELTAS of the window are summed over ranks with
  ONE all-reduce (RCCL over xGMI: one contiguous fp32 buffer of num_item*(k+1)+num_global floats)
  and every rank sets item_side = snapshot + sum(deltas).

With world == 1 the exchange is skipped and the result is the reference's sequential result bit
for bit.  With world > 1 item updates inside a window are computed against parameters that are up
to one window stale on the other ranks' contributions, so the acceptance bar is the RMSE tolerance
(|dRMSE| <= 1e-4 after equal passes), not bit parity; `windows` trades staleness for exchange cost.

The class is engine-agnostic: it drives an adaptor with train/delta methods, so the same code runs
on MI355X ranks (HipShard, RCCL) and in the world_size-2 gloo tests on CPU.
"""
from collections import namedtuple

import numpy as np

# rank pairs (user, positive item, negative item): the instance stream of BASELINE configs[4]
Pairs = namedtuple("Pairs", ["user", "pos", "neg"])


def shard_by_user(user, item, label, rank, world):
    """This rank's instances (user % world == rank), file order preserved."""
    if world == 1:
        return user, item, label
    m = (np.asarray(user) % world) == rank
    return user[m], item[m], label[m]


def window_bounds(n_total, windows):
    """Window w covers GLOBAL instance positions [b[w], b[w+1]) of the unsharded stream, so all ranks
    cut at the same points of the file."""
    return [(n_total * w) // windows for w in range(windows + 1)]


def shard_windows(user, item, label, rank, world, windows):
    """List of (u, i, r) per window for this rank."""
    n = len(label)
    b = window_bounds(n, windows)
    out = []
    for w in range(windows):
        s = slice(b[w], b[w + 1])
        out.append(shard_by_user(user[s], item[s], label[s], rank, world))
    return out


def split_by_item_range(user, item, label, num_item, parts):
    """One shard cut into `parts` pieces by item id range (piece p: items [num_item*p/parts, num_item*(p+1)/parts)), order kept.
    The pieces of a window touch disjoint item rows, so the exchange of one piece can overlap with training on another."""
    if parts == 1:
        return [(user, item, label)]
    piece = (np.asarray(item, np.int64) * parts) // max(int(num_item), 1)
    return [(user[piece == p], item[piece == p], label[piece == p]) for p in range(parts)]


def shard_windows_parts(user, item, label, rank, world, windows, num_item, parts):
    """List over windows of a list over item-range pieces of (u, i, r) for this rank."""
    return [split_by_item_range(u, i, r, num_item, parts) for (u, i, r) in shard_windows(user, item, label, rank, world, windows)]


def shard_pair_windows(user, pos, neg, rank, world, windows):
    """Rank pairs sharded like triples: this rank's pairs (user % world == rank) of every global window, order kept."""
    b = window_bounds(len(user), windows)
    out = []
    for w in range(windows):
        s = slice(b[w], b[w + 1])
        out.append(Pairs(*shard_by_user(user[s], pos[s], neg[s], rank, world)))
    return out


def shard_csr_windows(d, rank, world, windows):
    """General rows (data.CSRData with one user entry per row: global features, several item entries) sharded like triples: this rank's
    rows (user % world == rank) of every global window, order kept."""
    b = window_bounds(d.num_row, windows)
    owner = d.row_user() % np.uint32(world)
    idx = np.arange(d.num_row)
    return [d.select_rows((owner == rank) & (idx >= b[w]) & (idx < b[w + 1])) for w in range(windows)]


def block_window_bounds(ba, windows):
    """Window cuts of a user-group pass (data.BlockArrays) at BLOCK positions, the same on every rank: the even cut
    points are moved forward to the next position where no START..END span is open."""
    closed = ba.span_closed_before()
    nb = ba.num_block
    out = [0]
    for w in range(1, windows):
        p = max((nb * w) // windows, out[-1])
        while p < nb and not closed[p]:
            p += 1
        out.append(p)
    out.append(nb)
    return out


def shard_block_windows(ba, rank, world, windows):
    """User-group data (SVDPlusBlock streams, apex_svd_data.h:376-466): a block belongs to the rank of its user
    (user % world; a START..END span follows its START block), windows cut at global block positions.  Every rank keeps
    its blocks in file order, so per rank this is the reference's update(block) loop on a sub-stream."""
    b = block_window_bounds(ba, windows)
    owner = ba.block_user() % np.uint32(world)
    idx = np.arange(ba.num_block)
    out = []
    for w in range(windows):
        out.append(ba.select((owner == rank) & (idx >= b[w]) & (idx < b[w + 1])))
    return out


def item_block_bounds(num_item, world):
    """item block b = items [bounds[b], bounds[b + 1]) (the partition of svdf_item_delta_select)"""
    return [(int(num_item) * b) // world for b in range(world + 1)]


PER_ITEM_MAX = 128.0   # the most updates ANY item row may meet per window (svdf_wunit.cpp: window_per_target_max; calibrated on Zipf(0.7) items: 512 diverges)


def stratum_num_windows(items, lo, nblk, per_item):
    """windows of one stratum: `per_item` updates per item of the block on average -- over the block's ids for uniform catalogues (the
    round-3 rule), over the stratum's ENTRIES (sum c^2 / sum c) when popular items dominate -- and never more than PER_ITEM_MAX for any single
    item (round 5: with the id-average alone the schedule diverged to NaN on Zipf(0.7) items at the configs[2] size)."""
    m = len(items)
    if m == 0:
        return 1
    c = np.bincount(np.asarray(items, dtype=np.int64) - int(lo), minlength=int(nblk)).astype(np.float64)
    need = max(m / float(nblk) / float(per_item), float((c * c).sum()) / float(m) / (2.0 * float(per_item)), float(c.max()) / PER_ITEM_MAX)
    return max(1, int(np.ceil(need)))


def stratum_windows(user, item, label, rank, world, step, num_item, per_item=32.0, blocks_per_rank=1):
    """STRATIFIED schedule (DSGD-style): the instances of stratum (user block `rank`, item block (rank * P + step) % (world * P)) of one
    chunk, in file order, cut into windows of at most `per_item` updates per item of the block (P = blocks_per_rank).  Strata of one step
    share neither users nor items, so every rank trains its stratum against an item block it owns exclusively: no sum over ranks, the
    block is handed on afterwards.  Returns a list of (u, i, r) windows (possibly empty arrays, never an empty list)."""
    nblocks = world * blocks_per_rank
    b = (rank * blocks_per_rank + step) % nblocks
    bounds = item_block_bounds(num_item, nblocks)
    it = np.asarray(item)
    m = (it >= bounds[b]) & (it < bounds[b + 1])
    if world > 1:
        m &= (np.asarray(user) % world) == rank
    su, si, sr = user[m], item[m], label[m]
    nblk = max(bounds[b + 1] - bounds[b], 1)
    nwin = stratum_num_windows(si, bounds[b], nblk, per_item)
    cuts = [(len(sr) * w) // nwin for w in range(nwin + 1)]
    return [(su[cuts[w]:cuts[w + 1]], si[cuts[w]:cuts[w + 1]], sr[cuts[w]:cuts[w + 1]]) for w in range(nwin)]


def defer_tails(windows, num_user, num_item, min_frac=0.05):
    """Move the short tail of every window's conflict-free batch sequence into the next window.

    Inside a window a rank's instances form batches of rapidly shrinking size (a window of 390 K ratings: 55 K, 46 K,
    ... 2.4 K, 1.3 K, 582, 235, 94, 40, 13, 2): the last third of the launches carries 1 % of the instances and each
    costs the same ~8 us of launch latency as a full one.  Those instances (batches smaller than min_frac of the
    window's largest, at the end of the sequence) are handed to the next window instead, where they come first in
    file order; the last window of a pass keeps its tail, so every instance is still used exactly once per pass.
    Per rank this is exact SGD on a slightly permuted stream (an instance moves by at most one window); it only
    applies to multi-rank runs, whose acceptance bar is the RMSE tolerance anyway.  windows: [(u, i, r)] of ONE rank."""
    from . import schedule_resources
    out, carry = [], None
    for w, win in enumerate(windows):
        pairs = isinstance(win, Pairs)
        u, i, r = win
        if carry is not None and len(carry[2]):
            u, i, r = np.concatenate([carry[0], u]), np.concatenate([carry[1], i]), np.concatenate([carry[2], r])
        carry = None
        n = len(r)
        wrap = (lambda a, b, c: Pairs(a, b, c)) if pairs else (lambda a, b, c: (a, b, c))
        if w == len(windows) - 1 or n == 0 or min_frac <= 0:
            out.append(wrap(u, i, r))
            continue
        if pairs:   # three rows per pair: the user's and both items'
            res = np.empty(3 * n, np.uint32)
            res[0::3] = u
            res[1::3] = np.asarray(i, np.uint32) + np.uint32(num_user)
            res[2::3] = np.asarray(r, np.uint32) + np.uint32(num_user)
            order, level_ptr = schedule_resources(3 * np.arange(n + 1, dtype=np.int64), res, num_user + num_item)
        else:
            res = np.empty(2 * n, np.uint32)
            res[0::2] = u
            res[1::2] = np.asarray(i, np.uint32) + np.uint32(num_user)
            order, level_ptr = schedule_resources(2 * np.arange(n + 1, dtype=np.int64), res, num_user + num_item)
        sizes = np.diff(level_ptr)
        cut = len(sizes)
        floor = min_frac * float(sizes.max())
        while cut > 1 and sizes[cut - 1] < floor:
            cut -= 1
        keep = np.sort(order[:level_ptr[cut]])
        late = np.sort(order[level_ptr[cut]:])
        out.append(wrap(u[keep], i[keep], r[keep]))
        carry = (u[late], i[late], r[late])
    return out


class ShardedTrainer:
    """Runs passes of window-synchronous user-sharded SGD.

    adaptor protocol:
      train(window_handle)      run this rank's exact SGD over one window
      delta_begin()             snapshot the replicated (item-side) parameters
      delta_get() -> tensor     flat fp32 tensor holding current - snapshot (on the collective's device)
      delta_set(tensor)         replicated = snapshot + tensor
    optional:
      set_wire_half(bool)       the adaptor packs / unpacks the wire format itself (delta_get returns fp16)
      all_reduce(dist, tensor)  issue the collective in the adaptor's own stream order
      apply_refreshes_snapshot  delta_set also moves the snapshot, so delta_begin is needed once per pass only
    """

    def __init__(self, adaptor, window_handles, world, dist=None, force_exchange=False, half_delta=False, parts=1):
        self.a, self.windows, self.world, self.dist = adaptor, window_handles, world, dist
        # window-minibatch adaptors (HipShard(minibatch=True)): train() leaves the replicated side as it was at the window start
        # and collects its change, delta_get() returns the per-item sums, delta_set() ADDS the all-reduced sum.  The exchange step
        # is then part of the algorithm (the item side only moves through it), so it runs with one rank as well -- without the
        # collective -- and the result does not depend on the number of ranks beyond the order of fp32 additions.
        self.minibatch = bool(getattr(adaptor, "minibatch", False))
        # parts > 1: every window handle is a LIST of `parts` handles (pieces by item id range, split_by_item_range); the
        # all-reduce of piece p runs while piece p+1 trains.  A piece's item rows are not touched between its pack and its
        # unpack, so the values are those of the synchronous piece-by-piece schedule: no added staleness, only the order of
        # the instances inside a window changes (per rank still exact SGD on a permuted stream, like defer_tails)
        self.parts = parts
        self.force_exchange = force_exchange   # run the exchange even with one rank (plumbing tests)
        # exchange the window deltas as fp16 (parameters and all arithmetic stay fp32): halves the bytes on
        # xGMI; measured RMSE effect at configs[2] density: 5.32e-5 vs 5.33e-5 with fp32 deltas
        self.half_delta = half_delta
        self.native_wire = hasattr(adaptor, "set_wire_half")
        if self.native_wire:
            adaptor.set_wire_half(half_delta)

    def _reduce(self, d):
        if self.world == 1 and self.dist is None:
            return   # one rank, window-minibatch mode: the sum over ranks is the rank's own delta
        if hasattr(self.a, "all_reduce"):
            self.a.all_reduce(self.dist, d)   # ordered on the adaptor's stream, d already in the wire format
        elif self.half_delta:
            h = d.half()
            self.dist.all_reduce(h)
            d.copy_(h)
        else:
            self.dist.all_reduce(d)

    def gather_model(self, rank):
        """Complete the model on every rank (user rows live on their owners during training); call before save_model."""
        if self.world > 1 and hasattr(self.a, "gather_user_side"):
            self.a.gather_user_side(self.dist, rank, self.world)

    def _train_pass_parts(self, mark):
        a = self.a
        keep_snapshot = getattr(a, "apply_refreshes_snapshot", False) or self.minibatch
        pending = None   # (work handle or None, delta buffer, part): an exchange in flight

        def finish(p):
            work, d, part = p
            if work is not None:
                work.wait()          # the adaptor's stream waits for the collective
            mark("allreduce")
            a.delta_set(d, part)
            mark("unpack")

        for wi, w in enumerate(self.windows):
            if self.world == 1 and not self.force_exchange and not self.minibatch:
                for ds in w:
                    a.train(ds)
                mark("compute")
                continue
            if wi == 0 or not keep_snapshot:
                if pending is not None:
                    finish(pending)
                    pending = None
                a.delta_begin()
                mark("pack")
            for part, ds in enumerate(w):
                a.train(ds)
                mark("compute")
                d = a.delta_get(part)
                mark("pack")
                work = a.all_reduce_async(self.dist, d) if (hasattr(a, "all_reduce_async") and self.dist is not None) else self._reduce(d)
                if pending is not None:
                    finish(pending)   # enqueued AFTER this piece's training: the previous collective had all of it to overlap with
                pending = (work, d, part)
        if pending is not None:
            finish(pending)

    def train_pass(self, mark=None):
        """mark(phase): optional callback invoked on the host right after the work of a phase ("compute", "pack", "allreduce",
        "unpack") has been ENQUEUED -- bench.py records a HIP event on the adaptor's stream there, so that the stream time between
        consecutive events can be charged to the phase that ends at the later one."""
        mark = mark or (lambda phase: None)
        if self.parts > 1:
            return self._train_pass_parts(mark)
        keep_snapshot = getattr(self.a, "apply_refreshes_snapshot", False)
        for wi, w in enumerate(self.windows):
            if self.world == 1 and not self.force_exchange and not self.minibatch:
                self.a.train(w)
                mark("compute")
                continue
            if wi == 0 or not keep_snapshot:
                self.a.delta_begin()
                mark("pack")
            self.a.train(w)
            mark("compute")
            if getattr(self.a, "rccl_ready", False):   # the rank's own RCCL communicator, issued from C++ (svdf_rccl.cpp): pack -> ncclAllReduce -> add
                self.a.rccl_allreduce()
                mark("allreduce")
                continue
            if getattr(self.a, "ipc_ready", False):   # direct exchange through IPC-mapped buffers: no collective library (svdf_ipc.cpp)
                self.a.ipc_pack()
                mark("pack")
                self.a.ipc_reduce()
                mark("allreduce")
                self.a.ipc_apply()
                mark("unpack")
                continue
            d = self.a.delta_get()
            mark("pack")
            self._reduce(d)   # SUM over ranks, in place
            mark("allreduce")
            self.a.delta_set(d)
            mark("unpack")


def default_chunks(item, num_item, world):
    """File-order chunks per pass of the stratified schedule when the caller names none.  What the accuracy contract (|dRMSE| <= 1e-4 against the
    sequential pass) is sensitive to is how far the training order strays from the file order, and skewed catalogues are more sensitive than uniform
    ones: Zipf(0.7) items at the full configs[2] size on 8 ranks, 3 data seeds (profiles/r06_contract_zipf_c2.txt): 4 chunks max |d| 9.0e-5, 8: 7.0e-5,
    12: 4.4e-5, 16: 4.3e-5.  Skewed = the hottest item holds >= 16 x the ratings of a mean touched item (the test svdf_pivot.cpp uses): 12 chunks at
    every rank count; uniform streams keep 8 below 8 ranks and 4 from 8 ranks (profiles/r04_contract_seeds.txt: <= 6.3e-5).  Every rank computes
    this from the same item column, so the ranks agree without a message."""
    cnt = np.bincount(np.asarray(item), minlength=int(num_item))
    touched = int((cnt > 0).sum())
    skew = touched > 0 and float(cnt.max()) >= 16.0 * len(item) / touched
    return 12 if skew else (8 if world < 8 else 4)


def stratified_plan(user, item, label, rank, world, chunks, num_item, per_item=32.0, blocks_per_rank=1):
    """plan[c][t] = the windows (u, i, r) this rank trains in step t (0 .. world * P - 1) of file-order chunk c: the chunk's instances whose
    user is in user block `rank` and whose item is in item block (rank * P + t) % (world * P), file order, at most `per_item` updates per
    item per window."""
    n = len(label)
    plan = []
    for c in range(chunks):
        lo, hi = (n * c) // chunks, (n * (c + 1)) // chunks
        plan.append([stratum_windows(user[lo:hi], item[lo:hi], label[lo:hi], rank, world, t, num_item, per_item, blocks_per_rank)
                     for t in range(world * blocks_per_rank)])
    return plan


def stratified_plan_all_ranks(user, item, label, world, chunks, num_item, per_item=32.0, blocks_per_rank=1):
    """stratified_plan for EVERY rank at once (plans[rank][c][t] = windows): one stable sort of a chunk by (rank, step) instead of
    world x steps boolean passes over it -- the single-process simulation of N ranks at full size (tools/contract_seeds.py)."""
    n, P = len(label), int(blocks_per_rank)
    B = world * P
    bounds = np.asarray(item_block_bounds(num_item, B), np.int64)
    plans = [[] for _ in range(world)]
    for c in range(chunks):
        lo, hi = (n * c) // chunks, (n * (c + 1)) // chunks
        u, i, r = user[lo:hi], item[lo:hi], label[lo:hi]
        rk = (np.asarray(u) % world).astype(np.int64) if world > 1 else np.zeros(hi - lo, np.int64)
        ib = np.searchsorted(bounds, np.asarray(i, np.int64), side="right") - 1     # item block of every instance
        step = (ib - rk * P) % B
        key = rk * B + step
        order = np.argsort(key, kind="stable")
        cnt = np.bincount(key, minlength=world * B)
        off = np.concatenate([[0], np.cumsum(cnt)])
        su, si, sr = u[order], i[order], r[order]
        for rank in range(world):
            steps = []
            for t in range(B):
                a, b_ = off[rank * B + t], off[rank * B + t + 1]
                blk = (rank * P + t) % B
                nblk = max(int(bounds[blk + 1] - bounds[blk]), 1)
                m = int(b_ - a)
                nwin = stratum_num_windows(si[a:b_], bounds[blk], nblk, per_item)
                cuts = [a + (m * w) // nwin for w in range(nwin + 1)]
                steps.append([(su[cuts[w]:cuts[w + 1]], si[cuts[w]:cuts[w + 1]], sr[cuts[w]:cuts[w + 1]]) for w in range(nwin)])
            plans[rank].append(steps)
    return plans


class StratifiedTrainer:
    """STRATIFIED window-minibatch schedule (DSGD-style): no all-reduce.

    User block r = users with id % N == r (private to rank r, as everywhere in this module); the items are cut into B = N * P blocks by id
    range (P = blocks_per_rank).  A pass is cut into `chunks` file-order chunks; a chunk is B steps; in step t rank r trains stratum
    (r, (r P + t) % B) of the chunk with the window-minibatch step.  Strata of one step share neither users nor items, so the rank owns
    its item block exclusively: the per-item sums of a window are added to the model in place (adaptor.apply_local).  After the step the
    block goes to rank r - 1, which trains it P steps later, while block (r P + t + P) % B arrives from rank r + 1 for this rank's step
    t + P: with P = 1 the hand-over sits between two steps, with P = 2 it has a whole step of training to hide behind (the transfer is
    started right after the step and only waited for right before the block is trained).  One point-to-point transfer per rank and step
    (NI / B rows: 3.25 MB at configs[2] on 8 ranks with P = 1) against a 13 MB all-reduce per window.  Between passes every rank holds
    its P home blocks r P .. r P + P - 1; gather_blocks() completes the item side everywhere before predictions / a save.

    adaptor protocol: train(window), apply_local(window, block, nblocks), block_get(block, nblocks) -> tensor,
    block_set(block, nblocks, tensor), block_like(block, nblocks) -> empty tensor of that block's size,
    handoff_start(dist, out_tensor, dst_rank, in_tensor, src_rank) -> handle, handoff_wait(handle), broadcast(dist, tensor, src)."""

    minibatch = True

    def __init__(self, adaptor, plan_handles, world, rank, dist=None, blocks_per_rank=1):
        self.a, self.plan, self.world, self.rank, self.dist = adaptor, plan_handles, world, rank, dist
        self.P = int(blocks_per_rank)
        self.pending = {}   # block id -> (handle, incoming tensor): a block on its way to this rank

    def _arrive(self, b):
        if b in self.pending:
            handle, inc = self.pending.pop(b)
            if isinstance(handle, tuple) and handle[0] == "rccl-recv":
                self.a.rccl_arrive(b, self.world * self.P, handle[1])
                return
            self.a.handoff_wait(handle)
            self.a.block_set(b, self.world * self.P, handle if (isinstance(handle, tuple) and handle[0] == "ipc-recv") else inc)

    def train_pass(self, mark=None):
        fused = mark is None and hasattr(self.a, "stratum_step")   # the timed passes: one engine call per stratum step (phase marks need the separate calls)
        mark = mark or (lambda phase: None)
        a, N, P = self.a, self.world, self.P
        B = N * P
        for chunk in self.plan:
            for t in range(B):
                b = (self.rank * P + t) % B
                self._arrive(b)
                mark("allreduce")
                if fused:
                    native = getattr(a, "rccl_ready", False) and (N > 1 or getattr(a, "rccl_self_ring", False))
                    torch_ring = N > 1 and not native and not getattr(a, "ipc_ready", False)
                    out = a.stratum_step(chunk[t], b, B, torch_ring)
                    nxt = (b + P) % B
                    if native:
                        self.pending[nxt] = (a.rccl_handoff(b, nxt, B, (self.rank - 1) % N, (self.rank + 1) % N), None)
                    elif torch_ring:
                        inc = a.block_like(nxt, B)
                        self.pending[nxt] = (a.handoff_start(self.dist, out, (self.rank - 1) % N, inc, (self.rank + 1) % N), inc)
                    elif N > 1 or (getattr(a, "ipc_ready", False) and getattr(a, "ipc_self_ring", False)):   # IPC: the block goes straight into the neighbour's inbox
                        self.pending[nxt] = (a.handoff_start(self.dist, a.block_get(b, B, True), (self.rank - 1) % N, a.block_like(nxt, B, True), (self.rank + 1) % N), None)
                    continue
                for w in chunk[t]:
                    a.train(w)
                    mark("compute")
                    a.apply_local(w, b, B)
                    mark("pack")
                if getattr(a, "rccl_ready", False) and (N > 1 or getattr(a, "rccl_self_ring", False)):   # one C call per hand-over (svdf_rccl.cpp);
                    # rccl_self_ring: the one-rank ring of the tests -- the block goes to this rank itself and comes back P steps later
                    nxt = (b + P) % B
                    self.pending[nxt] = (a.rccl_handoff(b, nxt, B, (self.rank - 1) % N, (self.rank + 1) % N), None)
                    mark("pack")
                elif N > 1:
                    nxt = (b + P) % B
                    ipc = getattr(a, "ipc_ready", False)
                    out = a.block_get(b, B, True) if ipc else a.block_get(b, B)
                    inc = a.block_like(nxt, B, True) if ipc else a.block_like(nxt, B)
                    self.pending[nxt] = (a.handoff_start(self.dist, out, (self.rank - 1) % N, inc, (self.rank + 1) % N), inc)
                    mark("pack")
        for b in list(self.pending):   # the home blocks come back at the end of a chunk: have them in place between passes
            self._arrive(b)
        mark("allreduce")

    def gather_blocks(self):
        """every rank broadcasts the blocks it holds between passes (its P home blocks): the item side is complete everywhere afterwards"""
        a, N, P = self.a, self.world, self.P
        for b in range(N * P):
            owner = b // P
            buf = a.block_get(b, N * P) if owner == self.rank else a.block_like(b, N * P)
            a.broadcast(self.dist, buf, owner)
            if owner != self.rank:
                a.block_set(b, N * P, buf)


class HipShard:
    """Adaptor over svdfeature_amd.Trainer: windows are HBM-resident scheduled datasets; per window ONE kernel packs
    (current - snapshot) of all replicated ranges straight into a torch tensor in the wire format (fp32 or fp16),
    torch.distributed (backend nccl = RCCL) all-reduces it, ONE kernel sets current = snapshot + sum and moves the
    snapshot along.  Everything is ordered on ONE torch-owned HIP stream: a pass needs no host synchronisation."""

    apply_refreshes_snapshot = True

    def __init__(self, trainer, torch, device, parts=1, minibatch=False):
        self.t, self.torch, self.device = trainer, torch, device
        # window-minibatch mode (svdf_k_window.hip): windows are svdf_dataset_window_from_triples data sets; train = the users'
        # exact walks with the item side read-only, delta_get = per-item sum of the contributions straight into the wire buffer,
        # delta_set = replicated ranges += all-reduced buffer.  No snapshot, no pack.
        self.minibatch = bool(minibatch)
        self.last = None
        self._step_arrays = {}
        self.stream = torch.cuda.Stream(device=device)
        trainer.set_stream(self.stream.cuda_stream)
        self.buf = None
        self.bufs = {}
        self.half = False
        self.parts = parts

    def set_wire_half(self, half):
        self.half = bool(half)
        self.buf = None
        self.bufs = {}

    def make_windows(self, shards):
        """shards: per window (user, item, label) triples, Pairs(user, pos, neg) or a data.BlockArrays (user-group data)."""
        from .data import BlockArrays, CSRData
        out = []
        for sh in shards:
            if isinstance(sh, list):    # item-range pieces of one window
                mk = self.t.dataset_window_from_triples if self.minibatch else self.t.dataset_from_triples
                out.append([mk(*piece) for piece in sh])
                continue
            if self.minibatch:
                if isinstance(sh, BlockArrays):      # user-group (SVD++) blocks: user units with feedback lists (svdf_k_wunit.hip)
                    out.append(self.t.dataset_window_from_blocks(sh))
                elif isinstance(sh, CSRData):        # rows with global features / several item entries
                    out.append(self.t.dataset_window_from_csr(sh))
                else:
                    out.append(self.t.dataset_window_from_pairs(sh.user, sh.pos, sh.neg) if isinstance(sh, Pairs) else self.t.dataset_window_from_triples(*sh))
                continue
            if isinstance(sh, BlockArrays):
                out.append(self.t.dataset_from_blocks(sh))
            elif isinstance(sh, CSRData):
                out.append(self.t.dataset_from_csr(sh))
            elif isinstance(sh, Pairs):
                out.append(self.t.dataset_from_pairs(sh.user, sh.pos, sh.neg))
            else:
                out.append(self.t.dataset_from_triples(*sh))
        return out

    def train(self, ds):
        self.t.train_dataset(ds)
        self.last = ds

    def gather_user_side(self, dist, rank, world):
        """W_user / u_bias rows are private to their owner (user % world): sum the owners' rows over the ranks so that every
        rank (and a model file written by any of them) holds the complete model.  Supports are disjoint: a plain SUM."""
        torch = self.torch
        for name in ("W_user", "u_bias"):
            v = self.t.view(name)
            if v is None or v.size == 0:
                continue
            mine = (np.arange(v.shape[0]) % world) == rank
            v = np.where(mine.reshape((-1,) + (1,) * (v.ndim - 1)), v, np.float32(0))
            tns = torch.from_numpy(np.ascontiguousarray(v)).to(self.device)
            dist.all_reduce(tns)
            self.t.set_view(name, tns.cpu().numpy())

    # ---- stratified schedule (StratifiedTrainer): in-place sums into the owned item block, block hand-over between ranks
    def apply_local(self, ds, block, nblocks):
        self.t.item_delta_select(block, nblocks)
        self.t.window_delta_apply_local(ds)
        self.t.item_delta_select(0, 1)

    def stratum_step(self, windows, block, nblocks, want_out):
        """the whole stratum step in ONE engine call (svdf_stratum_step): train + sum in place for every window, then the block copied out for its
        hand-over (want_out); returns the out tensor or None.  Same launches as train / apply_local / block_get."""
        import ctypes as C
        key = id(windows)
        arr = self._step_arrays.get(key)
        if arr is None or arr[1] is not windows:
            arr = ((C.c_void_p * max(len(windows), 1))(*[w.h for w in windows]), windows)
            self._step_arrays[key] = arr
        out = None
        if want_out:
            k = ("out", block, nblocks)
            out = self.bufs.get(k)
            if out is None:
                self.t.item_delta_select(block, nblocks)
                n = self.t.item_block_count()
                self.t.item_delta_select(0, 1)
                with self.torch.cuda.stream(self.stream):
                    out = self.bufs[k] = self.torch.empty(n, device=self.device, dtype=self.torch.float32)
        self.t.stratum_step(arr[0], len(windows), block, nblocks, out.data_ptr() if out is not None else 0)
        if windows:
            self.last = windows[-1]
        return out

    def block_like(self, block, nblocks, for_handoff=False):
        if self.ipc_ready and for_handoff:
            return ("ipc-in", block, nblocks)
        key = ("in", block, nblocks)
        if key in self.bufs:
            return self.bufs[key]
        self.t.item_delta_select(block, nblocks)
        n = self.t.item_block_count()
        self.t.item_delta_select(0, 1)
        if key not in self.bufs:
            with self.torch.cuda.stream(self.stream):
                self.bufs[key] = self.torch.empty(n, device=self.device, dtype=self.torch.float32)
        return self.bufs[key]

    def block_get(self, block, nblocks, for_handoff=False):
        if self.ipc_ready and for_handoff:
            return ("ipc-out", block, nblocks)   # the block is stored straight into the neighbour's inbox by handoff_start
        self.t.item_delta_select(block, nblocks)
        n = self.t.item_block_count()
        key = ("out", block, nblocks)
        if key not in self.bufs:
            with self.torch.cuda.stream(self.stream):
                self.bufs[key] = self.torch.empty(n, device=self.device, dtype=self.torch.float32)
        self.t.item_block_get(self.bufs[key].data_ptr())
        self.t.item_delta_select(0, 1)
        return self.bufs[key]

    def block_set(self, block, nblocks, tensor):
        if not isinstance(tensor, tuple):
            self.t.item_block_set_at(block, nblocks, tensor.data_ptr())
            return
        self.t.item_delta_select(block, nblocks)
        if isinstance(tensor, tuple) and tensor[0] == "ipc-recv":   # wait for the neighbour's store, put the block in place, acknowledge the slot
            _, src, slot, seq = tensor
            self.t.ipc_block_recv(src, slot, seq)
        else:
            self.t.item_block_set(tensor.data_ptr())
        self.t.item_delta_select(0, 1)

    def handoff_start(self, dist, out, dst, inc, src):
        """start sending `out` to rank dst and receiving `inc` from rank src: ordered after the copy-out kernel on the trainer's stream,
        running beside whatever the trainer enqueues next"""
        if self.ipc_ready:
            _, block, nblocks = out
            self.t.item_delta_select(block, nblocks)
            self.t.ipc_block_send(dst, self.ipc_sent % 2)
            self.t.item_delta_select(0, 1)
            self.ipc_sent += 1
            self.ipc_received += 1
            return ("ipc-recv", src, (self.ipc_received - 1) % 2, self.ipc_received)
        with self.torch.cuda.stream(self.stream):
            return dist.batch_isend_irecv([dist.P2POp(dist.isend, out, dst), dist.P2POp(dist.irecv, inc, src)])

    def handoff_wait(self, reqs):
        if self.ipc_ready:
            return   # block_set does the waiting, on the stream
        with self.torch.cuda.stream(self.stream):   # the trainer's stream waits for the transfer; the host does not
            for q in reqs:
                q.wait()

    def broadcast(self, dist, buf, src):
        with self.torch.cuda.stream(self.stream):
            dist.broadcast(buf, src)

    # ---- the exchanges issued from C++ straight into RCCL (svdf_rccl.cpp): the rank's own communicator; torch.distributed only carries the id
    rccl_ready = False
    rccl_self_ring = False

    def rccl_open(self, dist, rank, world):
        import svdfeature_amd as sa
        assert self.minibatch, "the native RCCL exchange serves the window-minibatch step"
        box = [sa.rccl_unique_id() if rank == 0 else None]
        if dist is not None and world > 1:
            dist.broadcast_object_list(box, src=0)
        self.t.rccl_init(box[0], rank, world)
        if dist is not None and world > 1:
            dist.barrier()
        self.rccl_ready, self.rccl_sent = True, 0

    def rccl_close(self):
        if self.rccl_ready:
            self.t.rccl_close()
        self.rccl_ready = False

    def rccl_allreduce(self):
        self.t.rccl_window_allreduce(self.last, self.half)

    def rccl_handoff(self, block, in_block, nblocks, dst, src):
        slot = self.rccl_sent % 2
        self.t.item_delta_select(block, nblocks)
        self.t.rccl_block_handoff(dst, src, slot, in_block, nblocks)
        self.t.item_delta_select(0, 1)
        self.rccl_sent += 1
        return ("rccl-recv", slot)

    def rccl_arrive(self, block, nblocks, slot):
        self.t.item_delta_select(block, nblocks)
        self.t.rccl_block_arrive(slot)
        self.t.item_delta_select(0, 1)

    # ---- cross-process direct exchange (svdf_ipc.cpp): wire buffers / flag pages IPC-mapped into every rank's process
    ipc_ready = False

    def ipc_open(self, dist, rank, world, blocks=0, barrier=True):
        """every rank exports its wire buffer + flag page, the handles travel through the process group (all_gather_object), every rank
        maps the others'.  blocks > 0: also an inbox for the stratified schedule's item blocks (num_item / blocks rows each)."""
        assert self.minibatch, "the IPC exchange serves the window-minibatch step"
        n = self.t.item_delta_count()
        block_floats = 0
        if blocks > 0:
            for b in range(blocks):
                self.t.item_delta_select(b, blocks)
                block_floats = max(block_floats, self.t.item_block_count())
            self.t.item_delta_select(0, 1)
        mine = self.t.ipc_setup(rank, world, n * 4, block_floats)
        gathered = [None] * world
        if dist is None:   # the one-rank ring of the probes / tests
            gathered = [mine]
        else:
            dist.all_gather_object(gathered, mine)
        self.t.ipc_connect(b"".join(gathered))
        if dist is not None and barrier:   # (barrier=False: the caller agrees on the outcome first, then meets the others itself)
            dist.barrier()   # nobody signals into a page that is not mapped yet
        self.ipc_ready, self.ipc_rank, self.ipc_world = True, rank, world
        self.ipc_sent, self.ipc_received = 0, 0

    def ipc_pack(self):
        self.t.ipc_window_pack(self.last, self.half)

    def ipc_reduce(self):
        self.t.ipc_window_reduce(self.half)

    def ipc_apply(self):
        self.t.ipc_window_apply(self.half)

    def delta_begin(self):
        if not self.minibatch:
            self.t.item_delta_begin()

    def _pack(self, ptr):
        if self.minibatch:
            self.t.window_delta_pack(self.last, ptr, self.half)
        else:
            self.t.item_delta_pack(ptr, self.half)

    def _apply(self, ptr):
        if self.minibatch:
            self.t.window_delta_apply(ptr, self.half)
        else:
            self.t.item_delta_unpack(ptr, self.half, refresh_snapshot=True)

    def delta_get(self, part=None):
        if part is not None:   # one item-range piece of the exchange, a buffer of its own (its collective may still be in flight)
            self.t.item_delta_select(part, self.parts)
            if part not in self.bufs:
                with self.torch.cuda.stream(self.stream):
                    self.bufs[part] = self.torch.empty(self.t.item_delta_count(), device=self.device,
                                                       dtype=self.torch.float16 if self.half else self.torch.float32)
            self._pack(self.bufs[part].data_ptr())
            self.t.item_delta_select(0, 1)
            return self.bufs[part]
        if self.buf is None:
            with self.torch.cuda.stream(self.stream):
                self.buf = self.torch.empty(self.t.item_delta_count(), device=self.device,
                                            dtype=self.torch.float16 if self.half else self.torch.float32)
        self._pack(self.buf.data_ptr())
        return self.buf

    def all_reduce(self, dist, d):
        with self.torch.cuda.stream(self.stream):   # the collective is ordered after the pack kernel on our stream
            dist.all_reduce(d)

    def all_reduce_async(self, dist, d):
        """the collective starts after the pack kernel on our stream and runs beside what we enqueue next; .wait() on the
        returned handle (inside our stream context) makes our stream wait for it"""
        with self.torch.cuda.stream(self.stream):
            work = dist.all_reduce(d, async_op=True)
        torch, stream = self.torch, self.stream

        class _Wait:
            def wait(self_inner):
                with torch.cuda.stream(stream):
                    work.wait()
        return _Wait()

    def delta_set(self, d, part=None):
        if part is not None:
            self.t.item_delta_select(part, self.parts)
            self._apply(d.data_ptr())
            self.t.item_delta_select(0, 1)
            return
        self._apply(d.data_ptr())
