"""The exchanges of the one-process-per-GPU ranks issued from C++ straight into RCCL (svdf_rccl.cpp).  One GPU is what this box has and
RCCL refuses two ranks on one device, so what runs here is a ONE-rank communicator: ncclCommInitRank, ncclAllReduce (the identity), and the
stratified hand-over as a send-to-self / receive-from-self pair inside one group -- every call, stream, event and buffer of the path, with results
held against the torch-free reference forms of the same steps (window_delta_pack / apply through a plain device buffer; the block copied out and
put back).  More than one rank is the driver's 8-GPU run: `bench.py --gpus N` measures it as secondary.stratified_native / allreduce_minibatch_native."""
import numpy as np
import pytest

import cases
import svdfeature_amd as sa

pytestmark = pytest.mark.gpu


def _trainer(nu, ni, k):
    t = sa.Trainer(0, 0)
    t.seed(10)
    for kk, v in cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k):
        t.set_param(kk, v)
    t.init_model()
    t.init_trainer()
    return t


def _data(n, nu, ni, seed):
    rng = np.random.default_rng(seed)
    return (rng.integers(0, nu, n).astype(np.uint32), rng.integers(0, ni, n).astype(np.uint32), rng.integers(1, 6, n).astype(np.float32))


def _views(t):
    return {n: t.view(n).copy() for n in ("W_user", "W_item", "u_bias", "i_bias")}


def _same(a, b):
    for n in a:
        assert np.array_equal(a[n].view(np.uint32), b[n].view(np.uint32)), n


@pytest.mark.parametrize("half", [False, True])
def test_window_allreduce_with_one_rank_is_the_plain_step(half):
    torch = pytest.importorskip("torch")
    nu, ni, k, n = 500, 120, 64, 20000
    cols = _data(n, nu, ni, 1)
    a, b = _trainer(nu, ni, k), _trainer(nu, ni, k)
    a.rccl_init(sa.rccl_unique_id(), 0, 1)
    wa = [a.dataset_window_from_triples(*[c[s:s + 5000] for c in cols]) for s in range(0, n, 5000)]
    wb = [b.dataset_window_from_triples(*[c[s:s + 5000] for c in cols]) for s in range(0, n, 5000)]
    buf = torch.empty(b.item_delta_count(), device="cuda", dtype=torch.float16 if half else torch.float32)
    for _ in range(2):
        for x, y in zip(wa, wb):
            a.train_dataset(x)
            a.rccl_window_allreduce(x, half)
            b.train_dataset(y)
            b.window_delta_pack(y, buf.data_ptr(), half)
            torch.cuda.synchronize()
            b.window_delta_apply(buf.data_ptr(), half)
    assert a.rccl_counter(1) == 2 * len(wa)
    _same(_views(a), _views(b))
    a.rccl_close()


@pytest.mark.parametrize("nblocks", [2, 4, 7])
def test_block_handoff_to_self_round_trips_through_both_slots(nblocks):
    nu, ni, k = 300, 101, 64
    t = _trainer(nu, ni, k)
    t.rccl_init(sa.rccl_unique_id(), 0, 1)
    cols = _data(8000, nu, ni, 2)
    ds = t.dataset_from_triples(*cols)
    t.train_dataset(ds)
    rows = lambda b: (ni * b // nblocks, ni * (b + 1) // nblocks)
    for step in range(2 * nblocks):
        b = step % nblocks
        before = _views(t)
        t.item_delta_select(b, nblocks)
        t.rccl_block_handoff(0, 0, step % 2, b, nblocks)   # block b leaves, block b (from myself) arrives in slot step % 2
        t.item_delta_select(0, 1)
        t.train_dataset(ds)                                # the item side moves on while the block is in flight
        moved = _views(t)
        t.item_delta_select(b, nblocks)
        t.rccl_block_arrive(step % 2)
        t.item_delta_select(0, 1)
        after = _views(t)
        lo, hi = rows(b)
        assert np.array_equal(after["W_item"][lo:hi].view(np.uint32), before["W_item"][lo:hi].view(np.uint32))   # the block as it was sent
        assert np.array_equal(after["i_bias"][lo:hi].view(np.uint32), before["i_bias"][lo:hi].view(np.uint32))
        keep = np.ones(ni, bool); keep[lo:hi] = False
        assert np.array_equal(after["W_item"][keep].view(np.uint32), moved["W_item"][keep].view(np.uint32))      # nothing else touched
        assert np.array_equal(after["W_user"].view(np.uint32), moved["W_user"].view(np.uint32))
    assert t.rccl_counter(0) == 2 * nblocks
    with pytest.raises(sa.SvdfError, match="no hand-over is in flight"):
        t.rccl_block_arrive(0)
    t.item_delta_select(0, nblocks)
    t.rccl_block_handoff(0, 0, 0, 0, nblocks)
    with pytest.raises(sa.SvdfError, match="two hand-overs in flight"):
        t.rccl_block_handoff(0, 0, 0, 0, nblocks)
    t.rccl_block_arrive(0)
    t.item_delta_select(0, 1)
    t.rccl_close()


def test_calls_before_init_are_refused():
    t = _trainer(50, 20, 64)
    with pytest.raises(sa.SvdfError, match="svdf_rccl_init first"):
        t.rccl_block_arrive(0)


@pytest.mark.parametrize("k,chunks,P", [(64, 2, 2), (16, 3, 1)])
def test_stratified_trainer_over_the_native_ring_of_one_rank(k, chunks, P):
    """multi_gpu.StratifiedTrainer with the native transport on a ring of ONE rank (every block handed to this rank itself through ncclSend / ncclRecv and
    put back P steps later): the Python glue, slots and events of the path the 8-GPU run takes -- result == the same schedule without any exchange"""
    torch = pytest.importorskip("torch")
    from svdfeature_amd.multi_gpu import HipShard, StratifiedTrainer, stratified_plan
    nu, ni, n, passes = 900, 333, 40000, 2
    u, i, r = cases.planted_triples(n, nu, ni, seed=k)
    conf = cases.conf_with(cases.BASICMF_CONF, num_user=nu, num_item=ni, num_factor=k)
    dev = torch.device("cuda", 0)
    out = []
    for native in (False, True):
        t = sa.Trainer(0, 0)
        t.seed(10)
        for kk, v in conf:
            t.set_param(kk, v)
        t.init_model()
        t.init_trainer()
        ad = HipShard(t, torch, dev, minibatch=True)
        ad.set_wire_half(False)
        plan = [[ad.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, 0, 1, chunks, ni, 9.0, blocks_per_rank=P)]
        if native:
            ad.rccl_open(None, 0, 1)
            ad.rccl_self_ring = True
        st = StratifiedTrainer(ad, plan, 1, 0, None, blocks_per_rank=P)
        for _ in range(passes):
            st.train_pass()
        t.synchronize()
        if native:
            assert t.rccl_counter(0) == passes * chunks * P
            ad.rccl_close()
        out.append(_views(t))
    _same(out[0], out[1])
