"""Cross-PROCESS direct exchange (svdf_ipc.cpp) on one MI355X: two / three processes share GPU 0 (HIP IPC handles work
between processes on one device), their wire buffers and flag pages mapped into each other, the exchange of a window = pack -> sequence
flag -> peer-pointer reduce-scatter + all-gather -> flag -> add, the stratified hand-over = a store into the neighbour's inbox + flag +
acknowledgement -- no collective library on the data path (gloo only carries the 128 handle bytes).  Must equal the oracle-backed
simulation bit for bit (fp32 wire), like the gloo / RCCL transports."""
import os
import sys

import numpy as np
import pytest

import cases
from multi_rank_utils import simulate, simulate_stratified
from test_multi_rank import _free_port

pytestmark = pytest.mark.gpu
NU, NI, K = 1500, 400, 64
CONF = cases.conf_with(cases.BASICMF_CONF, num_user=NU, num_item=NI, num_factor=K)


def _worker(rank, world, port, mode, windows, passes, out_dir):
    import torch
    import torch.distributed as dist
    import svdfeature_amd as sa
    from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, StratifiedTrainer, shard_windows, stratified_plan
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = sa.Trainer(0, 0)
    t.seed(10)
    for k, v in CONF:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    ad = HipShard(t, torch, torch.device("cuda", 0), minibatch=True)
    ad.set_wire_half(False)
    if mode == "allreduce":
        ad.ipc_open(dist, rank, world)
        st = ShardedTrainer(ad, ad.make_windows(shard_windows(u, i, r, rank, world, windows)), world, dist)
    else:
        P = 2
        ad.ipc_open(dist, rank, world, blocks=world * P)
        plan = [[ad.make_windows(sub) for sub in chunk] for chunk in stratified_plan(u, i, r, rank, world, windows, NI, 9.0, P)]
        st = StratifiedTrainer(ad, plan, world, rank, dist, blocks_per_rank=P)
    for _ in range(passes):
        st.train_pass()
    t.synchronize()
    assert t.ipc_status() == 0
    if mode != "allreduce":
        dist.barrier()
        st.gather_blocks()
        t.synchronize()
    np.savez(os.path.join(out_dir, "rank%d.npz" % rank), **{n: t.view(n) for n in ("W_item", "i_bias", "W_user", "u_bias")})
    dist.barrier()
    t.ipc_close()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_all_reduce_step_equals_the_simulation(world, tmp_path):
    import torch.multiprocessing as mp
    windows, passes = 4, 2
    mp.spawn(_worker, args=(world, _free_port(), "allreduce", windows, passes, str(tmp_path)), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate(CONF, u, i, r, world, windows, passes, minibatch=True)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_stratified_hand_over_equals_the_simulation(world, tmp_path):
    import torch.multiprocessing as mp
    chunks, passes = 3, 2
    mp.spawn(_worker, args=(world, _free_port(), "stratified", chunks, passes, str(tmp_path)), nprocs=world, join=True)
    u, i, r = cases.planted_triples(40000, NU, NI, seed=9)
    sim = simulate_stratified(CONF, u, i, r, world, chunks, passes, NI, 9.0, blocks_per_rank=2)
    for rk in range(world):
        z = np.load(str(tmp_path / ("rank%d.npz" % rk)))
        for name in ("W_item", "i_bias", "W_user", "u_bias"):
            np.testing.assert_array_equal(z[name].view(np.uint32), sim[rk].t.view(name).view(np.uint32))


def _dead_peer_worker(rank, world, port, out_dir):
    import torch
    import torch.distributed as dist
    import svdfeature_amd as sa
    from svdfeature_amd.multi_gpu import HipShard, ShardedTrainer, shard_windows
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    t = sa.Trainer(0, 0)
    t.seed(10)
    for k, v in CONF:
        t.set_param(k, v)
    t.init_model()
    t.init_trainer()
    t.set_knob("ipc_spin_limit", 20000)
    before = t.view("W_item").copy()
    u, i, r = cases.planted_triples(8000, NU, NI, seed=9)
    ad = HipShard(t, torch, torch.device("cuda", 0), minibatch=True)
    ad.set_wire_half(False)
    ad.ipc_open(dist, rank, world)
    st = ShardedTrainer(ad, ad.make_windows(shard_windows(u, i, r, rank, world, 1)), world, dist)
    msgs = []
    if rank == 0:   # rank 1 never reaches the exchange: rank 0's wait must time out, say so, and leave rank 1's buffers alone
        st.train_pass()
        for call in (t.synchronize, t.ipc_close):
            try:
                call()
                msgs.append("no error")
            except Exception as e:   # noqa: BLE001
                msgs.append(str(e))
        assert t.ipc_status() == 0   # closed
    dist.barrier()
    if rank == 1:
        t.synchronize()
        assert t.ipc_status() == 0
        assert np.array_equal(before.view(np.uint32), t.view("W_item").view(np.uint32))
        t.ipc_close()
    with open(os.path.join(out_dir, "rank%d.txt" % rank), "w") as f:
        f.write("\n".join(msgs))
    dist.barrier()
    dist.destroy_process_group()


def test_ipc_wait_timeout_is_reported_at_synchronize_and_close(tmp_path):
    """a flag wait that hits its spin limit in the LAST window used to pass silently (no later svdf_ipc_* call) and the
    stream went on to sum incomplete buffers into every peer.  Now the signal / reduce / copy kernels are no-ops once the error word is
    raised and svdf_synchronize / svdf_ipc_close fail."""
    import torch.multiprocessing as mp
    mp.spawn(_dead_peer_worker, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    msgs = open(str(tmp_path / "rank0.txt")).read().split("\n")
    assert len(msgs) == 2 and "spin limit" in msgs[0] and "svdf_synchronize" in msgs[0], msgs
    assert "spin limit" in msgs[1] and "svdf_ipc_close" in msgs[1], msgs
