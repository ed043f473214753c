"""CPU, world_size 2 over gloo: the data-parallel bookkeeping of fit_mbgd
(partition, one all-reduce per step, identical update on every rank, loss
reduction) against the oracle's replay of the reference's master/worker step.

The HIP kernels cannot run here, so the per-rank gradient and the update are
substituted by the oracle / numpy through MbgdStepper's hooks; everything else is
the product's driver code."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem(z, T, seed):
    import oracle as orc
    rng = np.random.default_rng(seed)
    cubes = rng.random((T, 9 * z))
    onehot = np.zeros((T, 13), dtype=np.int32)
    onehot[np.arange(T), rng.integers(0, 13, T)] = 1
    orc.srand(seed)
    return orc.init_params_f64(z, 0), cubes, onehot


class OracleStepper:
    """Per-rank gradient from the oracle (sum over the sub-batch), numpy update."""

    def __init__(self, params, cubes, onehot, z):
        self.params, self.cubes, self.onehot, self.z = params, cubes, onehot, z

    def grad(self, first, nb, loss_values):
        import oracle as orc
        p = self.params.numpy().copy()
        lv, _, g = orc.mbgd_step(p, self.cubes, self.onehot, first, nb, self.z, 0.0)
        loss_values.copy_(torch.from_numpy(lv))
        return torch.from_numpy(g)

    def update(self, g, lr, divisor):
        self.params -= lr * g / divisor


def _worker(rank, world, port, z, T, B, epochs, out):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from gcn_amd import prep, train
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    params, cubes, onehot = _problem(z, T, 5)
    sl = T // world
    first, sub = prep.partition(T, B, world, rank, 0)
    assert first == rank * sl and sub == B // world
    lc, lo = np.ascontiguousarray(cubes[rank * sl:(rank + 1) * sl]), np.ascontiguousarray(onehot[rank * sl:(rank + 1) * sl])
    p = torch.from_numpy(params.copy())
    st = OracleStepper(p, lc, lo, z)
    hist = train.fit_mbgd(p, None, torch.from_numpy(lo), z, batch=B, lr=0.5, min_err=0.0, max_iter=epochs,
                          distributed=True, train_size=T, stepper=st)
    out[rank] = (p.numpy().copy(), hist.loss)
    dist.destroy_process_group()


@pytest.mark.parametrize("world,T,B", [(2, 37, 8), (2, 24, 6)])
def test_two_rank_fit_matches_master_worker_replay(world, T, B):
    import oracle as orc
    z, epochs = 30, 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, z, T, B, epochs, out), nprocs=world, join=True)
    params, cubes, onehot = _problem(z, T, 5)
    ref = params.copy()
    ref_loss = []
    for _ in range(epochs):
        tot = 0.0
        for i0 in range(T // B):
            tot += orc.mgpu_step_f64(ref, cubes, onehot, T, i0, B, world, z, 0.5)
        ref_loss.append(tot / T)
    p0, l0 = out[0]
    p1, l1 = out[1]
    assert np.array_equal(p0, p1)                       # every rank applied the identical update
    assert np.abs(p0 - ref).max() <= 1e-12 * np.abs(ref).max()
    assert np.allclose(l0, ref_loss, rtol=1e-12) and np.allclose(l1, ref_loss, rtol=1e-12)


def test_single_process_driver_bookkeeping():
    """world = 1 through the same hooks: tail batch dropped, loss divided by all T, schedule applied."""
    sys.path.insert(0, HERE)
    import oracle as orc
    from gcn_amd import train
    z, T, B = 30, 21, 8
    params, cubes, onehot = _problem(z, T, 9)
    p = torch.from_numpy(params.copy())
    hist = train.fit_mbgd(p, None, torch.from_numpy(onehot), z, batch=B, lr=0.5, min_err=0.0, max_iter=7,
                          stepper=OracleStepper(p, cubes, onehot, z))
    ref = params.copy()
    lr = 0.5
    sched = orc.Sched(np.float64, 1.0)
    for ep in range(7):
        tot = 0.0
        for i0 in range(T // B):                        # 2 batches; samples 16..20 never train
            lv, _, _ = orc.mbgd_step(ref, cubes, onehot, i0 * B, B, z, lr, want_grad=False)
            for v in lv:
                tot = tot + v
        assert abs(hist.loss[ep] - tot / T) <= 1e-12 * abs(tot / T)
        assert hist.lr[ep] == lr
        _, lr = sched.step(tot / T, lr)
    assert np.abs(p.numpy() - ref).max() <= 1e-12 * np.abs(ref).max()
