"""Worker of tests/test_gpu_comm.py: one rank of a node-local exchange test.  All ranks share
cuda:0 (the test box has one GPU); the rendezvous and the reference all-reduce use gloo."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import lib as L  # noqa: E402
from gcn_amd import ops  # noqa: E402
from gcn_amd.comm import NodeComm  # noqa: E402


def fit(rank, world, dev):
    """Two epochs of data-parallel fp64 MBGD through the product's driver (HIP gradient kernels,
    one-shot exchange fused with the update) against the oracle's master/worker replay."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle as orc
    from gcn_amd import train
    z, T, B, epochs = 48, 64 * world + 5, 8 * world, 2
    rng = np.random.default_rng(11)
    cubes = rng.random((T, 9 * z))
    onehot = np.zeros((T, 13), dtype=np.int32)
    onehot[np.arange(T), rng.integers(0, 13, T)] = 1
    orc.srand(11)
    params = orc.init_params_f64(z, 0)
    sl = T // world
    p = torch.from_numpy(params.copy()).to(dev)
    lc = torch.from_numpy(np.ascontiguousarray(cubes[rank * sl:(rank + 1) * sl])).to(dev)
    lo = torch.from_numpy(np.ascontiguousarray(onehot[rank * sl:(rank + 1) * sl])).to(dev)
    hist = train.fit_mbgd(p, lc, lo, z, batch=B, lr=0.5, min_err=0.0, max_iter=epochs, distributed=True, train_size=T)
    ref, ref_loss = params.copy(), []
    for _ in range(epochs):
        tot = 0.0
        for i0 in range(T // B):
            tot += orc.mgpu_step_f64(ref, cubes, onehot, T, i0, B, world, z, 0.5)
        ref_loss.append(tot / T)
    got = p.cpu().numpy()
    assert np.abs(got - ref).max() <= 1e-10 * np.abs(ref).max(), np.abs(got - ref).max()
    assert np.allclose(hist.loss, ref_loss, rtol=1e-10), (hist.loss, ref_loss)
    every = [torch.empty(p.numel(), dtype=p.dtype) for _ in range(world)]
    dist.all_gather(every, p.cpu())
    assert all(torch.equal(every[0], e) for e in every)


def stalled_peer(rank, world, dev):
    """Rank 1 never launches its step: rank 0's waits run out (CCNN_COMM_WAIT_MS), its parameters
    must stay bit for bit what they were, and the status word must say so - for the step that
    exchanges inside its reduction kernel and for the stand-alone exchange + update."""
    z = 60
    d = L.dims(z)
    for dtype in (torch.float32, torch.float64):
        comm = NodeComm(d.nparams, dtype)
        rng = np.random.default_rng(3)
        p0 = torch.from_numpy((rng.random(d.nparams) - 0.5) / 10).to(dtype).to(dev)
        B = 6
        cubes = torch.from_numpy(rng.random((B, 9 * z))).to(dtype).to(dev)
        oh = np.zeros((B, 13), dtype=np.int32)
        oh[np.arange(B), rng.integers(0, 13, B)] = 1
        oh = torch.from_numpy(oh).to(dev)
        p = p0.clone()
        assert comm.status() == 0
        if rank == 0:
            ops.mbgd_step(p, cubes, oh, 0, B, z, 0.04, comm=comm, divisor=B * world)
            torch.cuda.synchronize()
            assert comm.status() == 1, "the wait ran out but the status word does not say so"
            assert torch.equal(p, p0), "a timed-out exchange touched the parameters"
            # every later call gives up at once and leaves the parameters alone as well
            g = torch.ones_like(p)
            comm.allreduce_update(p, g, 0.04, z, divisor=B * world)
            ops.mbgd_step(p, cubes, oh, 0, B, z, 0.04, comm=comm, divisor=B * world)
            torch.cuda.synchronize()
            assert torch.equal(p, p0)
        dist.barrier()
        comm.close()
    # the stand-alone exchange on a fresh set of buffers
    comm = NodeComm(d.nparams, torch.float32)
    p0 = torch.full((d.nparams,), 0.25, dtype=torch.float32, device=dev)
    p = p0.clone()
    if rank == 0:
        comm.allreduce_update(p, torch.ones_like(p), 0.04, z)
        torch.cuda.synchronize()
        assert comm.status() == 1 and torch.equal(p, p0)
    dist.barrier()
    comm.close()


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if "--stalled-peer" in sys.argv:
        stalled_peer(rank, world, dev)
        if rank == 0:
            print("STALL-OK", world, flush=True)
        dist.destroy_process_group()
        return
    if "--fit" in sys.argv:
        fit(rank, world, dev)
        if rank == 0:
            print("FIT-OK", world, flush=True)
        dist.destroy_process_group()
        return
    z = 103
    d = L.dims(z)

    # A local fp64 gradient whose workspace is above 1 MB (the allocator's large blocks), computed
    # before any exchange buffer exists and again after every set has been given up: memory that
    # peers have had mapped must never come back as somebody's workspace (it did, when destroy
    # freed the buffers: gradients off by up to 20 %).
    def local_gradient():
        import numpy as np
        from gcn_amd import ops
        rng = np.random.default_rng(5)
        p = torch.from_numpy((rng.random(d.nparams) - 0.5) / 10).to(dev)
        cubes = torch.from_numpy(rng.random((40, d.cube))).to(dev)
        oh = np.zeros((40, 13), dtype=np.int32)
        oh[np.arange(40), rng.integers(0, 13, 40)] = 1
        g, lv = ops.mbgd_grad(p, cubes, torch.from_numpy(oh).to(dev), 0, 40, z)
        torch.cuda.synchronize()
        return g.clone(), lv.clone()

    g_first, lv_first = local_gradient()
    # three sets of exchange buffers in a row: each set is given up and the next created afresh
    # (over the same memory), so sequence numbers and stale contents are covered
    for dtype in (torch.float32, torch.float64, torch.float32):
        comm = NodeComm(d.nparams, dtype)
        a = torch.ones(8, dtype=dtype, device=dev)
        comm.all_reduce(a)
        torch.cuda.synchronize()
        assert comm.status() == 0, "a rank gave up waiting for its peers"
        assert float(a[0]) == world
        # the product's own acceptance tests, the in-step exchange first (it is the first thing a
        # training run does with fresh buffers) and at two batch sizes
        assert comm._self_test_step(z), "first in-step exchange on fresh buffers"
        assert comm._self_test_step(z, B=12)
        # a batch large enough for the sample-tile kernel and the one-pass batch sums, whose total
        # goes through the stand-alone exchange + update kernel
        assert comm._self_test_step(z, B=600), "data-parallel step of a large batch"
        assert comm._self_test_step(z, B=1600), "data-parallel step of a batch whose sums are split over workgroups"
        # both forms of the in-step exchange, whatever the default for this many ranks is:
        # everybody writes to everybody / reduce at an owner rank that sends the total round
        for via_owner in (True, False, True):
            comm.set_exchange(via_owner)
            assert comm._self_test_step(z, B=9), f"in-step exchange, via_owner={via_owner}"
            assert comm._self_test_step(z, B=16), f"in-step exchange, via_owner={via_owner}"
        comm.set_exchange(world >= 4)
        assert comm.self_test(z=z)
        # short vector (one partial chunk) and repeated calls (slot parity, sequence numbers)
        for n in (1, 5, 1023, 1025):
            for k in range(3):
                a = torch.full((n,), float(rank + 1 + k), dtype=dtype, device=dev)
                comm.all_reduce(a)
                torch.cuda.synchronize()
                want = sum(r + 1 + k for r in range(world))
                assert bool((a == want).all()), (n, k, a[:4], want)
        # fused update == all_reduce + stand-alone update, bit for bit
        g = torch.Generator().manual_seed(7 + rank)
        grad = (torch.rand(d.nparams, generator=g, dtype=torch.float64) - 0.5).to(dtype).to(dev)
        p0 = (torch.rand(d.nparams, generator=torch.Generator().manual_seed(3), dtype=torch.float64) - 0.5).to(dtype).to(dev)
        p_fused, p_ref = p0.clone(), p0.clone()
        comm.allreduce_update(p_fused, grad, 0.04, z, divisor=100)
        parts = [torch.empty(d.nparams, dtype=dtype) for _ in range(world)]
        dist.all_gather(parts, grad.cpu())
        total = torch.zeros_like(grad)
        for r in range(world):   # rank order, like the kernel
            total = total + parts[r].to(dev)
        ops.update(p_ref, total, 0.04, z, divisor=100)
        torch.cuda.synchronize()
        assert torch.equal(p_fused, p_ref), float((p_fused - p_ref).abs().max())
        # the step with the exchange inside its reduction kernel == gradient, exchange, update
        rng = np.random.default_rng(5 + rank)
        B = 12
        cubes = torch.from_numpy(rng.random((B, 9 * z))).to(dtype).to(dev)
        oh = np.zeros((B, 13), dtype=np.int32)
        oh[np.arange(B), rng.integers(0, 13, B)] = 1
        oh = torch.from_numpy(oh).to(dev)
        q_fused, q_ref = p0.clone(), p0.clone()
        # the ranks share one GPU here: a rank whose reduction kernel already waits for its peers
        # holds compute units the peers' per-sample kernels need, so the ranks must launch together
        torch.cuda.synchronize()
        dist.barrier()
        lv1 = ops.mbgd_step(q_fused, cubes, oh, 0, B, z, 0.04, comm=comm, divisor=B * world)
        g_loc, lv2 = ops.mbgd_grad(q_ref, cubes, oh, 0, B, z)
        comm.allreduce_update(q_ref, g_loc, 0.04, z, divisor=B * world)
        torch.cuda.synchronize()
        assert torch.equal(lv1, lv2)
        # same sums in the same order; the two kernels may contract p - lr*g differently (1 ulp)
        tol = 2e-7 if dtype == torch.float32 else 4e-16
        assert float((q_fused - q_ref).abs().max()) <= tol * float(q_ref.abs().max()), float((q_fused - q_ref).abs().max())
        every = [torch.empty(d.nparams, dtype=dtype) for _ in range(world)]
        dist.all_gather(every, q_fused.cpu())
        assert all(torch.equal(every[0], e) for e in every)
        # every rank ends with identical parameters
        every = [torch.empty(d.nparams, dtype=dtype) for _ in range(world)]
        dist.all_gather(every, p_fused.cpu())
        assert all(torch.equal(every[0], e) for e in every)
        assert comm.status() == 0
        comm.close()
        for _ in range(3):
            g, lv = local_gradient()
            assert torch.equal(g, g_first) and torch.equal(lv, lv_first), "a local gradient changed after an exchange set was given up"
    if rank == 0:
        print("COMM-OK", world, flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
