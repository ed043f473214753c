"""Training drivers: the host-side epoch loops of the reference's training()
functions over the C ABI.

    fit_sgd    gcn_sgd_fp64.cu:772-885    per-sample SGD, fp64
    fit_mbgd   gcn_mbgd_fp64.cu:923-1119 / gcn_mbgd_fp32.cu:917-1093 minibatch GD;
               with a process group it is the data-parallel form of
               hyper_version_mGPU_MBGD.cu:957-1200 (one all-reduce per step
               instead of weight broadcast + gradient-slab gather)
    evaluate   the test loop + count_err (gcn_mbgd_fp64.cu:1182-1231)

The loops keep the reference's bookkeeping: batch_num = T // B (the tail is
dropped), epoch loss = sum of the per-sample losses / T (all T, although the tail
never ran), FIFO-of-five plateau schedule, early stop below MIN_ERR.
"""
from dataclasses import dataclass, field

import numpy as np
import torch

from . import ops, prep

# learning rate, MIN_ERR, FIFO init, batch size per variant
# (gcn_sgd_fp64.cu:20-22, gcn_mbgd_fp64.cu:20-23, gcn_mbgd_fp32.cu:21-24, hyper_version_mGPU_MBGD.cu:20-23)
VARIANTS = {
    "sgd_fp64": dict(lr=0.008, min_err=0.0003, fifo=1.0, batch=1, dtype=torch.float64),
    "mbgd_fp64": dict(lr=0.5, min_err=0.0001, fifo=1.0, batch=100, dtype=torch.float64),
    "mbgd_fp32": dict(lr=0.04, min_err=0.001, fifo=100.0, batch=100, dtype=torch.float32),
    "mgpu_fp64": dict(lr=0.5, min_err=0.001, fifo=1.0, batch=100, dtype=torch.float64),
}


@dataclass
class History:
    loss: list = field(default_factory=list)        # epoch losses (logloss[] of the reference)
    lr: list = field(default_factory=list)          # learning rate in force during each epoch
    decays: list = field(default_factory=list)      # epochs after which lr was multiplied by 0.9
    stopped_early: bool = False


def fit_sgd(params, cubes, onehot, z, lr=0.008, min_err=0.0003, max_iter=300, log=None):
    """Per-sample SGD epochs; params updated in place.  Returns History."""
    n = onehot.shape[0]
    hist = History()
    sched = prep.LrSchedule(1.0, use_float=False)
    loss = torch.zeros(1, dtype=torch.float64, device=params.device)
    ws = ops.Workspace(params.device)
    for j in range(max_iter):
        ops.sgd(params, cubes, onehot, 0, n, z, lr, loss, workspace=ws)
        single_rate = loss.item() / n                     # one D2H per epoch (gcn_sgd_fp64.cu:857-858)
        if single_rate != single_rate:                    # NaN: the kernel's workgroups lost each other (or the run diverged)
            raise RuntimeError(f"epoch {j}: ccnn_sgd_f64 reported a NaN loss; the parameters are not a valid model")
        hist.loss.append(single_rate)
        hist.lr.append(lr)
        if single_rate < min_err:                         # checked before the schedule (:860-861)
            hist.stopped_early = True
            break
        if log:
            log(j, single_rate)
        decayed, lr = sched.step(single_rate, lr)
        if decayed:
            hist.decays.append(j)
    return hist


class _Local:
    """Single-process stand-in for the pieces of torch.distributed fit_mbgd uses."""
    rank, world = 0, 1

    def all_reduce(self, t):
        return t


class _Group:
    def __init__(self, group=None):
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.rank, self.world = dist.get_rank(group), dist.get_world_size(group)

    def all_reduce(self, t):
        if t.is_cuda and self.dist.get_backend(self.group) == "gloo":   # gloo reduces host tensors
            h = t.cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)
        return t


class MbgdStepper:
    """One minibatch step on this rank's samples.  `grad` and `update` go through
    the HIP library; tests of the multi-rank bookkeeping substitute them."""

    def __init__(self, params, cubes, onehot, z):
        self.params, self.cubes, self.onehot, self.z = params, cubes, onehot, z
        self.ws = ops.Workspace(params.device)
        self.g = torch.empty_like(params)

    def grad(self, first, nb, loss_values):
        ops.mbgd_grad(self.params, self.cubes, self.onehot, first, nb, self.z, grad=self.g,
                      loss_values=loss_values, workspace=self.ws)
        return self.g

    def step_fused(self, first, nb, lr, loss_values):
        ops.mbgd_step(self.params, self.cubes, self.onehot, first, nb, self.z, lr, loss_values=loss_values,
                      workspace=self.ws)

    def update(self, g, lr, divisor):
        ops.update(self.params, g, lr, self.z, divisor=divisor)


def fit_mbgd(params, cubes, onehot, z, batch=100, lr=None, min_err=None, max_iter=300, fifo_init=None,
             group=None, distributed=False, train_size=None, stepper=None, log=None, keep_tail=False):
    """Minibatch epochs; params updated in place (identically on every rank).

    Single process: cubes/onehot hold all T training samples.
    Distributed (distributed=True, optional process group): cubes/onehot hold THIS
    RANK'S slice of floor(T/n) samples (hyper_version_mGPU_MBGD.cu:934-940), every step
    consumes floor(batch/n) of them, gradients are all-reduced and every rank applies
    w -= lr * sum / batch (fp64) or w -= lr * sum (fp32).  train_size = T.
    keep_tail (single process only): the last T % batch samples form a short final batch whose
    mean is taken over its own size (hyper_version_MBGD.cu:780, 794-796) instead of being
    dropped (gcn_mbgd_fp64.cu:895).
    """
    f32 = params.dtype == torch.float32
    lr = (0.04 if f32 else 0.5) if lr is None else lr
    min_err = (0.001 if f32 else 0.0001) if min_err is None else min_err
    fifo_init = (100.0 if f32 else 1.0) if fifo_init is None else fifo_init
    comm = _Group(group) if distributed else _Local()
    n_local = onehot.shape[0]
    T = train_size if train_size is not None else n_local * comm.world
    sub_b = batch // comm.world
    if sub_b <= 0:
        raise ValueError("batch smaller than the number of ranks")
    batch_num = T // batch                                   # tail dropped (gcn_mbgd_fp64.cu:895)
    if batch_num * sub_b > n_local:
        raise ValueError("this rank's slice is shorter than batch_num sub-batches")
    st = stepper or MbgdStepper(params, cubes, onehot, z)
    node = None
    if distributed and stepper is None and params.is_cuda and comm.world > 1:
        # gradient exchange + update in one kernel over xGMI; the group's own all_reduce (RCCL)
        # stays the fallback when the exchange cannot be set up or fails its self-test
        from .comm import NodeComm
        try:
            node = NodeComm(params.numel(), params.dtype, group=group)
            if not node.self_test(rounds=2, z=z):
                node.close()
                node = None
        except Exception:   # raised on every rank alike (see NodeComm)
            node = None
    sched = prep.LrSchedule(fifo_init, use_float=f32)
    steps = [(i0 * sub_b, sub_b) for i0 in range(batch_num)]
    if keep_tail and T % batch:
        if comm.world != 1:
            raise ValueError("keep_tail is a single-process option")
        steps.append((batch_num * batch, T % batch))
    loss_d = torch.zeros((max(len(steps), 1), sub_b), dtype=torch.float64, device=params.device)
    hist = History()
    for it in range(max_iter):
        for i0, (first, nb) in enumerate(steps):
            if comm.world == 1 and stepper is None:
                st.step_fused(first, nb, lr, loss_d[i0])
            elif node is not None and not node.shared_device:
                ops.mbgd_step(params, cubes, onehot, first, nb, z, lr, loss_values=loss_d[i0], workspace=st.ws,
                              comm=node, divisor=batch)
            elif node is not None:
                # ranks that share a GPU: gradient first (waits for nobody), then the stand-alone
                # exchange + update kernel, which is small enough to leave room for the peers' work
                node.allreduce_update(params, st.grad(first, nb, loss_d[i0]), lr, z, divisor=batch)
            else:
                g = comm.all_reduce(st.grad(first, nb, loss_d[i0]))
                st.update(g, lr, batch if nb == sub_b else nb)
        if node is not None:
            # a wait that ran out inside the exchange left the parameters at the last completed
            # step (the kernels never apply a partial sum); no rank may train on from there
            bad = torch.tensor([node.status()], dtype=torch.float64, device=params.device)
            if comm.all_reduce(bad).item() != 0:
                raise RuntimeError(f"epoch {it}: a rank gave up waiting for a peer inside the gradient exchange "
                                   "(ccnn_comm_status != 0); parameters are those of the last completed step")
        # one D2H per epoch; the host adds the per-sample losses in sample order
        lv_all = loss_d[:len(steps)].cpu().numpy()
        lv = np.concatenate([lv_all[i, :nb] for i, (_, nb) in enumerate(steps)]) if steps else lv_all.ravel()[:0]
        if f32:
            acc = np.float32(0)                               # `float loss` (gcn_mbgd_fp32.cu:882, 1027)
            for v in lv:
                acc = np.float32(np.float64(acc) + v)
            total = float(acc)
        else:
            total = 0.0
            for v in lv:
                total = total + float(v)
        if comm.world > 1:
            t = torch.tensor([total], dtype=torch.float64, device=params.device)
            total = comm.all_reduce(t).item()
        single_rate = total / T                               # divides by all T (gcn_mbgd_fp64.cu:1094)
        hist.loss.append(single_rate)
        hist.lr.append(lr)
        if log:
            log(it, single_rate)
        decayed, lr = sched.step(single_rate, lr)
        if decayed:
            hist.decays.append(it)
        if single_rate < min_err:                             # after the schedule (gcn_mbgd_fp64.cu:1117-1118)
            hist.stopped_early = True
            break
    if node is not None:
        node.close()
    return hist


def evaluate(params, cubes, test_labels, z):
    """Fraction of test pixels with argmax+1 == label (count_err, gcn_mbgd_fp64.cu:628-652)."""
    n = test_labels.numel()
    if n == 0:
        return 0.0
    _, am = ops.forward(params, cubes, z, want_probs=False)
    return ops.count_correct(am, test_labels).item() / n


def prepare(scene, labels, x, y, z, dtype, seed, pool_bias_draws=None, device="cuda", n_classes=13):
    """The reference's prologue: scan, split, neighbour lists, shuffle, H2D, gather, init.
    scene: flat [z][y][x] numpy array; labels: x*y float64.  Returns a dict of device tensors;
    its "geometry" is what the training and inference calls take as `z` (the band count itself
    for the reference's 13 classes)."""
    from . import lib as _l
    npdt = np.float32 if dtype == torch.float32 else np.float64
    di = prep.scan_labels(labels, x, y)
    tr9, te9, onehot, test_labels = prep.build_index(labels, di, x, y, n_classes)
    prep.shuffle(tr9, onehot, seed)
    geom = _l.geometry(z, n_classes)
    d = _l.dims(geom)
    draws = d.pout if pool_bias_draws is None else pool_bias_draws
    params = prep.init_params(geom, npdt, draws)
    scene_d = torch.from_numpy(np.ascontiguousarray(scene, dtype=npdt)).to(device)
    out = dict(
        train_cubes=ops.gather(scene_d, torch.from_numpy(tr9).to(device), x, y, z),
        test_cubes=ops.gather(scene_d, torch.from_numpy(te9).to(device), x, y, z, sort=False),   # already in pixel order
        onehot=torch.from_numpy(onehot).to(device),
        test_labels=torch.from_numpy(test_labels).to(device),
        params=torch.from_numpy(params).to(device),
        train_idx9=tr9, test_idx9=te9, geometry=geom,
    )
    return out
