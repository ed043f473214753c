"""Node-local gradient exchange: the library's one-shot all-reduce over xGMI, one process per GPU.

`NodeComm` wires the ranks of a torch.distributed group together (IPC handles travel through
the group's own all_gather) and exposes the fused `allreduce_update`; `self_test()` checks it
against the group's all_reduce before anybody relies on it.
"""
import ctypes as C

import torch
import torch.distributed as dist

from . import lib as _l
from .ops import _ptr, _sfx, _stream

HANDLE_BYTES = 64


class NodeComm:
    def __init__(self, max_elems, dtype, group=None, device=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.dtype = dtype
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.max_elems = int(max_elems)
        sfx, es = _sfx(dtype)
        lib = _l.load()
        self._h = C.c_void_p()
        handle = (C.c_ubyte * HANDLE_BYTES)()
        # Every step below is agreed on by all ranks before anyone proceeds: a rank that failed
        # alone would leave the others waiting inside the next collective.
        with torch.cuda.device(self.device):
            rc = lib.ccnn_comm_create(self.world, self.rank, self.max_elems, es, C.byref(self._h), handle)
        every = [None] * self.world
        dist.all_gather_object(every, (rc, bytes(handle)), group=group)
        if any(r != 0 for r, _ in every):
            self._destroy()
            raise _l.CcnnError(next(r for r, _ in every if r != 0), "ccnn_comm_create (on some rank)")
        blob = (C.c_ubyte * (HANDLE_BYTES * self.world)).from_buffer_copy(b"".join(h for _, h in every))
        with torch.cuda.device(self.device):
            rc = lib.ccnn_comm_connect(self._h, blob)
        codes = [None] * self.world
        dist.all_gather_object(codes, rc, group=group)   # also: every rank has mapped every buffer
        if any(codes):
            self._destroy()
            raise _l.CcnnError(next(c for c in codes if c), "ccnn_comm_connect (on some rank)")
        sh = C.c_int(0)
        _l.check(lib.ccnn_comm_shares_device(self._h, C.byref(sh)), "ccnn_comm_shares_device")
        flags = [None] * self.world
        dist.all_gather_object(flags, bool(sh.value), group=group)
        # Several ranks on one GPU (tests on a one-GPU box): a kernel that waits for its peers inside
        # a launch then holds compute units its peers' kernels need.  The step that exchanges inside
        # its reduction kernel is only safe there while the ranks launch close together; drivers that
        # cannot promise that (fit_mbgd) take the gradient + stand-alone exchange route instead.
        self.shared_device = any(flags)

    def _destroy(self):
        if getattr(self, "_h", None):
            _l.load().ccnn_comm_destroy(self._h)
            self._h = None

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)   # nobody unmaps while a peer's kernel may still write
            _l.load().ccnn_comm_disconnect(self._h)
            dist.barrier(group=self.group)   # nobody recycles its buffer while a peer still maps it
            self._destroy()

    def set_exchange(self, via_owner):
        """How the in-step exchange moves its sums (ccnn_comm_set_exchange): False = every rank
        writes to every rank, True = reduce at an owner rank that sends the total round.  Same
        results; every rank must make the same call."""
        _l.check(_l.load().ccnn_comm_set_exchange(self._h, int(bool(via_owner))), "ccnn_comm_set_exchange")

    def status(self):
        st = C.c_int(0)
        _l.check(_l.load().ccnn_comm_status(self._h, C.byref(st)), "ccnn_comm_status")
        return st.value

    def all_reduce(self, t):
        """In-place sum over ranks (rank order, identical bits everywhere)."""
        sfx, _ = _sfx(t.dtype)
        assert t.dtype == self.dtype and t.is_contiguous() and t.numel() <= self.max_elems
        fn = getattr(_l.load(), f"ccnn_allreduce_{sfx}")
        _l.check(fn(self._h, _ptr(t), t.numel(), _stream()), f"ccnn_allreduce_{sfx}")
        return t

    def allreduce_update(self, params, grad, lr, z, divisor=1, freeze_conv_bias=True):
        """params -= lr * (sum over ranks of grad) [/ divisor for fp64], one kernel."""
        lib = _l.load()
        if params.dtype == torch.float32:
            _l.check(lib.ccnn_allreduce_update_f32(self._h, _ptr(params), _ptr(grad), float(lr), int(z),
                                                   int(freeze_conv_bias), _stream()), "ccnn_allreduce_update_f32")
        else:
            _l.check(lib.ccnn_allreduce_update_f64(self._h, _ptr(params), _ptr(grad), float(lr), int(divisor),
                                                   params.numel(), _stream()), "ccnn_allreduce_update_f64")

    def _agree(self, ok):
        flagged = [None] * self.world
        dist.all_gather_object(flagged, bool(ok), group=self.group)   # all ranks take the same branch
        return all(flagged)

    def _self_test_step(self, z, B=8):
        """One minibatch step with the exchange inside its reduction kernel against gradient +
        the group's all_reduce + update, from the same parameters and per-rank samples."""
        from . import ops
        import numpy as np
        d = _l.dims(z)
        npdt = np.float32 if self.dtype == torch.float32 else np.float64
        rng = np.random.default_rng(77)                      # same parameters on every rank
        p0 = torch.from_numpy(((rng.random(d.nparams) - 0.5) / 10).astype(npdt)).to(self.device)
        rng = np.random.default_rng(78 + self.rank)          # different samples
        cubes = torch.from_numpy(rng.random((B, d.cube)).astype(npdt)).to(self.device)
        oh = np.zeros((B, d.n_classes), dtype=np.int32)
        oh[np.arange(B), rng.integers(0, d.n_classes, B)] = 1
        oh = torch.from_numpy(oh).to(self.device)
        lr = 0.04
        p_dp, p_ref = p0.clone(), p0.clone()
        # The same step without the exchange first: the one-off costs of a process's first launch of
        # these kernels (code object load, LDS limit) must not count against the peers' bounded waits.
        ops.mbgd_step(p0.clone(), cubes, oh, 0, B, z, lr)
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)
        ops.mbgd_step(p_dp, cubes, oh, 0, B, z, lr, comm=self, divisor=B * self.world)
        torch.cuda.synchronize(self.device)
        if not self._agree(self.status() == 0):
            return False
        g, _ = ops.mbgd_grad(p_ref, cubes, oh, 0, B, z)
        if dist.get_backend(self.group) == "nccl":
            dist.all_reduce(g, group=self.group)
        else:
            h = g.cpu()
            dist.all_reduce(h, group=self.group)
            g.copy_(h)
        ops.update(p_ref, g, lr, z, divisor=B * self.world)
        torch.cuda.synchronize(self.device)
        tol = 1e-5 if self.dtype == torch.float32 else 1e-12     # the two sums differ in order only
        err = float((p_dp - p_ref).abs().max()) / float(p_ref.abs().max())
        if err > tol:
            import sys
            print(f"[rank {self.rank}] in-step exchange self-test: relative error {err:.3e} > {tol:.0e}", file=sys.stderr)
        return self._agree(err <= tol)

    def self_test(self, n=None, rounds=4, z=None):
        """True iff the one-shot exchange reproduces the group's all_reduce on random data, on
        every rank, and (with z) the minibatch step that exchanges inside its reduction kernel
        reproduces gradient + all_reduce + update.  Any failure (mismatch, time-out) means: do
        not use it."""
        if z is not None and not self._self_test_step(z):
            return False
        n = self.max_elems if n is None else n
        on_device = dist.get_backend(self.group) == "nccl"   # gloo reduces host tensors
        g = torch.Generator(device="cpu").manual_seed(1234 + self.rank)
        for k in range(rounds):
            # integers in +-2^10: sums are exact in either summation order
            a = torch.randint(-1024, 1025, (n,), generator=g).to(self.dtype).to(self.device)
            ref = a.clone() if on_device else a.cpu()
            dist.all_reduce(ref, group=self.group)
            self.all_reduce(a)
            torch.cuda.synchronize(self.device)
            ok = bool(torch.equal(a.to(ref.device), ref)) and self.status() == 0
            if not ok:
                import sys
                bad = int((a.to(ref.device) != ref).sum())
                print(f"[rank {self.rank}] all-reduce self-test round {k}: {bad} of {n} elements differ, status {self.status()}", file=sys.stderr)
            if not self._agree(ok):
                return False   # a failing call may have cost a whole wait limit: do not repeat it
        return True
