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

    def _destroy(self):
        if getattr(self, "_h", None):
            _l.load().ccnn_comm_destroy(self._h)
            self._h = None

    def close(self):
        if getattr(self, "_h", None):
            torch.cuda.synchronize(self.device)
            dist.barrier(group=self.group)   # nobody unmaps while a peer's kernel may still write
            self._destroy()

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

    def self_test(self, n=None, rounds=4):
        """True iff the one-shot exchange reproduces the group's all_reduce on random data, on
        every rank.  Any failure (mismatch, time-out) means: do not use it."""
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
            flagged = [None] * self.world
            dist.all_gather_object(flagged, ok, group=self.group)   # all ranks leave the loop together
            if not all(flagged):
                return False   # a failing call may have cost a whole wait limit: do not repeat it
        return True
