"""Tensor-level operators over libccnn_hip.so.

Each function takes CUDA (ROCm) torch tensors, checks shapes/dtypes on the host
so that a kernel never sees an operand it was not written for, and enqueues the
work on torch's current stream.  Names follow the reference's kernel groups:
gather (preprocessing), forward (the test loop), sgd, mbgd_grad, update.
"""
import ctypes as C
import os

import torch

from . import lib as _l

_F32, _F64 = torch.float32, torch.float64


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _req(t, dtype, name, numel=None):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise TypeError(f"{name} must be a CUDA tensor")
    if t.dtype != dtype:
        raise TypeError(f"{name} must be {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name} must be contiguous")
    if numel is not None and t.numel() != numel:
        raise ValueError(f"{name} has {t.numel()} elements, expected {numel}")
    return t


def _sfx(dtype):
    if dtype == _F32:
        return "f32", 4
    if dtype == _F64:
        return "f64", 8
    raise TypeError(f"unsupported dtype {dtype}")


_WS_BYTES = {}


def _mbgd_ws_bytes(lib, B, z, es):
    """ccnn_mbgd_workspace_bytes, remembered (the training loop asks on every step).  The size depends
    on which implementation of the step the library picks, and CCNN_MBGD_PATH / CCNN_TILE_WAVES can
    override that."""
    k = (B, z, es, os.environ.get("CCNN_MBGD_PATH"), os.environ.get("CCNN_TILE_WAVES"))
    n = _WS_BYTES.get(k)
    if n is None:
        n = _WS_BYTES[k] = lib.ccnn_mbgd_workspace_bytes(B, z, es)
    return n


class Workspace:
    """A reusable device scratch buffer (grown on demand, never shrunk)."""

    def __init__(self, device="cuda"):
        self.device = device
        self.buf = None

    def get(self, nbytes):
        if self.buf is None or self.buf.numel() < nbytes:
            self.buf = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=self.device)
        return self.buf


def gather(scene, idx9, x, y, z, out=None, workspace=None, sort=True):
    """scene: flat [z*y*x] (x fastest); idx9 int32 [n, 9] -> cubes [n, 9z].
    sort=True lets the library visit the samples in centre-pixel order (same result, fewer reads)."""
    sfx, _ = _sfx(scene.dtype)
    _req(scene, scene.dtype, "scene", x * y * z)
    _req(idx9, torch.int32, "idx9")
    n = idx9.numel() // 9
    if idx9.numel() != n * 9:
        raise ValueError("idx9 must hold 9 indices per sample")
    if out is None:
        out = torch.empty((n, 9 * z), dtype=scene.dtype, device=scene.device)
    _req(out, scene.dtype, "out", n * 9 * z)
    lib = _l.load()
    ws, ws_bytes = None, 0
    if sort:
        ws_bytes = lib.ccnn_gather_workspace_bytes(n)
        if ws_bytes:
            ws = (workspace or Workspace(scene.device)).get(ws_bytes)
    fn = getattr(lib, f"ccnn_gather_{sfx}")
    _l.check(fn(_ptr(scene), _ptr(idx9), n, x, y, z, _ptr(out), _ptr(ws), ws.numel() if ws is not None else 0,
                _stream()), f"ccnn_gather_{sfx}")
    return out


class GatherPlan:
    """One sample list prepared for repeated gathers (other scenes, other passes):
    `GatherPlan(idx9, x, y)` looks at the list once; `plan.gather(scene, z)` then runs only the
    gather kernel - tiles staged in LDS (`plan.dense`) for densely labelled lists of plain 3x3
    neighbourhoods, the ordered-list kernel otherwise.  Holds its own workspace; idx9 must stay
    alive and unchanged."""

    def __init__(self, idx9, x, y):
        _req(idx9, torch.int32, "idx9")
        self.idx9, self.x, self.y = idx9, x, y
        self.n = idx9.numel() // 9
        if idx9.numel() != self.n * 9:
            raise ValueError("idx9 must hold 9 indices per sample")
        lib = _l.load()
        nbytes = lib.ccnn_gather_plan_workspace_bytes(self.n, x, y)
        self.ws = torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=idx9.device)
        self._h = C.c_void_p()
        _l.check(lib.ccnn_gather_plan_create(_ptr(idx9), self.n, x, y, _ptr(self.ws), self.ws.numel(), _stream(),
                                             C.byref(self._h)), "ccnn_gather_plan_create")
        self.dense = lib.ccnn_gather_plan_kind(self._h) == 1

    def gather(self, scene, z, out=None):
        sfx, _ = _sfx(scene.dtype)
        _req(scene, scene.dtype, "scene", self.x * self.y * z)
        if out is None:
            out = torch.empty((self.n, 9 * z), dtype=scene.dtype, device=scene.device)
        _req(out, scene.dtype, "out", self.n * 9 * z)
        fn = getattr(_l.load(), f"ccnn_gather_planned_{sfx}")
        _l.check(fn(self._h, _ptr(scene), z, _ptr(out), _stream()), f"ccnn_gather_planned_{sfx}")
        return out

    def __del__(self):
        if getattr(self, "_h", None) and _l is not None:      # module globals are gone at interpreter exit
            _l.load().ccnn_gather_plan_destroy(self._h)
            self._h = None


def forward(params, cubes, z, want_probs=True, want_argmax=True, workspace=None):
    """Test-loop forward over cubes [n, 9z] -> (probs [n, classes] | None, argmax int32 [n] | None).
    z: the band count, or lib.geometry(z, n_classes) for a class count other than 13 (here and below)."""
    sfx, es = _sfx(params.dtype)
    d = _l.dims(z)
    _req(params, params.dtype, "params", d.nparams)
    n = cubes.numel() // d.cube
    _req(cubes, params.dtype, "cubes", n * d.cube)
    probs = torch.empty((n, d.n_classes), dtype=params.dtype, device=params.device) if want_probs else None
    argmax = torch.empty(n, dtype=torch.int32, device=params.device) if want_argmax else None
    if n == 0:
        return probs, argmax
    lib = _l.load()
    nbytes = lib.ccnn_forward_workspace_bytes(n, z, es)
    ws = (workspace or Workspace(params.device)).get(nbytes)
    fn = getattr(lib, f"ccnn_forward_{sfx}")
    _l.check(fn(_ptr(params), _ptr(cubes), n, z, _ptr(probs), _ptr(argmax), _ptr(ws), ws.numel(), _stream()),
             f"ccnn_forward_{sfx}")
    return probs, argmax


def forward_scene(params, scene, idx9, x, y, z, want_probs=True, want_argmax=True, workspace=None):
    """forward() on pixels of the scene: idx9 int32 [n, 9] neighbour lists; no cubes are stored."""
    sfx, es = _sfx(params.dtype)
    d = _l.dims(z)
    _req(params, params.dtype, "params", d.nparams)
    _req(scene, params.dtype, "scene", x * y * d.z)
    _req(idx9, torch.int32, "idx9")
    n = idx9.numel() // 9
    if idx9.numel() != n * 9:
        raise ValueError("idx9 must hold 9 indices per sample")
    probs = torch.empty((n, d.n_classes), dtype=params.dtype, device=params.device) if want_probs else None
    argmax = torch.empty(n, dtype=torch.int32, device=params.device) if want_argmax else None
    if n == 0:
        return probs, argmax
    lib = _l.load()
    ws = (workspace or Workspace(params.device)).get(lib.ccnn_forward_workspace_bytes(n, z, es))
    fn = getattr(lib, f"ccnn_forward_scene_{sfx}")
    _l.check(fn(_ptr(params), _ptr(scene), _ptr(idx9), n, x, y, z, _ptr(probs), _ptr(argmax), _ptr(ws), ws.numel(),
                _stream()), f"ccnn_forward_scene_{sfx}")
    return probs, argmax


def count_correct(argmax, labels):
    """Device count of argmax+1 == labels; returns a 1-element int32 tensor."""
    _req(argmax, torch.int32, "argmax")
    _req(labels, torch.int32, "labels", argmax.numel())
    count = torch.empty(1, dtype=torch.int32, device=argmax.device)
    _l.check(_l.load().ccnn_count_correct(_ptr(argmax), _ptr(labels), argmax.numel(), _ptr(count), _stream()),
             "ccnn_count_correct")
    return count


def sgd(params, cubes, onehot, first, count, z, lr, loss, workspace=None):
    """fp64 per-sample SGD over samples [first, first+count); params and the
    1-element fp64 tensor `loss` are updated in place."""
    d = _l.dims(z)
    _req(params, _F64, "params", d.nparams)
    n = cubes.numel() // d.cube
    _req(cubes, _F64, "cubes", n * d.cube)
    _req(onehot, torch.int32, "onehot", n * d.n_classes)
    _req(loss, _F64, "loss", 1)
    if first < 0 or count < 0 or first + count > n:
        raise ValueError(f"samples [{first}, {first + count}) outside [0, {n})")
    lib = _l.load()
    ws = (workspace or Workspace(params.device)).get(lib.ccnn_sgd_workspace_bytes(z))
    _l.check(lib.ccnn_sgd_f64(_ptr(params), _ptr(cubes), _ptr(onehot), first, count, z, float(lr), _ptr(loss),
                              _ptr(ws), ws.numel(), _stream()), "ccnn_sgd_f64")


def mbgd_grad(params, cubes, onehot, first, B, z, grad=None, loss_values=None, probs=None, workspace=None):
    """Batch-summed gradient of samples [first, first+B) -> (grad, loss_values[B] fp64)."""
    sfx, es = _sfx(params.dtype)
    d = _l.dims(z)
    _req(params, params.dtype, "params", d.nparams)
    n = cubes.numel() // d.cube
    _req(cubes, params.dtype, "cubes", n * d.cube)
    _req(onehot, torch.int32, "onehot", n * d.n_classes)
    if B <= 0 or first < 0 or first + B > n:
        raise ValueError(f"batch [{first}, {first + B}) outside [0, {n})")
    if grad is None:
        grad = torch.empty(d.nparams, dtype=params.dtype, device=params.device)
    _req(grad, params.dtype, "grad", d.nparams)
    if loss_values is None:
        loss_values = torch.empty(B, dtype=_F64, device=params.device)
    _req(loss_values, _F64, "loss_values")
    if loss_values.numel() < B:
        raise ValueError("loss_values shorter than the batch")
    if probs is not None:
        _req(probs, params.dtype, "probs", B * d.n_classes)
    lib = _l.load()
    ws = (workspace or Workspace(params.device)).get(_mbgd_ws_bytes(lib, B, z, es))
    fn = getattr(lib, f"ccnn_mbgd_grad_{sfx}")
    _l.check(fn(_ptr(params), _ptr(cubes), _ptr(onehot), first, B, z, _ptr(grad), _ptr(loss_values), _ptr(probs),
                _ptr(ws), ws.numel(), _stream()), f"ccnn_mbgd_grad_{sfx}")
    return grad, loss_values


def mbgd_step(params, cubes, onehot, first, B, z, lr, loss_values=None, workspace=None, comm=None, divisor=None):
    """Gradient and the variant's own update in one call; params updated in place.
    comm (a gcn_amd.comm.NodeComm): the data-parallel step - this rank's B samples, the sums
    exchanged with the other ranks inside the reduction kernel; fp64 then divides by `divisor`
    (the global batch)."""
    sfx, es = _sfx(params.dtype)
    d = _l.dims(z)
    _req(params, params.dtype, "params", d.nparams)
    n = cubes.numel() // d.cube
    _req(cubes, params.dtype, "cubes", n * d.cube)
    _req(onehot, torch.int32, "onehot", n * d.n_classes)
    if B <= 0 or first < 0 or first + B > n:
        raise ValueError(f"batch [{first}, {first + B}) outside [0, {n})")
    if loss_values is None:
        loss_values = torch.empty(B, dtype=_F64, device=params.device)
    _req(loss_values, _F64, "loss_values")
    if loss_values.numel() < B:
        raise ValueError("loss_values shorter than the batch")
    lib = _l.load()
    ws = (workspace or Workspace(params.device)).get(_mbgd_ws_bytes(lib, B, z, es))
    if comm is not None:
        fn = getattr(lib, f"ccnn_mbgd_step_dp_{sfx}")
        extra = () if params.dtype == torch.float32 else (int(divisor if divisor else B * comm.world),)
        _l.check(fn(comm._h, _ptr(params), _ptr(cubes), _ptr(onehot), first, B, z, float(lr), *extra,
                    _ptr(loss_values), _ptr(ws), ws.numel(), _stream()), f"ccnn_mbgd_step_dp_{sfx}")
        return loss_values
    fn = getattr(lib, f"ccnn_mbgd_step_{sfx}")
    _l.check(fn(_ptr(params), _ptr(cubes), _ptr(onehot), first, B, z, float(lr), _ptr(loss_values), _ptr(ws),
                ws.numel(), _stream()), f"ccnn_mbgd_step_{sfx}")
    return loss_values


class MbgdStepper:
    """mbgd_step for a training loop: everything that does not change from step to step (argument
    checks, workspace, pointers, the stream, the C function) is resolved once, so a step costs one
    foreign call - the host side of ops.mbgd_step is about 9 us per step, which is half of what the
    GPU needs for a step of 128 samples.  The tensors must stay alive and in place while the stepper
    is used; steps run on the stream that was current when the stepper was made.
        st = MbgdStepper(params, cubes, onehot, B, z, loss_values=ring)   # ring: [nb, B] float64 or None
        st.step(first, lr, row)                                           # loss of the step -> ring[row]
    comm / divisor as in mbgd_step (the data-parallel step)."""

    def __init__(self, params, cubes, onehot, B, z, loss_values=None, workspace=None, comm=None, divisor=None):
        sfx, es = _sfx(params.dtype)
        d = _l.dims(z)
        _req(params, params.dtype, "params", d.nparams)
        self.n = cubes.numel() // d.cube
        _req(cubes, params.dtype, "cubes", self.n * d.cube)
        _req(onehot, torch.int32, "onehot", self.n * d.n_classes)
        if B <= 0 or B > self.n:
            raise ValueError(f"batch of {B} does not fit {self.n} samples")
        if loss_values is None:
            loss_values = torch.empty((1, B), dtype=_F64, device=params.device)
        _req(loss_values, _F64, "loss_values")
        if loss_values.numel() < B or loss_values.numel() % B:
            raise ValueError("loss_values must hold whole rows of B values")
        lib = _l.load()
        self.B, self.z, self.rows = int(B), int(z), loss_values.numel() // B
        self._keep = (params, cubes, onehot, loss_values, workspace, comm)
        self._ws = (workspace or Workspace(params.device)).get(_mbgd_ws_bytes(lib, B, z, es))
        self._p, self._c, self._o = _ptr(params), _ptr(cubes), _ptr(onehot)
        self._lv, self._lv_stride = loss_values.data_ptr(), 8 * B
        self._wsp, self._wsn, self._st = _ptr(self._ws), self._ws.numel(), _stream()
        self._f32 = params.dtype == torch.float32
        if comm is not None:
            self._fn = getattr(lib, f"ccnn_mbgd_step_dp_{sfx}")
            self._comm = comm._h
            self._div = int(divisor if divisor else B * comm.world)
        else:
            self._fn = getattr(lib, f"ccnn_mbgd_step_{sfx}")
            self._comm = None
        self._what = self._fn.__name__

    def step(self, first, lr, row=0):
        if first < 0 or first + self.B > self.n or row < 0 or row >= self.rows:
            raise ValueError(f"batch [{first}, {first + self.B}) / loss row {row} out of range")
        lv = C.c_void_p(self._lv + row * self._lv_stride)
        if self._comm is None:
            rc = self._fn(self._p, self._c, self._o, first, self.B, self.z, lr, lv, self._wsp, self._wsn, self._st)
        elif self._f32:
            rc = self._fn(self._comm, self._p, self._c, self._o, first, self.B, self.z, lr, lv, self._wsp, self._wsn, self._st)
        else:
            rc = self._fn(self._comm, self._p, self._c, self._o, first, self.B, self.z, lr, self._div, lv, self._wsp,
                          self._wsn, self._st)
        if rc:
            _l.check(rc, self._what)


def update(params, grad, lr, z, divisor=None, freeze_conv_bias=True):
    """fp64: params -= lr*grad/divisor.  fp32: params -= lr*grad with the conv bias
    left untouched (the fp32 reference never updates it)."""
    d = _l.dims(z)
    _req(params, params.dtype, "params", d.nparams)
    _req(grad, params.dtype, "grad", d.nparams)
    lib = _l.load()
    if params.dtype == _F64:
        if not divisor or divisor <= 0:
            raise ValueError("fp64 update needs the batch size as divisor")
        _l.check(lib.ccnn_update_f64(_ptr(params), _ptr(grad), float(lr), int(divisor), d.nparams, _stream()),
                 "ccnn_update_f64")
    else:
        _sfx(params.dtype)
        _l.check(lib.ccnn_update_f32(_ptr(params), _ptr(grad), float(lr), z, int(bool(freeze_conv_bias)),
                                     _stream()), "ccnn_update_f32")


def update_adagrad(params, rms, grad, lr, it):
    _req(params, _F64, "params")
    _req(rms, _F64, "rms", params.numel())
    _req(grad, _F64, "grad", params.numel())
    _l.check(_l.load().ccnn_update_adagrad_f64(_ptr(params), _ptr(rms), _ptr(grad), float(lr), int(it),
                                               params.numel(), _stream()), "ccnn_update_adagrad_f64")
