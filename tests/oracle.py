"""ctypes binding of the CPU oracle (oracle/ccnn_oracle.c).

Test infrastructure only: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py.  The product package never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_ORACLE_DIR = os.path.join(os.path.dirname(_HERE), "oracle")
_LIB_PATHS = {13: os.path.join(_ORACLE_DIR, "_build", "libccnn_oracle.so"),
              16: os.path.join(_ORACLE_DIR, "_build", "libccnn_oracle_c16.so"),   # the oracle compiled with ORC_CLASSES=16
              9: os.path.join(_ORACLE_DIR, "_build", "libccnn_oracle_c9.so")}

FILTERS, COV_LEN, STRIDE, POOL_LEN, HIDDEN, PER_LAYER = 20, 19, 2, 2, 100, 9
FILTER_SIZE = PER_LAYER * COV_LEN


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "z", "cube", "re", "mre", "pin", "pout",
        "off_w0", "off_b0", "off_w1", "off_b1", "off_w2", "off_b2", "nparams")]


class SchedF64(C.Structure):
    _fields_ = [("fifo", C.c_double * 5), ("cur_min", C.c_double), ("count", C.c_int)]


class SchedF32(C.Structure):
    _fields_ = [("fifo", C.c_float * 5), ("cur_min", C.c_float), ("count", C.c_int)]


def build():
    """Compile the oracle (both class counts) if a shared object is missing or stale."""
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("ccnn_oracle.c", "ccnn_kernels.inc", "ccnn_oracle.h", "Makefile")]
    fresh = all(os.path.exists(p) and all(os.path.getmtime(p) >= os.path.getmtime(s) for s in srcs)
                for p in _LIB_PATHS.values())
    if not fresh:
        subprocess.check_call(["make", "-C", _ORACLE_DIR, "-s"])
    return _LIB_PATHS[13]


_libs = {}
CLASSES = 13          # the class count of the oracle build in use (see use_classes)


def lib():
    if CLASSES not in _libs:
        build()
        l = C.CDLL(_LIB_PATHS[CLASSES])
        l.orc_mgpu_step_f64.restype = C.c_double
        l.orc_mgpu_first_sample.restype = C.c_long
        _libs[CLASSES] = l
    return _libs[CLASSES]


class use_classes:
    """`with oracle.use_classes(16):` - every call inside goes to the oracle built with that class count."""

    def __init__(self, n):
        assert n in _LIB_PATHS, n
        self.n = n

    def __enter__(self):
        global CLASSES
        self.prev, CLASSES = CLASSES, self.n
        return self

    def __exit__(self, *exc):
        global CLASSES
        CLASSES = self.prev


def _p(a, ct):
    return a.ctypes.data_as(C.POINTER(ct)) if a is not None else None


def _chk(a, dtype):
    assert a.dtype == dtype and a.flags["C_CONTIGUOUS"], (a.dtype, dtype)
    return a


def dims(z):
    d = Dims()
    lib().orc_dims(C.c_int(z), C.byref(d))
    return d


def scan_labels(labels, x, y):
    """labels: float64 array of x*y values in the reference's linear (column-major) order."""
    labels = _chk(np.ascontiguousarray(labels, dtype=np.float64).ravel(), np.float64)
    idx = np.empty(x * y, dtype=np.int32)
    n = lib().orc_scan_labels(_p(labels, C.c_double), x, y, _p(idx, C.c_int))
    return idx[:n].copy()


def split_sizes(n):
    tr, te = C.c_int(), C.c_int()
    lib().orc_split_sizes(n, C.byref(tr), C.byref(te))
    return tr.value, te.value


def build_index(labels, data_index, x, y):
    labels = _chk(np.ascontiguousarray(labels, dtype=np.float64).ravel(), np.float64)
    data_index = _chk(np.ascontiguousarray(data_index, dtype=np.int32), np.int32)
    n = len(data_index)
    tr, te = split_sizes(n)
    train_idx9 = np.empty((tr, 9), dtype=np.int32)
    test_idx9 = np.empty((te, 9), dtype=np.int32)
    onehot = np.empty((tr, CLASSES), dtype=np.int32)
    test_labels = np.empty(te, dtype=np.int32)
    lib().orc_build_index(_p(labels, C.c_double), _p(data_index, C.c_int), n, x, y,
                          _p(train_idx9, C.c_int), _p(test_idx9, C.c_int), _p(onehot, C.c_int),
                          _p(test_labels, C.c_int))
    return train_idx9, test_idx9, onehot, test_labels


def shuffle(idx9, onehot, seed):
    """In place; leaves the libc rand() stream where the reference's weight init picks it up."""
    _chk(idx9, np.int32), _chk(onehot, np.int32)
    lib().orc_shuffle(_p(idx9, C.c_int), _p(onehot, C.c_int), len(idx9), C.c_uint(seed))


def srand(seed):
    C.CDLL(None).srand(C.c_uint(seed))


def init_params_f64(z, pool_bias_draws):
    p = np.empty(dims(z).nparams, dtype=np.float64)
    lib().orc_init_params_f64(_p(p, C.c_double), z, C.c_long(pool_bias_draws))
    return p


def init_params_f32(z):
    p = np.empty(dims(z).nparams, dtype=np.float32)
    lib().orc_init_params_f32(_p(p, C.c_float), z)
    return p


def gather(scene, idx9, x, y, z):
    """scene: flat [z][y][x] (x fastest) float32/float64; idx9 int32 [n][9]."""
    idx9 = _chk(np.ascontiguousarray(idx9, dtype=np.int32), np.int32)
    n = idx9.shape[0]
    scene = np.ascontiguousarray(scene).ravel()
    out = np.empty((n, 9 * z), dtype=scene.dtype)
    if scene.dtype == np.float64:
        lib().orc_gather_f64(_p(scene, C.c_double), _p(idx9, C.c_int), _p(out, C.c_double), x, y, z, C.c_long(n))
    else:
        assert scene.dtype == np.float32
        lib().orc_gather_f32(_p(scene, C.c_float), _p(idx9, C.c_int), _p(out, C.c_float), x, y, z, C.c_long(n))
    return out


def forward(params, cubes, z):
    cubes = np.ascontiguousarray(cubes)
    n = cubes.shape[0]
    probs = np.empty((n, CLASSES), dtype=params.dtype)
    argmax = np.empty(n, dtype=np.int32)
    if params.dtype == np.float64:
        lib().orc_forward_f64(_p(_chk(params, np.float64), C.c_double), _p(_chk(cubes, np.float64), C.c_double),
                              C.c_long(n), z, _p(probs, C.c_double), _p(argmax, C.c_int))
    else:
        lib().orc_forward_f32(_p(_chk(params, np.float32), C.c_float), _p(_chk(cubes, np.float32), C.c_float),
                              C.c_long(n), z, _p(probs, C.c_float), _p(argmax, C.c_int))
    return probs, argmax


def forward_trace_f64(params, cube, z):
    d = dims(z)
    a0 = np.empty(d.pin); pool = np.empty(d.pout); pidx = np.empty(d.pout, dtype=np.int32)
    a1 = np.empty(HIDDEN); probs = np.empty(CLASSES)
    lib().orc_forward_trace_f64(_p(_chk(params, np.float64), C.c_double),
                                _p(_chk(np.ascontiguousarray(cube), np.float64), C.c_double), z,
                                _p(a0, C.c_double), _p(pool, C.c_double), _p(pidx, C.c_int),
                                _p(a1, C.c_double), _p(probs, C.c_double))
    return a0, pool, pidx, a1, probs


def sgd_f64(params, cubes, onehot, first, count, z, lr, loss=0.0):
    """Updates params in place; returns the running device-loss scalar."""
    l = C.c_double(loss)
    lib().orc_sgd_f64(_p(_chk(params, np.float64), C.c_double), _p(_chk(cubes, np.float64), C.c_double),
                      _p(_chk(onehot, np.int32), C.c_int), C.c_long(first), C.c_long(count), z,
                      C.c_double(lr), C.byref(l))
    return l.value


def mbgd_step(params, cubes, onehot, first, B, z, lr, want_grad=True):
    """Updates params in place; returns (loss_values[B], probs[B,13], grad_sum or None)."""
    d = dims(z)
    loss_values = np.empty(B, dtype=np.float64)
    probs = np.empty((B, CLASSES), dtype=params.dtype)
    grad = np.empty(d.nparams, dtype=params.dtype) if want_grad else None
    _chk(onehot, np.int32)
    if params.dtype == np.float64:
        lib().orc_mbgd_step_f64(_p(_chk(params, np.float64), C.c_double), _p(_chk(cubes, np.float64), C.c_double),
                                _p(onehot, C.c_int), C.c_long(first), B, z, C.c_double(lr),
                                _p(loss_values, C.c_double), _p(probs, C.c_double), _p(grad, C.c_double))
    else:
        lib().orc_mbgd_step_f32(_p(_chk(params, np.float32), C.c_float), _p(_chk(cubes, np.float32), C.c_float),
                                _p(onehot, C.c_int), C.c_long(first), B, z, C.c_float(lr),
                                _p(loss_values, C.c_double), _p(probs, C.c_float), _p(grad, C.c_float))
    return loss_values, probs, grad


def mbgd_step_f32_ex(params, cubes, onehot, first, B, z, lr, flip=None, want_a0=False):
    """fp32 step with the test hooks: returns (loss_values, a0 [B, F*re] or None); `flip`
    int32 [B, F*mre] marks pooled features whose pair is decided the other way."""
    d = dims(z)
    loss_values = np.empty(B, dtype=np.float64)
    a0 = np.empty((B, d.pin), dtype=np.float32) if want_a0 else None
    if flip is not None:
        flip = _chk(np.ascontiguousarray(flip, dtype=np.int32), np.int32)
        assert flip.shape == (B, d.pout)
    lib().orc_mbgd_step_f32_ex(_p(_chk(params, np.float32), C.c_float), _p(_chk(cubes, np.float32), C.c_float),
                               _p(_chk(onehot, np.int32), C.c_int), C.c_long(first), B, z, C.c_float(lr),
                               _p(loss_values, C.c_double), None, None, _p(flip, C.c_int), _p(a0, C.c_float))
    return loss_values, a0


def mgpu_step_f64(params, cubes, onehot, train_size, i0, B, n_dev, z, lr):
    return lib().orc_mgpu_step_f64(_p(_chk(params, np.float64), C.c_double), _p(_chk(cubes, np.float64), C.c_double),
                                   _p(_chk(onehot, np.int32), C.c_int), C.c_long(train_size), C.c_long(i0),
                                   B, n_dev, z, C.c_double(lr))


def mgpu_first_sample(train_size, n_dev, B, slot, i0):
    return lib().orc_mgpu_first_sample(C.c_long(train_size), n_dev, B, slot, C.c_long(i0))


def adagrad_update_f64(params, rms, grad_sum, lr, it):
    lib().orc_adagrad_update_f64(_p(_chk(params, np.float64), C.c_double), _p(_chk(rms, np.float64), C.c_double),
                                 _p(_chk(grad_sum, np.float64), C.c_double), C.c_long(len(params)),
                                 C.c_double(lr), C.c_long(it))


def argmax(probs):
    probs = np.ascontiguousarray(probs)
    out = np.empty(len(probs), dtype=np.int32)
    for i, p in enumerate(probs):
        if probs.dtype == np.float64:
            out[i] = lib().orc_argmax_f64(_p(p, C.c_double))
        else:
            out[i] = lib().orc_argmax_f32(_p(p, C.c_float))
    return out


class Sched:
    """LR schedule + early-stop FIFO; init is 1 (SGD, fp64 MBGD, mGPU) or 100 (fp32 MBGD)."""

    def __init__(self, dtype, init):
        self.f32 = dtype == np.float32
        self.s = SchedF32() if self.f32 else SchedF64()
        if self.f32:
            lib().orc_sched_init_f32(C.byref(self.s), C.c_float(init))
        else:
            lib().orc_sched_init_f64(C.byref(self.s), C.c_double(init))

    def step(self, loss, lr):
        """Returns (decayed?, new lr)."""
        if self.f32:
            v = C.c_float(lr)
            r = lib().orc_sched_step_f32(C.byref(self.s), C.c_double(loss), C.byref(v))
        else:
            v = C.c_double(lr)
            r = lib().orc_sched_step_f64(C.byref(self.s), C.c_double(loss), C.byref(v))
        return bool(r), v.value


def num_threads():
    return lib().orc_num_threads()
