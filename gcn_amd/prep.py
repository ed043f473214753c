"""Sample preparation on the host (numpy in, numpy out) through the C ABI's
ccnn_host_* entry points: labelled-pixel scan, 80/20 split, mirrored 3x3
neighbour lists, libc-rand shuffle, weight initialisation, LR schedule and the
multi-GPU partition."""
import ctypes as C

import numpy as np

from . import lib as _l


def _p(a):
    return C.c_void_p(a.ctypes.data)


def scan_labels(labels, x, y):
    """labels: x*y float64 values in the reference's linear order (r + c*x)."""
    labels = np.ascontiguousarray(labels, dtype=np.float64).ravel()
    if labels.size != x * y:
        raise ValueError("labels must hold x*y values")
    idx = np.empty(x * y, dtype=np.int32)
    n = C.c_int()
    _l.check(_l.load().ccnn_host_scan_labels(_p(labels), x, y, _p(idx), C.byref(n)), "ccnn_host_scan_labels")
    return idx[:n.value].copy()


def split_sizes(n):
    tr, te = C.c_int(), C.c_int()
    _l.check(_l.load().ccnn_host_split_sizes(n, C.byref(tr), C.byref(te)), "ccnn_host_split_sizes")
    return tr.value, te.value


def build_index(labels, data_index, x, y, n_classes=_l.CLASSES):
    """-> train_idx9 [tr,9], test_idx9 [te,9], onehot [tr,n_classes], test_labels [te] (all int32)."""
    labels = np.ascontiguousarray(labels, dtype=np.float64).ravel()
    data_index = np.ascontiguousarray(data_index, dtype=np.int32)
    tr, te = split_sizes(len(data_index))
    train_idx9 = np.empty((tr, 9), dtype=np.int32)
    test_idx9 = np.empty((te, 9), dtype=np.int32)
    onehot = np.empty((tr, n_classes), dtype=np.int32)
    test_labels = np.empty(te, dtype=np.int32)
    _l.check(_l.load().ccnn_host_build_index_classes(_p(labels), _p(data_index), len(data_index), x, y, n_classes,
                                                     _p(train_idx9), _p(test_idx9), _p(onehot), _p(test_labels)),
             "ccnn_host_build_index_classes")
    return train_idx9, test_idx9, onehot, test_labels


def shuffle(idx9, onehot, seed):
    """In place; seeds libc rand() with `seed` and leaves the stream for init_params."""
    assert idx9.dtype == np.int32 and onehot.dtype == np.int32 and idx9.flags.c_contiguous and onehot.flags.c_contiguous
    n_classes = onehot.shape[1] if onehot.ndim == 2 else _l.CLASSES
    _l.check(_l.load().ccnn_host_shuffle_classes(_p(idx9), _p(onehot), len(idx9), n_classes, C.c_uint(seed)),
             "ccnn_host_shuffle_classes")


def init_params(z, dtype, pool_bias_draws=0):
    d = _l.dims(z)
    p = np.empty(d.nparams, dtype=dtype)
    if p.dtype == np.float64:
        _l.check(_l.load().ccnn_host_init_params_f64(_p(p), z, int(pool_bias_draws)), "ccnn_host_init_params_f64")
    elif p.dtype == np.float32:
        _l.check(_l.load().ccnn_host_init_params_f32(_p(p), z), "ccnn_host_init_params_f32")
    else:
        raise TypeError(dtype)
    return p


class LrSchedule:
    """FIFO-of-five plateau schedule; fifo_init 1 (fp64 trainers) or 100 (fp32 MBGD)."""

    def __init__(self, fifo_init=1.0, use_float=False):
        self.s = _l.LrSched()
        _l.check(_l.load().ccnn_host_sched_init(C.byref(self.s), float(fifo_init), int(use_float)),
                 "ccnn_host_sched_init")

    def step(self, epoch_loss, lr):
        v = C.c_double(lr)
        r = _l.load().ccnn_host_sched_step(C.byref(self.s), float(epoch_loss), C.byref(v))
        if r < 0:
            _l.check(r, "ccnn_host_sched_step")
        return bool(r), v.value


def partition(n_train, batch, n_ranks, rank, step):
    first, sub = C.c_long(), C.c_int()
    _l.check(_l.load().ccnn_host_partition(n_train, batch, n_ranks, rank, step, C.byref(first), C.byref(sub)),
             "ccnn_host_partition")
    return first.value, sub.value
