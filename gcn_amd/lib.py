"""ctypes binding of libccnn_hip.so.  There is no CPU fallback: if the shared
library is missing the import of any compute entry point fails loudly."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("CCNN_LIB_PATH") or os.path.join(_HERE, "lib", "libccnn_hip.so")

FILTERS, COV_LEN, STRIDE, POOL_LEN, HIDDEN, CLASSES, PER_LAYER, FILTER_SIZE = 20, 19, 2, 2, 100, 13, 9, 171


class CcnnError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what} failed with code {code}: {error_string(code)}")


class Dims(C.Structure):
    _fields_ = [(n, C.c_int) for n in (
        "z", "cube", "re", "mre", "pin", "pout",
        "off_w0", "off_b0", "off_w1", "off_b1", "off_w2", "off_b2", "nparams", "n_classes")]


def geometry(z, n_classes=CLASSES):
    """What the entry points take as `z` when the class count is not 13 (CCNN_GEOMETRY)."""
    return int(z) | (int(n_classes) << 16) if n_classes != CLASSES else int(z)


class LrSched(C.Structure):
    _fields_ = [("fifo", C.c_double * 5), ("cur_min", C.c_double), ("count", C.c_int), ("use_float", C.c_int)]


_P = C.c_void_p
_SIG = {
    # name: (restype, argtypes)
    "ccnn_get_dims": (C.c_int, [C.c_int, C.POINTER(Dims)]),
    "ccnn_error_string": (C.c_char_p, [C.c_int]),
    "ccnn_gather_workspace_bytes": (C.c_size_t, [C.c_long]),
    "ccnn_gather_f32": (C.c_int, [_P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "ccnn_gather_f64": (C.c_int, [_P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P, _P, C.c_size_t, _P]),
    "ccnn_gather_plan_workspace_bytes": (C.c_size_t, [C.c_long, C.c_int, C.c_int]),
    "ccnn_gather_plan_create": (C.c_int, [_P, C.c_long, C.c_int, C.c_int, _P, C.c_size_t, _P, C.POINTER(_P)]),
    "ccnn_gather_plan_kind": (C.c_int, [_P]),
    "ccnn_gather_plan_destroy": (C.c_int, [_P]),
    "ccnn_gather_planned_f32": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "ccnn_gather_planned_f64": (C.c_int, [_P, _P, C.c_int, _P, _P]),
    "ccnn_forward_workspace_bytes": (C.c_size_t, [C.c_long, C.c_int, C.c_int]),
    "ccnn_forward_f32": (C.c_int, [_P, _P, C.c_long, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_forward_f64": (C.c_int, [_P, _P, C.c_long, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_count_correct": (C.c_int, [_P, _P, C.c_long, _P, _P]),
    "ccnn_sgd_workspace_bytes": (C.c_size_t, [C.c_int]),
    "ccnn_sgd_f64": (C.c_int, [_P, _P, _P, C.c_long, C.c_long, C.c_int, C.c_double, _P, _P, C.c_size_t, _P]),
    "ccnn_mbgd_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "ccnn_mbgd_path": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "ccnn_mbgd_grad_f32": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_mbgd_grad_f64": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, _P, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_mbgd_step_f32": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_float, _P, _P, C.c_size_t, _P]),
    "ccnn_mbgd_step_noalloc_f32": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_float, _P, _P]),
    "ccnn_mbgd_step_f64": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_double, _P, _P, C.c_size_t, _P]),
    "ccnn_update_f64": (C.c_int, [_P, _P, C.c_double, C.c_int, C.c_long, _P]),
    "ccnn_update_f32": (C.c_int, [_P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "ccnn_update_adagrad_f64": (C.c_int, [_P, _P, _P, C.c_double, C.c_long, C.c_long, _P]),
    "ccnn_host_scan_labels": (C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(C.c_int)]),
    "ccnn_host_split_sizes": (C.c_int, [C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "ccnn_host_build_index": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "ccnn_host_shuffle": (C.c_int, [_P, _P, C.c_int, C.c_uint]),
    "ccnn_host_build_index_classes": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P, _P, _P]),
    "ccnn_host_shuffle_classes": (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_uint]),
    "ccnn_host_init_params_f64": (C.c_int, [_P, C.c_int, C.c_long]),
    "ccnn_host_init_params_f32": (C.c_int, [_P, C.c_int]),
    "ccnn_host_sched_init": (C.c_int, [C.POINTER(LrSched), C.c_double, C.c_int]),
    "ccnn_host_sched_step": (C.c_int, [C.POINTER(LrSched), C.c_double, C.POINTER(C.c_double)]),
    "ccnn_comm_create": (C.c_int, [C.c_int, C.c_int, C.c_long, C.c_int, C.POINTER(_P), _P]),
    "ccnn_comm_connect": (C.c_int, [_P, _P]),
    "ccnn_comm_connect_local": (C.c_int, [_P, C.POINTER(_P)]),
    "ccnn_comm_status": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "ccnn_comm_shares_device": (C.c_int, [_P, C.POINTER(C.c_int)]),
    "ccnn_comm_set_exchange": (C.c_int, [_P, C.c_int]),
    "ccnn_comm_disconnect": (C.c_int, [_P]),
    "ccnn_comm_destroy": (C.c_int, [_P]),
    "ccnn_allreduce_f32": (C.c_int, [_P, _P, C.c_long, _P]),
    "ccnn_allreduce_f64": (C.c_int, [_P, _P, C.c_long, _P]),
    "ccnn_allreduce_update_f32": (C.c_int, [_P, _P, _P, C.c_float, C.c_int, C.c_int, _P]),
    "ccnn_allreduce_update_f64": (C.c_int, [_P, _P, _P, C.c_double, C.c_int, C.c_long, _P]),
    "ccnn_mbgd_step_dp_f32": (C.c_int, [_P, _P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_float, _P, _P, C.c_size_t, _P]),
    "ccnn_mbgd_step_dp_f64": (C.c_int, [_P, _P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_double, C.c_int, _P, _P,
                                        C.c_size_t, _P]),
    "ccnn_forward_scene_f32": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_forward_scene_f64": (C.c_int, [_P, _P, _P, C.c_long, C.c_int, C.c_int, C.c_int, _P, _P, _P, C.c_size_t, _P]),
    "ccnn_profile_enable": (C.c_int, [C.c_int]),
    "ccnn_profile_report": (C.c_int, [C.c_int, C.c_char_p, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_long)]),
    "ccnn_host_partition": (C.c_int, [C.c_long, C.c_int, C.c_int, C.c_int, C.c_long,
                                      C.POINTER(C.c_long), C.POINTER(C.c_int)]),
}

EXPORTS = tuple(_SIG)

_lib = None


def load():
    """Load libccnn_hip.so and bind every declared entry point; raises if the
    library or any symbol is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    # torch first: its wheel carries its own copy of the HIP runtime, and a process must hold only one -
    # loaded after torch, this library binds to the runtime torch already brought in (same soname);
    # loaded before it, the two would each initialise a runtime of their own and this library's
    # launches would fail with "no ROCm-capable device" on tensors torch allocated
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIG.items():
        fn = getattr(lib, name)   # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def error_string(code):
    return load().ccnn_error_string(int(code)).decode()


def check(code, what):
    if code != 0:
        raise CcnnError(code, what)


_DIMS = {}


def dims(z):
    """ccnn_get_dims, remembered per geometry (every op asks on every call)."""
    d = _DIMS.get(z)
    if d is None:
        d = Dims()
        check(load().ccnn_get_dims(int(z), C.byref(d)), "ccnn_get_dims")
        _DIMS[z] = d
    return d


def profile_enable(on):
    load().ccnn_profile_enable(int(bool(on)))


def profile_report():
    """{kernel name: (total ms, launches)} for everything launched while timing was enabled."""
    lib = load()
    out = {}
    name = C.create_string_buffer(64)
    ms, cnt = C.c_double(), C.c_long()
    i = 0
    while lib.ccnn_profile_report(i, name, 64, C.byref(ms), C.byref(cnt)):
        out[name.value.decode()] = (ms.value, cnt.value)
        i += 1
    return out
