"""MI355X-native Cube-CNN hot path (drop-in for the jdjd1114/GCN trainers).

The compute lives in ``gcn_amd/lib/libccnn_hip.so`` (hand-written gfx950 HIP
kernels behind the C ABI of ``include/ccnn.h``).  This package is the Python
host: a ctypes binding (``gcn_amd.lib``), tensor-level operators that mirror the
reference's kernel groups (``gcn_amd.ops``), sample preparation
(``gcn_amd.prep``) and the training drivers (``gcn_amd.train``).  PyTorch is used
for device memory, streams and ``torch.distributed`` only.
"""
from . import lib  # noqa: F401
