"""Where fc_head_fwd<TRAIN> spends its time: per-workgroup stamps of a -DCCNN_CONV_CLOCK build
(tools/build_variant.py clock -DCCNN_CONV_CLOCK; CCNN_LIB_PATH=tools/_variants/libccnn_clock.so)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import lib as L, ops, prep  # noqa: E402

z, B = 103, int(sys.argv[1]) if len(sys.argv) > 1 else 2048
os.environ["CCNN_MBGD_PATH"] = "layers"
d = L.dims(z)
rng = np.random.default_rng(1)
cubes = torch.from_numpy(rng.random((B, 9 * z)).astype(np.float32)).cuda()
oh = np.zeros((B, 13), dtype=np.int32)
oh[np.arange(B), rng.integers(0, 13, B)] = 1
oh = torch.from_numpy(oh).cuda()
p = torch.from_numpy(prep.init_params(z, np.float32, d.pout)).cuda()
ws = ops.Workspace()
for i in range(30):
    ops.mbgd_step(p, cubes, oh, 0, B, z, 0.001, workspace=ws)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CCNN_LIB_PATH"])
w = (C.c_ulonglong * (4 * 4096))()
assert lib.ccnn_debug_fc_wg(w) == 0
a = np.array(w[:], dtype=np.int64).reshape(4096, 4)
a = a[a[:, 0] > 0]
us = (a - a[:, 0].min()) / 100.0
dd = np.diff(us, axis=1)
print(f"fc_head train B={B}: {len(a)} workgroups; entry median {np.median(us[:,0]):.1f} max {us[:,0].max():.1f}; exit max {us[:,3].max():.1f} us; "
      f"per workgroup: prologue {np.median(dd[:,0]):.1f}, K loop {np.median(dd[:,1]):.1f}, epilogue {np.median(dd[:,2]):.1f} us (medians)")
w = (C.c_ulonglong * (3 * 2048))()
assert lib.ccnn_debug_conv_wg(w) == 0
a = np.array(w[:], dtype=np.int64).reshape(2048, 3)
a = a[a[:, 0] > 0]
us = (a - a[:, 0].min()) / 100.0
print(f"conv_pool_fwd train: {len(a)} workgroups; entry median {np.median(us[:,0]):.1f} max {us[:,0].max():.1f}; "
      f"loop start median {np.median(us[:,1]):.1f} max {us[:,1].max():.1f}; loop end median {np.median(us[:,2]):.1f} max {us[:,2].max():.1f} us")
