"""In-kernel phase times of grad_reduce_split's dW1 workgroups (diagnostic; needs a
-DCCNN_SPLIT_TIMING build picked with CCNN_LIB_PATH).   usage: split_stamps.py [z] [B]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["CCNN_MBGD_PATH"] = "tile"
from gcn_amd import ops, prep, lib as L  # noqa: E402

z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 2, 9 * z)).astype(np.float32)).cuda()
onehot = np.zeros((B * 2, 13), dtype=np.int32)
onehot[np.arange(B * 2), rng.integers(0, 13, B * 2)] = 1
onehot = torch.from_numpy(onehot).cuda()
params = torch.from_numpy(prep.init_params(z, np.float32)).cuda()
ws = ops.Workspace()
lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(20):
    ops.mbgd_step(params, cubes, onehot, (i % 2) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CCNN_LIB_PATH"])
t = (C.c_ulonglong * (1024 * 8))()
assert lib.ccnn_debug_split_ticks(t) == 0
d = L.dims(z)
nblk = ((d.pout + 15) // 16) * 8 * ((B + 2047) // 2048)
a = np.array(t, dtype=np.int64).reshape(1024, 8)[:nblk]
t0 = a[:, 0].min()
rel = (a - t0) * 0.01
names = ["start", "loads issued", "chunk 0 in LDS", "products done", "partial acked", "arrived", "partials read", "updated"]
last = a[:, 7] > a[:, 0]          # workgroups that arrived last wrote stamps 6 and 7 in this launch
print(f"z={z} B={B}: {nblk} dW1 workgroups, {int(last.sum())} of them arrived last; times in us after the first workgroup's start")
for k in range(6):
    print(f"  {names[k]:16s} median {np.median(rel[:, k]):6.2f}  max {rel[:, k].max():6.2f}")
for k in (6, 7):
    print(f"  {names[k]:16s} median {np.median(rel[last, k]):6.2f}  max {rel[last, k].max():6.2f}   (last arrivers)")
