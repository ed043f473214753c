"""In-kernel phase times of tile_fwd_bwd (diagnostic; needs a -DCCNN_TILE_TIMING build picked with CCNN_LIB_PATH).
usage: tile_stamps.py [z] [B] [f32|f64]"""
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
DT = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 2, 9 * z)).astype(DT)).cuda()
onehot = np.zeros((B * 2, 13), dtype=np.int32)
onehot[np.arange(B * 2), rng.integers(0, 13, B * 2)] = 1
onehot = torch.from_numpy(onehot).cuda()
params = torch.from_numpy(prep.init_params(z, DT)).cuda()
ws = ops.Workspace()
lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(20):
    ops.mbgd_step(params, cubes, onehot, (i % 2) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize()
lib = C.CDLL(os.environ["CCNN_LIB_PATH"])
t = (C.c_ulonglong * 32)()
assert lib.ccnn_debug_tile_ticks(t) == 0
names = ["stage", "conv+pool", "fc1 K loop", "fc1 reduce", "w1 prefetch", "head", "dpool", "bias", "conv_bwd", "store"]
print(f"z={z} B={B} {DT.__name__}: " + "  ".join(f"{names[k]}={(t[k + 1] - t[k]) * 10}ns" for k in range(10)),
      f"total={(t[10] - t[0]) * 10}ns")
print("  stage: small operands stored +%dns, barrier (first loads landed) +%dns, filters scattered +%dns, cube in LDS +%dns, barrier +%dns"
      % tuple((t[k] - t[0]) * 10 for k in (11, 12, 13, 14, 15)))
print("  shader clock per phase (GHz): " + "  ".join(
    f"{names[k]}={(t[21 + k] - t[20 + k]) / ((t[k + 1] - t[k]) * 10.0):.2f}" for k in range(10) if t[k + 1] > t[k]))
if hasattr(lib, "ccnn_debug_tile_wg"):
    w = (C.c_ulonglong * (1024 * 17))()
    assert lib.ccnn_debug_tile_wg(w) == 0
    a = np.array(w[:], dtype=np.int64).reshape(1024, 17)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    start = (a[:, 0] - t0) / 100.0
    ends = np.where(a[:, 1:] > 0, a[:, 1:], a[:, :1])
    end = (ends.max(axis=1) - t0) / 100.0
    dur = end - start
    print(f"  {len(a)} workgroups: start median +{np.median(start):.2f} max +{start.max():.2f} us; own duration median {np.median(dur):.2f} "
          f"min {dur.min():.2f} max {dur.max():.2f} us; last end +{end.max():.2f} us")
    q = np.percentile(dur, [10, 50, 90, 99])
    print(f"  durations p10 {q[0]:.2f} p50 {q[1]:.2f} p90 {q[2]:.2f} p99 {q[3]:.2f}; by XCD (block mod 8), median: "
          + " ".join(f"{np.median(dur[i::8]):.1f}" for i in range(8)))
