"""fp64 MBGD step time (the reference's canonical build: gcn_mbgd_fp64.cu, B = 100).
usage: bench_mbgd_f64.py [z] [B]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import ops, prep, lib as L
z = int(sys.argv[1]) if len(sys.argv) > 1 else 176
B = int(sys.argv[2]) if len(sys.argv) > 2 else 100
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 8, 9 * z))).cuda()
onehot = np.zeros((B * 8, 13), dtype=np.int32); onehot[np.arange(B * 8), rng.integers(0, 13, B * 8)] = 1
onehot = torch.from_numpy(onehot).cuda()
import ctypes; ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, np.float64)).cuda()
ws = ops.Workspace(); lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(50): ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 500
for i in range(n): ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
L.profile_enable(True)
for i in range(50): ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize(); L.profile_enable(False)
print(f"fp64 z={z} B={B}: {dt * 1e6:.1f} us/step  {B / dt / 1e6:.2f} M pixels/s;",
      {k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in L.profile_report().items()})
