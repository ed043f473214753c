"""Inference throughput on synthetic cubes (diagnostic / profiling driver).
usage: bench_infer.py [z] [n] [f32|f64]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import ops, prep
import ctypes
z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dt = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((n, 9 * z)).astype(dt)).cuda()
ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, dt, 0)).cuda()
ws = ops.Workspace()
for _ in range(3): ops.forward(params, cubes, z, want_probs=False, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 20
for _ in range(reps): ops.forward(params, cubes, z, want_probs=False, workspace=ws)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
print(f"z={z} n={n} {np.dtype(dt).name}: {t*1e3:.3f} ms  {n/t/1e6:.1f} M pixels/s")
