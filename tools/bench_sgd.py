"""fp64 SGD throughput (one sample per step, sequential) on synthetic cubes (diagnostic driver).
usage: bench_sgd.py [z] [n]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import ops, prep
import ctypes
z = int(sys.argv[1]) if len(sys.argv) > 1 else 176
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((n, 9 * z))).cuda()
onehot = np.zeros((n, 13), dtype=np.int32); onehot[np.arange(n), rng.integers(0, 13, n)] = 1
onehot = torch.from_numpy(onehot).cuda()
ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, np.float64)).cuda()
ws = ops.Workspace()
loss = torch.zeros(1, dtype=torch.float64, device="cuda")
ops.sgd(params, cubes, onehot, 0, 256, z, 0.01, loss, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter()
ops.sgd(params, cubes, onehot, 0, n, z, 0.01, loss, workspace=ws)
torch.cuda.synchronize(); t = time.perf_counter() - t0
print(f"z={z} n={n}: {t/n*1e6:.2f} us/sample  {n/t:.0f} pixels/s  loss={float(loss):.4f}")
if os.environ.get("CCNN_SGD_TIMING"):
    raw = ws.buf[2 * 20 * 100 * 2 * 8:].cpu().numpy()
    t = raw[16:16 + 128].view(np.uint64).reshape(2, 8)
    names = ["stage", "conv+pool", "fc1+exchange", "head", "dpool", "updates"]
    for f in range(2):
        print(f"wg{f}: " + "  ".join(f"{names[k]}={t[f, k] * 10 / n:.0f}ns" for k in range(6)))
