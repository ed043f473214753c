"""N minibatch steps of batch B on synthetic cubes of a scene shape - the fixed workload the
profiler passes run (rocprofv3 ... -- python3 tools/run_steps.py pavia_u 2048 60).
    run_steps.py [workload] [B] [steps] [f32|f64]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from gcn_amd import lib as L, ops, prep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pavia_u"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dt = np.float64 if len(sys.argv) > 4 and sys.argv[4] == "f64" else np.float32
x, y, z, nl = synth.SCENES[name]
d = L.dims(z)
rng = np.random.default_rng(1)
n = max(4 * B, 8192)
cubes = torch.from_numpy(rng.random((n, 9 * z), dtype=dt)).cuda()
oh = np.zeros((n, 13), dtype=np.int32)
oh[np.arange(n), rng.integers(0, 13, n)] = 1
oh = torch.from_numpy(oh).cuda()
p = torch.from_numpy(prep.init_params(z, dt)).cuda()
lr = 0.04 * min(1.0, 128.0 / B) if dt is np.float32 else 0.5       # fp32: plain-sum update; fp64: mean over the batch
ws = ops.Workspace()
lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(steps):
    ops.mbgd_step(p, cubes, oh, (i % (n // B)) * B, B, z, lr, loss_values=lv, workspace=ws)
torch.cuda.synchronize()
assert torch.isfinite(p).all()
print(f"{name} B={B} {np.dtype(dt).name}: {steps} steps done")
