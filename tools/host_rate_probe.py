import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import ops, prep, lib as L
z, B = 103, 128
d = L.dims(z)
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 8, 9 * z)).astype(np.float32)).cuda()
oh = np.zeros((B * 8, 13), dtype=np.int32); oh[np.arange(B * 8), rng.integers(0, 13, B * 8)] = 1
oh = torch.from_numpy(oh).cuda()
p = torch.from_numpy(prep.init_params(z, np.float32)).cuda()
ws = ops.Workspace(); lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(200): ops.mbgd_step(p, cubes, oh, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize()
for n in (50, 2000):
    t0 = time.perf_counter()
    for i in range(n): ops.mbgd_step(p, cubes, oh, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"n={n}: enqueue {(t1 - t0) / n * 1e6:.1f} us/step, total {(t2 - t0) / n * 1e6:.1f} us/step")
st = ops.MbgdStepper(p, cubes, oh, B, z, loss_values=lv, workspace=ws)
for i in range(200): st.step((i % 8) * B, 0.0)
torch.cuda.synchronize()
for n in (50, 2000):
    t0 = time.perf_counter()
    for i in range(n): st.step((i % 8) * B, 0.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"MbgdStepper n={n}: enqueue {(t1 - t0) / n * 1e6:.1f} us/step, total {(t2 - t0) / n * 1e6:.1f} us/step")
