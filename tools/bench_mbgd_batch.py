"""Step time of ccnn_mbgd_step_f32 against the batch size, for both implementations of the
step (CCNN_MBGD_PATH=fused|tile|layers), with the per-kernel times from the library's hooks.
    python tools/bench_mbgd_batch.py [workload] [dtype]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth  # noqa: E402
from gcn_amd import lib as L, ops, prep  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "pavia_u"
dtype = np.float64 if len(sys.argv) > 2 and sys.argv[2] == "f64" else np.float32
x, y, z, nl = synth.SCENES[name]
d = L.dims(z)
rng = np.random.default_rng(1)
n = 8192
cubes = torch.from_numpy(rng.random((n, 9 * z)).astype(dtype)).cuda()
if os.environ.get("BENCH_ZERO_CUBES"):   # diagnostic: how much of the step time depends on the operand values (power, not work)
    cubes.zero_()
oh = np.zeros((n, 13), dtype=np.int32)
oh[np.arange(n), rng.integers(0, 13, n)] = 1
oh = torch.from_numpy(oh).cuda()
p0 = torch.from_numpy(prep.init_params(z, dtype, d.pout)).cuda()
for path in (os.environ.get("BENCH_PATHS", "fused,tile,layers").split(",")):
    os.environ["CCNN_MBGD_PATH"] = path
    for B in [int(b) for b in os.environ.get("BENCH_BATCHES", "64,128,256,512,1024,2048,4096").split(",")]:
        if path == "fused" and B > 1024:
            continue
        p = p0.clone()
        ws = ops.Workspace()
        lv = torch.zeros(B, dtype=torch.float64, device="cuda")
        nb = n // B
        lr = 0.04 * 128 / B if dtype == np.float32 else 0.5
        for i in range(20):
            ops.mbgd_step(p, cubes, oh, (i % nb) * B, B, z, lr, loss_values=lv, workspace=ws)
        torch.cuda.synchronize()
        steps = 300
        t = time.perf_counter()
        for i in range(steps):
            ops.mbgd_step(p, cubes, oh, (i % nb) * B, B, z, lr, loss_values=lv, workspace=ws)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / steps
        L.profile_enable(True)
        for i in range(50):
            ops.mbgd_step(p, cubes, oh, (i % nb) * B, B, z, lr, loss_values=lv, workspace=ws)
        torch.cuda.synchronize()
        L.profile_enable(False)
        rep = L.profile_report()
        ks = "  ".join(f"{k} {ms / c * 1e3:.1f}" for k, (ms, c) in rep.items())
        assert torch.isfinite(p).all()
        print(f"{name} {path:7s} B={B:5d}: {dt * 1e6:7.1f} us/step = {B / dt / 1e6:6.2f} M px/s   [{ks}]", flush=True)
