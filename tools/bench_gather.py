"""Gather throughput on a synthetic scene of a real scene's shape (diagnostic / profiling driver).
usage: bench_gather.py [pavia_u|indian_pines|ksc] [f32|f64] [reps]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from gcn_amd import ops, prep
name = sys.argv[1] if len(sys.argv) > 1 else "pavia_u"
dt = np.float64 if (len(sys.argv) > 2 and sys.argv[2] == "f64") else np.float32
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
x, y, z, nl = synth.SCENES[name]
rng = np.random.default_rng(1)
labels = synth.make_labels(x, y, nl, rng)
scene = torch.from_numpy(synth.make_scene(x, y, z, dt, rng)).cuda()
di = prep.scan_labels(labels, x, y)
tr9, te9, oh, _ = prep.build_index(labels, di, x, y)
prep.shuffle(tr9, oh, 1)
es = np.dtype(dt).itemsize
for tag, idx in (("train(shuffled)", tr9), ("test(ordered)", te9)):
    idx_d = torch.from_numpy(idx).cuda()
    out = torch.empty((len(idx) + 64, 9 * z), dtype=scene.dtype, device="cuda")[:len(idx)]   # slack behind it for padded-row experiments
    ws = ops.Workspace()
    for srt in (False, True):
        for _ in range(3): ops.gather(scene, idx_d, x, y, z, out=out, workspace=ws, sort=srt)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): ops.gather(scene, idx_d, x, y, z, out=out, workspace=ws, sort=srt)
        torch.cuda.synchronize(); dt_s = (time.perf_counter() - t0) / reps
        alg = len(idx) * (2 * 9 * z * es + 36)
        print(f"{name} {np.dtype(dt).name} {tag} sort={srt}: n={len(idx)} {dt_s*1e6:.1f} us  {alg/dt_s/1e9:.0f} GB/s algorithmic ({alg/1e6:.1f} MB)")
