"""Whole-call time of ccnn_gather_planned_f32 on the shuffled training list of a scene shape.
usage: bench_gather_planned.py [workload] [reps] [full]   (CCNN_LIB_PATH selects a build variant;
"full" labels every pixel and gathers all of them; CCNN_GATHER_PLAN=dense|sorted forces a kernel)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import synth
from gcn_amd import ops, prep
name = sys.argv[1] if len(sys.argv) > 1 else "pavia_u"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
x, y, z, nl = synth.SCENES[name]
rng = np.random.default_rng(20244)
labels = synth.make_labels(x, y, nl, rng)
full = len(sys.argv) > 3 and sys.argv[3] in ("full", "fullsorted")
if full:
    labels = rng.integers(1, 14, x * y).astype(np.int32)
scene = torch.from_numpy(synth.make_scene(x, y, z, np.float32, rng)).cuda()
di = prep.scan_labels(labels, x, y)
tr9, te9, oh, _ = prep.build_index(labels, di, x, y)
prep.shuffle(tr9, oh, 20240)
if full:
    tr9 = np.concatenate([tr9.reshape(-1, 9), te9.reshape(-1, 9)])
    tr9 = np.ascontiguousarray(tr9[rng.permutation(len(tr9))] if sys.argv[3] == "full" else tr9[np.argsort(tr9[:, 4])])
    tr9 = tr9[: min(len(tr9), (6 << 30) // (9 * z * 4))]     # keep the cubes under 6 GiB
idx_d = torch.from_numpy(tr9).cuda()
out = torch.empty((len(tr9), 9 * z), dtype=torch.float32, device="cuda")
plan = ops.GatherPlan(idx_d, x, y)
best = 1e9
for trial in range(5):
    for _ in range(3): plan.gather(scene, z, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): plan.gather(scene, z, out=out)
    torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / reps)
alg = len(tr9) * (2 * 9 * z * 4 + 36)
print(f"{os.environ.get('CCNN_LIB_PATH', 'default').split('/')[-1]} {name}{' full' if full else ''}: {'tiles' if plan.dense else 'ordered list'} n={len(tr9)} {best*1e6:.1f} us  {alg/best/1e9:.0f} GB/s = {alg/best/8e12:.3f} of 8 TB/s")
