"""Full-scene inference: every pixel of a synthetic scene classified (a) from pre-gathered cubes
(gather + forward, chunked so the cubes of a chunk fit a fixed buffer) and (b) straight from the
scene (forward_scene).  usage: bench_scene_infer.py [x] [y] [z]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gcn_amd import ops, prep
import ctypes
x = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
y = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
z = int(sys.argv[3]) if len(sys.argv) > 3 else 224
rng = np.random.default_rng(0)
scene = torch.rand((z * y * x,), dtype=torch.float32, device="cuda")
labels = np.ones(x * y, dtype=np.float64)                    # every pixel is a sample
di = prep.scan_labels(labels, x, y)
tr9, te9, _, _ = prep.build_index(labels, di, x, y)
idx = torch.from_numpy(np.concatenate([tr9, te9])).cuda()     # all x*y neighbourhoods
n = idx.shape[0]
ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, np.float32)).cuda()
ws, gws = ops.Workspace(), ops.Workspace()
CH = 262144
def cubes_path():
    out = []
    buf = torch.empty((CH, 9 * z), dtype=torch.float32, device="cuda")
    for s in range(0, n, CH):
        m = min(CH, n - s)
        ops.gather(scene, idx[s:s + m], x, y, z, out=buf[:m], workspace=gws, sort=False)
        out.append(ops.forward(params, buf[:m], z, want_probs=False, workspace=ws)[1])
    return torch.cat(out)
def scene_path():
    return ops.forward_scene(params, scene, idx, x, y, z, want_probs=False, workspace=ws)[1]
for name, fn in (("gather+forward", cubes_path), ("forward_scene", scene_path)):
    a = fn(); torch.cuda.synchronize(); t0 = time.perf_counter(); a = fn(); torch.cuda.synchronize()
    t = time.perf_counter() - t0
    print(f"{x}x{y}x{z} {name}: {t*1e3:.1f} ms  {n/t/1e6:.1f} M pixels/s")
    if name == "gather+forward": ref = a
assert torch.equal(ref, a)
