"""Gather of every pixel of a scene (the list a full-scene job would use): kernel time from the
library's dispatch events.  usage: bench_gather_full.py [x] [y] [z] [f32|f64]"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import ops, prep, lib as L
x = int(sys.argv[1]) if len(sys.argv) > 1 else 610
y = int(sys.argv[2]) if len(sys.argv) > 2 else 340
z = int(sys.argv[3]) if len(sys.argv) > 3 else 103
dt = torch.float64 if (len(sys.argv) > 4 and sys.argv[4] == "f64") else torch.float32
scene = torch.rand((z * y * x,), dtype=dt, device="cuda")
labels = np.ones(x * y, dtype=np.float64)
di = prep.scan_labels(labels, x, y)
tr9, te9, oh, _ = prep.build_index(labels, di, x, y)
prep.shuffle(tr9, oh, 1)
es = scene.element_size()
for tag, idx in (("4/5 of the pixels, shuffled", tr9), ("1/5 of the pixels, ordered", te9)):
    idx_d = torch.from_numpy(idx).cuda()
    n = idx_d.shape[0]
    out = torch.empty((n, 9 * z), dtype=dt, device="cuda")
    ws = ops.Workspace()
    for _ in range(3): ops.gather(scene, idx_d, x, y, z, out=out, workspace=ws)
    torch.cuda.synchronize()
    L.profile_enable(True)
    for _ in range(10): ops.gather(scene, idx_d, x, y, z, out=out, workspace=ws)
    torch.cuda.synchronize()
    L.profile_enable(False)
    rep = L.profile_report()
    ms, cnt = rep["gather_tile"]
    alg = n * (2 * 9 * z * es + 36)
    t = ms / cnt * 1e-3
    print(f"{x}x{y}x{z} {tag}: n={n}  gather_tile {t*1e6:.1f} us  {alg/t/1e9:.0f} GB/s algorithmic = {alg/t/8e12*100:.0f} % of 8 TB/s; "
          + ", ".join(f"{k} {v[0]/v[1]*1e3:.1f} us" for k, v in rep.items() if k != "gather_tile"))
