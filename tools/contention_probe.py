"""Two (or more) processes on one GPU repeat the same fp64/fp32 minibatch gradient and compare it
with the CPU oracle's: any mismatch is a kernel whose result depends on what else runs on the CU.
torchrun --nproc-per-node 2 tools/contention_probe.py [f64|f32] [B] [reps]"""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from gcn_amd import lib as L, ops
dist.init_process_group("gloo")
rank = dist.get_rank()
torch.cuda.set_device(0)
f64 = len(sys.argv) < 2 or sys.argv[1] == "f64"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
z = 103
d = L.dims(z)
npdt = np.float64 if f64 else np.float32
rng = np.random.default_rng(77)
p0 = ((rng.random(d.nparams) - 0.5) / 10).astype(npdt)
rng = np.random.default_rng(78 + rank)
cubes = rng.random((B, d.cube)).astype(npdt)
oh = np.zeros((B, 13), dtype=np.int32); oh[np.arange(B), rng.integers(0, 13, B)] = 1
pr = p0.copy()
lv_ref, probs_ref, g_ref = orc.mbgd_step(pr, cubes, oh, 0, B, z, npdt(0.0))
p_d, c_d, o_d = torch.from_numpy(p0).cuda(), torch.from_numpy(cubes).cuda(), torch.from_numpy(oh).cuda()
tol = 1e-11 if f64 else 2e-5
bad = 0
dist.barrier()
for it in range(reps):
    probs = torch.empty((B, 13), dtype=p_d.dtype, device="cuda")
    g, lv = ops.mbgd_grad(p_d, c_d, o_d, 0, B, z, probs=probs)
    torch.cuda.synchronize()
    eg = float(np.abs(g.cpu().numpy() - g_ref).max() / np.abs(g_ref).max())
    ep = np.abs(probs.cpu().numpy() - probs_ref).max(axis=1) / np.abs(probs_ref).max()
    if eg > tol or ep.max() > tol:
        bad += 1
        if bad <= 5:
            offs = [0, d.off_b0, d.off_w1, d.off_b1, d.off_w2, d.off_b2, d.nparams]
            gg = np.abs(g.cpu().numpy() - g_ref)
            ge = [float(gg[offs[i]:offs[i + 1]].max() / np.abs(g_ref).max()) for i in range(6)]
            print(f"[{rank}] it {it}: grad err {eg:.1e} groups {['%.0e' % e for e in ge]}; samples with wrong probs: {np.nonzero(ep > tol)[0].tolist()} ({ep.max():.1e})", flush=True)
print(f"[{rank}] {'f64' if f64 else 'f32'} B={B} split={os.environ.get('CCNN_FB_SPLIT')}: {bad} of {reps} calls differ from the oracle", flush=True)
dist.barrier()
dist.destroy_process_group()
