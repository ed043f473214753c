"""Times the data-parallel minibatch step (exchange inside the reduction kernel) with one rank
(diagnostic; pick the build with CCNN_LIB_PATH).  usage: bench_dp_step.py [z] [B]"""
import os, sys, time
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import ops, prep, lib as L
from gcn_amd.comm import NodeComm
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29577")
dist.init_process_group("gloo", rank=0, world_size=1)
z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = L.dims(z)
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 8, 9 * z)).astype(np.float32)).cuda()
onehot = np.zeros((B * 8, 13), dtype=np.int32); onehot[np.arange(B * 8), rng.integers(0, 13, B * 8)] = 1
onehot = torch.from_numpy(onehot).cuda()
import ctypes; ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, np.float32)).cuda()
grad = torch.empty_like(params)
ws = ops.Workspace(); lv = torch.zeros(B, dtype=torch.float64, device="cuda")
comm = NodeComm(d.nparams, torch.float32)
def run(mode, n):
    for i in range(n):
        if mode == "fused1": ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
        elif mode == "dp": ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws, comm=comm)
        else:
            ops.mbgd_grad(params, cubes, onehot, (i % 8) * B, B, z, grad=grad, loss_values=lv, workspace=ws)
            comm.allreduce_update(params, grad, 0.0, z)
for mode in ("fused1", "dp", "grad+allreduce_update"):
    run(mode, 50); torch.cuda.synchronize(); t0 = time.perf_counter(); run(mode, 500); torch.cuda.synchronize()
    print(os.environ.get("CCNN_LIB_PATH", "default"), mode, f"{(time.perf_counter() - t0) / 500 * 1e6:.1f} us/step")
assert comm.status() == 0
comm.close(); dist.destroy_process_group()
