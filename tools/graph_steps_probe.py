"""Step time of the fused minibatch step when a run of consecutive steps is captured into ONE HIP
graph and replayed, against the same steps issued call by call (diagnostic).
    graph_steps_probe.py [z] [B] [steps per graph]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import lib as L, ops, prep  # noqa: E402

z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
G = int(sys.argv[3]) if len(sys.argv) > 3 else 100
rng = np.random.default_rng(0)
nb = 64
cubes = torch.from_numpy(rng.random((B * nb, 9 * z)).astype(np.float32)).cuda()
oh = np.zeros((B * nb, 13), dtype=np.int32)
oh[np.arange(B * nb), rng.integers(0, 13, B * nb)] = 1
oh = torch.from_numpy(oh).cuda()
p0 = torch.from_numpy(prep.init_params(z, np.float32)).cuda()
lv = torch.zeros((nb, B), dtype=torch.float64, device="cuda")


def run(params, n, stepper):
    for i in range(n):
        stepper.step((i % nb) * B, 0.01, i % nb)


p_plain, p_graph = p0.clone(), p0.clone()
st_plain = ops.MbgdStepper(p_plain, cubes, oh, B, z, loss_values=lv, workspace=ops.Workspace())
ws_g = ops.Workspace()
st_graph = ops.MbgdStepper(p_graph, cubes, oh, B, z, loss_values=lv.clone(), workspace=ws_g)
run(p_plain, G, st_plain)
run(p_graph, G, st_graph)          # warm-up: sizes the workspace, sets the kernels' attributes
torch.cuda.synchronize()
assert torch.equal(p_plain, p_graph)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    st_cap = ops.MbgdStepper(p_graph, cubes, oh, B, z, loss_values=st_graph.loss_values if hasattr(st_graph, "loss_values") else lv, workspace=ws_g)
    run(p_graph, G, st_cap)
torch.cuda.synchronize()
reps = max(1, 2000 // G)
g.replay()
run(p_plain, G, st_plain)
torch.cuda.synchronize()
assert torch.equal(p_plain, p_graph), "replayed steps differ from issued steps"
t = time.perf_counter()
for _ in range(reps):
    g.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t) / (reps * G)
t = time.perf_counter()
run(p_plain, reps * G, st_plain)
torch.cuda.synchronize()
tp = (time.perf_counter() - t) / (reps * G)
print(f"z={z} B={B}: {G} steps per graph: {tg * 1e6:.2f} us/step replayed, {tp * 1e6:.2f} us/step call by call")
