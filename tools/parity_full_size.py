"""Full-size parity run on the GPU box: a KSC-shaped synthetic scene (512x614x176, 5211 labelled
pixels), the whole prologue + one fp64 SGD epoch + two fp64 MBGD epochs (B = 100) + inference,
HIP library against the CPU oracle on identical inputs.  Prints the figures the scope contract
sets bars for (cubes / labels / argmax bit-exact; fp64 weights and losses 1e-12 relative)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc, synth
from gcn_amd import ops, prep

x, y, z, nl = synth.SCENES["ksc"]
rng = np.random.default_rng(20242)
labels = synth.make_labels(x, y, nl, rng)
scene = synth.make_scene(x, y, z, np.float64, rng)
di = orc.scan_labels(labels, x, y)
tr9, te9, onehot, tl = orc.build_index(labels, di, x, y)
orc.shuffle(tr9, onehot, 7)
d = orc.dims(z)
T = len(onehot)
dev = torch.device("cuda")
t0 = time.perf_counter()
ref_train, ref_test = orc.gather(scene, tr9, x, y, z), orc.gather(scene, te9, x, y, z)
t_cpu_gather = time.perf_counter() - t0
sc = torch.from_numpy(scene).to(dev)
g_train = ops.gather(sc, torch.from_numpy(tr9).to(dev), x, y, z)
g_test = ops.gather(sc, torch.from_numpy(te9).to(dev), x, y, z, sort=False)
print(f"train {T} / test {len(tl)} pixels; cubes bit-exact: train {np.array_equal(g_train.cpu().numpy(), ref_train)}, "
      f"test {np.array_equal(g_test.cpu().numpy(), ref_test)}  (oracle gather {t_cpu_gather:.2f} s)")
oh_d = torch.from_numpy(onehot).to(dev)

def rel(a, b):
    return float(np.abs(a - b).max() / np.abs(b).max())

# ---- one SGD epoch (gcn_sgd_fp64.cu)
orc.srand(7)
p_ref = orc.init_params_f64(z, d.pout)
p_gpu = torch.from_numpy(p_ref.copy()).to(dev)
t0 = time.perf_counter()
loss_ref = orc.sgd_f64(p_ref, ref_train, onehot, 0, T, z, 0.008) / T
t_cpu = time.perf_counter() - t0
sloss = torch.zeros(1, dtype=torch.float64, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
ops.sgd(p_gpu, g_train, oh_d, 0, T, z, 0.008, sloss)
torch.cuda.synchronize(); t_gpu = time.perf_counter() - t0
print(f"SGD epoch: weights rel err {rel(p_gpu.cpu().numpy(), p_ref):.2e}, loss {float(sloss) / T:.12f} vs {loss_ref:.12f} "
      f"(rel {abs(float(sloss) / T - loss_ref) / abs(loss_ref):.1e}); GPU {t_gpu * 1e3:.1f} ms, oracle {t_cpu:.2f} s")
pr_ref, am_ref = orc.forward(p_ref, ref_test, z)
pr, am = ops.forward(p_gpu, g_test, z)
margin = np.sort(pr_ref, axis=1)
safe = (margin[:, -1] - margin[:, -2]) > 1e-9
print(f"  inference after it: probs rel err {rel(pr.cpu().numpy(), pr_ref):.2e}; argmax equal on {int((am.cpu().numpy() == am_ref)[safe].sum())}"
      f" of {int(safe.sum())} pixels with a top-2 margin > 1e-9 ({len(tl)} in all); accuracy {np.mean(am_ref + 1 == tl) * 100:.3f} % both"
      if np.array_equal(am.cpu().numpy()[safe], am_ref[safe]) else "  ARGMAX MISMATCH")

# ---- two MBGD epochs, B = 100 (gcn_mbgd_fp64.cu)
B = 100
orc.srand(7)
p_ref = orc.init_params_f64(z, d.pout * B)
p_gpu = torch.from_numpy(p_ref.copy()).to(dev)
ws = ops.Workspace(dev)
lv = torch.zeros(B, dtype=torch.float64, device=dev)
worst_loss = 0.0
t_cpu = t_gpu = 0.0
for ep in range(2):
    for first in range(0, T - T % B, B):
        t0 = time.perf_counter()
        lref, _, _ = orc.mbgd_step(p_ref, ref_train, onehot, first, B, z, 0.5, want_grad=False)
        t_cpu += time.perf_counter() - t0
        t0 = time.perf_counter()
        ops.mbgd_step(p_gpu, g_train, oh_d, first, B, z, 0.5, loss_values=lv, workspace=ws)
        torch.cuda.synchronize(); t_gpu += time.perf_counter() - t0
        worst_loss = max(worst_loss, rel(lv.cpu().numpy(), np.asarray(lref)))
print(f"MBGD 2 epochs x {T // B} steps: weights rel err {rel(p_gpu.cpu().numpy(), p_ref):.2e}, worst per-step loss rel err {worst_loss:.2e}; "
      f"GPU {t_gpu * 1e3:.0f} ms (with a sync per step), oracle {t_cpu:.1f} s")

# ---- fp32 MBGD on Pavia-U- and Indian-Pines-shaped scenes (gcn_mbgd_fp32.cu), B = 128: per-step error
# with the GPU restarted from the oracle's weights every step, then one free-running epoch and the accuracy
del sc, g_train, g_test


def fp32_epoch(name, seed):
    x, y, z, nl = synth.SCENES[name]
    rng = np.random.default_rng(seed)
    labels = synth.make_labels(x, y, nl, rng)
    scene = synth.make_scene(x, y, z, np.float32, rng)
    di = orc.scan_labels(labels, x, y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, x, y)
    orc.shuffle(tr9, onehot, 9)
    T, B = len(onehot), 128
    ref_train, ref_test = orc.gather(scene, tr9, x, y, z), orc.gather(scene, te9, x, y, z)
    sc = torch.from_numpy(scene).to(dev)
    g_train = ops.gather(sc, torch.from_numpy(tr9).to(dev), x, y, z)
    g_test = ops.gather(sc, torch.from_numpy(te9).to(dev), x, y, z, sort=False)
    print(f"{name}: train {T} / test {len(tl)}; cubes bit-exact: {np.array_equal(g_train.cpu().numpy(), ref_train)} / "
          f"{np.array_equal(g_test.cpu().numpy(), ref_test)}")
    oh_d = torch.from_numpy(onehot).to(dev)
    orc.srand(9)
    p_ref = orc.init_params_f32(z)
    p_free = torch.from_numpy(p_ref.copy()).to(dev)
    lv = torch.zeros(B, dtype=torch.float64, device=dev)
    worst_w = worst_l = 0.0
    nsteps = T // B
    for i in range(nsteps):
        p_step = torch.from_numpy(p_ref.copy()).to(dev)          # this step from the oracle's weights
        ops.mbgd_step(p_step, g_train, oh_d, i * B, B, z, 0.04, loss_values=lv, workspace=ws)
        ops.mbgd_step(p_free, g_train, oh_d, i * B, B, z, 0.04, workspace=ws)
        lref, _, _ = orc.mbgd_step(p_ref, ref_train, onehot, i * B, B, z, np.float32(0.04), want_grad=False)
        e_step = rel(p_step.cpu().numpy(), p_ref)
        if e_step > worst_w and os.environ.get("PARITY_DETAIL"):
            dd = orc.dims(z)
            ps_h = p_step.cpu().numpy()
            grp = {"W0": (0, dd.off_b0), "b0": (dd.off_b0, dd.off_w1), "W1": (dd.off_w1, dd.off_b1), "b1": (dd.off_b1, dd.off_w2),
                   "W2": (dd.off_w2, dd.off_b2), "b2": (dd.off_b2, dd.nparams)}
            print(f"  step {i}: " + ", ".join(f"{k} err {np.abs(ps_h[a:b] - p_ref[a:b]).max():.2e} (max |w| {np.abs(p_ref[a:b]).max():.2e})"
                                             for k, (a, b) in grp.items()))
            e0 = np.abs(ps_h[:dd.off_b0] - p_ref[:dd.off_b0]).reshape(20, 171).max(axis=1)
            print("    filters whose error exceeds 1e-7:", [(int(f), float(f"{e0[f]:.2e}")) for f in np.nonzero(e0 > 1e-7)[0]])
        worst_w = max(worst_w, e_step)
        worst_l = max(worst_l, rel(lv.cpu().numpy(), np.asarray(lref, dtype=np.float64)))
    pr_ref, am_ref = orc.forward(p_ref, ref_test, z)
    _, am = ops.forward(p_free, g_test, z)
    acc_ref, acc = np.mean(am_ref + 1 == tl) * 100, np.mean(am.cpu().numpy() + 1 == tl) * 100
    print(f"fp32 MBGD, {nsteps} steps: worst per-step weights rel err {worst_w:.2e}, loss {worst_l:.2e}; after the free-running epoch "
          f"weights differ by {rel(p_free.cpu().numpy(), p_ref):.2e}, accuracy {acc:.3f} % vs oracle {acc_ref:.3f} %, "
          f"argmax differs on {int((am.cpu().numpy() != am_ref).sum())} of {len(tl)} pixels")


fp32_epoch("pavia_u", 20244)
fp32_epoch("indian_pines", 20243)
