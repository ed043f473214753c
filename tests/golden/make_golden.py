"""Regenerates the golden fixtures in this directory from the CPU oracle.

Run in the authoring container (needs /root/reference/model/KSCmodel.mat for the
checkpoint fixture):  python tests/golden/make_golden.py

The reference ships no golden vectors, so these pin the ORACLE (regression) and
give the GPU tests fixed inputs; the forward fixture is additionally cross-checked
here against the independent numpy restatement in tests/np_ref.py.
"""
import hashlib
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import np_ref  # noqa: E402
import oracle as orc  # noqa: E402
import synth  # noqa: E402

KSC_MODEL = "/root/reference/model/KSCmodel.mat"


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def index_fixture(x, y, z, seed, keep_cubes):
    scene, labels = synth.small_scene(x, y, z, seed)
    di = orc.scan_labels(labels, x, y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, x, y)
    ctr = orc.gather(scene, tr9, x, y, z)
    cte = orc.gather(scene, te9, x, y, z)
    out = dict(x=x, y=y, z=z, seed=seed, labels=labels, data_index=di, train_idx9=tr9, test_idx9=te9,
               onehot=onehot, test_labels=tl, train_sha=sha(ctr), test_sha=sha(cte))
    if keep_cubes:
        out.update(scene=scene, train_cubes=ctr, test_cubes=cte)
    else:
        out.update(train_cubes_head=ctr[:4], test_cubes_head=cte[:4])
    np.savez_compressed(os.path.join(HERE, f"index_{x}x{y}x{z}.npz"), **out)


def ksc_forward_fixture():
    import scipy.io as sio
    m = sio.loadmat(KSC_MODEL)   # MAT v5, plain numeric arrays
    z = 176
    params = np.concatenate([m["kernel"].ravel(order="F"), m["bias0"].ravel(), m["omega1"].ravel(order="F"),
                             m["bias1"].ravel(), m["omega2"].ravel(order="F"), m["bias2"].ravel()])
    params = np.ascontiguousarray(params, dtype=np.float64)
    rng = np.random.default_rng(20241)
    cubes = rng.random((8, 9 * z))
    probs, argmax = orc.forward(params, cubes, z)
    tr = [orc.forward_trace_f64(params, c, z) for c in cubes]
    for i, c in enumerate(cubes):   # independent cross-check
        f = np_ref.forward(params, c, z)
        assert np.abs(f["probs"] - probs[i]).max() < 1e-13 and (f["pidx"] == tr[i][2]).all()
        assert np.abs(f["a1"] - tr[i][3]).max() < 1e-13
    np.savez_compressed(os.path.join(HERE, "ksc_forward.npz"), z=z, params=params, cubes=cubes, probs=probs,
                        argmax=argmax, a0=np.stack([t[0] for t in tr]), pool=np.stack([t[1] for t in tr]),
                        pidx=np.stack([t[2] for t in tr]), a1=np.stack([t[3] for t in tr]),
                        loss_curve=m["loss"].ravel())


def train_fixture():
    z, n, seed = 48, 12, 7
    rng = np.random.default_rng(seed)
    cubes = rng.random((n, 9 * z))
    onehot = np.zeros((n, 13), dtype=np.int32)
    onehot[np.arange(n), rng.integers(0, 13, n)] = 1
    d = orc.dims(z)
    out = dict(z=z, seed=seed, cubes=cubes, onehot=onehot)
    orc.srand(seed)
    p0 = orc.init_params_f64(z, d.pout)
    out["sgd_p0"] = p0.copy()
    p = p0.copy()
    out["sgd_loss"] = orc.sgd_f64(p, cubes, onehot, 0, 3, z, 0.008)
    out["sgd_p3"] = p
    orc.srand(seed)
    p0 = orc.init_params_f64(z, d.pout * 4)
    out["mbgd64_p0"] = p0.copy()
    p = p0.copy()
    lv = []
    for i0 in range(2):
        l, _, g = orc.mbgd_step(p, cubes, onehot, i0 * 4, 4, z, 0.5)
        lv.append(l)
        out[f"mbgd64_grad{i0}"] = g
    out["mbgd64_p2"] = p
    out["mbgd64_loss"] = np.stack(lv)
    orc.srand(seed)
    p0 = orc.init_params_f32(z)
    c32 = cubes.astype(np.float32)
    out["mbgd32_p0"] = p0.copy()
    p = p0.copy()
    lv = []
    for i0 in range(2):
        l, _, g = orc.mbgd_step(p, c32, onehot, i0 * 4, 4, z, np.float32(0.04))
        lv.append(l)
        out[f"mbgd32_grad{i0}"] = g
    out["mbgd32_p2"] = p
    out["mbgd32_loss"] = np.stack(lv)
    np.savez_compressed(os.path.join(HERE, "train_small.npz"), **out)


def partition_fixture():
    T, B = 37, 8
    out = {}
    for n in (2, 4):
        tab = np.array([[orc.mgpu_first_sample(T, n, B, slot, i0) for slot in range(n)] for i0 in range(T // B)])
        out[f"first_n{n}"] = tab
    np.savez_compressed(os.path.join(HERE, "partition.npz"), T=T, B=B, **out)


if __name__ == "__main__":
    index_fixture(7, 6, 24, 11, keep_cubes=True)
    index_fixture(16, 12, 176, 12, keep_cubes=False)
    ksc_forward_fixture()
    train_fixture()
    partition_fixture()
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
