"""CPU: the oracle against its committed golden fixtures and against the
independent numpy restatement (tests/np_ref.py)."""
import hashlib

import numpy as np
import pytest

import np_ref
import oracle as orc
import synth


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def test_dims_match_known_scenes():
    # z -> (re, mre, pout): KSC, Indian Pines, Pavia U, AVIRIS
    for z, (re, mre, pout, npar) in {176: (79, 40, 800, 84853), 200: (91, 46, 920, 96853),
                                      103: (42, 21, 420, 46853), 224: (103, 52, 1040, 108853)}.items():
        d = orc.dims(z)
        assert (d.re, d.mre, d.pout, d.nparams) == (re, mre, pout, npar)
        assert np_ref.dims(z) == (re, mre)


@pytest.mark.parametrize("name,keep", [("index_7x6x24.npz", True), ("index_16x12x176.npz", False)])
def test_index_and_gather_golden(golden, name, keep):
    g = golden(name)
    x, y, z, seed = int(g["x"]), int(g["y"]), int(g["z"]), int(g["seed"])
    scene, labels = synth.small_scene(x, y, z, seed)
    assert np.array_equal(labels, g["labels"])
    di = orc.scan_labels(labels, x, y)
    assert np.array_equal(di, g["data_index"])
    tr9, te9, onehot, tl = orc.build_index(labels, di, x, y)
    for a, k in ((tr9, "train_idx9"), (te9, "test_idx9"), (onehot, "onehot"), (tl, "test_labels")):
        assert np.array_equal(a, g[k]), k
    # every neighbour index lies inside the scene and mirrors about the centre
    for idx in (tr9, te9):
        assert idx.min() >= 0 and idx.max() < x * y
    ctr = orc.gather(scene, tr9, x, y, z)
    cte = orc.gather(scene, te9, x, y, z)
    assert sha(ctr) == str(g["train_sha"]) and sha(cte) == str(g["test_sha"])
    if keep:
        assert np.array_equal(ctr, g["train_cubes"]) and np.array_equal(cte, g["test_cubes"])
    # gather is a pure copy: numpy fancy indexing gives the same bytes
    sc = scene.reshape(z, x * y)
    ref = sc[:, tr9].transpose(1, 0, 2).reshape(len(tr9), 9 * z)
    assert np.array_equal(ref, ctr)


def test_split_rule():
    for n in (1, 2, 5, 6, 37, 5211, 42776):
        tr, te = orc.split_sizes(n)
        assert te == (n - 1) // 5 + 1 and tr + te == n
        assert te == len(range(0, n, 5))


def test_forward_golden_and_numpy(golden):
    g = golden("ksc_forward.npz")
    z = int(g["z"])
    params, cubes = g["params"], g["cubes"]
    probs, argmax = orc.forward(params, cubes, z)
    assert np.array_equal(probs, g["probs"]) and np.array_equal(argmax, g["argmax"])
    for i, c in enumerate(cubes):
        f = np_ref.forward(params, c, z)
        a0, pool, pidx, a1, p = orc.forward_trace_f64(params, c, z)
        assert np.array_equal(pidx, f["pidx"])
        for a, b in ((a0, f["a0"]), (pool, f["pool"]), (a1, f["a1"]), (p, f["probs"])):
            assert np.abs(a - b).max() < 1e-13
        assert abs(p.sum() - 1) < 1e-14
        assert argmax[i] == int(np.argmax(p))
    # the checkpoint's own loss curve is only a sanity envelope
    lc = g["loss_curve"]
    assert lc[0] > 0.2 and lc[-1] < 0.01


def test_argmax_follows_count_err():
    p = np.zeros((3, 13))
    p[1, 4] = 0.5
    p[1, 7] = 0.5          # tie: first maximum wins (strict '>')
    p[2, :] = -1.0         # nothing above the initial max of 0 -> index 0
    assert list(orc.argmax(p)) == [0, 4, 0]


def test_training_golden(golden):
    g = golden("train_small.npz")
    z = int(g["z"])
    cubes, onehot = g["cubes"], g["onehot"]
    d = orc.dims(z)
    orc.srand(int(g["seed"]))
    p = orc.init_params_f64(z, d.pout)
    assert np.array_equal(p, g["sgd_p0"])
    loss = orc.sgd_f64(p, cubes, onehot, 0, 3, z, 0.008)
    assert np.array_equal(p, g["sgd_p3"]) and loss == float(g["sgd_loss"])

    p = g["mbgd64_p0"].copy()
    for i0 in range(2):
        lv, _, gr = orc.mbgd_step(p, cubes, onehot, i0 * 4, 4, z, 0.5)
        assert np.array_equal(lv, g["mbgd64_loss"][i0]) and np.array_equal(gr, g[f"mbgd64_grad{i0}"])
    assert np.array_equal(p, g["mbgd64_p2"])

    p = g["mbgd32_p0"].copy()
    c32 = cubes.astype(np.float32)
    for i0 in range(2):
        lv, _, gr = orc.mbgd_step(p, c32, onehot, i0 * 4, 4, z, np.float32(0.04))
        assert np.array_equal(lv, g["mbgd32_loss"][i0]) and np.array_equal(gr, g[f"mbgd32_grad{i0}"])
    assert np.array_equal(p, g["mbgd32_p2"])
    assert np.all(p[d.off_b0:d.off_w1] == 0)      # fp32 conv bias never moves


def test_gradients_against_numpy(golden):
    g = golden("train_small.npz")
    z = int(g["z"])
    cubes, onehot = g["cubes"], g["onehot"]
    d = orc.dims(z)
    p0 = g["mbgd64_p0"]
    gs = sum(np_ref.grad(p0, cubes[i], onehot[i], z)[0] for i in range(4))
    assert np.abs(gs - g["mbgd64_grad0"]).max() <= 1e-13 * np.abs(gs).max()
    p1 = p0 - 0.5 * gs / 4
    gs1 = sum(np_ref.grad(p1, cubes[4 + i], onehot[4 + i], z)[0] for i in range(4))
    assert np.abs((p1 - 0.5 * gs1 / 4) - g["mbgd64_p2"]).max() <= 1e-13
    # fp32 variant: reversed filter forward, plain-sum update, frozen conv bias
    p0 = g["mbgd32_p0"].astype(np.float64)
    c = cubes.astype(np.float32).astype(np.float64)
    gs = sum(np_ref.grad(p0, c[i], onehot[i], z, flip=True)[0] for i in range(4))
    rel = np.abs(gs - g["mbgd32_grad0"]).max() / np.abs(gs).max()
    assert rel < 5e-6
    # loss of the fp64 variant is the BCE-over-softmax expression
    f = np_ref.forward(g["mbgd64_p0"], cubes[0], z)
    assert abs(np_ref.bce_loss(f["probs"], onehot[0]) - g["mbgd64_loss"][0][0]) < 1e-14


def test_sgd_against_numpy(golden):
    g = golden("train_small.npz")
    z = int(g["z"])
    q = g["sgd_p0"].copy()
    loss = 0.0
    for i in range(3):
        gr, f = np_ref.grad(q, g["cubes"][i], g["onehot"][i], z)
        loss += np_ref.bce_loss(f["probs"], g["onehot"][i])
        q = q - 0.008 * gr
    assert np.abs(q - g["sgd_p3"]).max() < 1e-14
    assert abs(loss - float(g["sgd_loss"])) < 1e-13


def test_mgpu_step_equals_single_device_when_partition_is_contiguous():
    # with n_dev = 1 the master/worker step is the plain fp64 MBGD step
    z, T, B = 40, 16, 8
    rng = np.random.default_rng(3)
    cubes = rng.random((T, 9 * z))
    onehot = np.zeros((T, 13), dtype=np.int32)
    onehot[np.arange(T), rng.integers(0, 13, T)] = 1
    orc.srand(1)
    p0 = orc.init_params_f64(z, 0)
    a, b = p0.copy(), p0.copy()
    l1 = orc.mgpu_step_f64(a, cubes, onehot, T, 1, B, 1, z, 0.5)
    lv, _, _ = orc.mbgd_step(b, cubes, onehot, B, B, z, 0.5)
    assert np.array_equal(a, b)
    assert abs(l1 - lv.sum()) < 1e-12
    # two devices: each takes B/2 samples from its own half of the data
    a = p0.copy()
    orc.mgpu_step_f64(a, cubes, onehot, T, 0, B, 2, z, 0.5)
    sel = np.r_[0:4, 8:12]
    gs = sum(np_ref.grad(p0, cubes[i], onehot[i], z)[0] for i in sel)
    assert np.abs(a - (p0 - 0.5 * gs / B)).max() < 1e-14


def test_partition_golden(golden):
    g = golden("partition.npz")
    T, B = int(g["T"]), int(g["B"])
    for n in (2, 4):
        tab = np.array([[orc.mgpu_first_sample(T, n, B, s, i0) for s in range(n)] for i0 in range(T // B)])
        assert np.array_equal(tab, g[f"first_n{n}"])


def test_lr_schedule_plateau():
    s = orc.Sched(np.float64, 1.0)
    lr = 0.5
    hist = []
    for loss in [0.9, 0.8, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 0.7, 0.6]:
        dec, lr = s.step(loss, lr)
        hist.append(dec)
    assert sum(hist) >= 1 and lr < 0.5
