"""GPU: the `gcn` command line end to end on a small synthetic .mat dataset --
stdout lines in the reference's format (tools/run.sh field positions), model.mat
contents and accuracy against the oracle's replay of the same run."""
import os
import re
import subprocess

import numpy as np
import pytest
import scipy.io as sio

import oracle as orc
import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GCN = os.path.join(ROOT, "gcn_amd", "bin", "gcn")
X, Y, Z, SEED = 24, 20, 60, 77


def make_dataset(tmp_path, dtype):
    scene, labels = synth.small_scene(X, Y, Z, 3)
    ds = scene.reshape(Z, Y, X).transpose(2, 1, 0).astype(dtype)      # MATLAB x-by-y-by-z, column-major on disk
    lab = labels.reshape(Y, X).T
    path = str(tmp_path / "DATASET.mat")
    sio.savemat(path, {"DataSet": ds, "labels": lab}, do_compression=True)
    return path, scene, labels


def run_gcn(tmp_path, args, env=None):
    assert os.path.exists(GCN), "build the CLI with __graft_entry__.build()"
    r = subprocess.run([GCN] + args, cwd=tmp_path, capture_output=True, text=True, timeout=120,
                       env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr
    return r.stdout


def oracle_run(scene, labels, dtype, B, lr, epochs, sgd=False, fifo=1.0, keep_tail=False):
    d = orc.dims(Z)
    di = orc.scan_labels(labels, X, Y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, X, Y)
    orc.shuffle(tr9, onehot, SEED)
    p = orc.init_params_f32(Z) if dtype == np.float32 else orc.init_params_f64(Z, d.pout if sgd else d.pout * B)
    sc = scene.astype(dtype)
    train, test = orc.gather(sc, tr9, X, Y, Z), orc.gather(sc, te9, X, Y, Z)
    T = len(onehot)
    losses = []
    cur = dtype(lr)
    sched = orc.Sched(dtype, fifo)
    for _ in range(epochs):
        if sgd:
            rate = orc.sgd_f64(p, train, onehot, 0, T, Z, float(cur)) / T
        else:
            tot = dtype(0) if dtype == np.float32 else 0.0
            for first in range(0, T if keep_tail else T - T % B, B):
                lv, _, _ = orc.mbgd_step(p, train, onehot, first, min(B, T - first), Z, cur, want_grad=False)
                for v in lv:
                    tot = dtype(np.float64(tot) + v) if dtype == np.float32 else tot + v
            rate = float(tot) / T
        losses.append(rate)
        _, nl = sched.step(rate, float(cur))
        cur = dtype(nl)
    _, am = orc.forward(p, test, Z)
    return p, losses, float(np.mean(am + 1 == tl)), d


def parse(stdout):
    losses = [float(m) for m in re.findall(r"Iteration \d+,\s+loss = ([0-9.eE+-]+);", stdout)]
    acc_line = [l for l in stdout.splitlines() if "Accuracy" in l][0]
    glob_line = [l for l in stdout.splitlines() if "Global" in l][0]
    return losses, acc_line, glob_line


@pytest.mark.parametrize("variant,dtype,lr,fifo", [("mbgd_fp32", np.float32, 0.04, 100.0), ("mbgd_fp64", np.float64, 0.5, 1.0)])
def test_mbgd_cli_matches_oracle(tmp_path, variant, dtype, lr, fifo):
    path, scene, labels = make_dataset(tmp_path, dtype)
    B, epochs = 20, 3
    out = run_gcn(tmp_path, [f"--variant={variant}", f"--seed={SEED}", f"--epochs={epochs}", f"--batch={B}", path, "0"])
    p_ref, l_ref, acc_ref, d = oracle_run(scene, labels, dtype, B, lr, epochs, fifo=fifo)
    losses, acc_line, glob_line = parse(out)
    assert len(losses) == epochs
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6)          # printed with %lf (6 decimals)
    # tools/run.sh: accuracy = field 10, test time = field 7 of the Accuracy line; train time = field 16 (MBGD)
    f = acc_line.split(" ")
    assert f[8] == "Accuracy:" and f[9].endswith("%") and float(f[6]) >= 0
    assert abs(float(f[9].rstrip("%")) - 100 * acc_ref) <= 0.1 + 1e-9
    g = glob_line.split(" ")
    assert g[11] == "Global" and g[12] == "Exesution" and float(g[15]) > 0
    assert f"BatchSize = {B}]" in glob_line
    m = sio.loadmat(str(tmp_path / "model" / "model.mat"))
    assert m["filters"].shape == (171, 20) and m["omega1"].shape == (100, d.pout) and m["loss"].shape == (300, 1)
    got = np.concatenate([m[k].ravel(order="F") for k in ("filters", "bias0", "omega1", "bias1", "omega2", "bias2")])
    tol = 1e-5 if dtype == np.float32 else 1e-11
    assert np.abs(got - p_ref).max() <= tol * np.abs(p_ref).max()
    assert np.allclose(m["loss"].ravel()[:epochs], l_ref, rtol=10 * tol)


def test_sgd_cli_and_inference_mode(tmp_path):
    path, scene, labels = make_dataset(tmp_path, np.float64)
    epochs = 2
    out = run_gcn(tmp_path, ["--variant=sgd_fp64", f"--seed={SEED}", f"--epochs={epochs}", path, "0"])
    p_ref, l_ref, acc_ref, d = oracle_run(scene, labels, np.float64, 1, 0.008, epochs, sgd=True)
    losses, acc_line, glob_line = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6)
    g = glob_line.split(" ")
    assert g[7] == "Global" and float(g[11]) > 0                      # run.sh field 12 for SGD
    acc = float(acc_line.split(" ")[9].rstrip("%"))
    assert abs(acc - 100 * acc_ref) <= 0.1 + 1e-9
    m = sio.loadmat(str(tmp_path / "model.mat"))                      # SGD writes ./model.mat with its own names
    got = np.concatenate([m[k].ravel(order="F") for k in ("filters", "conv_bias", "fulconnect_weights",
                                                          "fulconnect_bias", "output_weights", "output_bias")])
    assert np.abs(got - p_ref).max() <= 1e-10 * np.abs(p_ref).max()
    # load the saved model back and only run the test loop
    out2 = run_gcn(tmp_path, ["--variant=sgd_fp64", f"--seed={SEED}", "--infer", "--model=model.mat", path, "0"])
    assert "Iteration" not in out2
    assert abs(float([l for l in out2.splitlines() if "Accuracy" in l][0].split(" ")[9].rstrip("%")) - acc) < 1e-9


@pytest.mark.parametrize("extra", [[], ["--exchange=oneshot"]])
def test_multi_gpu_form_with_one_device(tmp_path, extra):
    """`gcn DATASET.mat device_num master_dev_id` with device_num = 1 is the fp64 minibatch trainer;
    --exchange=oneshot sends the step through the gradient / one-kernel exchange + update path."""
    path, scene, labels = make_dataset(tmp_path, np.float64)
    B, epochs = 20, 2
    out = run_gcn(tmp_path, [f"--seed={SEED}", f"--epochs={epochs}", f"--batch={B}", *extra, path, "1", "0"])
    _, l_ref, acc_ref, _ = oracle_run(scene, labels, np.float64, B, 0.5, epochs)
    losses, acc_line, _ = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6)
    assert abs(float(acc_line.split(" ")[9].rstrip("%")) - 100 * acc_ref) <= 0.1 + 1e-9


@pytest.mark.parametrize("extra", [[], ["--exchange=oneshot"]])
def test_multi_gpu_form_fp32_with_one_device(tmp_path, extra):
    """`gcn DATASET.mat 1 0 --variant=mbgd_fp32`: the data-parallel driver around the fp32 trainer's
    arithmetic (plain-sum update, frozen conv bias, float loss accumulation) - with one device it
    must reproduce the single-GPU fp32 trainer, through RCCL's route and through the exchange.  The
    scene reaches the device in chunks of 1000 values here (the streaming upload crosses chunk ends)."""
    path, scene, labels = make_dataset(tmp_path, np.float32)
    B, epochs = 20, 2
    out = run_gcn(tmp_path, ["--variant=mbgd_fp32", f"--seed={SEED}", f"--epochs={epochs}", f"--batch={B}", *extra, path, "1", "0"],
                  env={"CCNN_SCENE_CHUNK": "1000"})
    _, l_ref, acc_ref, _ = oracle_run(scene, labels, np.float32, B, 0.04, epochs, fifo=100.0)
    losses, acc_line, _ = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6)
    assert abs(float(acc_line.split(" ")[9].rstrip("%")) - 100 * acc_ref) <= 0.1 + 1e-9


@pytest.mark.parametrize("variant,dtype,lr,fifo,ndev", [("mbgd_fp64", np.float64, 0.5, 1.0, 2), ("mbgd_fp32", np.float32, 0.04, 100.0, 2),
                                                       ("mbgd_fp64", np.float64, 0.5, 1.0, 4)])
def test_multi_gpu_form_with_several_device_slots(tmp_path, variant, dtype, lr, fifo, ndev):
    """`gcn DATASET.mat n 0` with n device slots (CCNN_CLI_DEVICE_MAP puts them all on the one GPU
    of the test box): one host thread per slot, every slot its own slice, streams and parameter
    copy, the sums traded inside the step.  Against the oracle's replay of the reference's
    master / worker step (hyper_version_mGPU_MBGD.cu:903-1171: slot k owns samples
    [k T/n, (k+1) T/n) and takes B/n of them per step; mean over the global batch in fp64) - in fp32
    against the same sub-batches pushed through the single-device trainer's plain-sum update."""
    path, scene, labels = make_dataset(tmp_path, dtype)
    B, epochs = 20, 2
    out = run_gcn(tmp_path, [f"--variant={variant}", f"--seed={SEED}", f"--epochs={epochs}", f"--batch={B}",
                             "--exchange=oneshot", path, str(ndev), "0"],
                  env={"CCNN_CLI_DEVICE_MAP": ",".join(["0"] * ndev)})
    d = orc.dims(Z)
    di = orc.scan_labels(labels, X, Y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, X, Y)
    orc.shuffle(tr9, onehot, SEED)
    p = orc.init_params_f32(Z) if dtype == np.float32 else orc.init_params_f64(Z, d.pout * B)
    sc = scene.astype(dtype)
    train, test = orc.gather(sc, tr9, X, Y, Z), orc.gather(sc, te9, X, Y, Z)
    T, sub_b, slice_len = len(onehot), B // ndev, len(onehot) // ndev
    sched = orc.Sched(dtype, fifo)
    cur = dtype(lr)
    l_ref = []
    for _ in range(epochs):
        per_dev = [[] for _ in range(ndev)]
        for i0 in range(T // B):
            firsts = [orc.mgpu_first_sample(T, ndev, B, k, i0) for k in range(ndev)]
            assert firsts == [k * slice_len + i0 * sub_b for k in range(ndev)]
            # the step's B samples side by side: one oracle step over them is the summed update
            rows = np.concatenate([np.arange(f, f + sub_b) for f in firsts])
            lv, _, _ = orc.mbgd_step(p, np.ascontiguousarray(train[rows]), np.ascontiguousarray(onehot[rows]), 0,
                                     len(rows), Z, cur, want_grad=False)
            for k in range(ndev):
                per_dev[k].extend(lv[k * sub_b:(k + 1) * sub_b])
        tot = dtype(0) if dtype == np.float32 else 0.0
        for k in range(ndev):                 # the host adds device after device, sample after sample
            for v in per_dev[k]:
                tot = dtype(np.float64(tot) + v) if dtype == np.float32 else tot + v
        l_ref.append(float(tot) / T)
        _, nl = sched.step(l_ref[-1], float(cur))
        cur = dtype(nl)
    _, am = orc.forward(p, test, Z)
    losses, acc_line, _ = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6), (losses, l_ref)
    assert abs(float(acc_line.split(" ")[9].rstrip("%")) - 100 * float(np.mean(am + 1 == tl))) <= 0.1 + 1e-9
    assert f"Training implemented on {ndev} Device(s), master 0." in out


def test_tail_batch_kept(tmp_path):
    """--tail=keep trains the last T % B samples as a short batch (hyper_version_MBGD.cu:780, 794-796)."""
    path, scene, labels = make_dataset(tmp_path, np.float64)
    B, epochs = 32, 2
    out = run_gcn(tmp_path, ["--variant=mbgd_fp64", f"--seed={SEED}", f"--epochs={epochs}", f"--batch={B}",
                             "--tail=keep", path, "0"])
    _, l_ref, acc_ref, _ = oracle_run(scene, labels, np.float64, B, 0.5, epochs, keep_tail=True)
    _, l_drop, _, _ = oracle_run(scene, labels, np.float64, B, 0.5, epochs)
    losses, acc_line, _ = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6) and not np.allclose(l_ref, l_drop, rtol=1e-4)
    assert abs(float(acc_line.split(" ")[9].rstrip("%")) - 100 * acc_ref) <= 0.1 + 1e-9


def test_preprocess_dump(tmp_path):
    """`gcn --preprocess DATASET.mat 0` writes what pre-treatment/preprocess048.cu:240-261 writes:
    cubes of all labelled pixels in scan order, one-hot training labels, plain test labels."""
    path, scene, labels = make_dataset(tmp_path, np.float64)
    out = run_gcn(tmp_path, ["--preprocess", path, "0"])
    di = orc.scan_labels(labels, X, Y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, X, Y)
    assert f"train_size:{len(tr9)}  test_size:{len(te9)}" in out
    tr = sio.loadmat(str(tmp_path / "traindata.mat"))
    te = sio.loadmat(str(tmp_path / "testdata.mat"))
    assert tr["data"].shape == (9 * Z, len(tr9)) and te["data"].shape == (9 * Z, len(te9))
    assert np.array_equal(tr["data"].T, orc.gather(scene, tr9, X, Y, Z))          # bit-exact copy
    assert np.array_equal(te["data"].T, orc.gather(scene, te9, X, Y, Z))
    assert np.array_equal(tr["labels"].T, onehot.astype(np.float64))
    assert np.array_equal(te["labels"].ravel(), tl.astype(np.float64))


def test_argument_errors(tmp_path):
    r = subprocess.run([GCN, "only_one_arg"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and "3 input arguments required!" in r.stderr      # gcn_mbgd_fp64.cu:1269-1272
    r = subprocess.run([GCN, "missing.mat", "0"], cwd=tmp_path, capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "cannot open" in r.stderr


@pytest.mark.parametrize("variant,dtype,lr,fifo,explicit", [("mbgd_fp32", np.float32, 0.04, 100.0, False),
                                                            ("mbgd_fp64", np.float64, 0.5, 1.0, True),
                                                            ("sgd_fp64", np.float64, 0.008, 1.0, False)])
def test_sixteen_class_label_map_trains(tmp_path, variant, dtype, lr, fifo, explicit):
    """A label map with 16 classes (the reference needs NEU_NUM2 = 16 and a rebuild; the oracle is
    that rebuild): picked up from the labels or given with --classes=16; losses, weights, model.mat
    shapes and accuracy against the oracle's replay."""
    scene, labels = synth.small_scene(X, Y, Z, 3)
    rng = np.random.default_rng(8)
    lab16 = np.where(labels > 0, rng.integers(1, 17, labels.size), 0).astype(labels.dtype)
    assert lab16.max() == 16
    if explicit:                     # no label above 13 left: only the option can say 16
        lab16 = np.where(lab16 > 13, lab16 - 3, lab16)
    ds = scene.reshape(Z, Y, X).transpose(2, 1, 0).astype(dtype)
    path = str(tmp_path / "DATASET16.mat")
    sio.savemat(path, {"DataSet": ds, "labels": lab16.reshape(Y, X).T}, do_compression=True)
    sgd = variant == "sgd_fp64"
    B, epochs = (1, 2) if sgd else (20, 3)
    args = [f"--variant={variant}", f"--seed={SEED}", f"--epochs={epochs}"] + ([] if sgd else [f"--batch={B}"])
    out = run_gcn(tmp_path, args + (["--classes=16"] if explicit else []) + [path, "0"])
    with orc.use_classes(16):
        p_ref, l_ref, acc_ref, d = oracle_run(scene, lab16, dtype, B, lr, epochs, sgd=sgd, fifo=fifo)
    losses, acc_line, _ = parse(out)
    assert np.allclose(losses, l_ref, rtol=2e-5, atol=1e-6)
    assert abs(float(acc_line.split(" ")[9].rstrip("%")) - 100 * acc_ref) <= 0.1 + 1e-9
    names = (("filters", "conv_bias", "fulconnect_weights", "fulconnect_bias", "output_weights", "output_bias") if sgd
             else ("filters", "bias0", "omega1", "bias1", "omega2", "bias2"))
    m = sio.loadmat(str(tmp_path / ("model.mat" if sgd else os.path.join("model", "model.mat"))))
    assert m[names[4]].shape == (16, 100) and m[names[5]].shape == (16, 1)
    got = np.concatenate([m[k].ravel(order="F") for k in names])
    tol = 1e-5 if dtype == np.float32 else 1e-10
    assert np.abs(got - p_ref).max() <= tol * np.abs(p_ref).max()
