"""GPU: the benchmark configurations at FULL size against the CPU oracle, through the C ABI,
with the bars per parameter group (tests/parity.py).

  config 2  KSC shape 512x614x176, fp64: gather, one SGD epoch, inference after it
  config 3  Indian-Pines shape 145x145x200, fp32 MBGD B = 128: every step of an epoch
  config 4  Pavia-U shape 610x340x103, fp32 MBGD B = 128 (one GPU): every step of an epoch

fp32: each GPU step starts from the oracle's weights, so a step's error is its own; a second
replica runs free for the accuracy / argmax comparison at the end of the epoch."""
import numpy as np
import pytest
import torch

import oracle as orc
import parity
import synth
from gcn_amd import ops

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def prepared(name, seed, shuffle_seed, dtype):
    x, y, z, nl = synth.SCENES[name]
    rng = np.random.default_rng(seed)
    labels = synth.make_labels(x, y, nl, rng)
    scene = synth.make_scene(x, y, z, dtype, rng)
    di = orc.scan_labels(labels, x, y)
    tr9, te9, onehot, tl = orc.build_index(labels, di, x, y)
    orc.shuffle(tr9, onehot, shuffle_seed)
    ref_train, ref_test = orc.gather(scene, tr9, x, y, z), orc.gather(scene, te9, x, y, z)
    sc = dev(scene)
    g_train = ops.gather(sc, dev(tr9), x, y, z)
    g_test = ops.gather(sc, dev(te9), x, y, z, sort=False)
    assert g_train.cpu().numpy().tobytes() == ref_train.tobytes(), "training cubes differ"
    assert g_test.cpu().numpy().tobytes() == ref_test.tobytes(), "test cubes differ"
    return z, onehot, tl, ref_train, ref_test, g_train, g_test


def assert_predictions(params_gpu, g_test, params_ref, ref_test, tl, z, tol):
    pr_ref, am_ref = orc.forward(params_ref, ref_test, z)
    _, am = ops.forward(params_gpu, g_test, z)
    am = am.cpu().numpy()
    top2 = np.sort(pr_ref.astype(np.float64), axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 100 * tol * top2[:, 1]
    assert clear.mean() > 0.9
    assert np.array_equal(am[clear], am_ref[clear]), "argmax differs on pixels with a clear top-2 margin"
    acc, acc_ref = np.mean(am + 1 == tl) * 100, np.mean(am_ref + 1 == tl) * 100
    assert abs(acc - acc_ref) <= 0.1, (acc, acc_ref)
    return int((am != am_ref).sum())


def test_config2_ksc_fp64_sgd_epoch_and_inference():
    z, onehot, tl, ref_train, ref_test, g_train, g_test = prepared("ksc", 20242, 7, np.float64)
    d = orc.dims(z)
    T = len(onehot)
    assert (T, len(tl)) == (4168, 1043)
    orc.srand(7)
    p_ref = orc.init_params_f64(z, d.pout)
    p_gpu = dev(p_ref)
    loss_ref = orc.sgd_f64(p_ref, ref_train, onehot, 0, T, z, 0.008)
    sloss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.sgd(p_gpu, g_train, dev(onehot), 0, T, z, 0.008, sloss)
    parity.assert_groups(p_gpu.cpu().numpy(), p_ref, d, 1e-12, "SGD epoch")
    assert abs(sloss.item() - loss_ref) <= 1e-12 * abs(loss_ref)
    pr_ref, _ = orc.forward(p_ref, ref_test, z)
    pr, _ = ops.forward(p_gpu, g_test, z)
    assert parity.relerr(pr.cpu().numpy(), pr_ref) <= 1e-12
    assert_predictions(p_gpu, g_test, p_ref, ref_test, tl, z, 1e-12)


@pytest.mark.parametrize("name,seed,counts", [("indian_pines", 20243, (8199, 2050)), ("pavia_u", 20244, (34220, 8556))])
def test_config34_fp32_mbgd_b128_every_step_of_an_epoch(name, seed, counts, mbgd_path):
    z, onehot, tl, ref_train, ref_test, g_train, g_test = prepared(name, seed, 9, np.float32)
    T, B, lr = len(onehot), 128, np.float32(0.04)
    assert (T, len(tl)) == counts
    oh_d = dev(onehot)
    orc.srand(9)
    p_ref = orc.init_params_f32(z)
    p_free = dev(p_ref)
    lv = torch.zeros(B, dtype=torch.float64, device="cuda")
    ws = ops.Workspace()
    flipped_steps = 0
    for i in range(T // B):
        p_step = dev(p_ref)                              # this step from the oracle's weights
        ops.mbgd_step(p_step, g_train, oh_d, i * B, B, z, float(lr), loss_values=lv, workspace=ws)
        ops.mbgd_step(p_free, g_train, oh_d, i * B, B, z, float(lr), workspace=ws)
        p_ref, nflip = parity.check_fp32_step(p_ref, p_step.cpu().numpy(), lv.cpu().numpy(), ref_train, onehot,
                                              i * B, B, z, lr)
        flipped_steps += nflip > 0
    print(f"{name}: {T // B} steps, {flipped_steps} of them decided by the near-tie rule")
    assert flipped_steps <= 3, f"{flipped_steps} steps needed the near-tie rule: that is not rounding any more"
    # the free-running replica: one epoch of accumulated rounding, same predictions
    assert parity.relerr(p_free.cpu().numpy(), p_ref) <= 1e-3
    assert_predictions(p_free, g_test, p_ref, ref_test, tl, z, 1e-5)
