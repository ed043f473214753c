"""Run-time class count (2..16 through CCNN_GEOMETRY) against the oracle compiled with the same
NEU_NUM2 (the reference fixes it at compile time, gcn_mbgd_fp64.cu:18): forward, minibatch
gradient / fused step on both training paths, per-sample SGD, the host prologue.  Parity
unpinned like the 13-class tests: the reference holds no outputs for any class count."""
import numpy as np
import pytest
import torch

import oracle as orc
import parity
import synth
from gcn_amd import lib as L
from gcn_amd import ops, prep

pytestmark = pytest.mark.gpu
TOL = {np.float32: 1e-5, np.float64: 1e-12}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    return float(np.max(np.abs(a - b)) / max(np.max(np.abs(b)), 1e-300))


def make_problem(z, n, seed, dtype, classes):
    rng = np.random.default_rng(seed)
    cubes = rng.random((n, 9 * z)).astype(dtype)
    onehot = np.zeros((n, classes), dtype=np.int32)
    onehot[np.arange(n), rng.integers(0, classes, n)] = 1
    orc.srand(seed)
    params = orc.init_params_f64(z, 0) if dtype == np.float64 else orc.init_params_f32(z)
    return params, cubes, onehot


@pytest.mark.parametrize("classes", [16, 9])
@pytest.mark.parametrize("dtype,z,n", [(np.float32, 103, 70), (np.float64, 176, 33), (np.float32, 200, 600)])
def test_forward_matches_oracle(dtype, z, n, classes):
    with orc.use_classes(classes):
        params, cubes, _ = make_problem(z, n, 100 + z, dtype, classes)
        params = (params * 5).astype(dtype)
        pr, am = orc.forward(params, cubes, z)
    probs, argmax = ops.forward(dev(params), dev(cubes), L.geometry(z, classes))
    assert probs.shape == (n, classes)
    assert relerr(probs.cpu().numpy(), pr) < TOL[dtype]
    top2 = np.sort(pr, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 10 * TOL[dtype] * top2[:, 1]
    assert np.array_equal(argmax.cpu().numpy()[clear], am[clear]) and clear.mean() > 0.9


@pytest.mark.parametrize("classes", [16, 9])
@pytest.mark.parametrize("dtype,z,B,lr", [(np.float32, 103, 128, 0.04), (np.float64, 176, 100, 0.5),
                                          (np.float32, 200, 37, 0.04), (np.float64, 60, 5, 0.5)])
def test_mbgd_matches_oracle(dtype, z, B, lr, classes, mbgd_path):
    """Gradient, probabilities and losses, then the fused step, three batches in a row."""
    steps = 3
    g = L.geometry(z, classes)
    with orc.use_classes(classes):
        params, cubes, onehot = make_problem(z, B * steps, 7 + z, dtype, classes)
        d = orc.dims(z)
        p_ref, p_ref2 = params.copy(), params.copy()
        p_gpu, p_gpu2, c_gpu, o_gpu = dev(params), dev(params), dev(cubes), dev(onehot)
        ws = ops.Workspace()
        for i0 in range(steps):
            lv_ref, pr_ref, g_ref = orc.mbgd_step(p_ref, cubes, onehot, i0 * B, B, z, dtype(lr))
            probs = torch.empty((B, classes), dtype=p_gpu.dtype, device="cuda")
            grad, lv = ops.mbgd_grad(p_gpu, c_gpu, o_gpu, i0 * B, B, g, probs=probs, workspace=ws)
            parity.assert_groups(grad.cpu().numpy(), g_ref, d, TOL[dtype], f"gradient, step {i0}")
            assert relerr(lv.cpu().numpy(), lv_ref) < TOL[dtype], i0
            assert relerr(probs.cpu().numpy(), pr_ref) < TOL[dtype], i0
            ops.update(p_gpu, grad, lr, g, divisor=B)
            parity.assert_groups(p_gpu.cpu().numpy(), p_ref, d, TOL[dtype], f"parameters, step {i0}")
            lv2_ref, _, _ = orc.mbgd_step(p_ref2, cubes, onehot, i0 * B, B, z, dtype(lr), want_grad=False)
            lv2 = ops.mbgd_step(p_gpu2, c_gpu, o_gpu, i0 * B, B, g, lr, workspace=ws)
            assert relerr(lv2.cpu().numpy(), lv2_ref) < TOL[dtype], i0
            parity.assert_groups(p_gpu2.cpu().numpy(), p_ref2, d, TOL[dtype], f"fused step {i0}")


@pytest.mark.parametrize("classes", [16, 9])
@pytest.mark.parametrize("z,n", [(103, 40), (176, 25)])
def test_sgd_matches_oracle(z, n, classes):
    with orc.use_classes(classes):
        params, cubes, onehot = make_problem(z, n, 31 + z, np.float64, classes)
        p_ref = params.copy()
        loss_ref = orc.sgd_f64(p_ref, cubes, onehot, 0, n, z, 0.008)
    p_gpu = dev(params)
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.sgd(p_gpu, dev(cubes), dev(onehot), 0, n, L.geometry(z, classes), 0.008, loss)
    assert relerr(p_gpu.cpu().numpy(), p_ref) < 1e-10
    assert abs(loss.item() - loss_ref) < 1e-10 * max(1.0, abs(loss_ref))
