"""GPU: the HIP library (through the C ABI) against the CPU oracle and the golden
fixtures.  Tolerances: gather / labels / indices bit-exact; fp64 weights, losses
and probabilities 1e-12 relative to the tensor's largest magnitude; fp32 1e-5.
argmax must agree wherever the oracle's top-2 margin exceeds the tolerance."""
import os

import numpy as np
import pytest
import torch

import oracle as orc
import parity
import synth
from gcn_amd import lib as L
from gcn_amd import ops, prep

pytestmark = pytest.mark.gpu

TOL = {np.float64: 1e-12, np.float32: 1e-5}


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def assert_argmax(am_gpu, am_ref, probs_ref, tol):
    top2 = np.sort(probs_ref, axis=1)[:, -2:]
    clear = (top2[:, 1] - top2[:, 0]) > 10 * tol * top2[:, 1]
    assert np.array_equal(am_gpu[clear], am_ref[clear])
    assert clear.mean() > 0.9      # near-ties are rare; the check must not be vacuous


def make_problem(z, n, seed, dtype):
    rng = np.random.default_rng(seed)
    cubes = rng.random((n, 9 * z)).astype(dtype)
    onehot = np.zeros((n, 13), dtype=np.int32)
    onehot[np.arange(n), rng.integers(0, 13, n)] = 1
    orc.srand(seed)
    params = orc.init_params_f64(z, 0) if dtype == np.float64 else orc.init_params_f32(z)
    return params, cubes, onehot


# --------------------------------------------------------------------------- gather
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("name", ["index_7x6x24.npz", "index_16x12x176.npz"])
def test_gather_golden_bit_exact(golden, name, dtype):
    g = golden(name)
    x, y, z, seed = int(g["x"]), int(g["y"]), int(g["z"]), int(g["seed"])
    scene, _ = synth.small_scene(x, y, z, seed)
    scene = scene.astype(dtype)
    for key in ("train_idx9", "test_idx9"):
        idx = g[key]
        out = ops.gather(dev(scene), dev(idx), x, y, z).cpu().numpy()
        ref = orc.gather(scene, idx, x, y, z)
        assert out.tobytes() == ref.tobytes()
        if dtype == np.float64 and "train_cubes" in g and key == "train_idx9":
            assert out.tobytes() == g["train_cubes"].tobytes()


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gather_ragged_and_empty(dtype):
    x, y, z = 37, 29, 103
    scene, labels = synth.small_scene(x, y, z, 5, dtype)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    for idx in (tr9, te9, tr9[:1], tr9[:0]):
        out = ops.gather(dev(scene), dev(idx.reshape(-1, 9)), x, y, z).cpu().numpy()
        assert out.tobytes() == orc.gather(scene, idx.reshape(-1, 9), x, y, z).tobytes()


@pytest.mark.parametrize("force", [None, "dense", "sorted"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gather_planned_equals_gather(dtype, force, monkeypatch):
    """One look at the list, several gathers (two scenes with the same label mask, different band
    counts), with the plan's own choice of kernel and with either kernel forced."""
    if force:
        monkeypatch.setenv("CCNN_GATHER_PLAN", force)
    x, y = 37, 29
    _, labels = synth.small_scene(x, y, 21, 5, dtype)
    di = prep.scan_labels(labels, x, y)
    tr9, _, oh, _ = prep.build_index(labels, di, x, y)
    prep.shuffle(tr9, oh, 4)
    for idx in (tr9, tr9[:16], tr9[:17], tr9[:0]):
        idx_d = dev(np.ascontiguousarray(idx.reshape(-1, 9)))
        plan = ops.GatherPlan(idx_d, x, y)
        if force and len(idx) > 16:
            assert plan.dense == (force == "dense")
        for z, seed in ((103, 1), (40, 2), (103, 3)):
            scene = np.random.default_rng(seed).random(x * y * z).astype(dtype)
            out = plan.gather(dev(scene), z).cpu().numpy()
            assert out.tobytes() == orc.gather(scene, idx.reshape(-1, 9), x, y, z).tobytes(), (len(idx), z)


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_gather_dense_fully_labelled_scene(dtype, monkeypatch):
    """Every pixel of a 64 x 50 x 224 scene labelled: the plan picks the tile kernel; mirrored
    borders, partial tiles on both axes, a list that repeats pixels, and a list with one sample
    that is not a plain neighbourhood (the plan must fall back to the ordered list)."""
    x, y, z = 64, 50, 224
    monkeypatch.setenv("CCNN_GATHER_PLAN", "dense")     # 3200 samples: below the length the plan asks for by itself
    rng = np.random.default_rng(77)
    labels = rng.integers(1, 14, x * y).astype(np.int32)
    scene = rng.random(x * y * z).astype(dtype)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    every = np.concatenate([tr9.reshape(-1, 9), te9.reshape(-1, 9)])
    assert len(every) == x * y
    every = every[rng.permutation(len(every))]
    repeated = np.concatenate([every, every[:1500], every[:700]])
    repeated = repeated[rng.permutation(len(repeated))]
    odd = every.copy()
    odd[1234, 0], odd[1234, 8] = odd[1234, 8], odd[1234, 0]
    sc = dev(scene)
    for idx, dense in ((every, True), (repeated, True), (odd, False)):
        idx = np.ascontiguousarray(idx)
        plan = ops.GatherPlan(dev(idx), x, y)
        assert plan.dense == dense
        out = plan.gather(sc, z).cpu().numpy()
        assert out.tobytes() == orc.gather(scene, idx, x, y, z).tobytes()
        # a second scene with fewer bands through the same plan
        out = plan.gather(sc[: x * y * 33].contiguous(), 33).cpu().numpy()
        assert out.tobytes() == orc.gather(scene[: x * y * 33], idx, x, y, 33).tobytes()


def test_gather_plan_picks_tiles_for_a_large_fully_labelled_scene():
    """610 x 340 pixels all labelled, 20 bands: the plan chooses the tile kernel by itself and the
    result equals plain indexing, checked on the device."""
    x, y, z = 610, 340, 20
    rng = np.random.default_rng(5)
    labels = rng.integers(1, 14, x * y).astype(np.int32)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    idx = np.concatenate([tr9.reshape(-1, 9), te9.reshape(-1, 9)])
    idx = np.ascontiguousarray(idx[rng.permutation(len(idx))])
    sc = torch.rand(x * y * z, dtype=torch.float32, device="cuda")
    idx_d = dev(idx)
    plan = ops.GatherPlan(idx_d, x, y)
    assert plan.dense
    out = plan.gather(sc, z)
    ref = sc.view(z, x * y)[:, idx_d.long()].permute(1, 0, 2).reshape(len(idx), 9 * z)
    assert torch.equal(out, ref)
    assert torch.equal(out, ops.gather(sc, idx_d, x, y, z))


def test_gather_full_size_property():
    """Indian-Pines-sized scene: gather equals plain indexing (a pure copy), checked on the device."""
    x, y, z, nl = synth.SCENES["indian_pines"]
    rng = np.random.default_rng(20243)
    labels = synth.make_labels(x, y, nl, rng)
    scene = synth.make_scene(x, y, z, np.float32, rng)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    prep.shuffle(tr9, np.zeros((len(tr9), 13), dtype=np.int32), 1)
    sc = dev(scene)
    out = ops.gather(sc, dev(tr9), x, y, z)
    ref = sc.view(z, x * y)[:, dev(tr9).long()].permute(1, 0, 2).reshape(len(tr9), 9 * z)
    assert torch.equal(out, ref)
    # centre-pixel spectrum of every sample equals the scene's spectrum at that pixel
    assert torch.equal(out.view(-1, z, 9)[:, :, 4], sc.view(z, x * y)[:, dev(tr9[:, 4]).long()].t())


# --------------------------------------------------------------------------- forward
def test_forward_ksc_checkpoint(golden):
    g = golden("ksc_forward.npz")
    z = int(g["z"])
    probs, am = ops.forward(dev(g["params"]), dev(g["cubes"]), z)
    assert relerr(probs.cpu().numpy(), g["probs"]) < 1e-12
    assert np.array_equal(am.cpu().numpy(), g["argmax"])


@pytest.mark.parametrize("dtype,z", [(np.float32, 103), (np.float64, 60), (np.float32, 224), (np.float64, 176)])
def test_forward_scene_equals_gather_then_forward(dtype, z):
    """The scene-fed forward (neighbourhoods read inside the conv kernel) is bit-identical to
    gathering cubes first, on shuffled, ordered, 1-element and empty lists with border pixels."""
    x, y = 23, 19
    scene, labels = synth.small_scene(x, y, z, 5, dtype)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, oh, _ = prep.build_index(labels, di, x, y)
    prep.shuffle(tr9, oh, 3)
    orc.srand(2)
    params = dev(orc.init_params_f32(z) if dtype == np.float32 else orc.init_params_f64(z, 0))
    sc = dev(scene)
    for idx in (tr9, te9, tr9[:1], tr9[:7], tr9[:0]):
        idx_d = dev(np.ascontiguousarray(idx.reshape(-1, 9)))
        p1, a1 = ops.forward(params, ops.gather(sc, idx_d, x, y, z), z)
        p2, a2 = ops.forward_scene(params, sc, idx_d, x, y, z)
        assert torch.equal(p1, p2) and torch.equal(a1, a2)
    # and against the oracle, through the cubes the oracle gathers itself
    cubes = orc.gather(scene, tr9, x, y, z)
    p_ref, a_ref = orc.forward(params.cpu().numpy(), cubes, z)
    p2, a2 = ops.forward_scene(params, sc, dev(tr9), x, y, z)
    assert relerr(p2.cpu().numpy(), p_ref) < (1e-5 if dtype == np.float32 else 1e-12)


@pytest.mark.parametrize("dtype,z", [(np.float32, 103), (np.float64, 40), (np.float32, 230)])
def test_forward_long_lists_every_row_once(dtype, z):
    """Lists much longer than the conv kernel's fixed shares (three rows per resident wave): the
    rows beyond them are handed out through work counters.  Every pixel of a scene, repeated, from
    the scene and from cubes; both must agree bit for bit with the same pixels classified in
    short lists (which never touch the counters)."""
    x, y = 64, 50
    scene, labels = synth.small_scene(x, y, z, 11, dtype)
    labels[:] = 1                                         # every pixel is a sample
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    idx = np.concatenate([tr9.reshape(-1, 9), te9.reshape(-1, 9)])
    idx = np.ascontiguousarray(np.tile(idx, (12, 1)))     # 38400 rows
    orc.srand(4)
    params = dev(orc.init_params_f32(z) if dtype == np.float32 else orc.init_params_f64(z, 0))
    sc, idx_d = dev(scene), dev(idx)
    p_scene, a_scene = ops.forward_scene(params, sc, idx_d, x, y, z)
    p_cubes, a_cubes = ops.forward(params, ops.gather(sc, idx_d, x, y, z), z)
    assert torch.equal(p_scene, p_cubes) and torch.equal(a_scene, a_cubes)
    n1 = x * y
    for s0 in range(0, n1, 800):                          # short lists: fixed shares only
        p_short, _ = ops.forward_scene(params, sc, idx_d[s0:s0 + 800].contiguous(), x, y, z)
        for rep in (0, 5, 11):
            assert torch.equal(p_scene[rep * n1 + s0:rep * n1 + s0 + 800], p_short)


@pytest.mark.parametrize("dtype,z,n", [(np.float64, 176, 67), (np.float32, 200, 130), (np.float32, 103, 33),
                                       (np.float64, 21, 5), (np.float32, 224, 17), (np.float64, 22, 3),
                                       (np.float32, 448, 9), (np.float64, 448, 5), (np.float32, 60, 1)])
def test_forward_matches_oracle(dtype, z, n):
    params, cubes, _ = make_problem(z, n, 100 + z, dtype)
    params = (params * 5).astype(dtype)          # spread the probabilities
    pr, am = orc.forward(params, cubes, z)
    probs, argmax = ops.forward(dev(params), dev(cubes), z)
    assert relerr(probs.cpu().numpy(), pr) < TOL[dtype]
    assert_argmax(argmax.cpu().numpy(), am, pr, TOL[dtype])
    # optional outputs
    p2, a2 = ops.forward(dev(params), dev(cubes), z, want_probs=False)
    assert p2 is None and torch.equal(a2, argmax)


def test_forward_chunked_large_batch():
    """More samples than one workspace chunk; every chunk must agree with a direct call."""
    z, n = 103, 20000
    params, cubes, _ = make_problem(z, 64, 9, np.float32)
    big = np.tile(cubes, (n // 64 + 1, 1))[:n]
    os.environ["CCNN_FWD_CHUNK"] = "16384"   # the default chunk holds far more than 20000 samples
    try:
        probs, am = ops.forward(dev(params), dev(big), z)
    finally:
        del os.environ["CCNN_FWD_CHUNK"]
    p64, a64 = ops.forward(dev(params), dev(cubes), z)
    assert torch.equal(probs.view(-1)[:64 * 13], p64.view(-1))
    assert torch.equal(probs[-(n % 64):], p64[:n % 64])
    labels = dev((orc.forward(params, cubes, z)[1] + 1).astype(np.int32))
    cnt = ops.count_correct(am[:64].contiguous(), labels)
    assert int(cnt.item()) == 64
    assert int(ops.count_correct(am[:0].contiguous(), labels[:0].contiguous()).item()) == 0


# --------------------------------------------------------------------------- MBGD
@pytest.mark.parametrize("dtype,key,lr", [(np.float64, "mbgd64", 0.5), (np.float32, "mbgd32", 0.04)])
def test_mbgd_golden(golden, dtype, key, lr, mbgd_path):
    g = golden("train_small.npz")
    z = int(g["z"])
    cubes = dev(g["cubes"].astype(dtype))
    onehot = dev(g["onehot"])
    params = dev(g[f"{key}_p0"])
    for i0 in range(2):
        grad, lv = ops.mbgd_grad(params, cubes, onehot, i0 * 4, 4, z)
        assert relerr(grad.cpu().numpy(), g[f"{key}_grad{i0}"]) < TOL[dtype]
        assert relerr(lv.cpu().numpy(), g[f"{key}_loss"][i0]) < TOL[dtype]
        ops.update(params, grad, lr, z, divisor=4)
    assert relerr(params.cpu().numpy(), g[f"{key}_p2"]) < TOL[dtype]
    d = orc.dims(z)
    if dtype == np.float32:
        assert np.all(params.cpu().numpy()[d.off_b0:d.off_w1] == 0)


@pytest.mark.parametrize("dtype,z,B,lr", [(np.float64, 176, 100, 0.5), (np.float32, 200, 128, 0.04),
                                          (np.float32, 103, 128, 0.04), (np.float64, 103, 37, 0.5),
                                          (np.float32, 224, 16, 0.04), (np.float32, 21, 5, 0.04),
                                          (np.float64, 448, 6, 0.5), (np.float32, 448, 3, 0.04),
                                          (np.float64, 60, 1, 0.5)])
def test_mbgd_steps_match_oracle(dtype, z, B, lr, mbgd_path):
    steps = 3
    params, cubes, onehot = make_problem(z, B * steps, 7 + z, dtype)
    p_ref = params.copy()
    p_gpu, c_gpu, o_gpu = dev(params), dev(cubes), dev(onehot)
    ws = ops.Workspace()
    d = orc.dims(z)
    for i0 in range(steps):
        lv_ref, pr_ref, g_ref = orc.mbgd_step(p_ref, cubes, onehot, i0 * B, B, z, dtype(lr))
        probs = torch.empty((B, 13), dtype=p_gpu.dtype, device="cuda")
        grad, lv = ops.mbgd_grad(p_gpu, c_gpu, o_gpu, i0 * B, B, z, probs=probs, workspace=ws)
        parity.assert_groups(grad.cpu().numpy(), g_ref, d, TOL[dtype], f"gradient, step {i0}")
        assert relerr(lv.cpu().numpy(), lv_ref) < TOL[dtype], i0
        assert relerr(probs.cpu().numpy(), pr_ref) < TOL[dtype], i0
        ops.update(p_gpu, grad, lr, z, divisor=B)
        parity.assert_groups(p_gpu.cpu().numpy(), p_ref, d, TOL[dtype], f"parameters, step {i0}")


@pytest.mark.parametrize("dtype,z,B,lr", [(np.float64, 176, 100, 0.5), (np.float32, 103, 128, 0.04),
                                          (np.float32, 200, 50, 0.04), (np.float64, 30, 7, 0.5),
                                          (np.float32, 103, 700, 0.01), (np.float64, 60, 530, 0.5),
                                          (np.float32, 224, 300, 0.01)])
def test_mbgd_fused_step_matches_oracle(dtype, z, B, lr, mbgd_path):
    """ccnn_mbgd_step = gradient + the variant's own update in one call."""
    steps = 3
    params, cubes, onehot = make_problem(z, B * steps, 17 + z, dtype)
    p_ref = params.copy()
    p_gpu, c_gpu, o_gpu = dev(params), dev(cubes), dev(onehot)
    ws = ops.Workspace()
    d = orc.dims(z)
    for i0 in range(steps):
        lv_ref, _, _ = orc.mbgd_step(p_ref, cubes, onehot, i0 * B, B, z, dtype(lr), want_grad=False)
        lv = ops.mbgd_step(p_gpu, c_gpu, o_gpu, i0 * B, B, z, lr, workspace=ws)
        assert relerr(lv.cpu().numpy(), lv_ref) < TOL[dtype], i0
        out = p_gpu.cpu().numpy()
        parity.assert_groups(out, p_ref, d, TOL[dtype], f"parameters, step {i0}")
        if dtype == np.float32:
            assert np.all(out[d.off_b0:d.off_w1] == 0)


@pytest.mark.parametrize("dtype,z,batches", [
    (np.float32, 103, (1, 2, 3, 7, 8, 9, 15, 17, 31, 33, 63, 65, 127, 129, 255, 256, 257, 511, 512, 513)),
    (np.float64, 60, (1, 2, 5, 8, 9, 16, 17, 63, 65, 129, 255, 256, 257, 258)),
    (np.float32, 200, (1, 6, 10, 130, 250, 257, 510))])
def test_mbgd_step_at_the_edges_of_every_batch_split(dtype, z, batches):
    """One step at each batch size around the places where the work is cut: a single sample, batches
    below and just over the eight samples of one round of workgroups, sizes one off a multiple of the
    16-sample product steps and of the eight-way split of the batch sum, the largest batch whose
    samples are shared by two workgroups, and the sizes either side of the change of kernel."""
    nmax = max(batches)
    params, cubes, onehot = make_problem(z, nmax + 40, 5 + z, dtype)
    d = orc.dims(z)
    c_gpu, o_gpu = dev(cubes), dev(onehot)
    ws = ops.Workspace()
    lr = 0.04 if dtype == np.float32 else 0.5
    for k, B in enumerate(batches):
        i0 = k % 40                      # a different, unaligned start in the pool each time
        p_ref = params.copy()
        lv_ref, _, _ = orc.mbgd_step(p_ref, cubes, onehot, i0, B, z, dtype(lr / max(1, B // 64)), want_grad=False)
        p_gpu = dev(params)
        lv = ops.mbgd_step(p_gpu, c_gpu, o_gpu, i0, B, z, lr / max(1, B // 64), workspace=ws)
        assert relerr(lv.cpu().numpy(), lv_ref) < TOL[dtype], B
        parity.assert_groups(p_gpu.cpu().numpy(), p_ref, d, TOL[dtype], f"parameters, batch {B}")


@pytest.mark.parametrize("dtype,z,B,lr", [(np.float32, 103, 301, 0.01),      # the smallest batches of the tile path: two samples per tile, the last one ragged
                                          (np.float32, 103, 2045, 0.002),    # full tiles of eight, the last one ragged
                                          (np.float32, 200, 1797, 0.002),    # full tiles, delta image in four chunks
                                          (np.float32, 224, 1024, 0.004),    # four samples per tile
                                          (np.float32, 224, 2043, 0.002),    # the largest shape that fits eight samples
                                          (np.float32, 448, 1000, 0.002),    # four waves per workgroup
                                          (np.float64, 60, 2048, 0.5),       # fp64, full tiles
                                          (np.float64, 176, 1024, 0.5)])     # fp64, four waves
def test_mbgd_large_batches_match_oracle(dtype, z, B, lr, monkeypatch):
    """Batches large enough for every form of the sample-tile kernel (full tiles, ragged last tile,
    chunked delta images, four-wave workgroups) and for the one-pass batch reduction, two steps each,
    on the path the library picks by itself and on the layer kernels."""
    params, cubes, onehot = make_problem(z, 2 * B, 23 + z, dtype)
    d = orc.dims(z)
    c_gpu, o_gpu = dev(cubes), dev(onehot)
    p_ref = params.copy()
    refs = []
    for i0 in range(2):
        lv_ref, _, _ = orc.mbgd_step(p_ref, cubes, onehot, i0 * B, B, z, dtype(lr), want_grad=False)
        refs.append((lv_ref, p_ref.copy()))
    for path in (None, "tile", "tile8", "layers"):
        monkeypatch.delenv("CCNN_TILE_WAVES", raising=False)
        if path == "tile8":       # full fp32 tiles run sixteen waves per workgroup unless told otherwise
            monkeypatch.setenv("CCNN_MBGD_PATH", "tile")
            monkeypatch.setenv("CCNN_TILE_WAVES", "8")
        elif path:
            monkeypatch.setenv("CCNN_MBGD_PATH", path)
        else:
            monkeypatch.delenv("CCNN_MBGD_PATH", raising=False)
        p_gpu = dev(params)
        ws = ops.Workspace()
        for i0 in range(2):
            lv = ops.mbgd_step(p_gpu, c_gpu, o_gpu, i0 * B, B, z, lr, workspace=ws)
            assert relerr(lv.cpu().numpy(), refs[i0][0]) < TOL[dtype], (path, i0)
            parity.assert_groups(p_gpu.cpu().numpy(), refs[i0][1], d, TOL[dtype], f"parameters, path {path}, step {i0}")


def test_large_batch_step_does_not_depend_on_what_the_workspace_held(monkeypatch):
    """A step of 1600 samples on the sample-tile path and on the layer kernels: a workspace full of
    ones, or one that an earlier, larger call has used, gives the same bits as a zeroed one (every
    row and counter a launch reads has been written or cleared inside the same step)."""
    z, B = 103, 1600
    params, cubes, onehot = make_problem(z, B, 31, np.float32)
    c_gpu, o_gpu = dev(cubes), dev(onehot)
    for path in ("tile", "layers"):
        monkeypatch.setenv("CCNN_MBGD_PATH", path)
        outs = []
        for fill in (0x00, 0xff, None):
            ws = ops.Workspace()
            nbytes = L.load().ccnn_mbgd_workspace_bytes(2048, z, 4)
            ws.get(nbytes).fill_(fill if fill is not None else 0)
            p_gpu = dev(params)
            if fill is None:      # a larger batch first: its rows and counters stay behind
                ops.mbgd_step(dev(params), dev(np.concatenate([cubes, cubes[:448]])), dev(np.concatenate([onehot, onehot[:448]])),
                              0, 2048, z, 0.001, workspace=ws)
            lv = ops.mbgd_step(p_gpu, c_gpu, o_gpu, 0, B, z, 0.002, workspace=ws)
            outs.append((p_gpu.cpu(), lv.cpu()))
        for p_o, lv_o in outs[1:]:
            assert torch.equal(p_o, outs[0][0]) and torch.equal(lv_o, outs[0][1]), path
    p_ref = params.copy()
    lv_ref, _, _ = orc.mbgd_step(p_ref, cubes, onehot, 0, B, z, np.float32(0.002), want_grad=False)
    parity.assert_groups(outs[0][0].numpy(), p_ref, orc.dims(z), 1e-5, "batch sums of 1600 samples")


@pytest.mark.parametrize("dtype,z,B,lr", [(np.float32, 103, 128, 0.04), (np.float64, 176, 100, 0.5), (np.float32, 200, 37, 0.04)])
def test_pair_exchange_does_not_depend_on_what_the_workspace_held(dtype, z, B, lr):
    """At the reference's batch sizes the two workgroups of a sample trade their halves of the hidden
    layer's sums through tagged words in the caller's workspace.  A reader clears what it has read and
    every launch has a tag of its own, so a workspace of zeros, of ones, of random bytes, or one that an
    earlier, differently sized call has used gives the same bits - three steps in a row each, and the
    result is the oracle's."""
    params, cubes, onehot = make_problem(z, 3 * B, 41, dtype)
    c_gpu, o_gpu = dev(cubes), dev(onehot)
    es = np.dtype(dtype).itemsize
    assert L.load().ccnn_mbgd_path(B, z, es) == 0
    nbytes = L.load().ccnn_mbgd_workspace_bytes(128, z, es)
    outs = []
    for fill in (0x00, 0xff, "random", "used"):
        ws = ops.Workspace()
        buf = ws.get(nbytes)
        if fill == "random":
            buf.copy_(torch.randint(0, 256, (buf.numel(),), dtype=torch.uint8, generator=torch.Generator().manual_seed(5)).to(buf.device))
        elif fill == "used":
            buf.fill_(0)
            ops.mbgd_step(dev(params), c_gpu, o_gpu, 0, min(128, B + 19), z, lr, workspace=ws)
        else:
            buf.fill_(fill)
        p_gpu = dev(params)
        lvs = [ops.mbgd_step(p_gpu, c_gpu, o_gpu, i0 * B, B, z, lr, workspace=ws).cpu() for i0 in range(3)]
        outs.append((p_gpu.cpu(), torch.stack(lvs)))
    for p_o, lv_o in outs[1:]:
        assert torch.equal(p_o, outs[0][0]) and torch.equal(lv_o, outs[0][1])
    p_ref = params.copy()
    for i0 in range(3):
        orc.mbgd_step(p_ref, cubes, onehot, i0 * B, B, z, dtype(lr), want_grad=False)
    parity.assert_groups(outs[0][0].numpy(), p_ref, orc.dims(z), TOL[dtype] * 3, "three steps through the pair exchange")


def test_a_pair_whose_partner_never_answers_poisons_the_step(monkeypatch):
    """CCNN_FB_TEST_MUTE_PARTNER makes the second workgroup of every pair publish nothing: the first
    waits its two seconds, the call still returns, and the step cannot pass for a good one - every
    loss value is a NaN, and so are the parameters the sample sums reach."""
    z, B = 103, 16
    params, cubes, onehot = make_problem(z, B, 47, np.float32)
    p_gpu = dev(params)
    monkeypatch.setenv("CCNN_FB_TEST_MUTE_PARTNER", "1")
    lv = ops.mbgd_step(p_gpu, dev(cubes), dev(onehot), 0, B, z, 0.04)
    torch.cuda.synchronize()
    monkeypatch.delenv("CCNN_FB_TEST_MUTE_PARTNER")
    assert torch.isnan(lv).all()
    d = orc.dims(z)
    out = p_gpu.cpu().numpy()
    assert np.isnan(out[d.off_w1:d.off_b1]).all() and np.isnan(out[d.off_w2:]).all()
    # and the next step on the same workspace-less path is sound again
    p2 = dev(params)
    lv2 = ops.mbgd_step(p2, dev(cubes), dev(onehot), 0, B, z, 0.04)
    assert torch.isfinite(lv2).all() and torch.isfinite(p2).all()


def test_forward_pass_split_over_a_pair_agrees_with_the_unsplit_form(monkeypatch):
    """CCNN_FB_FSPLIT=0 keeps the whole forward pass in both workgroups of a sample (the form for
    launches that do not fit the chip at once); the split form adds the two halves of the hidden
    layer's sums in another order and must agree to rounding."""
    z, B = 103, 96
    params, cubes, onehot = make_problem(z, B, 43, np.float32)
    c_gpu, o_gpu = dev(cubes), dev(onehot)
    # (the switch is read once per process: the unsplit form runs in a child process)
    import subprocess, sys, tempfile, os as _os
    p_gpu = dev(params)
    lv = ops.mbgd_step(p_gpu, c_gpu, o_gpu, 0, B, z, 0.04)
    with tempfile.TemporaryDirectory() as td:
        np.savez(_os.path.join(td, "in.npz"), params=params, cubes=cubes, onehot=onehot)
        code = ("import sys, numpy as np, torch; sys.path.insert(0, %r); from gcn_amd import ops; d = np.load(%r);"
                "p = torch.from_numpy(d['params']).cuda(); lv = ops.mbgd_step(p, torch.from_numpy(d['cubes']).cuda(),"
                "torch.from_numpy(d['onehot']).cuda(), 0, %d, %d, 0.04); np.savez(%r, p=p.cpu().numpy(), lv=lv.cpu().numpy())"
                % (_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), _os.path.join(td, "in.npz"), B, z, _os.path.join(td, "out.npz")))
        env = dict(_os.environ, CCNN_FB_FSPLIT="0")
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=240)
        out = np.load(_os.path.join(td, "out.npz"))
    parity.assert_groups(p_gpu.cpu().numpy(), out["p"], orc.dims(z), 1e-5, "split against unsplit forward pass")
    assert relerr(lv.cpu().numpy(), out["lv"]) < 1e-5


def test_adagrad_update_matches_oracle():
    rng = np.random.default_rng(1)
    n = 5000
    p, r, g = rng.normal(size=n), np.abs(rng.normal(size=n)), rng.normal(size=n)
    pg, rg = dev(p), dev(r)
    for it in (1, 2, 3):
        orc.adagrad_update_f64(p, r, g, 0.2, it)
        ops.update_adagrad(pg, rg, dev(g), 0.2, it)
    assert relerr(pg.cpu().numpy(), p) < 1e-14 and relerr(rg.cpu().numpy(), r) < 1e-14


# --------------------------------------------------------------------------- SGD
def test_sgd_golden(golden):
    g = golden("train_small.npz")
    z = int(g["z"])
    params = dev(g["sgd_p0"])
    loss = torch.zeros(1, dtype=torch.float64, device="cuda")
    ops.sgd(params, dev(g["cubes"]), dev(g["onehot"]), 0, 3, z, 0.008, loss)
    assert relerr(params.cpu().numpy(), g["sgd_p3"]) < 1e-12
    assert abs(loss.item() - float(g["sgd_loss"])) < 1e-12 * abs(float(g["sgd_loss"]))


@pytest.mark.parametrize("z,n", [(176, 40), (103, 25), (21, 6), (448, 5)])
def test_sgd_matches_oracle_across_calls(z, n):
    params, cubes, onehot = make_problem(z, n, 31 + z, np.float64)
    d = orc.dims(z)
    p_ref = params.copy()
    l_ref = orc.sgd_f64(p_ref, cubes, onehot, 0, n, z, 0.008)
    p_gpu = dev(params)
    loss = torch.full((1,), 123.0, dtype=torch.float64, device="cuda")   # sample 0 must reset it
    half = n // 2
    ws = ops.Workspace()
    ops.sgd(p_gpu, dev(cubes), dev(onehot), 0, half, z, 0.008, loss, workspace=ws)
    ops.sgd(p_gpu, dev(cubes), dev(onehot), half, n - half, z, 0.008, loss, workspace=ws)
    out = p_gpu.cpu().numpy()
    for lo, hi in ((0, d.off_b0), (d.off_b0, d.off_w1), (d.off_w1, d.off_b1), (d.off_b1, d.off_w2),
                   (d.off_w2, d.off_b2), (d.off_b2, d.nparams)):
        assert relerr(out[lo:hi], p_ref[lo:hi]) < 1e-12, (lo, hi)
    assert abs(loss.item() - l_ref) < 1e-12 * abs(l_ref)


# --------------------------------------------------------------------------- argument checks
def test_bad_arguments_fail_loudly():
    from gcn_amd.lib import CcnnError
    params, cubes, onehot = make_problem(103, 4, 1, np.float32)
    with pytest.raises((ValueError, TypeError)):
        ops.mbgd_grad(dev(params), dev(cubes), dev(onehot), 2, 4, 103)          # batch runs past the data
    with pytest.raises((ValueError, TypeError)):
        ops.forward(dev(params.astype(np.float64)), dev(cubes), 103)            # dtype mismatch
    with pytest.raises(CcnnError):
        ops.forward(dev(params), dev(cubes), 5)                                 # too few bands
