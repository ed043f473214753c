"""CPU: the product's host-side sample preparation (ccnn_host_*) against the oracle."""
import numpy as np
import pytest

import oracle as orc
import synth
from gcn_amd import lib as L
from gcn_amd import prep


@pytest.mark.parametrize("x,y,z,seed", [(7, 6, 24, 11), (16, 12, 30, 12), (2, 2, 21, 1), (3, 9, 21, 2), (33, 2, 21, 5)])
def test_scan_split_index_match_oracle(x, y, z, seed):
    _, labels = synth.small_scene(x, y, z, seed)
    di_o = orc.scan_labels(labels, x, y)
    di = prep.scan_labels(labels, x, y)
    assert np.array_equal(di, di_o)
    assert prep.split_sizes(len(di)) == orc.split_sizes(len(di))
    a = prep.build_index(labels, di, x, y)
    b = orc.build_index(labels, di_o, x, y)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_patchy_scene_counts():
    x, y, z, nl = synth.SCENES["indian_pines"]
    labels = synth.make_labels(x, y, nl, np.random.default_rng(20243))
    di = prep.scan_labels(labels, x, y)
    assert len(di) == nl
    tr, te = prep.split_sizes(nl)
    assert (tr, te) == (8199, 2050)
    a = prep.build_index(labels, di, x, y)
    b = orc.build_index(labels, di, x, y)
    for u, v in zip(a, b):
        assert np.array_equal(u, v)


def test_bad_arguments_are_rejected():
    from gcn_amd.lib import CcnnError
    labels = np.ones(4)
    with pytest.raises(CcnnError):
        prep.build_index(labels, np.arange(4, dtype=np.int32), 1, 4)   # x < 2: reference reads outside the scene
    labels = np.full(16, 14.0)                                          # label beyond the 13 classes
    with pytest.raises(CcnnError):
        prep.build_index(labels, np.arange(16, dtype=np.int32), 4, 4)


def test_shuffle_and_init_share_the_rand_stream():
    x, y, z, seed = 16, 12, 48, 12
    _, labels = synth.small_scene(x, y, z, seed)
    di = prep.scan_labels(labels, x, y)
    d = orc.dims(z)
    for dtype, draws in ((np.float64, d.pout), (np.float64, d.pout * 100), (np.float32, 0)):
        tr9, _, oh, _ = prep.build_index(labels, di, x, y)
        tr9o, oho = tr9.copy(), oh.copy()
        prep.shuffle(tr9, oh, 4242)
        p = prep.init_params(z, dtype, draws)
        orc.shuffle(tr9o, oho, 4242)
        po = orc.init_params_f64(z, draws) if dtype == np.float64 else orc.init_params_f32(z)
        assert np.array_equal(tr9, tr9o) and np.array_equal(oh, oho)
        assert np.array_equal(p, po)
        assert not np.array_equal(tr9[:, 4], np.sort(tr9[:, 4]))       # it did shuffle
        assert (oh.sum(axis=1) == 1).all()
    if True:
        # fp32: weights within 1/10, biases exactly zero; fp64: within 1/50
        assert np.abs(p).max() <= 0.1 and np.all(p[d.off_b1:d.off_w2] == 0) and np.all(p[d.off_b2:] == 0)


def test_lr_schedule_matches_oracle():
    rng = np.random.default_rng(0)
    losses = np.abs(np.cumsum(rng.normal(-0.01, 0.02, 200))) + 0.01
    for use_float, init, lr0 in ((False, 1.0, 0.5), (False, 1.0, 0.008), (True, 100.0, 0.04)):
        a = prep.LrSchedule(init, use_float)
        b = orc.Sched(np.float32 if use_float else np.float64, init)
        lra = lrb = np.float32(lr0).item() if use_float else lr0
        for l in losses:
            da, lra = a.step(l, lra)
            db, lrb = b.step(l, lrb)
            assert da == db and lra == lrb
        assert lra < lr0


def test_partition_matches_reference_slices(golden):
    g = golden("partition.npz")
    T, B = int(g["T"]), int(g["B"])
    for n in (2, 4):
        for i0 in range(T // B):
            for r in range(n):
                first, sub = prep.partition(T, B, n, r, i0)
                assert first == g[f"first_n{n}"][i0, r] and sub == B // n
    # every rank's batches stay inside its own slice and never overlap another rank's
    T, B, n = 34220, 100, 8
    for r in range(n):
        first, sub = prep.partition(T, B, n, r, T // B - 1)
        assert sub == 12 and r * (T // n) <= first and first + sub <= (r + 1) * (T // n)


@pytest.mark.parametrize("classes", [16, 9])
def test_other_class_counts_match_the_oracle_rebuilt_for_them(classes):
    """Run-time class count (CCNN_GEOMETRY, ccnn_host_*_classes) against the oracle compiled with the
    same NEU_NUM2 (a compile-time constant of the reference, gcn_sgd_fp64.cu:18)."""
    z = 103
    with orc.use_classes(classes):
        d_ref = orc.dims(z)
        g = L.geometry(z, classes)
        d = L.dims(g)
        assert (d.z, d.n_classes, d.nparams, d.off_w2, d.off_b2) == (z, classes, d_ref.nparams, d_ref.off_w2, d_ref.off_b2)
        x, y = 21, 17
        rng = np.random.default_rng(3)
        labels = rng.integers(0, classes + 1, x * y).astype(np.float64)
        di = prep.scan_labels(labels, x, y)
        got = prep.build_index(labels, di, x, y, classes)
        ref = orc.build_index(labels, orc.scan_labels(labels, x, y), x, y)
        for a, b in zip(got, ref):
            assert np.array_equal(a, b)
        prep.shuffle(got[0], got[2], 11)
        p32 = prep.init_params(g, np.float32)
        orc.shuffle(ref[0], ref[2], 11)
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2])
        assert np.array_equal(p32, orc.init_params_f32(z))
        orc.srand(5)
        p64 = prep.init_params(g, np.float64, 7)
        orc.srand(5)
        assert np.array_equal(p64, orc.init_params_f64(z, 7))
    # 13 classes reject a label of 14; the 16-class build takes it
    labels[0] = 14
    di = prep.scan_labels(labels, x, y)
    if classes == 16:
        with pytest.raises(L.CcnnError):
            prep.build_index(labels, di, x, y)
        prep.build_index(labels, di, x, y, 16)
    with pytest.raises(L.CcnnError):
        L.dims(L.geometry(z, 17))
