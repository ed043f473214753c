"""GPU: the entry point that allocates nothing (ccnn_mbgd_step_noalloc_f32, its per-sample rows
in a fixed device array of the library) against ccnn_mbgd_step_f32 with a caller-owned
workspace, through the C ABI: two calls in a row, the second with a smaller batch, must leave
bit-identical parameters and losses; a batch beyond the fixed array is rejected."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle as orc
from gcn_amd import lib as L

pytestmark = pytest.mark.gpu


def _p(t):
    return C.c_void_p(t.data_ptr())


# (128, 37): the fused per-sample kernel, two workgroups per sample that trade tagged words through the
# array, one scratch row per sample - rows 37..127 are stale in the second call.  (400, 140): the same
# kernel with one workgroup per sample, two per compute unit, then one per compute unit.  (1200, 600):
# the sample-tile kernel out of the same array - the conv-gradient region then holds one partial row
# per TILE (150, then 75) and sits at another offset for each batch size, so the second call reads
# rows the first wrote at other places; every row a call reads it has written itself.
@pytest.mark.parametrize("calls", [((0, 128), (128, 37)), ((0, 400), (400, 140)), ((0, 1200), (1200, 600))])
def test_noalloc_step_equals_workspace_step_twice_second_smaller(calls):
    lib = L.load()
    z = 103
    d = L.dims(z)
    rng = np.random.default_rng(5)
    n = calls[1][0] + calls[1][1]
    bmax = calls[0][1]
    assert lib.ccnn_mbgd_path(bmax, z, 4) == (0 if bmax <= 512 else 1)
    cubes = torch.from_numpy(rng.random((n, 9 * z)).astype(np.float32)).cuda()
    onehot_h = np.zeros((n, 13), dtype=np.int32)
    onehot_h[np.arange(n), rng.integers(0, 13, n)] = 1
    onehot = torch.from_numpy(onehot_h).cuda()
    orc.srand(5)
    p0 = orc.init_params_f32(z)
    p_ws, p_na = torch.from_numpy(p0).cuda(), torch.from_numpy(p0).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # inputs and outputs exist before the calls under test; from here on nothing is allocated for the no-alloc calls
    l_ws = torch.zeros(bmax, dtype=torch.float64, device="cuda")
    l_na = torch.zeros(bmax, dtype=torch.float64, device="cuda")
    ws_bytes = max(lib.ccnn_mbgd_workspace_bytes(B, z, 4) for _, B in calls)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
    lr = 0.04 * min(1.0, 128.0 / bmax)
    torch.cuda.synchronize()
    for first, B in calls:                          # the second call is smaller: it finds the first call's rows in the scratch
        mem_before = torch.cuda.memory_allocated()
        L.check(lib.ccnn_mbgd_step_noalloc_f32(_p(p_na), _p(cubes), _p(onehot), first, B, z, lr, _p(l_na), st),
                "ccnn_mbgd_step_noalloc_f32")
        assert torch.cuda.memory_allocated() == mem_before
        L.check(lib.ccnn_mbgd_step_f32(_p(p_ws), _p(cubes), _p(onehot), first, B, z, lr, _p(l_ws), _p(ws),
                                       ws.numel(), st), "ccnn_mbgd_step_f32")
        torch.cuda.synchronize()
        assert torch.equal(p_na, p_ws), (first, B)
        assert torch.equal(l_na[:B], l_ws[:B]), (first, B)
    assert not torch.equal(p_na, torch.from_numpy(p0).cuda())      # the steps did move the weights
    # and the pair agrees with the oracle to the fp32 bar
    p_ref = p0.copy()
    cubes_h = cubes.cpu().numpy()
    for first, B in calls:
        orc.mbgd_step(p_ref, cubes_h, onehot_h, first, B, z, np.float32(lr), want_grad=False)
    out = p_na.cpu().numpy()
    for lo, hi in ((0, d.off_b0), (d.off_w1, d.off_b1), (d.off_b1, d.off_w2), (d.off_w2, d.off_b2),
                   (d.off_b2, d.nparams)):
        assert np.abs(out[lo:hi] - p_ref[lo:hi]).max() <= 1e-5 * np.abs(p_ref[lo:hi]).max(), (lo, hi)


def test_noalloc_step_under_stream_capture():
    """The caller this entry point is for: a step captured into a HIP graph (no allocation may
    happen during capture) after one warm-up call, replayed twice, against two plain calls."""
    lib = L.load()
    z, B = 103, 64
    rng = np.random.default_rng(8)
    cubes = torch.from_numpy(rng.random((B, 9 * z)).astype(np.float32)).cuda()
    onehot_h = np.zeros((B, 13), dtype=np.int32)
    onehot_h[np.arange(B), rng.integers(0, 13, B)] = 1
    onehot = torch.from_numpy(onehot_h).cuda()
    orc.srand(8)
    p0 = torch.from_numpy(orc.init_params_f32(z)).cuda()
    p_graph, p_plain = p0.clone(), p0.clone()
    l_graph = torch.zeros(B, dtype=torch.float64, device="cuda")
    l_plain = torch.zeros(B, dtype=torch.float64, device="cuda")
    scratch_p = p0.clone()
    st0 = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    # warm-up on the same device: resolves the scratch array's address and the kernels' attributes
    L.check(lib.ccnn_mbgd_step_noalloc_f32(_p(scratch_p), _p(cubes), _p(onehot), 0, B, z, 0.04, _p(l_plain), st0), "warm-up")
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        rc = lib.ccnn_mbgd_step_noalloc_f32(_p(p_graph), _p(cubes), _p(onehot), 0, B, z, 0.04, _p(l_graph), st)
    assert rc == 0, L.error_string(rc)
    torch.cuda.synchronize()
    assert torch.equal(p_graph, p0), "capture must not execute the step"
    for _ in range(2):
        g.replay()
        L.check(lib.ccnn_mbgd_step_noalloc_f32(_p(p_plain), _p(cubes), _p(onehot), 0, B, z, 0.04, _p(l_plain), st0), "plain")
    torch.cuda.synchronize()
    assert not torch.equal(p_graph, p0)
    assert torch.equal(p_graph, p_plain) and torch.equal(l_graph, l_plain)


def test_noalloc_step_rejects_batches_beyond_the_fixed_array():
    lib = L.load()
    z = 448
    B = 400                                          # 8 MiB holds 359 samples' rows at z = 448
    assert lib.ccnn_mbgd_workspace_bytes(B, z, 4) > (8 << 20)
    d = L.dims(z)
    params = torch.zeros(d.nparams, dtype=torch.float32, device="cuda")
    cubes = torch.zeros((B, 9 * z), dtype=torch.float32, device="cuda")
    onehot = torch.zeros((B, 13), dtype=torch.int32, device="cuda")
    loss = torch.zeros(B, dtype=torch.float64, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rc = lib.ccnn_mbgd_step_noalloc_f32(_p(params), _p(cubes), _p(onehot), 0, B, z, 0.04, _p(loss), st)
    assert rc == -4 and "scratch" in L.error_string(rc)
    torch.cuda.synchronize()
    assert float(params.abs().max()) == 0.0          # nothing ran
