"""Inference throughput on synthetic cubes (diagnostic / profiling driver).
usage: bench_infer.py [z] [n] [f32|f64]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import ops, prep
import ctypes
z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
n = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
dt = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == "f64") else np.float32
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((n, 9 * z)).astype(dt)).cuda()
ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, dt, 0)).cuda()
ws = ops.Workspace()
for _ in range(3): ops.forward(params, cubes, z, want_probs=False, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter(); reps = 20
for _ in range(reps): ops.forward(params, cubes, z, want_probs=False, workspace=ws)
torch.cuda.synchronize(); t = (time.perf_counter() - t0) / reps
print(f"z={z} n={n} {np.dtype(dt).name}: {t*1e3:.3f} ms  {n/t/1e6:.1f} M pixels/s")
if os.environ.get("CCNN_CONV_CLOCK"):
    import ctypes as C
    lib = C.CDLL(os.environ["CCNN_LIB_PATH"])
    t = (C.c_ulonglong * 4)()
    for _ in range(200): ops.forward(params, cubes, z, want_probs=False, workspace=ws)   # let the clock settle under load
    torch.cuda.synchronize()
    assert lib.ccnn_debug_conv_clock(t) == 0
    print(f"sustained back-to-back launches: in-kernel clock {(t[2] - t[0]) / ((t[3] - t[1]) * 10):.3f} GHz, loop {(t[3] - t[1]) / 100:.1f} us")
    from gcn_amd import lib as L
    L.profile_enable(True)
    for _ in range(20): ops.forward(params, cubes, z, want_probs=False, workspace=ws)
    torch.cuda.synchronize()
    L.profile_enable(False)
    print({k: round(ms / cnt * 1e3, 1) for k, (ms, cnt) in L.profile_report().items()}, "us per launch (dispatch events)")
    assert lib.ccnn_debug_conv_clock(t) == 0
    cyc, ticks = t[2] - t[0], t[3] - t[1]
    print(f"conv loop of one wave: {cyc} shader cycles in {ticks * 10} ns -> {cyc / (ticks * 10):.3f} GHz")
    w = (C.c_ulonglong * (3 * 2048))()
    assert lib.ccnn_debug_conv_wg(w) == 0
    a = np.array(w[:], dtype=np.int64).reshape(2048, 3)
    a = a[a[:, 0] > 0]
    t0 = a[:, 0].min()
    us = (a - t0) / 100.0
    print(f"{len(a)} workgroups; entry: min {us[:,0].min():.1f} median {np.median(us[:,0]):.1f} max {us[:,0].max():.1f} us; "
          f"loop start: median {np.median(us[:,1]):.1f} max {us[:,1].max():.1f}; loop end: min {us[:,2].min():.1f} "
          f"median {np.median(us[:,2]):.1f} max {us[:,2].max():.1f}")
    late = us[us[:, 0] > 20]
    print(f"workgroups entering later than 20 us: {len(late)}; their loop time median {np.median(late[:,2]-late[:,1]) if len(late) else 0:.1f} us; "
          f"early ones' loop time median {np.median((us[:,2]-us[:,1])[us[:,0] <= 20]):.1f} us")
    w = (C.c_ulonglong * (4 * 4096))()
    assert lib.ccnn_debug_fc_wg(w) == 0
    a = np.array(w[:], dtype=np.int64).reshape(4096, 4)
    a = a[a[:, 0] > 0]
    us = (a - a[:, 0].min()) / 100.0
    d = np.diff(us, axis=1)
    print(f"fc_head: {len(a)} workgroups; entry median {np.median(us[:,0]):.1f} max {us[:,0].max():.1f}; exit max {us[:,3].max():.1f} us; "
          f"per workgroup: prologue {np.median(d[:,0]):.1f}, K loop {np.median(d[:,1]):.1f}, epilogue {np.median(d[:,2]):.1f} us (medians)")
    order = np.sort(us[:, 0])
    print("entry times (us) at 10 % steps:", [round(float(order[int(q * (len(order) - 1))]), 1) for q in np.linspace(0, 1, 11)])
