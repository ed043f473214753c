"""Times ccnn_mbgd_step for one library build (diagnostic; pick the build with CCNN_LIB_PATH).
usage: phase_probe.py [z] [B] [f32|f64]"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gcn_amd import ops, prep, lib as L
z = int(sys.argv[1]) if len(sys.argv) > 1 else 103
B = int(sys.argv[2]) if len(sys.argv) > 2 else 128
d = L.dims(z)
DT = np.float64 if (len(sys.argv) > 3 and sys.argv[3] == 'f64') else np.float32
rng = np.random.default_rng(0)
cubes = torch.from_numpy(rng.random((B * 8, 9 * z)).astype(DT)).cuda()
onehot = np.zeros((B * 8, 13), dtype=np.int32); onehot[np.arange(B * 8), rng.integers(0, 13, B * 8)] = 1
onehot = torch.from_numpy(onehot).cuda()
import ctypes; ctypes.CDLL(None).srand(1)
params = torch.from_numpy(prep.init_params(z, DT)).cuda()
ws = ops.Workspace(); lv = torch.zeros(B, dtype=torch.float64, device="cuda")
for i in range(50): ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize(); t0 = time.perf_counter(); n = 500
for i in range(n): ops.mbgd_step(params, cubes, onehot, (i % 8) * B, B, z, 0.0, loss_values=lv, workspace=ws)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(os.environ.get("CCNN_LIB_PATH", "default"), f"z={z} B={B}: {dt * 1e6:.1f} us/step  {B / dt / 1e6:.2f} M pixels/s")
if os.environ.get("CCNN_FB_TIMING"):
    import ctypes as C
    lib = C.CDLL(os.environ["CCNN_LIB_PATH"])
    t = (C.c_ulonglong * 32)()
    torch.cuda.synchronize()
    assert lib.ccnn_debug_fb_ticks(t) == 0
    names = ["stage", "conv+pool", "fc1", "a1", "out layer", "softmax", "delta1", "dpool", "conv_bwd", "bias+tail"]
    print("  ".join(f"{names[k]}={(t[k + 1] - t[k]) * 10}ns" for k in range(10)), f"total={(t[10] - t[0]) * 10}ns")
    if t[11]:
        print("inside stage: loads issued +%dns, LDS zeroed +%dns, labels stored (first wait) +%dns, cube staged +%dns, barrier +%dns, filters scattered + barrier +%dns"
              % tuple((t[k] - t[0]) * 10 for k in (11, 12, 13, 14, 15, 1)))
    if hasattr(lib, "ccnn_debug_fb_wg"):
        w = (C.c_ulonglong * (1024 * 9))()
        assert lib.ccnn_debug_fb_wg(w) == 0
        a = np.array(w[:], dtype=np.int64).reshape(1024, 9)
        grid = 16 * ((B + 7) // 8) if L.load().ccnn_mbgd_path(B, z, np.dtype(DT).itemsize) == 0 and B <= 128 else B
        a = a[:min(grid, 1024)]
        a = a[a[:, 0] > 0]
        t0 = a[:, 0].min()
        start = (a[:, 0] - t0) / 100.0
        end = (a[:, 1:].max(axis=1) - t0) / 100.0
        dur = end - start
        wave_end = (a[:, 1:] - a[:, :1]) / 100.0
        print(f"  {len(a)} workgroups: start median +{np.median(start):.2f} max +{start.max():.2f} us; own duration median {np.median(dur):.2f} "
              f"min {dur.min():.2f} max {dur.max():.2f} us; last end +{end.max():.2f} us; waves' ends after their start (median over workgroups): "
              + " ".join(f"{np.median(wave_end[:, k]):.2f}" for k in range(8)))
    if hasattr(lib, "ccnn_debug_gr_wg"):
        w2 = (C.c_ulonglong * (1024 * 3))()
        assert lib.ccnn_debug_gr_wg(w2) == 0
        r = np.array(w2[:], dtype=np.int64).reshape(1024, 3)
        r = r[r[:, 0] > 0]
        fb_t0, fb_end = a[:, 0].min(), a[:, 1:].max()
        st, sm, en = (r[:, 0] - fb_t0) / 100.0, (r[:, 1] - fb_t0) / 100.0, (r[:, 2] - fb_t0) / 100.0
        print(f"  grad_reduce, {len(r)} workgroups, us after the first start of sample_fwd_bwd (whose last wave ended at +{(fb_end - fb_t0) / 100.0:.2f}): "
              f"first start +{st.min():.2f}, last start +{st.max():.2f}; sums done median +{np.median(sm):.2f} last +{sm.max():.2f}; "
              f"last end +{en.max():.2f}; per workgroup: sums {np.median(sm - st):.2f} (max {(sm - st).max():.2f}), update {np.median(en - sm):.2f} (max {(en - sm).max():.2f})")
        nb_w1 = 7 * ((d.pout + 15) // 16 + 1)
        nb_w0 = (20 * 171 + 20 + 63) // 64
        w3 = np.array(w2[:], dtype=np.int64).reshape(1024, 3)
        for name, lo, hi in (("dW1 and dW2 tiles", 0, nb_w1), ("conv-gradient column sums", nb_w1, nb_w1 + nb_w0), ("db1, db2", nb_w1 + nb_w0, nb_w1 + nb_w0 + 3)):
            k = w3[lo:hi]
            print(f"    {name} ({hi - lo} workgroups): start +{(k[:, 0].min() - fb_t0) / 100.0:.2f}..+{(k[:, 0].max() - fb_t0) / 100.0:.2f}, sums {np.median(k[:, 1] - k[:, 0]) / 100.0:.2f} "
                  f"(max {(k[:, 1] - k[:, 0]).max() / 100.0:.2f}), update {np.median(k[:, 2] - k[:, 1]) / 100.0:.2f} (max {(k[:, 2] - k[:, 1]).max() / 100.0:.2f}), last end +{(k[:, 2].max() - fb_t0) / 100.0:.2f}")
