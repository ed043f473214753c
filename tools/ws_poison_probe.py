"""Every training entry point with its workspace pre-filled with 0xFF bytes (NaN as floats): results
must equal those from a zeroed workspace bit for bit - a difference is a read of workspace memory
that the call itself never wrote."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gcn_amd import lib as L, ops
bad = 0
import ctypes
_dbg = ctypes.CDLL(os.environ["CCNN_LIB_PATH"]) if os.environ.get("LDS_POISON") else None


def poison_lds():
    if _dbg is not None:
        assert _dbg.ccnn_debug_poison_lds(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)) == 0


for dtype in (np.float64, np.float32):
    for z in (103, 200):
        d = L.dims(z)
        for B in (1, 7, 9, 12, 16, 32, 100, 128, 129, 300, 700, 800, 2048):
            for path in ("", "batched"):
                if path:
                    os.environ["CCNN_MBGD_PATH"] = path
                else:
                    os.environ.pop("CCNN_MBGD_PATH", None)
                rng = np.random.default_rng(B)
                p0 = torch.from_numpy(((rng.random(d.nparams) - 0.5) / 10).astype(dtype)).cuda()
                cubes = torch.from_numpy(rng.random((B, d.cube)).astype(dtype)).cuda()
                oh = np.zeros((B, 13), dtype=np.int32); oh[np.arange(B), rng.integers(0, 13, B)] = 1
                oh = torch.from_numpy(oh).cuda()
                outs = []
                for fill in (0, 0xFF):
                    ws = ops.Workspace()
                    nbytes = L.load().ccnn_mbgd_workspace_bytes(B, z, np.dtype(dtype).itemsize)
                    ws.get(nbytes).fill_(fill)
                    probs = torch.empty((B, 13), dtype=p0.dtype, device="cuda")
                    if fill:
                        poison_lds()
                    g, lv = ops.mbgd_grad(p0, cubes, oh, 0, B, z, probs=probs, workspace=ws)
                    ws.get(nbytes).fill_(fill)
                    if fill:
                        poison_lds()
                    p = p0.clone()
                    lv2 = ops.mbgd_step(p, cubes, oh, 0, B, z, 0.01, workspace=ws)
                    torch.cuda.synchronize()
                    outs.append((g.clone(), lv.clone(), probs.clone(), p, lv2.clone()))
                same = all(torch.equal(a, b) for a, b in zip(*outs))
                finite = all(bool(torch.isfinite(t).all()) for t in outs[1])
                if not same or not finite:
                    bad += 1
                    print(f"{np.dtype(dtype).name} z={z} B={B} path={path or 'auto'}: same={same} finite={finite}", flush=True)
print("workspace poison probe:", "clean" if bad == 0 else f"{bad} cases depend on workspace contents")
