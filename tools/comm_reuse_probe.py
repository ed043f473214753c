"""Two sets of exchange buffers in a row in every rank process (fp32, then fp64), both forms of the
in-step exchange at several batch sizes.  Per call: this rank's local gradient before and after the
data-parallel step against the CPU oracle, the step itself against the oracle gradients of all ranks,
and the inputs against their host copies.  This is the reproducer of what a freed exchange buffer does
to the owner's later allocations (with buffers freed at destroy: wrong fp64 gradients at B = 32, the
first workspace above 1 MB, in 8 of 12 runs; with the library's buffer pool: 0 of 10).
    torchrun --nproc-per-node 2 tools/comm_reuse_probe.py     (WS_EARLY=1: one workspace allocated first)
CCNN_COMM_POOL=free makes the library free its buffers at destroy (the failing configuration);
CCNN_COMM_ALLOC=plain allocates them as ordinary instead of uncached device memory.  On a bad local
gradient the probe runs the same gradient once more into the suspect workspace and once into a
workspace that was allocated before any exchange buffer existed, and reports where the two
workspaces differ: address ranges, the region (pool / a1 / d1 / d2 / probs / conv-gradient rows) and
whether the differing 8-byte words look like the exchange's tagged words (payload | sequence << 32)."""
import os, sys
import numpy as np, torch, torch.distributed as dist
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle as orc
from gcn_amd import lib as L, ops
from gcn_amd.comm import NodeComm
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
z = 103
d = L.dims(z)
nbad = 0
GOOD = ops.Workspace(); GOOD.get(96 << 20)     # allocated before any exchange buffer exists: never on recycled pages
WS = None
if os.environ.get("WS_EARLY"):       # one workspace for everything, allocated before any exchange buffer exists
    WS = ops.Workspace(); WS.get(64 << 20)
for dtype in (torch.float32, torch.float64):
    comm = NodeComm(d.nparams, dtype)
    npdt = np.float32 if dtype == torch.float32 else np.float64
    tol = 2e-5 if dtype == torch.float32 else 1e-11
    for it, (via, B) in enumerate(((0, 8), (1, 9), (1, 16), (1, 16), (0, 16), (1, 12), (1, 16), (1, 32), (0, 32), (1, 9), (0, 12))):
        comm.set_exchange(via)
        rng = np.random.default_rng(77)
        p0_h = ((rng.random(d.nparams) - 0.5) / 10).astype(npdt)
        rng = np.random.default_rng(78 + rank)
        cubes_h = rng.random((B, d.cube)).astype(npdt)
        oh_h = np.zeros((B, 13), dtype=np.int32); oh_h[np.arange(B), rng.integers(0, 13, B)] = 1
        _, _, g_ref = orc.mbgd_step(p0_h.copy(), cubes_h, oh_h, 0, B, z, npdt(0.0))
        p0, cubes, oh = torch.from_numpy(p0_h).cuda(), torch.from_numpy(cubes_h).cuda(), torch.from_numpy(oh_h).cuda()
        def local_err():
            g, _ = ops.mbgd_grad(p0, cubes, oh, 0, B, z, workspace=WS)      # WS None: a fresh block per call, dropped after it
            torch.cuda.synchronize()
            return float(np.abs(g.cpu().numpy() - g_ref).max() / np.abs(g_ref).max())
        e_a = local_err()
        p_dp = p0.clone()
        dist.barrier()
        ops.mbgd_step(p_dp, cubes, oh, 0, B, z, 0.04, comm=comm, divisor=B * world, workspace=WS)
        torch.cuda.synchronize()
        e_b = local_err()
        # expected in-step result from the oracle gradients of all ranks
        gl = [torch.empty(d.nparams, dtype=dtype) for _ in range(world)]
        dist.all_gather(gl, torch.from_numpy(g_ref))
        gs = gl[0].clone()
        for t in gl[1:]:
            gs += t
        want = p0_h - (npdt(0.04) * gs.numpy() / (B * world) if dtype == torch.float64 else npdt(0.04) * gs.numpy())
        if dtype == torch.float32:
            want[d.off_b0:d.off_w1] = p0_h[d.off_b0:d.off_w1]
        e_dp = float(np.abs(p_dp.cpu().numpy() - want).max() / np.abs(want).max())
        ok_in = torch.equal(cubes.cpu(), torch.from_numpy(cubes_h)) and torch.equal(p0.cpu(), torch.from_numpy(p0_h)) and torch.equal(oh.cpu(), torch.from_numpy(oh_h))
        if max(e_a, e_b, e_dp) > tol or not ok_in or comm.status():
            nbad += 1
            e_c = local_err()
            if e_a > tol or e_b > tol:
                # the same gradient into a fresh (suspect) workspace and into the early one: diff the rows the kernels leave behind
                es = 4 if dtype == torch.float32 else 8
                nb = L.load().ccnn_mbgd_workspace_bytes(B, z, es)
                # (the allocator hands the block that the failed call has just dropped to the next request of its size)
                sus = ops.Workspace(); sus.get(nb); sus.buf.fill_(0x5a)
                GOOD.buf[:nb].fill_(0x5a)
                gs_, _ = ops.mbgd_grad(p0, cubes, oh, 0, B, z, workspace=sus)
                gg_, _ = ops.mbgd_grad(p0, cubes, oh, 0, B, z, workspace=GOOD)
                torch.cuda.synchronize()
                a = sus.buf[:nb].cpu().numpy().view(np.uint64); b = GOOD.buf[:nb].cpu().numpy().view(np.uint64)
                diff = np.flatnonzero(a != b)
                al = lambda v: (v + 255) // 256 * 256
                offs, o = {}, 0
                for nm, cnt in (("pool", B * d.pout), ("a1", B * 100), ("d1", B * 100), ("d2", B * 13), ("probs", B * 13), ("slab", B * 3440)):
                    offs[nm] = (o, o + cnt * es); o = al(o + cnt * es)
                where = {nm: int(((diff * 8 >= lo) & (diff * 8 < hi)).sum()) for nm, (lo, hi) in offs.items()}
                eg = float(np.abs(gs_.cpu().numpy() - g_ref).max() / np.abs(g_ref).max()); eo = float(np.abs(gg_.cpu().numpy() - g_ref).max() / np.abs(g_ref).max())
                print(f"[{rank}]   suspect workspace {sus.buf.data_ptr():#x}+{nb:#x} grad err {eg:.1e}; early workspace {GOOD.buf.data_ptr():#x} grad err {eo:.1e}; "
                      f"{len(diff)} of {len(a)} 8-byte words differ, by region {where}", flush=True)
                for i in diff[:6]:
                    print(f"[{rank}]     word {int(i) * 8:#x}: suspect {int(a[i]):#018x} (as f64 {a[i:i+1].view(np.float64)[0]:.6g}) early {int(b[i]):#018x} ({b[i:i+1].view(np.float64)[0]:.6g})", flush=True)
            print(f"[{rank}] {dtype} it {it} via={via} B={B}: local before {e_a:.1e}, after {e_b:.1e}, again {e_c:.1e}; in-step {e_dp:.1e}; inputs intact {ok_in}; status {comm.status()}", flush=True)
        dist.barrier()
    comm.close()
print(f"[{rank}] done: {nbad} bad iterations", flush=True)
dist.destroy_process_group()
