#!/usr/bin/env python3
"""Throughput of the Cube-CNN hot path on MI355X: fp32 minibatch training
(forward + backward + update) in pixels/s, plus inference pixels/s.

    python bench.py --gpus N --steps K --warmup W [--workload pavia_u|indian_pines|ksc] [--batch 128]

One process per GPU (the driver launches N>1 through torch.distributed.run);
data-parallel weak scaling: every rank owns 1/N of the training pixels and runs
`--batch` samples per step, gradients are all-reduced over RCCL, every rank applies
the identical update.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import synth  # noqa: E402  (seeded synthetic scenes)
from gcn_amd import lib as L  # noqa: E402
from gcn_amd import ops, prep  # noqa: E402

PEAK_FP32_MATRIX_TFLOPS = 157.3     # MI355X dense f32 MFMA
PEAK_FP64_MATRIX_TFLOPS = 78.6      # MI355X dense f64 MFMA
PEAK_HBM_GBS = 8000.0               # HBM3E spec
SEED_BASE = 20240
WORKLOAD_CONFIG = {"ksc": 2, "indian_pines": 3, "pavia_u": 4, "aviris": 5}


def kernel_work(d, z):
    """Algorithmic FLOPs (or bytes) per PIXEL of each kernel (update: per launch)."""
    conv = 2.0 * 171 * d.re * 20
    fc1 = 2.0 * d.pout * 100
    return {
        # conv fwd + conv bwd, FC1 fwd + delta, FC2 fwd + delta
        "sample_fwd_bwd": ("mfma", 2 * conv + 2 * fc1 + 4.0 * 1300),
        # the same chain for a tile of samples per workgroup (large batches)
        "tile_fwd_bwd": ("mfma", 2 * conv + 2 * fc1 + 4.0 * 1300),
        # dW1 = pool^T d1, dW2 = a1^T d2
        "grad_reduce": ("mfma", fc1 + 2.0 * 1300),
        "conv_pool_fwd": ("mfma", conv),
        "fc_head_fwd": ("mfma", fc1 + 2.0 * 1300),
        # the batched training path: conv fwd | FC1 + FC2 fwd + delta1 | delta at the pooled features | conv gradient
        "conv_pool_fwd_train": ("mfma", conv),
        "fc_head_fwd_train": ("mfma", fc1 + 4.0 * 1300),
        "dpool_bwd": ("mfma", fc1),
        "conv_bwd_batch": ("mfma", conv),
        # 9z values read + 9z written + 9 int32 indices
        "gather_tile": ("hbm", 2.0 * 9 * z * 4 + 36),
        # per launch: read params, read gradient, write params
        "update_f32_kernel": ("hbm", None),
        # per launch (filled in by the caller): the same plus world slots written and read
        "allreduce_apply": ("hbm", None),
    }


KERNEL_SOURCES = ("mbgd.hip", "tile.hip", "forward.hip", "ccnn_device.h")   # what the training kernels are built from


def kernel_source_digest():
    """sha256 over the sources of the training kernels: identifies the build a PMC pass profiled."""
    import hashlib
    h = hashlib.sha256()
    for name in KERNEL_SOURCES:
        with open(os.path.join(ROOT, "gcn_amd", "csrc", name), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def committed_traffic(workload, batch, kernel, dtype="f32"):
    """(HBM bytes per launch of `kernel`, where the figure comes from).  The bytes are those of the
    committed PMC passes (rocprofv3 --pmc cannot run inside this process): None when no pass was
    recorded for this workload and batch, and None - with the reason - when the kernels' sources
    have changed since the pass (tools/make_traffic.py stamps their digest into the file)."""
    try:
        with open(os.path.join(ROOT, "profiles", "traffic.json")) as f:
            doc = json.load(f)
        for rec in doc["records"]:
            if (rec["workload"], rec["batch"], rec["kernel"], rec.get("dtype", "f32")) == (workload, batch, kernel, dtype):
                where = f"committed {doc.get('summaries', 'profiles/')} (profiles/traffic.json), kernels @{doc.get('kernel_source_digest')}"
                try:
                    now = kernel_source_digest()
                except OSError:
                    now = None
                if doc.get("kernel_source_digest") != now:
                    return None, f"stale: {where}, sources now @{now}"
                return rec["hbm_bytes_per_launch"], where
    except (OSError, KeyError, ValueError):
        pass
    return None, "no PMC pass recorded for this workload, batch and precision"


def kernel_rows(rep, d, z, pixels, passes=None, world=1, es=4):
    """Per-kernel table from the library's event timings: average duration, achieved rate on the
    kernel's algorithmic work and its fraction of the peak that bounds it (es: element size)."""
    work = kernel_work(d, z)
    peak_mfma = PEAK_FP32_MATRIX_TFLOPS if es == 4 else PEAK_FP64_MATRIX_TFLOPS
    rows = {}
    for name, (ms, cnt) in rep.items():
        if name not in work or cnt == 0 or (work[name][1] is not None and name not in pixels):
            continue
        bound, per_px = work[name]
        if per_px is None:
            factor = {"allreduce_apply": 3.0 + 2.0 * world}.get(name, 3.0)
            units = factor * d.nparams * es * cnt
        else:
            # a pass over n pixels may be several launches (the library works through its workspace in chunks)
            units = per_px * (es / 4.0 if bound == "hbm" else 1.0) * pixels[name] * (passes.get(name, cnt) if passes else cnt)
        tot_s = ms * 1e-3
        if bound == "mfma":
            ach, peak, unit = units / tot_s / 1e12, peak_mfma, "TFLOP/s"
        else:
            ach, peak, unit = units / tot_s / 1e9, PEAK_HBM_GBS, "GB/s"
        rows[name] = {"avg_us": round(ms / cnt * 1e3, 3), "launches": cnt, "bound": bound,
                      "achieved": round(ach, 4), "peak": peak, "unit": unit, "frac": round(ach / peak, 5)}
    return rows


TRAIN_KERNELS = ("sample_fwd_bwd", "tile_fwd_bwd", "grad_reduce", "update_f32_kernel", "allreduce_apply", "conv_pool_fwd_train",
                 "fc_head_fwd_train", "dpool_bwd", "conv_bwd_batch")
STEP_PATHS = {0: "fused per-sample kernel", 1: "sample-tile kernel", 2: "batched layer kernels"}   # ccnn_mbgd_path


def dominant(rows, workload, B, dtype="f32"):
    train_k = [k for k in TRAIN_KERNELS if k in rows]
    if not train_k:
        return None
    dom = max(train_k, key=lambda k: rows[k]["avg_us"])
    k = rows[dom]
    traffic, source = committed_traffic(workload, B, dom, dtype)
    return {"kernel": dom, "bound": k["bound"], "achieved": k["achieved"], "peak": k["peak"], "unit": k["unit"],
            "frac": k["frac"], "traffic": traffic, "traffic_source": source, "avg_us": k["avg_us"]}


def timed_steps(fn, n):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(n):
        fn(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n


def train_rate(params_h, cubes, onehot_d, z, B, steps, workload, profile=True):
    """MBGD steps of B samples over `cubes` (already on the device) in the precision of params_h:
    pixels/s, and the per-kernel table of the same steps.  fp32: the step size is scaled with 128/B
    (plain-sum update); fp64: the reference's 0.5 (mean over the batch)."""
    d = L.dims(z)
    n = cubes.shape[0]
    nb = n // B
    if nb < 1:
        return None
    es = params_h.dtype.itemsize
    p = torch.from_numpy(params_h).to(cubes.device)
    ws = ops.Workspace(cubes.device)
    lv = torch.zeros((nb, B), dtype=torch.float64, device=cubes.device)
    lr = 0.04 * min(1.0, 128.0 / B) if es == 4 else 0.5

    stepper = ops.MbgdStepper(p, cubes, onehot_d, B, z, loss_values=lv, workspace=ws)

    def step(i):
        b = i % nb
        stepper.step(b * B, lr, b)
    for i in range(min(steps, 30)):
        step(i)
    dt = timed_steps(step, steps)
    assert torch.isfinite(p).all(), "training diverged"
    out = {"batch": B, "train_pixels_per_s": round(B / dt, 1), "us_per_step": round(dt * 1e6, 2),
           "path": STEP_PATHS.get(L.load().ccnn_mbgd_path(B, z, es), "?")}
    if profile:
        L.profile_enable(True)
        psteps = min(steps, 100)
        for i in range(psteps):
            step(i)
        torch.cuda.synchronize()
        L.profile_enable(False)
        rows = kernel_rows(L.profile_report(), d, z, {k: B for k in TRAIN_KERNELS}, es=es)
        out["roofline"] = dominant(rows, workload, B, "f32" if es == 4 else "f64")
        out["kernels"] = {k: {"avg_us": v["avg_us"], "frac": v["frac"]} for k, v in rows.items()}
    return out


def gather_call_rate(scene_d, idx_d, x, y, z, out, reps=5):
    """Whole gather calls on the list, wall clock: with the list's order kept from one ordering
    pass (ccnn_gather_plan once, then ccnn_gather_planned per call - the headline figure: algorithmic
    bytes per second and the fraction of the HBM peak), and the one-shot ccnn_gather call that
    orders the list every time."""
    n = idx_d.shape[0]
    byts = n * (2.0 * 9 * z * 4 + 36)
    torch.cuda.synchronize()
    t = time.perf_counter()
    plan = ops.GatherPlan(idx_d, x, y)
    torch.cuda.synchronize()
    t_plan = time.perf_counter() - t
    plan.gather(scene_d, z, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        plan.gather(scene_d, z, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    gws = ops.Workspace(scene_d.device)
    ops.gather(scene_d, idx_d, x, y, z, out=out, workspace=gws)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.gather(scene_d, idx_d, x, y, z, out=out, workspace=gws)
    torch.cuda.synchronize()
    dt1 = (time.perf_counter() - t) / reps
    return {"us_per_call": round(dt * 1e6, 1), "pixels": n, "achieved": round(byts / dt / 1e9, 1), "peak": PEAK_HBM_GBS,
            "unit": "GB/s", "frac": round(byts / dt / 1e9 / PEAK_HBM_GBS, 4),
            "kernel": "gather_dense (tiles)" if plan.dense else "gather_tile (ordered list)",
            "what": "ccnn_gather_planned_f32, wall clock of the whole call (the list's order comes from one ccnn_gather_plan)",
            "plan_us_once": round(t_plan * 1e6, 1),
            "one_shot": {"us_per_call": round(dt1 * 1e6, 1), "frac": round(byts / dt1 / 1e9 / PEAK_HBM_GBS, 4),
                         "what": "ccnn_gather_f32: ordering pass (gather_keys + radix sort) + tile kernel on every call"}}


def gather_fully_labelled(scene_d, x, y, z, reps=5):
    """Every pixel of the scene labelled and gathered in shuffled order through one plan (the plan
    picks the tile kernel for such a list): wall clock of the whole call."""
    rng = np.random.default_rng(SEED_BASE + 99)
    labels = rng.integers(1, 14, x * y).astype(np.int32)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, _, _ = prep.build_index(labels, di, x, y)
    idx = np.concatenate([tr9.reshape(-1, 9), te9.reshape(-1, 9)])
    idx = idx[rng.permutation(len(idx))][: (4 << 30) // (9 * z * 4)]      # at most 4 GiB of cubes
    idx_d = torch.from_numpy(np.ascontiguousarray(idx)).to(scene_d.device)
    out = torch.empty((len(idx), 9 * z), dtype=torch.float32, device=scene_d.device)
    plan = ops.GatherPlan(idx_d, x, y)
    plan.gather(scene_d, z, out=out)
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        plan.gather(scene_d, z, out=out)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    byts = len(idx) * (2.0 * 9 * z * 4 + 36)
    return {"pixels": int(len(idx)), "kernel": "gather_dense (tiles)" if plan.dense else "gather_tile (ordered list)",
            "us_per_call": round(dt * 1e6, 1), "achieved": round(byts / dt / 1e9, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
            "frac": round(byts / dt / 1e9 / PEAK_HBM_GBS, 4),
            "what": "ccnn_gather_planned_f32 over every pixel of the scene, wall clock of the whole call"}


def workload_report(name, dev, full=True):
    """Training (fp32 MBGD, B = 128 and 2048), whole-call gather and full-scene inference on one
    scene shape, everything resident on the device; rank 0 only."""
    x, y, z, nl = synth.SCENES[name]
    d = L.dims(z)
    rng = np.random.default_rng(SEED_BASE + WORKLOAD_CONFIG[name])
    labels = synth.make_labels(x, y, nl, rng)
    if name == "aviris":
        # 939 M values: drawn on the device (3.8 GB in fp32), never on the host
        g = torch.Generator(device=dev).manual_seed(SEED_BASE + WORKLOAD_CONFIG[name])
        scene_d = torch.rand(x * y * z, generator=g, device=dev, dtype=torch.float32)
    else:
        scene_d = torch.from_numpy(synth.make_scene(x, y, z, np.float32, rng)).to(dev)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, onehot, _ = prep.build_index(labels, di, x, y)
    prep.shuffle(tr9, onehot, SEED_BASE)
    params_h = prep.init_params(z, np.float32)
    n_train = min(len(tr9), 65536)                  # a slice of the training list that fits comfortably
    idx_d = torch.from_numpy(np.ascontiguousarray(tr9[:n_train])).to(dev)
    cubes = torch.empty((n_train, 9 * z), dtype=torch.float32, device=dev)
    out = {"scene": f"{x}x{y}x{z}", "labelled_pixels": int(len(di)), "train_pixels_used": int(n_train)}
    if name == "aviris":
        # config 5's gather: the scene is fully labelled, so a rank gathers a solid block of it - here
        # 600 whole columns (1 228 800 pixels, 9.9 GB of cubes, output offsets past 2^31 elements) in
        # the shuffled order of the sample list; the plan picks the LDS-staged tile kernel for it
        blk = np.concatenate([tr9, te9])
        blk = blk[(blk[:, 4] // x) < 600]
        blk = blk[np.random.default_rng(SEED_BASE + 5).permutation(len(blk))]
        blk_d = torch.from_numpy(np.ascontiguousarray(blk)).to(dev)
        big = torch.empty((len(blk), 9 * z), dtype=torch.float32, device=dev)
        out["gather_call"] = gather_call_rate(scene_d, blk_d, x, y, z, big, reps=3)
        out["gather_call"]["list"] = "600 whole columns of the fully labelled scene, shuffled"
        del big, blk_d, blk
        torch.cuda.empty_cache()
        # the first 65 536 samples of the shuffled training list: a 1.5 % sparse sample of the scene
        # (neighbouring samples share no rows) - what the training-rate lines below run on
        out["gather_call_sparse_sample"] = gather_call_rate(scene_d, idx_d, x, y, z, cubes)
    else:
        out["gather_call"] = gather_call_rate(scene_d, idx_d, x, y, z, cubes)
    oh_d = torch.from_numpy(np.ascontiguousarray(onehot[:n_train])).to(dev)
    for B in (128, 2048):
        r = train_rate(params_h, cubes, oh_d, z, B, 300 if B == 128 else 100, name, profile=False)
        if r:
            out[f"train_b{B}_pixels_per_s"] = r["train_pixels_per_s"]
    del cubes, oh_d, idx_d
    # every pixel of the scene classified straight from the scene (no cubes in memory)
    all_lab = np.ones(x * y, dtype=np.float64)
    a_tr9, a_te9, _, _ = prep.build_index(all_lab, prep.scan_labels(all_lab, x, y), x, y)
    all_idx = torch.from_numpy(np.concatenate([a_tr9, a_te9])).to(dev)
    del a_tr9, a_te9
    p = torch.from_numpy(params_h).to(dev)
    fws = ops.Workspace(dev)
    ops.forward_scene(p, scene_d, all_idx, x, y, z, want_probs=False, workspace=fws)
    reps = max(1, int(4e6 // all_idx.shape[0]))
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        ops.forward_scene(p, scene_d, all_idx, x, y, z, want_probs=False, workspace=fws)
    torch.cuda.synchronize()
    out["infer_full_scene_pixels_per_s"] = round(all_idx.shape[0] * reps / (time.perf_counter() - t), 1)
    out["scene_pixels"] = int(all_idx.shape[0])
    if name == "ksc":
        # config 2: fp64 per-sample SGD (sequential by definition)
        ns = 2048
        idx64 = torch.from_numpy(np.ascontiguousarray(tr9[:ns])).to(dev)
        c64 = ops.gather(scene_d.double(), idx64, x, y, z)
        o64 = torch.from_numpy(np.ascontiguousarray(onehot[:ns])).to(dev)
        p64 = torch.from_numpy(prep.init_params(z, np.float64)).to(dev)
        sl = torch.zeros(1, dtype=torch.float64, device=dev)
        sws = ops.Workspace(dev)
        ops.sgd(p64, c64, o64, 0, 256, z, 0.008, sl, workspace=sws)
        torch.cuda.synchronize()
        t = time.perf_counter()
        ops.sgd(p64, c64, o64, 0, ns, z, 0.008, sl, workspace=sws)
        torch.cuda.synchronize()
        out["sgd_fp64_pixels_per_s"] = round(ns / (time.perf_counter() - t), 1)
        # the reference's canonical trainer on its own scene shape: fp64 MBGD, B = 100 (gcn_mbgd_fp64.cu)
        out["mbgd_fp64_b100"] = train_rate(prep.init_params(z, np.float64, 100 * d.pout), c64, o64, z, 100, 200, name)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--workload", default="pavia_u", choices=["pavia_u", "indian_pines", "ksc"])
    ap.add_argument("--no-extras", action="store_true",
                    help="skip by_batch / by_workload (the other batch sizes and scene shapes, rank 0 only)")
    ap.add_argument("--batch", type=int, default=128, help="samples per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N>1 step (gradient, all-reduce, update) even with one rank")
    args = ap.parse_args()
    # stdout carries exactly one line, the result: whatever else writes to file descriptor 1
    # (RCCL prints a version banner there at its first collective) is sent to stderr instead
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one process per GPU)")
        args.gpus = world
    # rehearsal on a box with fewer GPUs than ranks: CCNN_BENCH_DEVICE pins every rank to one device
    # and CCNN_BENCH_BACKEND=gloo replaces RCCL (which refuses two ranks on one GPU)
    if os.environ.get("CCNN_BENCH_DEVICE"):
        local = int(os.environ["CCNN_BENCH_DEVICE"])
    backend = os.environ.get("CCNN_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(local)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29533")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    L.load()
    t_start = time.perf_counter()
    if os.environ.get("CCNN_BENCH_TRACE_AFTER"):   # rehearsals: where is a rank that stopped making progress?
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["CCNN_BENCH_TRACE_AFTER"]), exit=False, file=sys.stderr)

    def log(msg):
        # progress on stderr: a long run must not look hung to whoever watches it
        print(f"[bench rank {rank} +{time.perf_counter() - t_start:6.1f}s] {msg}", file=sys.stderr, flush=True)

    # ---- synthetic scene of the real scene's shape, prepared exactly like training() does
    x, y, z, nl = synth.SCENES[args.workload]
    rng = np.random.default_rng(SEED_BASE + WORKLOAD_CONFIG[args.workload])
    labels = synth.make_labels(x, y, nl, rng)
    scene = synth.make_scene(x, y, z, np.float32, rng)
    di = prep.scan_labels(labels, x, y)
    tr9, te9, onehot, test_labels = prep.build_index(labels, di, x, y)
    prep.shuffle(tr9, onehot, SEED_BASE)
    params_h = prep.init_params(z, np.float32)
    d = L.dims(z)
    B = args.batch
    T = len(tr9)
    slice_len = T // world
    lo = rank * slice_len
    nb = slice_len // B
    assert nb >= 1, "slice smaller than one batch"

    dev = torch.device("cuda", local)
    scene_d = torch.from_numpy(scene).to(dev)
    cubes = ops.gather(scene_d, torch.from_numpy(tr9[lo:lo + slice_len]).to(dev), x, y, z)
    onehot_d = torch.from_numpy(np.ascontiguousarray(onehot[lo:lo + slice_len])).to(dev)
    te_lo, te_n = rank * (len(te9) // world), len(te9) // world
    tcubes = ops.gather(scene_d, torch.from_numpy(te9[te_lo:te_lo + te_n]).to(dev), x, y, z)
    params = torch.from_numpy(params_h).to(dev)
    grad = torch.empty_like(params)
    loss_ring = torch.zeros((nb, B), dtype=torch.float64, device=dev)
    ws = ops.Workspace(dev)
    lr = 0.04 / world      # plain-sum update: keep the step size independent of the global batch

    log(f"inputs resident: {slice_len} training cubes, {te_n} test cubes")
    # N>1: the library's one-shot exchange over xGMI, fused with the update; RCCL if it cannot be
    # set up on this node or fails its self-test against RCCL (CCNN_ALLREDUCE=rccl forces RCCL)
    comm = None
    if use_dist and os.environ.get("CCNN_ALLREDUCE", "oneshot") != "rccl":
        from gcn_amd.comm import NodeComm
        os.environ.setdefault("CCNN_COMM_WAIT_MS", "5000")   # a broken exchange must fail fast, not stall the bench
        try:
            comm = NodeComm(d.nparams, torch.float32)
            if not comm.self_test(rounds=2, z=z):
                comm.close()
                comm = None
        except Exception as e:   # raised on every rank alike (see NodeComm)
            if rank == 0:
                print(f"one-shot exchange unavailable ({e}); using RCCL", file=sys.stderr)
            comm = None

    # route of the N>1 step: "instep" (exchange inside the reduction kernel, every rank writing to
    # every rank), "instep_owner" (the same with the sums meeting at an owner rank that sends the total
    # round: fewer bytes per link, one more hop), "standalone" (gradient, then the library's exchange +
    # update kernel) or "rccl" (gradient, RCCL all-reduce, update)
    mode = "instep" if comm is not None else "rccl"     # the library's default form; the warm-up below times the others

    def set_mode(m):
        nonlocal mode
        mode = m
        if comm is not None and m.startswith("instep"):
            comm.set_exchange(m == "instep_owner")
    log(f"exchange set up: mode {mode}" + (f", ranks share a device: {comm.shared_device}" if comm is not None else ""))

    # the step's unchanging arguments resolved once (ops.MbgdStepper): one foreign call per step
    st_local = ops.MbgdStepper(params, cubes, onehot_d, B, z, loss_values=loss_ring, workspace=ws)
    st_dp = ops.MbgdStepper(params, cubes, onehot_d, B, z, loss_values=loss_ring, workspace=ws, comm=comm) if comm is not None else None

    def step(i):
        b = i % nb
        if not use_dist:
            # gradient + update fused (nothing has to happen in between on one GPU)
            st_local.step(b * B, lr, b)
        elif mode in ("instep", "instep_owner"):
            # two launches: the reduction kernel trades its sums with the other ranks and updates
            st_dp.step(b * B, lr, b)
        else:
            ops.mbgd_grad(params, cubes, onehot_d, b * B, B, z, grad=grad, loss_values=loss_ring[b], workspace=ws)
            if mode == "standalone":
                comm.allreduce_update(params, grad, lr, z)
            else:
                if backend == "gloo":   # rehearsal backend: gloo reduces host tensors
                    h = grad.cpu()
                    dist.all_reduce(h)
                    grad.copy_(h)
                else:
                    dist.all_reduce(grad)
                ops.update(params, grad, lr, z)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(n, first=0):
        fence()
        t = time.perf_counter()
        for i in range(n):
            step(first + i)
        fence()
        t = time.perf_counter() - t
        if world > 1:
            tt = torch.tensor([t], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t = tt.item()
        return t

    def stalled():
        fence()
        st = [None] * world
        dist.all_gather_object(st, comm.status())
        return any(st)

    if comm is not None:
        # three steps first: an exchange that passed its self-tests and stalls all the same must
        # not be waited for two hundred times
        for i in range(3):
            step(i)
        if stalled():
            if rank == 0:
                print("one-shot exchange timed out in its first steps; using RCCL", file=sys.stderr)
            mode = "rccl"
    for i in range(args.warmup):
        step(i)
    fence()
    log("warm-up done")
    calibration = None
    if comm is not None and mode == "rccl":
        comm.close()
        comm = None
        params.copy_(torch.from_numpy(params_h).to(dev))
        for i in range(args.warmup):
            step(i)
        fence()
    if comm is not None:
        # The exchange passed its self-tests.  Had it stalled during warm-up anyway, finish on RCCL;
        # otherwise let the rest of the warm-up decide which of the three routes this node runs
        # fastest (untimed steps of the same work; every rank sees the same max-over-ranks times).
        st_all = [None] * world
        dist.all_gather_object(st_all, comm.status())
        if any(st_all):
            if rank == 0:
                print("one-shot exchange timed out during warm-up; using RCCL", file=sys.stderr)
            mode = "rccl"
        elif os.environ.get("CCNN_ALLREDUCE", "") != "oneshot":
            ncal = max(args.warmup, 50)
            times = {}
            for m in ("instep", "instep_owner", "standalone", "rccl"):
                set_mode(m)
                for i in range(3):
                    step(i)
                if m != "rccl" and stalled():            # a wait timed out: not a candidate, and not worth 200 more waits
                    times[m] = float("inf")
                    continue
                for i in range(20):
                    step(i)
                times[m] = timed(ncal)
                if m != "rccl" and stalled():
                    times[m] = float("inf")
            calibration = {"steps": ncal, **{f"{m}_us_per_step": (round(t / ncal * 1e6, 2) if t != float("inf") else None)
                                             for m, t in times.items()}}
            set_mode(min(times, key=times.get))
        if mode == "rccl":
            comm.close()
            comm = None
        params.copy_(torch.from_numpy(params_h).to(dev))
        for i in range(args.warmup):
            step(i)
        fence()
    log(f"timing {args.steps} steps, mode {mode}")
    dt = timed(args.steps, args.warmup)
    train_pps = world * B * args.steps / dt
    # how much the step time scatters around that mean: the same steps again (untimed for `value`), an
    # event on the launch stream after every GROUP steps (an event per step would add its own couple
    # of microseconds to every interval), this rank's GPU only
    GROUP = 10
    ngr = max(10, min(args.steps, 400) // GROUP)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(ngr + 1)]
    fence()
    evs[0].record()
    for g in range(ngr):
        for i in range(GROUP):
            step(args.warmup + g * GROUP + i)
        evs[g + 1].record()
    fence()
    per = np.array([evs[g].elapsed_time(evs[g + 1]) * 1e3 / GROUP for g in range(ngr)])
    spread = {"groups": ngr, "steps_per_group": GROUP, "median_us": round(float(np.median(per)), 2),
              "p10_us": round(float(np.percentile(per, 10)), 2), "p90_us": round(float(np.percentile(per, 90)), 2),
              "max_us": round(float(per.max()), 2),
              "what": "per-step time of groups of ten steps (events between the groups) in a second, untimed pass"}
    log(f"timed region done: {dt / args.steps * 1e6:.2f} us/step")
    assert torch.isfinite(params).all(), "training diverged"
    replicas_identical = True
    rank_devices = None
    if use_dist:
        # which physical GPU every rank ran on (PCI bus id), as the group saw it
        rank_devices = [None] * world
        dist.all_gather_object(rank_devices, f"{torch.cuda.get_device_properties(local).name} "
                                             f"pci {torch.cuda.get_device_properties(local).pci_bus_id:02x}")
        # every rank applied the same sums in the same order: the replicas must agree bit for bit
        digest = params.view(torch.int32).to(torch.int64).sum().reshape(1)
        every = [torch.empty_like(digest) for _ in range(world)]
        dist.all_gather(every, digest)
        replicas_identical = all(bool(torch.equal(every[0], e)) for e in every)
        if comm is not None:
            assert comm.status() == 0, "a rank gave up waiting for a peer during the exchange"

    # ---- inference over this rank's test pixels
    fws = ops.Workspace(dev)
    ops.forward(params, tcubes, z, want_probs=False, workspace=fws)
    reps = max(1, int(2e6 // max(te_n, 1)))
    fence()
    t1 = time.perf_counter()
    for _ in range(reps):
        ops.forward(params, tcubes, z, want_probs=False, workspace=fws)
    fence()
    dti = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dti], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dti = t.item()
    infer_pps = world * te_n * reps / dti
    # the same pixels classified straight from the scene (no cubes in memory)
    te_idx_d = torch.from_numpy(np.ascontiguousarray(te9[te_lo:te_lo + te_n])).to(dev)
    ops.forward_scene(params, scene_d, te_idx_d, x, y, z, want_probs=False, workspace=fws)
    fence()
    t1 = time.perf_counter()
    for _ in range(reps):
        ops.forward_scene(params, scene_d, te_idx_d, x, y, z, want_probs=False, workspace=fws)
    fence()
    dts = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dts], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dts = t.item()
    infer_scene_pps = world * te_n * reps / dts
    # every pixel of the scene (this rank's 1/N of them), again straight from the scene
    all_lab = np.ones(x * y, dtype=np.float64)
    a_tr9, a_te9, _, _ = prep.build_index(all_lab, prep.scan_labels(all_lab, x, y), x, y)
    all9 = np.concatenate([a_tr9, a_te9])
    fs_n = len(all9) // world
    all_idx_d = torch.from_numpy(np.ascontiguousarray(all9[rank * fs_n:(rank + 1) * fs_n])).to(dev)
    ops.forward_scene(params, scene_d, all_idx_d, x, y, z, want_probs=False, workspace=fws)
    fs_reps = max(1, int(2e6 // max(fs_n, 1)))
    fence()
    t1 = time.perf_counter()
    for _ in range(fs_reps):
        ops.forward_scene(params, scene_d, all_idx_d, x, y, z, want_probs=False, workspace=fws)
    fence()
    dtf = time.perf_counter() - t1
    if world > 1:
        t = torch.tensor([dtf], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dtf = t.item()
    infer_full_pps = world * fs_n * fs_reps / dtf
    del all_idx_d

    log("inference timed")
    # ---- fp64 per-sample SGD on the same cubes (sequential by definition: replicas only, rank 0 reports)
    sgd_pps = None
    if rank == 0:
        ns = min(slice_len, 4096)
        cubes64 = cubes[:ns].double()
        oh64 = onehot_d[:ns].contiguous()
        p64 = torch.from_numpy(prep.init_params(z, np.float64)).to(dev)
        sloss = torch.zeros(1, dtype=torch.float64, device=dev)
        sws = ops.Workspace(dev)
        ops.sgd(p64, cubes64, oh64, 0, min(ns, 256), z, 0.008, sloss, workspace=sws)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ops.sgd(p64, cubes64, oh64, 0, ns, z, 0.008, sloss, workspace=sws)
        torch.cuda.synchronize()
        sgd_pps = ns / (time.perf_counter() - t2)
        assert torch.isfinite(p64).all()
        del cubes64

    # ---- fp64 MBGD (the reference's canonical build, gcn_mbgd_fp64.cu: B = 100) on the same cubes, rank 0:
    # rate, and its dominant kernel against the fp64 matrix peak
    mbgd64 = None
    if rank == 0:
        B64 = 100
        n64 = min(slice_len // B64, 40)
        c64 = cubes[:n64 * B64].double()
        o64 = onehot_d[:n64 * B64].contiguous()
        mbgd64 = train_rate(prep.init_params(z, np.float64, B64 * d.pout), c64, o64, z, B64, 400, args.workload)
        del c64

    # ---- per-kernel timing (HIP events on the launch stream, recorded inside the library):
    # the same training steps, one inference pass and one gather of the training list
    roof = None
    kernels = {}
    fence()   # rank 0 ran the single-GPU extras above: nobody starts exchanging (with bounded waits) before it is back
    if rank == 0:
        L.profile_enable(True)
    psteps = min(args.steps, 300)
    for i in range(psteps):
        step(i)
    fence()
    if rank == 0:
        preps = 5
        gws = ops.Workspace(dev)
        idx_d = torch.from_numpy(tr9[lo:lo + slice_len]).to(dev)
        for _ in range(preps):
            ops.forward(params, tcubes, z, want_probs=False, workspace=fws)
            ops.gather(scene_d, idx_d, x, y, z, out=cubes, workspace=gws)
        torch.cuda.synchronize()
        L.profile_enable(False)
        pixels = {k: B for k in TRAIN_KERNELS}
        pixels.update({"conv_pool_fwd": te_n, "fc_head_fwd": te_n, "gather_tile": slice_len})
        kernels = kernel_rows(L.profile_report(), d, z, pixels, world=world,
                              passes={"conv_pool_fwd": preps, "fc_head_fwd": preps, "gather_tile": preps})
        roof = dominant(kernels, args.workload, B)
        kernels["gather_call"] = gather_call_rate(scene_d, idx_d, x, y, z, cubes)

    log("per-kernel timing done")
    # ---- the other batch sizes (same cubes) and the other scene shapes, rank 0 only
    by_batch, by_workload = None, None
    if rank == 0 and world == 1 and not args.no_extras:
        by_batch = []
        for Bx in (128, 512, 2048):
            r = train_rate(params_h, cubes, onehot_d, z, Bx, 300 if Bx < 2048 else 150, args.workload)
            if r:
                by_batch.append(r)
            log(f"batch {Bx} done")
        kernels["gather_call"]["fully_labelled"] = gather_fully_labelled(scene_d, x, y, z)
        by_workload = {}
        for name in ("ksc", "indian_pines", "pavia_u", "aviris"):
            if name == args.workload:
                continue
            by_workload[name] = workload_report(name, dev)
            torch.cuda.empty_cache()
            log(f"workload {name} done")

    # ---- CPU baseline: the oracle's fp32 minibatch step on the host cores (rank 0, N=1 only)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        import oracle as orc
        cubes_h = cubes[:min(slice_len, 16 * B)].cpu().numpy()
        onehot_h = np.ascontiguousarray(onehot[lo:lo + len(cubes_h)])
        p = params_h.copy()
        nbh = len(cubes_h) // B
        orc.mbgd_step(p, cubes_h, onehot_h, 0, B, z, np.float32(0.04), want_grad=False)   # warm-up
        n_done, tc0 = 0, time.perf_counter()
        while time.perf_counter() - tc0 < args.cpu_seconds:
            orc.mbgd_step(p, cubes_h, onehot_h, (n_done % nbh) * B, B, z, np.float32(0.04), want_grad=False)
            n_done += 1
        tc = time.perf_counter() - tc0
        cpu = {"value": round(n_done * B / tc, 1), "unit": "train pixels/s", "cores": orc.num_threads(),
               "kind": "port", "sample": f"{n_done} fp32 MBGD steps of {B} {args.workload} pixels "
                                         f"(oracle/ccnn_oracle.c, OpenMP over samples)"}

    if rank == 0:
        out = {
            "metric": "train pixels/sec (fwd+bwd+update), MBGD fp32",
            "value": round(train_pps, 1),
            "unit": "pixels/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 5),
            "step_spread": spread,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": f"{args.workload}-shaped {x}x{y}x{z} fp32 MBGD", "batch_per_gpu": B,
                       "global_batch": B * world, "train_pixels": T, "parallelism": f"dp{world}",
                       "exchange": ("none" if not use_dist else
                                    {"instep": "xgmi one-shot exchange inside the reduction kernel",
                                     "instep_owner": "xgmi exchange inside the reduction kernel, sums reduced at an owner rank and sent round",
                                     "standalone": "xgmi one-shot exchange + update kernel after the gradient",
                                     "rccl": "rccl all-reduce + update"}[mode]),
                       "exchange_calibration": calibration, "replicas_identical": replicas_identical,
                       "ranks_in_group": (dist.get_world_size() if use_dist else 1),
                       "backend": (dist.get_backend() if use_dist else None), "rank_devices": rank_devices},
            "infer_pixels_per_s": round(infer_pps, 1),
            "infer_from_scene_pixels_per_s": round(infer_scene_pps, 1),
            "infer_full_scene_pixels_per_s": round(infer_full_pps, 1),
            "sgd_fp64_pixels_per_s": round(sgd_pps, 1),
            "mbgd_fp64_b100_pixels_per_s": mbgd64["train_pixels_per_s"],
            "by_dtype": {"f64": {args.workload: mbgd64,
                                 **({"ksc": by_workload["ksc"].get("mbgd_fp64_b100")} if by_workload and "ksc" in by_workload else {}),
                                 "what": "fp64 MBGD, B = 100 (the reference's canonical trainer); roofline: dominant kernel "
                                         "against the 78.6 TFLOP/s fp64 matrix peak, from the dispatch events"}},
            "roofline": roof,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "by_batch": by_batch,
            "by_workload": by_workload,
        }
        if by_workload is not None:
            by_workload[args.workload] = {
                "scene": f"{x}x{y}x{z}", "labelled_pixels": int(len(di)), "train_pixels_used": int(slice_len),
                "gather_call": kernels.get("gather_call"), "train_b128_pixels_per_s": round(train_pps, 1) if B == 128 else None,
                "train_b2048_pixels_per_s": next((r["train_pixels_per_s"] for r in (by_batch or []) if r["batch"] == 2048), None),
                "infer_full_scene_pixels_per_s": round(infer_full_pps, 1), "scene_pixels": int(fs_n * world)}
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if comm is not None:
        comm.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
