"""Node-local one-shot all-reduce (gcn_amd/csrc/comm.hip) with 2 and 4 ranks.  The ranks are
separate processes that map each other's exchange buffers through IPC handles, exactly as on a
multi-GPU node; here they all sit on the one GPU of the test box."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, *args, wait_ms="2000"):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="1", CCNN_COMM_WAIT_MS=wait_ms)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29610 + world),
           os.path.join(ROOT, "tests", "comm_worker.py"), *args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=110)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_oneshot_allreduce_ranks(world):
    r = _run(world)
    assert r.returncode == 0 and f"COMM-OK {world}" in r.stdout, r.stdout[-2000:] + r.stderr[-12000:]


@pytest.mark.gpu
def test_two_rank_fit_on_gpu_matches_master_worker_replay():
    r = _run(2, "--fit")
    assert r.returncode == 0 and "FIT-OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-12000:]


@pytest.mark.gpu
def test_timed_out_exchange_leaves_the_parameters_alone():
    """One rank never launches: the other's waits run out, its parameters stay bit-identical and
    ccnn_comm_status reports 1."""
    r = _run(2, "--stalled-peer", wait_ms="200")
    assert r.returncode == 0 and "STALL-OK 2" in r.stdout, r.stdout[-2000:] + r.stderr[-12000:]
