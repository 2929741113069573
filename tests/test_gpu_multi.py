"""RCCL through the real path, the first time a box with >= 2 GPUs runs the suite (the pool's test boxes have one: the test
skips itself there).  tests/multi_gpu_worker.py under ``torch.distributed.run``: ``segment_plot(dist=...)`` and
``classify_sharded`` over ``nccl`` must reproduce the single-process result bit for bit on every rank."""
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _launch(world, extra, timeout):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "multi_gpu_worker.py"), *extra]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.gpu
def test_sharded_paths_over_rccl_match_single_process():
    n = torch.cuda.device_count()
    if n < 2:
        pytest.skip(f"{n} GPU visible: the RCCL path needs two (covered by the gloo tests of tests/test_host_cpu.py)")
    world = 2 if n < 4 else 4
    r = _launch(world, [], 900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "all_ranks_ok=1" in r.stdout


def test_multi_gpu_worker_dry_run_gloo():
    """The worker script's own logic on the CPU (gloo, stand-in model and vote), world 2."""
    r = _launch(2, ["--dry", "--points", "40000"], 600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "all_ranks_ok=1" in r.stdout and "plot_equal=True" in r.stdout and "classify_equal=True" in r.stdout
