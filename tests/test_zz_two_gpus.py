"""Tests that need TWO visible GPUs (the driver's multi-GPU node; every one is skipped on the one-GPU boxes of a build session).
In a file of their own that sorts last: this code path has never run on hardware during a build, and a failure here must not keep
`pytest -x` from running the rest of the suite first."""
import os

import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("bounds", [None, "0,0,45", "0,31,45"])
def test_bench_two_gpus_rccl_exchange_inside_libmgs(bounds):
    """Runs only where two GPUs are visible (the driver's multi-GPU node; skipped on the one-GPU boxes of a build session):
    bench.py --gpus 2 over RCCL, i.e. mgs_scene_comm_init + mgs_render_gathered — libmgs's own grouped in-place ncclBroadcast with
    more than one rank — with cost-balanced strips, an EMPTY strip (rank 0 renders nothing and still takes part) and unequal
    strips; --check-gather asserts that the frame every rank holds afterwards equals the single-GPU frame bit for bit."""
    import json
    import subprocess
    import sys
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29731", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2", "--width", "1280",
           "--height", "720", "--splats", "400000", "--backend", "nccl", "--check-gather", "--inflight", "2", "--no-cpu-baseline"]
    if bounds:
        cmd += ["--strip-bounds", bounds]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, cwd=root, env=dict(os.environ, MASTER_ADDR="127.0.0.1"))
    assert out.returncode == 0, out.stdout[-1500:] + out.stderr[-3000:]
    assert out.stderr.count("gathered frame == full frame: True") == 2, out.stderr[-3000:]
    assert "libmgs: grouped ncclBroadcast" in out.stderr, out.stderr[-3000:]  # not the torch.distributed fallback
    d = json.loads([l for l in out.stdout.splitlines() if l.strip().startswith("{")][0])
    assert d["n_gpus"] == 2 and d["error_flags"] == 0 and d["value"] > 0
