"""RCCL on the one GPU a test box has: a group of one rank, every collective helper forced through its device branch."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_rccl_group_of_one_runs_every_collective_of_the_multi_gpu_path(device):
    """A fresh interpreter initialises the "nccl" backend (RCCL) on cuda:0 with world_size 1 and calls
    allreduce_gradients (in place on the two flat gradient vectors of a real backward pass, and through the bucket),
    broadcast_parameters, all_gather_rows, render_sharded and render_path_sharded with force_collective: the
    library loads, the exact calls of the 8-GPU run execute on device tensors, and every result equals its input."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    run = subprocess.run([sys.executable, os.path.join(root, "tests", "rccl_single_rank.py")], cwd=root, env=env,
                         capture_output=True, text=True, timeout=280)
    lines = [ln for ln in run.stdout.splitlines() if ln.startswith("{")]
    assert lines, (run.returncode, run.stdout[-1000:], run.stderr[-3000:])
    res = json.loads(lines[-1])
    assert res["backend"] == "nccl" and res["world"] == 1
    assert res["ok"] and run.returncode == 0, (res, run.stderr[-2000:])
