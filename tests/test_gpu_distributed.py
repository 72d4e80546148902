"""GPU: the RCCL ("nccl" backend) branch of bench.py on the one card of the test box -- process group init with device_id,
all_gather, barrier, max-reduce -- as a world-size-1 job in a child process (the test process itself stays out of the group)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu


def test_bench_rccl_world_size_one():
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--rehearse"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert out["backend"] == "nccl" and out["rccl_ranks"] == 1 and out["all_gather_ok"] is True
