"""GPU (-m gpu): the ray-sharded iteration on a real RCCL communicator (world size 1 on the one leased GPU; tests/rccl_world1_probe.py)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_world1_rccl_iteration_is_the_plain_iteration_bit_for_bit():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29561", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_world1_probe.py")], capture_output=True, text=True, timeout=600, env=env)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_WORLD1 ")]
    assert r.returncode == 0 and line, (r.stdout[-2000:], r.stderr[-4000:])
    out = json.loads(line[-1][len("RCCL_WORLD1 "):])
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "rccl_world1.json"), "w"), indent=1)
    assert out["ok"], out
    for k in ("onecall", "stagewise", "graph", "onecall_rows", "graph_rows", "onecall_dense_forced"):
        assert all(out[k]["equal"].values()) and out[k]["backend"] == "rccl" and not out[k]["invalid"], (k, out[k])
    assert out["onecall"]["rows_cap"] == "dense" and isinstance(out["onecall_rows"]["rows_cap"], int) and out["onecall_dense_forced"]["rows_cap"] == "dense"
