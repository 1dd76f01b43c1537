"""GPU: bench.py itself, as the driver launches it, in a FRESH child process (never a re-exec of this process, which has touched the GPU).

* the N > 1 loop of bench.py (RCCL gather -> gathered views -> drain -> slice checks) rehearsed on one GPU with EFTB_BENCH_FORCE_COMM=1:
  one rank exchanging with itself over RCCL.  Not a scaling measurement (DESIGN section 6 keeps saying "unmeasured").
* the N = 1 loop: every timed step bit-identical to the synchronous path, the dependent-sampler key present.
"""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _bench(extra_env, *flags):
    env = dict(os.environ, **extra_env)
    env.pop("EFTB_BENCH_DEPTH", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras", *flags]
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-1500:]
    return json.loads(lines[0])


def test_bench_multi_rank_loop_with_rccl_self_exchange():
    """bench.py's `exchange == "rccl"` loop: eftb_gather_plk every step, fetch_gathered(copy=False) views DEPTH steps back, the drain, and the
    asserts on the root's own slice of the gathered block (inside bench.py) -- all on one GPU."""
    j = _bench({"EFTB_BENCH_FORCE_COMM": "1"})
    assert j["valid"] is True and j["n_gpus"] == 1
    assert "rccl" in j["config"]["parallelism"]
    assert j["timed_steps_checked_against_sync_path"] == 4     # every timed step of the gathered loop == the synchronous path, bit for bit
    assert j["value"] > 0 and j["roofline"]["launches_timed"] >= 1


def test_bench_single_gpu_loop_checks_every_step():
    j = _bench({})
    assert j["valid"] is True and j["config"]["parallelism"] == "single GPU"
    assert j["timed_steps_checked_against_sync_path"] == 4
    assert j["sync_step_evaluations_per_s"] > 0 and j["sync_step_ms"] > 0
    assert 0.0 < j["roofline"]["frac"] < 1.0 and j["dtype"] == "f64"
