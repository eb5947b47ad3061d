"""GPU: failure paths of bench.py that a healthy run never takes.

An invalidated hipGraph capture cannot be ended and leaves the process unable to synchronise the device or empty the allocator's
cache (steady_state.py), so bench.py treats it as fatal for the PROCESS: at N = 1 it starts a fresh child with `--graph off`
and forwards the child's JSON line; at N > 1 it exits non-zero at once.  The injection hook makes the first capture attempt
report an invalidated capture."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.timeout(600)
def test_invalidated_capture_reruns_in_a_fresh_child_with_eager_launches():
    env = dict(os.environ, SS_BENCH_INJECT_CAPTURE_INVALIDATED="1", SS_BENCH_PAGE_IN="0")
    env.pop("SS_BENCH_CHILD", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--no-secondary", "--no-cpu-baseline",
                        "--no-pmc"], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=550)
    err = r.stderr.decode(errors="replace")
    assert r.returncode == 0, err[-2000:]
    assert "hipGraph capture invalidated: injected" in err and "re-running in a fresh child process with eager launches" in err
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines                                     # ONE JSON line: the child's
    res = json.loads(lines[0])
    assert res["config"]["execution"].startswith("eager launches") and res["steps"] == 2 and res["value"] > 1e6
    assert "roofline" in res and res["n_gpus"] == 1
