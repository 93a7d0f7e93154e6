"""bench.py's supervisor on a machine without a GPU: an ORDINARY error exit of the measuring child (here: no HIP device) is reported once and
passed on — only a child that dies of a signal (how a GPU fault ends a process) is started again (tests/test_bench_line_gpu.py has that case)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_an_ordinary_error_is_not_retried():
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("a device is present: the child would measure")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "0"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 1
    assert r.stdout.strip() == ""                                     # no JSON line without a measurement
    assert r.stderr.count("bench.py needs a HIP device") == 1          # one attempt
    assert "attempt 1 ended with return code 1" in r.stderr and "attempt 2" not in r.stderr


def test_the_launcher_contract_is_checked_before_anything_runs():
    env = dict(os.environ, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode != 0 and "n_gpus must be what was asked for" in r.stderr
