"""bench.py's own contract on the device: the driver's command prints exactly one JSON line with the fields the contract names, and a
process death inside a secondary leg (how a GPU memory fault ends a process; the round-4 driver run) still leaves the measured
headline on stdout."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(extra, env=None):
    e = dict(os.environ)
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1", "--instances", "1024"] + extra,
                       env=e, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    return r, lines


def test_one_json_line_with_the_contract_fields():
    r, lines = _bench(["--cpu-sample", "0.5"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 1e8
    assert d["histories_valid"] == d["histories_checked"] == 3 * 1024 and d["instances_flagged"] == 0
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    ck = d["roofline"]["checker"]   # the set-full checker alone on the chip: the step's one bandwidth kernel (a third of the HBM peak at the headline's 4096 histories; this run has 1024)
    assert ck["bound"] == "hbm" and 0.02 < ck["frac"] < 1 and abs(ck["achieved"] / ck["peak"] - ck["frac"]) < 1e-9, ck
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0
    assert d["value_incl_fetch"] and d["history_gather"]["bytes"] > 0
    assert "attempts" not in d


@pytest.mark.parametrize("leg", ["incl_fetch", "history_gather"])
def test_a_death_inside_a_secondary_leg_keeps_the_headline(leg):
    r, lines = _bench(["--cpu-sample", "0"], env={"MSIM_BENCH_TEST_ABORT_IN": leg})
    assert r.returncode == 0, r.stderr[-2000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] > 1e8 and d["histories_valid"] == 3 * 1024
    assert d["attempts"]["failed"][0]["died_in"] == leg and d["attempts"]["n"] == 2   # the second attempt ran without that leg
    if leg == "incl_fetch":
        assert d["value_incl_fetch"] is None and d["history_gather"]["bytes"] > 0
    else:
        assert "history_gather" not in d and d["value_incl_fetch"]
