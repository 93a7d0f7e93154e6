"""maelstrom.checker/availability-checker (checker.clj:6-39): :ok-fraction and the three --availability modes (core.clj:149),
on oracle histories through the host entry point; the device entry point is compared with it in the GPU tier."""
import numpy as np
import pytest

from maelstrom_amd import engine as E
import oracle_lib as O


def _ref(history, a):
    """checker.clj:15-39 over decoded op maps"""
    ok = sum(1 for op in history if op["type"] == ":ok")
    inv = sum(1 for op in history if op["type"] == ":invoke")
    frac = 1.0 if inv == 0 else float(np.float32(ok / inv))
    if a is None:
        return True, frac
    if a == "total":
        return frac == 1.0, frac
    return a <= frac, frac


def test_availability_modes_on_lossy_and_healthy_histories():
    healthy = E.test_config("broadcast", node_count=5, rate=20, time_limit=5, seed=1)
    lossy = E.test_config("echo", node_count=3, rate=20, time_limit=8, p_loss=0.2, seed=5)
    for cfg, expect_total in ((healthy, True), (lossy, False)):
        ora = O.run(cfg, 0, 4)
        for i in range(4):
            rows, pay = ora.history(i)
            hist = E.decode_history(rows, pay, cfg.n_nodes, cfg.workload)
            for a in (None, "total", 0.5, 0.99, 1.0, 0.0):
                got = E.check_availability_rows(rows, a)
                want_valid, want_frac = _ref(hist, a)
                assert got["valid?"] == want_valid and got["ok-fraction"] == pytest.approx(want_frac, abs=0)
            assert E.check_availability_rows(rows, "total")["valid?"] == expect_total
            assert E.check_availability_rows(rows, None)["valid?"] is True


def test_empty_history_and_bad_arguments():
    empty = np.zeros(0, dtype=E.OP_DT)
    assert E.check_availability_rows(empty, "total") == {"valid?": True, "ok-fraction": 1.0, "ok-count": 0, "invoke-count": 0}
    with pytest.raises(E.EngineError):
        E.check_availability_rows(empty, "mostly")
    with pytest.raises(E.EngineError):
        E.check_availability_rows(empty, 1.5)
