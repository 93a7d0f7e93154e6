"""unique-ids workload (workload/unique_ids.clj over demo/clojure/flake_ids.clj), CPU side."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _check_rows(rows):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    assert A.load().msim_check_unique_rows(rows.ctypes.data_as(C.c_void_p), len(rows), C.byref(res)) == 0
    return res


@pytest.mark.parametrize("kw", [dict(), dict(latency=30, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=3)])
def test_flake_ids_are_unique_and_shaped_like_the_reference(kw):
    cfg = E.test_config("unique-ids", node_count=3, rate=200, time_limit=5, seed=8, **kw)
    assert cfg.node_program == A.NODE_FLAKE_IDS
    r = O.run(cfg, 0, 3)
    for i in range(3):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, 3, A.WL_UNIQUE_IDS) if o["process"] != ":nemesis"]
        oks = [o for o in ops if o["type"] == ":ok"]
        assert len(oks) > (5 if kw else 500) and all(o["f"] == ":generate" for o in ops)   # a lost message stalls its worker for 5 s
        for o in oks:   # [time count node-id], flake_ids.clj:30-31; the client is pinned to node process mod n
            t, c, n = o["value"]
            assert n == f"n{o['process'] % 3}" and 0 <= t <= 6 and t == o["time"] // 10**9
        # the counter restarts every second, per node
        by = {}
        for o in oks:
            by.setdefault((o["value"][2], o["value"][0]), []).append(o["value"][1])
        # (the state starts as {:time 0 :count 0}, flake_ids.clj:13-14: in virtual second 0 the first id already counts 1)
        assert all(v == list(range(1 if t == 0 else 0, len(v) + (1 if t == 0 else 0))) for (_, t), v in by.items())
        res = _check_rows(rows)
        assert res.valid == 1 and res.duplicated_count == 0 and res.ok_count == len(oks) == res.attempt_count - res.info_count - res.fail_count


def test_checker_finds_duplicates():
    rows = np.zeros(8, dtype=E.OP_DT)
    vals = [5, 7, 5, 9, 7, 5]
    for i, v in enumerate(vals):
        rows["packed"][i] = A.T_OK | (A.F_GENERATE << 2) | (i << 12)
        rows["value"][i] = v
    rows["packed"][6] = A.T_INVOKE | (A.F_GENERATE << 2)
    rows["packed"][7] = A.T_INFO | (A.F_GENERATE << 2)
    res = _check_rows(rows)
    assert res.valid == 0 and res.duplicated_count == 2 and res.ok_count == 6 and res.attempt_count == 1   # 5 and 7 repeat
    assert list(res.stable_latency_ms)[:2] == [5, 9]


def test_lin_tso_hands_out_every_integer_once_in_completion_order_per_client():
    """service.clj:116-132 through the oracle's node + service (pinned by a real node process on the bridge, tests/test_process_bridge.py):
    the ids of a run are 0, 1, 2, ... without gaps while nothing is lost, each client sees its own ids grow, one ts / ts_ok pair per id."""
    cfg = E.test_config("unique-ids", bin="tso-ids", node_count=3, rate=200, time_limit=5, latency=5, seed=3)
    r = O.run(cfg, 0, 3)
    for i in range(3):
        assert r.meta["flags"][i] == 0
        ops = E.decode_history(*r.history(i), cfg.n_nodes, cfg.workload, cfg.node_program)
        ids = [o["value"] for o in ops if o["type"] == ":ok"]
        assert len(ids) > 500 and sorted(ids) == list(range(len(ids)))
        per = {}
        for o in ops:
            if o["type"] == ":ok":
                assert per.get(o["process"], -1) < o["value"]
                per[o["process"]] = o["value"]
        assert int(r.stats[i]["servers_send"]) == 2 * len(ids)
        res = E.check_unique_history(r.history(i)[0]) if hasattr(E, "check_unique_history") else None
        assert res is None or res["valid?"] is True


def test_lin_tso_under_loss_still_never_repeats_an_id():
    cfg = E.test_config("unique-ids", bin="tso-ids", node_count=4, concurrency=8, rate=300, time_limit=6, latency=10, latency_dist="exponential", p_loss=0.1, seed=4)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        ops = E.decode_history(*r.history(i), cfg.n_nodes, cfg.workload, cfg.node_program)
        ids = [o["value"] for o in ops if o["type"] == ":ok"]
        assert len(set(ids)) == len(ids) > 20 and any(o["type"] == ":info" for o in ops)   # (a lost message costs its worker the 5 s timeout)
