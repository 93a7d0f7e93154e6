"""The device pass of the rw-register analysis (csrc/rw_check_dev.hip, msim_check for txn-rw-register / msim_check_rw_batch)
against the host analysis (msim_check_rw_rows, itself compared with tests/elle_ref.py in test_txn_rw_register.py; [upstream] elle is
not vendored: parity unpinned): the same :valid?, the same counts, and never an anomaly the host does not see; whatever the device
cannot prove valid it hands to the host, whose result is then the result."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

import oracle_lib as O

pytestmark = pytest.mark.gpu

RW = A.WL_TXN_RW_REGISTER
R = ":r"
MODELS = ["read-uncommitted", "read-committed", "snapshot-isolation", "serializable", "strict-serializable"]


def _cfg(**kw):
    args = dict(workload="txn-rw-register", node_count=2, rate=100.0, time_limit=6.0, seed=11)
    args.update(kw)
    return E.test_config(**args)


def _agree(hs, model):
    """device + host fallback == host, history by history (verdict, counts; the device's anomalies are a subset of the host's)"""
    res, n_host = E.check_rw_batch(hs, model)
    proscribed = A.load().msim_proscribed_anomalies(E.CONSISTENCY_MODELS[model])
    decided = 0
    for i, (rows, pay) in enumerate(hs):
        host = E.check_rw_history(rows, pay, model)
        g = res[i]
        assert {1: True, 0: False, 2: "unknown"}[int(g["valid"])] == host["valid?"], (i, model, g, host)
        assert int(g["ok_count"]) == host["ok-count"] and int(g["attempt_count"]) == host["txn-count"], (i, g, host)
        hb = int(host["anomaly-bits"])
        assert int(g["error_count"]) & ~hb == 0, (i, model, g, host)
        if int(g["error_count"]) != hb or int(g["lost_count"]) != host["edge-count"]:
            decided += 1    # not the host's record: the device decided this one, so nothing proscribed may be in it
            assert host["valid?"] is not False and hb & proscribed == 0, (i, model, g, host)
    return res, n_host, decided


@pytest.mark.parametrize("kw", [dict(), dict(nemesis=("partition",), nemesis_interval=2.0), dict(node_count=5, latency=10, latency_dist="uniform"),
                                dict(node_count=3, nemesis=("partition",), nemesis_interval=2.0, latency=20, latency_dist="exponential", p_loss=0.02)])
def test_device_pass_agrees_with_host_on_oracle_histories(lib, kw):
    cfg = _cfg(**kw)
    o = O.run(cfg, 0, 6)
    hs = [o.history(i) for i in range(6)]
    for model in MODELS:
        res, n_host, decided = _agree(hs, model)
        if model == "read-committed":     # what the reference's demo asks for (core.clj:115-121): decided on the device
            assert n_host == 0 and (res["valid"] == 1).all()


def test_device_pass_hands_proscribed_anomalies_to_the_host(lib):
    """corrupt reads / failed writers of real histories: G1a, G1b, internal, cycles — the verdicts stay the host's"""
    cfg = _cfg(node_count=3, rate=60.0, time_limit=4.0, latency=5)
    o = O.run(cfg, 0, 3)
    rng = np.random.default_rng(5)
    hs = []
    for i in range(3):
        ops = E.decode_history(*o.history(i), 3, RW)
        for trial in range(12):
            mut = [dict(op, value=[list(m) for m in op["value"]]) if op["f"] == ":txn" else op for op in ops]
            for _ in range(1 + trial % 3):
                cands = [op for op in mut if op["type"] == ":ok" and op["f"] == ":txn"]
                op = cands[rng.integers(len(cands))]
                m = op["value"][rng.integers(len(op["value"]))]
                if m[0] == R:
                    m[2] = None if rng.integers(4) == 0 else int(rng.integers(1, 6))
                elif rng.integers(3) == 0:
                    op["type"] = ":fail"
            hs.append(E.encode_txn_history(mut, rw=True))
    invalid = 0
    for model in MODELS:
        res, n_host, decided = _agree(hs, model)
        invalid += int((res["valid"] == 0).sum())
    assert invalid > 0


def test_engine_check_uses_the_device_pass(lib):
    cfg = _cfg(node_count=3, rate=80.0, time_limit=5.0, latency=5)
    with E.Engine(cfg) as eng:
        eng.run(0, 16)
        eng.check()
        eng.fetch()
        res = eng.check_results().copy()
        assert eng.check_host_rechecks() == 0 and (res["valid"] == 1).all()
        for i in range(16):
            rows, pay = eng.raw_history(i)
            host = E.check_rw_history(rows.copy(), pay.copy(), "read-committed")
            assert host["valid?"] is True and int(res[i]["ok_count"]) == host["ok-count"]
