"""pn-counter workload (workload/pn_counter.clj) on the CPU side.

KAT-9: the checker is pinned by the reference's OWN test vectors (test/maelstrom/workload/pn_counter_test.clj:10-36):
same histories in, same :valid? / :final-reads / :acceptable out.  The oracle's node program is pinned by golden
transitions recorded from demo/js/crdt_pn_counter.js (tests/test_golden_transitions.py)."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _check(ops):
    return E.check_pn_history(E.encode_pn_history(ops))


def test_kat9_empty_history():
    res = _check([])
    assert res["valid?"] is True and res["final-reads"] == [] and res["acceptable"] == [[0, 0]]   # pn_counter_test.clj:10-15


def test_kat9_definite_adds():
    res = _check([{"type": ":ok", "f": ":add", "value": 2}, {"type": ":ok", "f": ":add", "value": 3},
                  {"type": ":ok", "f": ":read", "final?": True, "value": 5},
                  {"type": ":ok", "f": ":read", "final?": True, "value": 4}])
    # pn_counter_test.clj:17-25: {:valid? false, :errors [the read of 4], :final-reads [5 4], :acceptable [[5 5]]}
    assert res["valid?"] is False and res["error-count"] == 1
    assert res["final-reads"] == [5, 4] and res["acceptable"] == [[5, 5]]


def test_kat9_indefinite_adds():
    res = _check([{"type": ":ok", "f": ":add", "value": 10}, {"type": ":info", "f": ":add", "value": 5},
                  {"type": ":info", "f": ":add", "value": -1}, {"type": ":info", "f": ":add", "value": -1},
                  {"type": ":ok", "f": ":read", "final?": True, "value": 11},
                  {"type": ":ok", "f": ":read", "final?": True, "value": 15}])
    # pn_counter_test.clj:27-36: {:valid? false, :errors [the read of 11], :final-reads [11 15], :acceptable [[8 10] [13 15]]}
    assert res["valid?"] is False and res["error-count"] == 1
    assert res["final-reads"] == [11, 15] and res["acceptable"] == [[8, 10], [13, 15]]


def test_non_final_reads_and_failed_adds_do_not_count():
    res = _check([{"type": ":ok", "f": ":add", "value": 4}, {"type": ":fail", "f": ":add", "value": 100},
                  {"type": ":ok", "f": ":read", "value": 0}, {"type": ":ok", "f": ":read", "final?": True, "value": 4}])
    assert res["valid?"] is True and res["acceptable"] == [[4, 4]] and res["final-reads"] == [4]


@pytest.mark.parametrize("kw", [dict(), dict(latency=20), dict(latency=50, latency_dist="exponential", p_loss=0.1),
                                dict(latency=10, nemesis=["partition"], nemesis_interval=4, time_limit=20)])
def test_oracle_histories_converge_and_pass_the_checker(kw):
    """After the 10 s quiesce (two replicate ticks) every node's final read is the sum of the completed adds (+ some
    subset of the timed-out ones): the CRDT converged (pn_counter.rb merge = element-wise max)."""
    base = dict(node_count=5, rate=20, time_limit=12, seed=17)
    base.update(kw)
    cfg = E.test_config("pn-counter", **base)
    assert cfg.node_program == A.NODE_PN_COUNTER and cfg.max_values == 64 * 5
    r = O.run(cfg, 0, 4)
    for i in range(4):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, 5, A.WL_PN_COUNTER) if o["process"] != ":nemesis"]
        adds = [o for o in ops if o["f"] == ":add" and o["type"] == ":invoke"]
        assert adds and all(-5 <= o["value"] <= 4 for o in adds)               # (- (rand-int 10) 5)
        finals = [o for o in ops if o.get("final?") and o["type"] == ":ok"]
        assert len(finals) >= 1 and all(o["f"] == ":read" for o in finals)
        res = E.check_pn_history(rows)
        assert res["valid?"] is True, res
        if not kw.get("p_loss"):
            total = sum(o["value"] for o in ops if o["f"] == ":add" and o["type"] == ":ok")
            assert len(finals) == 5 and {o["value"] for o in finals} == {total}
        # g-counter style message accounting (KAT-4 shape): every tick N*(N-1) replicate messages
        assert int(r.stats["servers_send"][i]) % (5 * 4) == 0


def test_g_counter_never_adds_a_negative_delta():
    """workload/g_counter.clj:37-41: the pn-counter workload with negative adds filtered out of the generator."""
    cfg = E.test_config("g-counter", node_count=5, rate=50, time_limit=10, latency=10, seed=3)
    assert cfg.node_program == A.NODE_PN_COUNTER
    r = O.run(cfg, 0, 3)
    for i in range(3):
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, 5, A.WL_G_COUNTER) if o["process"] != ":nemesis"]
        adds = [o["value"] for o in ops if o["f"] == ":add" and o["type"] == ":invoke"]
        reads = [o for o in ops if o["f"] == ":read" and o["type"] == ":invoke" and not o.get("final?")]
        assert adds and min(adds) >= 0 and max(adds) <= 4
        assert 1.3 < len(reads) / len(adds) < 3.2      # 1/2 : 1/4 after the filter
        assert E.check_pn_history(rows)["valid?"] is True
        finals = [o["value"] for o in ops if o.get("final?") and o["type"] == ":ok"]
        assert finals == [sum(o["value"] for o in ops if o["f"] == ":add" and o["type"] == ":ok")] * 5
