"""Edge cases of the hot path through the C-ABI, engine vs oracle (bit-exact): empty workloads, single-node clusters,
the widest clusters one wavefront holds, far-away instance ids, capacity overflows (reported identically by both, never
silently truncated), repeated and asynchronous runs."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O
from test_parity_gpu import _compare

pytestmark = pytest.mark.gpu


def test_no_client_operations(lib):
    """--rate 0: (gen/sleep time-limit), core.clj:69 — only init/topology traffic, then the final reads."""
    ora = _compare(E.test_config("broadcast", node_count=5, rate=0, time_limit=3, seed=1), 0, 4)
    assert (ora.meta["n_rows"] == 10).all()          # 5 final reads, invoke + ok
    ora = _compare(E.test_config("echo", node_count=3, rate=0, time_limit=2, seed=1), 0, 2)
    assert (ora.meta["n_rows"] == 0).all()           # an empty history
    _compare(E.test_config("lin-kv", bin="raft", node_count=3, rate=0, time_limit=5, seed=1), 0, 2)


def test_single_node_clusters(lib):
    _compare(E.test_config("broadcast", node_count=1, rate=20, time_limit=3, seed=2), 0, 4)
    _compare(E.test_config("g-set", node_count=1, rate=20, time_limit=6, seed=2), 0, 4)
    _compare(E.test_config("echo", node_count=1, rate=20, time_limit=3, seed=2), 0, 4)
    _compare(E.test_config("txn-list-append", node_count=1, rate=20, time_limit=3, latency=3, seed=2), 0, 4)
    _compare(E.test_config("pn-counter", node_count=1, rate=20, time_limit=6, seed=2), 0, 4)


def test_widest_clusters_of_each_layout(lib):
    _compare(E.test_config("broadcast", node_count=32, rate=50, time_limit=3, topology="total", latency=5, seed=3), 0, 2)
    _compare(E.test_config("broadcast", node_count=32, rate=50, time_limit=3, topology="line", seed=3), 0, 2)
    _compare(E.test_config("broadcast", node_count=16, concurrency=48, rate=100, time_limit=3, latency=10, seed=3), 0, 2)   # 16 + 48 = 64 lanes
    _compare(E.test_config("txn-list-append", node_count=31, rate=200, time_limit=3, latency=2, seed=3), 0, 2)             # 31 nodes + the service
    _compare(E.test_config("g-set", node_count=127, rate=100, time_limit=6, latency=20, latency_dist="uniform", seed=3), 0, 1)
    _compare(E.test_config("broadcast", node_count=127, rate=100, time_limit=4, latency=20, latency_dist="uniform", seed=3), 0, 1)


def test_far_away_instance_ids_and_large_seeds(lib):
    cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=3, latency=5, latency_dist="exponential", seed=2**63 + 12345)
    _compare(cfg, 2**40 + 7, 4)
    _compare(cfg, 2**32 - 2, 4)   # crosses 2^32


def _flags_agree(cfg, n):
    ora = O.run(cfg, 0, n)
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.check()
        eng.fetch()
        res = eng.check_results()
        for i in range(n):
            assert eng.meta(i).flags == int(ora.meta["flags"][i]) != 0, (i, eng.meta(i).flags, int(ora.meta["flags"][i]))
            assert int(res["valid"][i]) == 0   # a truncated history is never reported valid
    return ora


def test_capacity_overflows_are_flagged_not_truncated(lib):
    ora = _flags_agree(E.test_config("broadcast", node_count=5, rate=50, time_limit=5, seed=4, max_rows=64), 4)
    assert (ora.meta["flags"] & A.FLAG_ROWS_OVERFLOW).all()
    ora = _flags_agree(E.test_config("broadcast", node_count=5, rate=50, time_limit=5, seed=4, max_payload_words=40), 4)
    assert (ora.meta["flags"] & A.FLAG_PAYLOAD_OVERFLOW).all()
    ora = _flags_agree(E.test_config("broadcast", node_count=5, rate=50, time_limit=5, seed=4, max_values=32), 4)
    assert (ora.meta["flags"] & A.FLAG_VALUES_OVERFLOW).all()
    ora = _flags_agree(E.test_config("broadcast", node_count=9, rate=200, time_limit=4, latency=200, topology="total", seed=4,
                                     inbox_capacity=1, spill_capacity=1), 4)
    assert (ora.meta["flags"] & A.FLAG_INBOX_OVERFLOW).all()


@pytest.mark.parametrize("kw", [dict(bin="raft", p_loss=0.97), dict(bin="raft", p_loss=0.97, concurrency=12), dict(p_loss=0.95)])
def test_lin_kv_runs_out_of_keys_after_256(lib, kw):
    """lin-kv keys travel in 8 bits of an op's value; [upstream] jepsen.tests.linearizable-register retires a key after 20 processes,
    and a process is retired by every :info.  Nearly total loss for half an hour of virtual time uses up 256 keys: the run stops with
    VALUES_OVERFLOW instead of aliasing key 256 onto key 0 — in the oracle, in the Raft kernels (four clusters per wavefront, and one:
    12 clients do not fit a 16-lane group) and in the proxy kernel alike, histories identical up to that point."""
    kw = dict(dict(node_count=3, concurrency=6, rate=200, time_limit=1700, seed=3, max_rows=40000), **kw)
    cfg = E.test_config("lin-kv", **kw)
    ora = _flags_agree(cfg, 3)
    assert (ora.meta["flags"] == A.FLAG_VALUES_OVERFLOW).all()
    with E.Engine(cfg) as eng:   # and everything up to the stop is the oracle's
        eng.run(0, 3)
        eng.fetch()
        for i in range(3):
            m, om = eng.meta(i), ora.meta[i]
            assert (m.n_rows, m.n_payload_words, m.n_rounds) == (om["n_rows"], om["n_payload_words"], om["n_rounds"])
            assert eng.raw_history(i)[0].tobytes() == ora.history(i)[0].tobytes()
    assert all(int((ora.history(i)[0]["value"] & 0xFF).max()) == 255 for i in range(3))


def test_repeated_runs_reuse_the_context(lib):
    cfg = E.test_config("g-set", node_count=5, rate=20, time_limit=6, latency=10, seed=5)
    ora = O.run(cfg, 0, 12)
    with E.Engine(cfg) as eng:
        for first, n in ((0, 4), (4, 8), (2, 3)):      # growing and shrinking batches
            eng.run(first, n)
            eng.fetch()
            for i in range(n):
                rows, pay = eng.raw_history(i)
                orows, opay = ora.history(first + i)
                assert rows.tobytes() == orows.tobytes() and pay.tobytes() == opay.tobytes()


def test_api_misuse_is_reported(lib):
    cfg = E.test_config("echo", node_count=3, rate=5, time_limit=2, seed=6)
    with E.Engine(cfg) as eng:
        with pytest.raises(E.EngineError, match="before msim_run"):
            eng.fetch()
        with pytest.raises(E.EngineError, match="before msim_run"):
            eng.check()
        with pytest.raises(E.EngineError, match="n_instances"):
            eng.run(0, 0)
        eng.run(0, 2)
        with pytest.raises(E.EngineError):
            eng.raw_history(2)     # out of range
    with pytest.raises(E.EngineError, match="one wavefront|at most 32 nodes"):
        E.Engine(E.test_config("echo", node_count=40, rate=5, time_limit=2))   # (the broadcast programs, g-set and the counters go up to 127 nodes)
