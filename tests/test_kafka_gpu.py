"""kafka workload (workload/kafka.clj over demo/clojure/kafka.clj): the HIP engine against the CPU oracle, bit for bit."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
from test_parity_gpu import _compare

pytestmark = pytest.mark.gpu

SHAPES = [
    dict(node_count=3, rate=50, time_limit=5, latency=5, seed=3),
    dict(node_count=5, rate=100, time_limit=6, latency=0, seed=4),
    dict(node_count=5, rate=100, time_limit=6, latency=20, latency_dist="exponential", seed=5),
    dict(node_count=4, rate=60, time_limit=8, latency=10, latency_dist="uniform", p_loss=0.05, seed=6),
    dict(node_count=5, rate=80, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=3, seed=7),
    dict(node_count=2, rate=200, time_limit=4, latency=2, key_count=2, max_writes_per_key=40, seed=8),   # keys retire, chunks fill
    dict(node_count=5, rate=100, time_limit=5, latency=150, seed=9),                                    # slow lin-kv: client timeouts
    dict(node_count=3, rate=30, time_limit=4, latency=3, journal=True, seed=10),
    dict(node_count=7, rate=150, time_limit=8, latency=30, latency_dist="exponential", p_loss=0.02, nemesis=["partition"], nemesis_interval=2, seed=19),
]


@pytest.mark.parametrize("kw", SHAPES)
def test_kafka_parity(lib, kw):
    kw = dict(kw)
    journal = kw.pop("journal", False)
    cfg = E.test_config("kafka", **kw)
    if journal:
        cfg.journal_capacity = 60000
    ora = _compare(cfg, 0, 6)
    if not journal:
        _compare(cfg, 0, 11, dev_flags=0x400)   # eight clusters per wavefront (csrc/kafka8.hip; large batches take it unasked)
    ops = E.decode_history(*ora.history(0), cfg.n_nodes, A.WL_KAFKA)
    fs = {op["f"] for op in ops}
    assert {":send", ":poll", ":assign"} <= fs


def test_kafka_engine_check(lib):
    """msim_check for kafka (csrc/kafka_check.cpp on the host cores) over the histories of a run: all valid, as the one-history entry says"""
    cfg = E.test_config("kafka", node_count=4, rate=100, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2, seed=12)
    with E.Engine(cfg) as eng:
        eng.run(0, 64)
        eng.check()
        eng.fetch()
        res = eng.check_results().copy()
        assert (res["valid"] == 1).all() and (res["error_count"] == 0).all() and (res["stable_count"] > 20).all()
        for i in (0, 17, 63):
            rows, pay = eng.raw_history(i)
            one = E.check_kafka_history(rows.copy(), pay.copy())
            assert one["valid?"] is True and one["acked-count"] == int(res[i]["stable_count"]) and one["unobserved-count"] == int(res[i]["never_read_count"])
