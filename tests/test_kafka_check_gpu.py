"""kafka: the device pass of msim_check (csrc/kafka_check_dev.hip — the checker's tables in LDS, one workgroup per history) against the
host checker (csrc/kafka_check.cpp, pinned by tests/test_kafka.py): it proves clean histories clean with the host's counts, and never
calls a history clean that the host finds an anomaly in — every such history goes to the host checker."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

import oracle_lib as O
from test_kafka import _cfg, _h, _ops, _poll, _send

pytestmark = pytest.mark.gpu

FIELDS = ("valid", "attempt_count", "stable_count", "lost_count", "never_read_count", "duplicated_count", "error_count", "op_count", "ok_count",
          "fail_count", "info_count")


def _host(rows, pay):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows); pay = np.ascontiguousarray(pay, dtype=np.uint32)
    rc = A.load().msim_check_kafka_rows(rows.ctypes.data, len(rows), pay.ctypes.data, len(pay), res)
    assert rc == 0
    return {f: int(getattr(res, f)) for f in FIELDS}


def _same(rec, host):
    return all(int(rec[f]) == host[f] for f in FIELDS)


@pytest.mark.parametrize("kw", [dict(), dict(latency=0, rate=120.0), dict(node_count=5, latency=20, latency_dist="exponential"),
                                dict(nemesis=("partition",), nemesis_interval=2.0), dict(p_loss=0.05, latency=10, latency_dist="uniform"),
                                dict(key_count=2, max_writes_per_key=40, rate=150.0, node_count=2), dict(node_count=5, latency=150, rate=100.0)])
def test_device_pass_proves_oracle_histories_clean_with_the_hosts_counts(lib, kw):
    cfg = _cfg(**kw)
    o = O.run(cfg, 0, 6)
    hs = [o.history(i) for i in range(6)]
    out, n_host = E.check_kafka_batch(hs, cfg.n_nodes)
    assert n_host == 0
    acked = 0
    for (rows, pay), rec in zip(hs, out):
        host = _host(rows, pay)
        assert host["valid"] == 1 and host["error_count"] == 0
        assert _same(rec, host), (rec, host)
        acked += host["stable_count"]
    assert acked > 20   # (the counts compared are not all zero)


def _anomalies():
    a = [{"type": ":invoke", "process": 0, "f": ":assign", "value": ["5"]}, {"type": ":ok", "process": 0, "f": ":assign", "value": ["5"]}]
    ref = _h(*(sum((_send(1, "5", m, m - 1) for m in range(1, 10)), [])), *_poll(0, {"5": [[4, 5], [5, 6]]}), *_poll(0, {"5": [[8, 9]]}),
             *_poll(2, {"5": [[o_, o_ + 1] for o_ in range(9)]}))
    bad = {
        "lost": _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_send(0, "1", 3, 2), *_poll(1, {"1": [[0, 1]]}), *_poll(2, {"1": [[2, 3]]})),
        "nm_poll": _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[0, 1], [1, 2]]}), *_poll(1, {"1": [[1, 2]]})),
        "nm_send": _h(*_send(0, "1", 1, 3), *_send(0, "1", 2, 3)),
        "int_skip": _h(*(sum((_send(0, "1", m, m - 1) for m in range(1, 6)), [])), *_poll(1, {"1": [[0, 1], [3, 4]]})),
        "int_nm": _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[1, 2], [0, 1]]})),
        "incons": _h(*_send(0, "1", 1, 0), *_poll(1, {"1": [[0, 7]]})),
        "dup": _h(*_send(0, "1", 1, 0), *_poll(1, {"1": [[0, 1], [1, 1]]})),
        "aborted": _h(*_send(0, "1", 1, typ=":fail"), *_poll(1, {"1": [[0, 1]]})),
        "poll_skip": ref,                                                                 # workload/kafka.clj:42-60
        # a process that comes back after another process of its worker: not one stream per worker — the host's to judge
        "returning": _h(*_send(0, "1", 1, 0), *_send(3, "1", 2, 1), *_send(0, "1", 3, 0)),
    }
    good = {
        "unobserved": _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[0, 1]]})),
        "across_assign": _h(*ref[:20], *a, *ref[20:]),                                     # :64-70
        "sparse": _h(*_send(1, "5", 5, 4), *_send(1, "5", 6, 5), *_send(1, "5", 9, 8), *_poll(0, {"5": [[4, 5], [5, 6]]}), *_poll(0, {"5": [[8, 9]]})),
        "info_send": _h(*_send(0, "1", 1, typ=":info"), *_poll(1, {"1": [[0, 1]]})),
        "next_process": _h(*_send(0, "1", 1, 5), *_send(3, "1", 2, 0)),                    # worker 0 crashed: process 3 starts afresh
        "nothing": _h({"type": ":invoke", "process": 0, "f": ":poll", "value": [[":poll"]]}, {"type": ":info", "process": 0, "f": ":poll", "value": [[":poll"]]}),
    }
    return bad, good


def test_hand_made_anomalies_go_to_the_host_and_clean_ones_stay(lib):
    bad, good = _anomalies()
    for name, ops in bad.items():
        rows, pay = E.encode_kafka_history(ops)
        out, n_host = E.check_kafka_batch([(rows, pay)], 3)
        host = _host(rows, pay)
        assert n_host == 1, name
        assert _same(out[0], host), (name, out[0], host)
        assert host["error_count"] != 0 or name == "returning", name
    for name, ops in good.items():
        rows, pay = E.encode_kafka_history(ops)
        out, n_host = E.check_kafka_batch([(rows, pay)], 3)
        host = _host(rows, pay)
        assert host["error_count"] == 0 and n_host == 0, (name, host)
        assert _same(out[0], host), (name, out[0], host)
    # all of them in one launch: the verdicts do not depend on the neighbours
    hs = [E.encode_kafka_history(ops) for ops in list(bad.values()) + list(good.values())]
    out, n_host = E.check_kafka_batch(hs, 3)
    assert n_host == len(bad)
    for (rows, pay), rec in zip(hs, out):
        assert _same(rec, _host(rows, pay))


def test_corrupted_histories_equal_the_host_checker(lib):
    cfg = _cfg(node_count=3, rate=80.0, time_limit=5.0)
    o = O.run(cfg, 0, 3)
    rng = np.random.default_rng(9)
    hs, unclean = [], 0
    for i in range(3):
        ops = _ops(*o.history(i), 3)
        for trial in range(24):
            mut = [dict(op) for op in ops]
            for _ in range(1 + trial % 3):
                op = mut[int(rng.integers(len(mut)))]
                if op["f"] == ":poll" and op["type"] == ":ok" and len(op["value"][0]) > 1:
                    msgs = {k: [list(p) for p in v] for k, v in op["value"][0][1].items()}
                    ks = [k for k in msgs if msgs[k]]
                    if ks:
                        k = ks[int(rng.integers(len(ks)))]
                        what = int(rng.integers(4))
                        if what == 0:
                            del msgs[k][int(rng.integers(len(msgs[k])))]
                        elif what == 1:
                            msgs[k][-1][0] += int(rng.integers(1, 4))
                        elif what == 2:
                            msgs[k][-1][1] = int(rng.integers(1, 30))
                        else:
                            msgs[k] = msgs[k][1:] + msgs[k][:1]                       # out of order inside one poll
                    op["value"] = [[":poll", msgs]]
                elif op["f"] == ":send" and op["type"] == ":ok":
                    v = op["value"][0]
                    op["value"] = [[":send", v[1], [max(0, v[2][0] - int(rng.integers(0, 3))), v[2][1]]]]
                elif op["f"] == ":send" and op["type"] == ":info" and trial % 4 == 0:
                    op["type"] = ":fail"                                                # its message may have been polled: aborted read
            mut = [dict(op, index=n_) for n_, op in enumerate(mut)]
            hs.append(E.encode_kafka_history(mut))
    out, n_host = E.check_kafka_batch(hs, 3)
    for (rows, pay), rec in zip(hs, out):
        host = _host(rows, pay)
        assert _same(rec, host), (rec, host)
        unclean += host["error_count"] != 0
    assert unclean >= 20 and n_host >= unclean   # (every unclean history was the host's; the device may hand over clean ones too)
    # one by one: a history with an anomaly never stays on the device
    for rows, pay in hs[::5]:
        rec, nh = E.check_kafka_batch([(rows, pay)], 3)
        if _host(rows, pay)["error_count"] != 0:
            assert nh == 1


def test_engine_check_on_the_device_equals_the_host_checker(lib):
    cfg = E.test_config("kafka", node_count=4, rate=100, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2, seed=12)
    with E.Engine(cfg) as eng:
        eng.run(0, 96)
        eng.check()
        dev = eng.check_results().copy()
        assert eng.check_host_rechecks() == 0
        eng.set_dev_flags(0x800)   # the host checker
        eng.check()
        host = eng.check_results().copy()
        for f in FIELDS:
            assert (dev[f] == host[f]).all(), f
        assert (dev["valid"] == 1).all() and (dev["stable_count"] > 20).all()
