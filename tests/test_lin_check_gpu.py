"""lin-kv's linearizability search on the device (csrc/lin_check_dev.hip, behind msim_check) against the host search
(csrc/lin_check.cpp, msim_check_lin_kv_rows — itself pinned to the pure-Python restatement in tests/test_oracle_raft.py):
every field of every history's result, on engine histories, on corrupted ones, and on shapes that exceed what a wavefront holds."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

pytestmark = pytest.mark.gpu

FIELDS = ("valid", "attempt_count", "error_count", "op_count", "ok_count", "fail_count", "info_count")


def _host(rows):
    res = A.CheckResult()
    rows = np.ascontiguousarray(rows)
    assert A.load().msim_check_lin_kv_rows(rows.ctypes.data, len(rows), C.byref(res)) == 0
    return res


def _same(dev, host, ctx):
    for f in FIELDS:
        assert int(dev[f]) == int(getattr(host, f)), (ctx, f, int(dev[f]), int(getattr(host, f)))


@pytest.mark.parametrize("kw", [
    dict(bin="raft", node_count=5, rate=30, time_limit=30),
    dict(bin="raft", node_count=5, rate=30, time_limit=40, latency=10, nemesis=["partition"], nemesis_interval=6),
    dict(bin="raft", node_count=3, concurrency=6, rate=60, time_limit=30, latency=20, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=4),
    dict(node_count=5, rate=100, time_limit=20, latency=5),                                          # lin-kv over the proxy node
    dict(node_count=3, rate=200, time_limit=20, latency=30, latency_dist="exponential", p_loss=0.1),  # timeouts: many indeterminate calls
])
def test_device_search_equals_host_search_on_engine_histories(lib, kw):
    cfg = E.test_config("lin-kv", seed=41, **kw)
    n = 48
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.check()
        res = eng.check_results()
        rechecks = eng.check_host_rechecks()
        eng.fetch()
        for i in range(n):
            rows, _ = eng.raw_history(i)
            h = _host(rows)
            if eng.meta(i).flags:
                assert int(res[i]["valid"]) == 0
                continue
            _same(res[i], h, (kw, i))
        assert (res["attempt_count"] >= 1).all()
        assert rechecks <= n // 2, rechecks   # 64, then 512 configurations in registers are enough for most histories


def _histories(n=12, **kw):
    cfg = E.test_config("lin-kv", seed=7, **kw)
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.fetch()
        return [eng.raw_history(i)[0].copy() for i in range(n)]


def test_corrupted_histories_are_caught_like_on_the_host(lib):
    hs = _histories(bin="raft", node_count=5, rate=30, time_limit=30, latency=10, nemesis=["partition"], nemesis_interval=6)
    rng = np.random.default_rng(5)
    bad = []
    for rows in hs:
        rows2 = rows.copy()
        typ, f = rows2["packed"] & 3, (rows2["packed"] >> 2) & 31
        cand = np.flatnonzero((typ == A.T_OK) & (f == A.F_READ))
        for j in rng.choice(cand, size=min(3, len(cand)), replace=False):   # flip the value seen by some :ok reads
            v = (int(rows2["value"][j]) >> 8) & 0xFF
            rows2["value"][j] = (int(rows2["value"][j]) & ~0xFF00) | ((((v + 1) % 5) if v != 0xFF else 1) << 8)
        bad.append(rows2)
    dev = E.check_lin_kv_batch(hs + bad)
    n_invalid = 0
    for i, rows in enumerate(hs + bad):
        _same(dev[i], _host(rows), i)
        n_invalid += int(dev[i]["valid"]) == 0
    assert (dev["valid"][:len(hs)] == 1).all()
    assert n_invalid > 0


def _row(t, typ, f, proc, key, v1=0xFF, v2=0xFF):
    return (t, typ | (f << 2) | (proc << 12), key | (v1 << 8) | (v2 << 16))


def _mk(rows):
    a = np.zeros(len(rows), dtype=E.OP_DT)
    for i, (t, pk, val) in enumerate(rows):
        a[i]["time_len"] = t; a[i]["packed"] = pk; a[i]["value"] = val
    return a


def test_shapes_beyond_a_wavefront_go_to_the_host_and_agree(lib):
    hs = []
    # 70 calls pending at once on one key: more than the 64 slots of either search -> :unknown on both sides
    rows = [_row(i, A.T_INVOKE, A.F_WRITE, i, 0, i % 5) for i in range(70)]
    hs.append(_mk(rows))
    # indeterminate writes and cas, then a sequential reader: every read sees the value of a write nobody has seen yet (so a
    # linearization exists), or — every other history — once a value nobody wrote: wide searches, dominance pruning at work
    rng = np.random.default_rng(3)
    for trial in range(8):
        rows, t = [], 0
        writes = []
        for i in range(14):   # calls that never return
            f = A.F_WRITE if i < 8 or rng.random() < 0.5 else A.F_CAS
            v1, v2 = int(rng.integers(5)), int(rng.integers(5))
            if f == A.F_WRITE:
                writes.append(v1)
            rows.append(_row(t, A.T_INVOKE, f, 1000 + i, 0, v1, v2)); t += 1
            rows.append(_row(t, A.T_INFO, f, 1000 + i, 0, v1, v2)); t += 1
        order = list(rng.permutation(len(writes)))[:6]
        for n_read, wi in enumerate(order):
            v = writes[wi] if not (trial % 2 == 1 and n_read == 4) else 9
            rows.append(_row(t, A.T_INVOKE, A.F_READ, 1, 0)); t += 1
            rows.append(_row(t, A.T_OK, A.F_READ, 1, 0, v)); t += 1
        hs.append(_mk(rows))
    # interleaved keys, double invocations, completions without invocations: the pairing rules
    rows = [_row(0, A.T_INVOKE, A.F_WRITE, 1, 3, 1), _row(1, A.T_INVOKE, A.F_WRITE, 1, 4, 2), _row(2, A.T_OK, A.F_WRITE, 1, 3, 1),
            _row(3, A.T_OK, A.F_READ, 2, 3, 1), _row(4, A.T_INVOKE, A.F_READ, 2, 3), _row(5, A.T_OK, A.F_READ, 2, 3, 1),
            _row(6, A.T_INVOKE, A.F_READ, 3, 4), _row(7, A.T_OK, A.F_READ, 3, 4, 0xFF), _row(8, A.T_INVOKE, A.F_WRITE, 1, 3, 2),
            _row(9, A.T_INVOKE, A.F_WRITE, 1, 3, 4), _row(10, A.T_FAIL, A.F_WRITE, 1, 3, 4), _row(11, A.T_INVOKE, A.F_READ, 2, 3), _row(12, A.T_OK, A.F_READ, 2, 3, 2)]
    hs.append(_mk(rows))
    hs.append(_mk([]))
    dev = E.check_lin_kv_batch(hs)
    for i, rows in enumerate(hs):
        _same(dev[i], _host(rows), i)
    assert int(dev[0]["valid"]) == 2
    assert [int(v) for v in dev["valid"][1:9]] == [1, 0] * 4
