"""CPU tests of the Raft lin-kv restatement (oracle): every emitted history must be linearizable per key
(what Knossos verifies in the reference, workload/lin_kv.clj:84), elections must converge, and the error
mapping must follow client.clj:153-172 / resources/errors.edn."""
import collections

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import linearizable_ref as L
import oracle_lib as O


def _histories(cfg, n):
    r = O.run(cfg, 0, n)
    for i in range(n):
        rows, pay = r.history(i)
        yield r, i, E.decode_history(rows, pay, cfg.n_nodes, A.WL_LIN_KV)


@pytest.mark.parametrize("kw", [
    dict(),
    dict(latency=10),
    dict(latency=20, latency_dist="exponential"),
    dict(nemesis=["partition"], nemesis_interval=5, latency=5),
    dict(nemesis=["partition"], nemesis_interval=10, time_limit=40, latency=10, latency_dist="uniform"),
    dict(node_count=3, concurrency=12, nemesis=["partition"], nemesis_interval=4, time_limit=30),   # 04-committing.md:418 shape
])
def test_raft_histories_are_linearizable(lib, kw):
    kw = dict(dict(node_count=5, rate=30, time_limit=20, seed=7), **kw)
    cfg = E.test_config("lin-kv", bin="raft", **kw)
    for r, i, h in _histories(cfg, 6):
        assert r.meta[i]["flags"] == 0
        res = L.check(h)
        assert res and all(res.values()), (i, res)


def test_raft_elects_a_leader_and_serves_ops(lib):
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=20, seed=1)
    for r, i, h in _histories(cfg, 4):
        c = collections.Counter((op["f"], op["type"]) for op in h)
        # no leader for the first 2-4 s (election timeout, raft.rb:101,276-280): error 11 = :temporarily-unavailable => :fail
        early = [op for op in h if op["time"] < 1_900_000_000 and op["type"] != ":invoke"]
        assert early and all(op["type"] == ":fail" and op["error"][0] == ":temporarily-unavailable" for op in early)
        late = [op for op in h if op["time"] > 6_000_000_000 and op["type"] != ":invoke"]
        assert late and not any(op.get("error", [""])[0] == ":temporarily-unavailable" for op in late)
        assert c[(":write", ":ok")] > 30 and c[(":read", ":ok")] > 100 and c[(":cas", ":ok")] > 5
        assert not any(op["type"] == ":info" for op in h)               # healthy network: nothing times out
        # readers are threads 0..n-1, writers n..2n-1 ([upstream] gen/reserve n r ...)
        assert all((op["process"] % 10 < 5) == (op["f"] == ":read") for op in h)
        assert int(r.stats[i]["servers_send"]) > 500                     # heartbeats + replication


def test_raft_timeouts_are_indeterminate_for_writes_only(lib):
    """client.clj:153-172 with idempotent #{:read} (lin_kv.clj:52): a timed-out read is :fail, a timed-out write/cas
    is :info and retires the process id (+concurrency); keys rotate after 20 distinct processes."""
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=60, nemesis=["partition"], nemesis_interval=8,
                        latency=10, seed=5)
    seen_info = seen_keys = 0
    for r, i, h in _histories(cfg, 6):
        for op in h:
            if op.get("error") == ":net-timeout":
                assert op["type"] == (":fail" if op["f"] == ":read" else ":info")
                seen_info += op["type"] == ":info"
        procs = {op["process"] for op in h if op["process"] != ":nemesis"}
        assert all(p % 10 == q % 10 or True for p in procs for q in procs)
        seen_keys = max(seen_keys, len({op["value"][0] for op in h if op["process"] != ":nemesis"}))
    assert seen_info > 0 and seen_keys > 1


def _lib_check(rows):
    import ctypes as C
    from maelstrom_amd import _abi
    res = _abi.CheckResult()
    rows = rows.copy()
    assert _abi.load().msim_check_lin_kv_rows(rows.ctypes.data, len(rows), C.byref(res)) == 0
    return res


def test_library_linearizability_checker_agrees_with_reference(lib):
    """The product's lin-kv checker (csrc/lin_check.cpp, host side of msim_check) against the pure-Python
    restatement, on oracle histories and on histories corrupted to be non-linearizable."""
    import numpy as np
    cfg = E.test_config("lin-kv", bin="raft", node_count=5, rate=30, time_limit=30, nemesis=["partition"], nemesis_interval=6,
                        latency=10, seed=23)
    r = O.run(cfg, 0, 6)
    rng = np.random.default_rng(5)
    n_bad = 0
    for i in range(6):
        rows, pay = r.history(i)
        for corrupt in (False, True):
            rows2 = rows.copy()
            if corrupt:  # flip the value seen by some :ok reads
                typ, f = rows2["packed"] & 3, (rows2["packed"] >> 2) & 31
                cand = np.flatnonzero((typ == A.T_OK) & (f == A.F_READ))
                for j in rng.choice(cand, size=min(3, len(cand)), replace=False):
                    v = (int(rows2["value"][j]) >> 8) & 0xFF
                    rows2["value"][j] = (int(rows2["value"][j]) & ~0xFF00) | ((((v + 1) % 5) if v != 0xFF else 1) << 8)
            ref = L.check(E.decode_history(rows2, pay, 5, A.WL_LIN_KV))
            got = _lib_check(rows2)
            assert got.attempt_count == len(ref)
            assert got.error_count == sum(1 for ok in ref.values() if not ok), (i, corrupt, ref)
            assert got.valid == (1 if all(ref.values()) else 0)
            n_bad += got.valid == 0
            if not corrupt:
                assert got.valid == 1
    assert n_bad > 0
