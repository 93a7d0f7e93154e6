"""txn-rw-register (workload/txn_rw_register.clj) over the highly-available-transactions node
(demo/clojure/txn_rw_register_hat.clj) — CPU side: the oracle's restatement behaves like a HAT system, and the host
rw-register checker (msim_check_rw_rows) agrees with an independent Python restatement and with hand-made anomalies."""
import ctypes as C

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

import elle_ref
import oracle_lib as O

RW = A.WL_TXN_RW_REGISTER


def _cfg(**kw):
    args = dict(workload="txn-rw-register", node_count=2, rate=100.0, time_limit=6.0, seed=11)
    args.update(kw)
    return E.test_config(**args)


def _ops(*txns):
    """[(process, type, mops)] -> invoke/completion op maps; completion type None = never completes"""
    ops = []
    for p, typ, req, done in txns:
        ops.append({"type": ":invoke", "process": p, "f": ":txn", "value": req})
        if typ:
            ops.append({"type": typ, "process": p, "f": ":txn", "value": done})
    return ops


def _check(ops, model):
    rows, pay = E.encode_txn_history(ops, rw=True)
    return E.check_rw_history(rows, pay, model)


def test_defaults_follow_the_reference_demo():
    cfg = _cfg()
    # core.clj:115-121: the demo runs txn_rw_register_hat.clj and asks for read-committed
    assert cfg.node_program == A.NODE_TXN_RW_HAT and cfg.consistency_model == A.CM_READ_COMMITTED
    assert cfg.max_txn_length == 4 and cfg.max_writes_per_key == 16 and cfg.replication_words > 0
    with pytest.raises(E.EngineError):
        _cfg(node_count=1)   # replicate-step! would send to nil (txn_rw_register_hat.clj:85-105)


@pytest.mark.parametrize("kw", [dict(), dict(nemesis=("partition",), nemesis_interval=2.0), dict(node_count=5, latency=10, latency_dist="uniform"),
                                dict(node_count=3, nemesis=("partition",), nemesis_interval=2.0, latency=20, latency_dist="exponential", p_loss=0.02)])
def test_oracle_histories_are_read_committed_but_not_serializable(kw):
    cfg = _cfg(**kw)
    o = O.run(cfg, 0, 6)
    weak = 0
    for i in range(6):
        rows, pay = o.history(i)
        assert o.meta[i]["flags"] == 0
        rc = E.check_rw_history(rows, pay, "read-committed")
        assert rc["valid?"] is True, rc
        # whatever is wrong with these histories is an anti-dependency cycle (lost updates, write skew), nothing worse
        assert set(rc["anomalies"]) <= {"G-single", "G2", "realtime"}, rc
        ref = elle_ref.analyse_rw(E.decode_history(rows, pay, cfg.n_nodes, RW))
        assert ref["anomalies"] == set(rc["anomalies"]) and ref["ok-count"] == rc["ok-count"]
        weak += E.check_rw_history(rows, pay, "strict-serializable")["valid?"] is False
    if not kw.get("p_loss"):
        assert weak >= 4   # two nodes taking writes independently: serializability does not survive


def test_txn_semantics_in_the_history():
    """reads return the last write the node has seen (or nil); a transaction sees its own writes; writes echo"""
    cfg = _cfg(node_count=2, rate=50.0)
    o = O.run(cfg, 3, 1)
    ops = E.decode_history(*o.history(0), 2, RW)
    inv = {}
    n_ok = 0
    for op in ops:
        if op["type"] == ":invoke":
            inv[op["process"]] = op["value"]
            continue
        assert op["type"] == ":ok"   # a healthy network loses nothing: total availability (core.clj:119)
        req, done, own = inv[op["process"]], op["value"], {}
        assert len(req) == len(done)
        for (f, k, v), (f2, k2, v2) in zip(req, done):
            assert f == f2 and k == k2
            if f == ":w":
                assert v == v2
                own[k] = v
            elif k in own:
                assert v2 == own[k]
        n_ok += 1
    assert n_ok > 200


def test_replication_converges():
    """once every node has had all its txns acknowledged, the registers agree everywhere (last write wins by timestamp)"""
    lib = O.load()
    cfg = _cfg(node_count=3, rate=2.0, time_limit=20.0, latency=5, p_loss=0.1, nemesis=("partition",), nemesis_interval=2.0)
    converged = 0
    for inst in range(24):
        rows = np.zeros(cfg.max_rows, dtype=O.OP_DT); pay = np.zeros(cfg.max_payload_words, dtype=np.uint32)
        stats = np.zeros(1, dtype=O.STATS_DT); meta = np.zeros(1, dtype=O.META_DT)
        kv = np.zeros((3, cfg.max_values), dtype=np.uint32); lam = np.zeros(3, dtype=np.uint32); npend = np.zeros(3, dtype=np.uint32)
        rc = lib.oracle_hat_state(C.byref(cfg), inst, rows.ctypes.data_as(C.c_void_p), pay.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p),
                                  meta.ctypes.data_as(C.c_void_p), kv.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p), npend.ctypes.data_as(C.c_void_p), None)
        assert rc == 0 and meta[0]["flags"] == 0
        if npend.sum() == 0:
            converged += 1
            assert (kv[0] == kv[1]).all() and (kv[1] == kv[2]).all()
            assert kv.any()
    assert converged >= 3


# ---- the checker on hand-made histories (Adya's phenomena over registers) ----
R, W = ":r", ":w"


def test_checker_clean_history():
    ops = _ops((0, ":ok", [[W, 1, 1]], [[W, 1, 1]]), (1, ":ok", [[R, 1, None], [W, 1, 2]], [[R, 1, 1], [W, 1, 2]]), (0, ":ok", [[R, 1, None]], [[R, 1, 2]]))
    res = _check(ops, "strict-serializable")
    assert res["valid?"] is True and res["anomalies"] == [] and res["txn-count"] == 3


def test_checker_g1a_g1b_internal():
    g1a = _ops((0, ":fail", [[W, 1, 1]], [[W, 1, 1]]), (1, ":ok", [[R, 1, None]], [[R, 1, 1]]))
    assert _check(g1a, "read-committed")["anomalies"] == ["G1a"] and _check(g1a, "read-committed")["valid?"] is False
    assert _check(g1a, "read-uncommitted")["valid?"] is True
    g1b = _ops((0, ":ok", [[W, 1, 1], [W, 1, 2]], [[W, 1, 1], [W, 1, 2]]), (1, ":ok", [[R, 1, None]], [[R, 1, 1]]))
    assert "G1b" in _check(g1b, "read-committed")["anomalies"] and _check(g1b, "read-committed")["valid?"] is False
    internal = _ops((0, ":ok", [[W, 1, 1], [R, 1, None]], [[W, 1, 1], [R, 1, None]]))
    res = _check(internal, "snapshot-isolation")
    assert res["anomalies"] == ["internal"] and res["valid?"] is False
    assert _check(internal, "read-committed")["valid?"] is True   # Adya's PL-2 says nothing about a txn's own reads


def test_checker_cycles_by_class():
    g0 = _ops((0, ":ok", [[R, 1, None], [W, 1, 2], [W, 2, 1]], [[R, 1, 1], [W, 1, 2], [W, 2, 1]]),
              (1, ":ok", [[R, 2, None], [W, 2, 2], [W, 1, 1]], [[R, 2, 1], [W, 2, 2], [W, 1, 1]]))
    assert "G0" in _check(g0, "read-uncommitted")["anomalies"] and _check(g0, "read-uncommitted")["valid?"] is False
    g1c = _ops((0, ":ok", [[W, 1, 1], [R, 2, None]], [[W, 1, 1], [R, 2, 1]]), (1, ":ok", [[W, 2, 1], [R, 1, None]], [[W, 2, 1], [R, 1, 1]]))
    assert _check(g1c, "read-committed")["anomalies"] == ["G1c"] and _check(g1c, "read-committed")["valid?"] is False
    # read skew: T0 misses T1's write of key 1 but sees its write of key 2
    gsingle = _ops((0, ":ok", [[R, 1, None], [R, 2, None]], [[R, 1, None], [R, 2, 1]]), (1, ":ok", [[W, 1, 1], [W, 2, 1]], [[W, 1, 1], [W, 2, 1]]))
    assert _check(gsingle, "read-committed") == {**_check(gsingle, "read-committed"), "valid?": True, "anomalies": ["G-single"]}
    assert _check(gsingle, "snapshot-isolation")["valid?"] is False
    # write skew
    g2 = _ops((0, ":ok", [[R, 1, None], [W, 2, 1]], [[R, 1, None], [W, 2, 1]]), (1, ":ok", [[R, 2, None], [W, 1, 1]], [[R, 2, None], [W, 1, 1]]))
    assert _check(g2, "snapshot-isolation")["anomalies"] == ["G2"] and _check(g2, "snapshot-isolation")["valid?"] is True
    assert _check(g2, "serializable")["valid?"] is False
    for h in (g0, g1c, gsingle, g2):
        assert elle_ref.analyse_rw(h)["anomalies"] == set(_check(h, "strict-serializable")["anomalies"])


def test_checker_realtime_and_cyclic_versions():
    # T0 completes before T1 starts, yet T1 does not see its write: only strict serializability minds
    stale = [{"type": ":invoke", "process": 0, "f": ":txn", "value": [[W, 1, 1]]}, {"type": ":ok", "process": 0, "f": ":txn", "value": [[W, 1, 1]]},
             {"type": ":invoke", "process": 1, "f": ":txn", "value": [[R, 1, None]]}, {"type": ":ok", "process": 1, "f": ":txn", "value": [[R, 1, None]]}]
    res = _check(stale, "strict-serializable")
    assert res["anomalies"] == ["G-single", "realtime"] and res["valid?"] is False
    assert _check(stale, "serializable")["valid?"] is True
    cyc = _ops((0, ":ok", [[R, 1, None], [W, 1, 1]], [[R, 1, 2], [W, 1, 1]]), (1, ":ok", [[R, 1, None], [W, 1, 2]], [[R, 1, 1], [W, 1, 2]]))
    res = _check(cyc, "read-uncommitted")
    assert "cyclic-versions" in res["anomalies"] and res["valid?"] is False
    assert "cyclic-versions" in elle_ref.analyse_rw(cyc)["anomalies"]


def test_checker_agrees_with_reference_on_mutated_histories():
    """corrupt reads of real histories: both implementations must name the same anomalies"""
    cfg = _cfg(node_count=3, rate=60.0, time_limit=4.0, latency=5)
    o = O.run(cfg, 0, 3)
    rng = np.random.default_rng(5)
    seen = set()
    for i in range(3):
        ops = E.decode_history(*o.history(i), 3, RW)
        for trial in range(25):
            mut = [dict(op, value=[list(m) for m in op["value"]]) if op["f"] == ":txn" else op for op in ops]
            for _ in range(1 + trial % 3):
                cands = [op for op in mut if op["type"] == ":ok" and op["f"] == ":txn"]
                op = cands[rng.integers(len(cands))]
                m = op["value"][rng.integers(len(op["value"]))]
                if m[0] == R:
                    m[2] = None if rng.integers(4) == 0 else int(rng.integers(1, 6))
                elif rng.integers(3) == 0:
                    op["type"] = ":fail"
            rows, pay = E.encode_txn_history(mut, rw=True)
            got = E.check_rw_history(rows, pay, "strict-serializable")
            ref = elle_ref.analyse_rw(mut)
            assert set(got["anomalies"]) == ref["anomalies"], (i, trial, got, ref)
            seen |= ref["anomalies"]
    assert {"G1a", "internal"} <= seen and seen & {"G1c", "G0", "G-single", "G2"}


def test_proscribed_sets_nest():
    lib = A.load()
    lib.msim_proscribed_anomalies.restype = C.c_uint32
    sets = [lib.msim_proscribed_anomalies(m) for m in (A.CM_READ_UNCOMMITTED, A.CM_READ_COMMITTED, A.CM_SNAPSHOT_ISOLATION, A.CM_SERIALIZABLE, A.CM_STRICT_SERIALIZABLE)]
    for weaker, stronger in zip(sets, sets[1:]):
        assert weaker & stronger == weaker and weaker != stronger


# ---- the oracle's node against an independent transliteration of txn_rw_register_hat.clj ----
def _replay_through_model(cfg, inst):
    """Runs the oracle with the journal on, then replays its network schedule (what was sent, what was delivered and when,
    journal.clj:220-239) through tests/hat_ref.py: every message a node emits must be the one the oracle's node emitted."""
    import collections
    import hat_ref
    lib = O.load()
    N = cfg.n_nodes
    rows = np.zeros(cfg.max_rows, dtype=O.OP_DT); pay = np.zeros(cfg.max_payload_words, dtype=np.uint32)
    stats = np.zeros(1, dtype=O.STATS_DT); meta = np.zeros(1, dtype=O.META_DT)
    kv = np.zeros((N, cfg.max_values), dtype=np.uint32); lam = np.zeros(N, dtype=np.uint32); npend = np.zeros(N, dtype=np.uint32)
    journal = np.zeros(cfg.journal_capacity, dtype=O.EVENT_DT)
    rc = lib.oracle_hat_state(C.byref(cfg), inst, rows.ctypes.data_as(C.c_void_p), pay.ctypes.data_as(C.c_void_p), stats.ctypes.data_as(C.c_void_p),
                              meta.ctypes.data_as(C.c_void_p), kv.ctypes.data_as(C.c_void_p), lam.ctypes.data_as(C.c_void_p), npend.ctypes.data_as(C.c_void_p),
                              journal.ctypes.data_as(C.c_void_p))
    assert rc == 0 and meta[0]["flags"] == 0 and meta[0]["n_events"] <= cfg.journal_capacity
    T = {name: i for i, name in enumerate(A.MSG_TYPES)}
    created = {}
    nodes = [hat_ref.HatNode(i, range(N), created) for i in range(N)]
    out = [collections.deque() for _ in range(N)]
    content, next_tick, n_replicated = {}, 100000, 0
    for ev in journal[: meta[0]["n_events"]]:
        t, msg, a, route = int(ev["time_us"]), int(ev["msg"]), int(ev["a"]), int(ev["route"])
        mid, recv, typ = msg >> 8, (msg >> 7) & 1, msg & 0x7F
        src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
        while next_tick <= t:   # the replication thread wakes every 100 ms (:107-118), before anything else at that instant
            for n in range(N):
                step = nodes[n].replicate_step()
                if step:
                    out[n].append(("replicate", step[0], step[1]))
            next_tick += 100000
        if not recv:
            if src >= N:    # a client's request
                content[mid] = ("txn", E.decode_rw_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])) if typ == T["txn"] else ("init", None)
                continue
            assert out[src], f"node {src} sent {A.MSG_TYPES[typ]} at {t} us, the model had nothing to send"
            kind, to, body = out[src].popleft()
            assert (kind, to) == (A.MSG_TYPES[typ], dest), (t, src, kind, to, A.MSG_TYPES[typ], dest)
            if kind == "txn_ok":
                got = E.decode_rw_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])
                assert got == [[":" + f, k, v] for f, k, v in body], (t, src, got, body)
            elif kind in ("replicate", "replicate_ack"):
                assert len(body) & 0xFFFF == b, (t, src, kind, len(body), b)
                n_replicated += kind == "replicate"
            content[mid] = (kind, body)
        elif dest < N:
            kind, body = content[mid]
            node = nodes[dest]
            if kind == "init":
                out[dest].append(("init_ok", src, None))
            elif kind == "txn":
                out[dest].append(("txn_ok", src, node.on_txn([[f[1:], k, v] for f, k, v in body])))
            elif kind == "replicate":
                for to, tss in node.on_replicate(body):
                    out[dest].append(("replicate_ack", to, tss))
            elif kind == "replicate_ack":
                node.on_replicate_ack(src, body)
    assert not any(out), [len(o) for o in out]
    for n, node in enumerate(nodes):
        assert node.lamport == lam[n] and len(node.unreplicated) == npend[n]
        want = {k: (int(w) >> 11, (int(w) >> 8) & 7, int(w) & 0xFF) for k, w in enumerate(kv[n]) if w}
        assert {k: (r["ts"][0], r["ts"][1], r["value"]) for k, r in node.kv.items()} == want
    return n_replicated, len(created)


@pytest.mark.parametrize("kw", [dict(), dict(nemesis=("partition",), nemesis_interval=1.5), dict(node_count=3, latency=30, latency_dist="exponential", p_loss=0.1),
                                dict(node_count=5, latency=10, nemesis=("partition",), nemesis_interval=2.0), dict(node_count=8, rate=40.0, latency=40, latency_dist="uniform")])
def test_oracle_node_equals_transliterated_reference_node(kw):
    cfg = _cfg(journal_capacity=400000, **kw)
    for inst in range(3):
        n_rep, n_txn = _replay_through_model(cfg, inst)
        assert n_txn > 20 and n_rep > 10
