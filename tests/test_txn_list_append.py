"""txn-list-append (BASELINE configs[4]) on the CPU side: the oracle's single-root
transactional node + lin-kv service against the reference's known answers, and the host list-append checker
(restating [upstream] elle, txn_list_append.clj:142) against hand-made anomalous histories.

The node + service transition functions are pinned separately by golden vectors recorded from the reference's own
demo/js/single_key_txn.js (tests/test_golden_transitions.py); pinned here: message counts per transaction (read + cas per
txn, service.clj / single_key_txn.clj:171-180) and strict serializability of every history."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O


def _cfg(**kw):
    base = dict(node_count=5, rate=50, time_limit=10, seed=5)
    base.update(kw)
    return E.test_config("txn-list-append", **base)


@pytest.mark.parametrize("kw", [dict(), dict(latency=5), dict(latency=20, latency_dist="exponential"),
                                dict(latency=5, nemesis=["partition"], nemesis_interval=3), dict(node_count=3, rate=200)])
def test_messages_per_transaction_and_verdict(kw):
    cfg = _cfg(**kw)
    r = O.run(cfg, 0, 4)
    for i in range(4):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, cfg.n_nodes, A.WL_TXN_LIST_APPEND) if o["process"] != ":nemesis"]
        inv = [o for o in ops if o["type"] == ":invoke"]
        done = [o for o in ops if o["type"] != ":invoke"]
        assert len(inv) == len(done) > 50 and not any(o["type"] == ":info" for o in done)
        st = r.stats[i]
        # every txn: client -> node, node -> client; node -> lin-kv read, reply, cas, reply (single_key_txn.clj:171-180)
        assert int(st["clients_send"]) == 2 * (len(inv) + cfg.n_nodes)
        assert int(st["servers_send"]) == 4 * len(inv)
        assert int(st["all_recv"]) == int(st["all_send"])
        # completed transactions echo the requested micro-ops, reads filled in (txn_list_append.clj:40-52)
        by_proc = {}
        for o in ops:
            if o["type"] == ":invoke":
                by_proc[o["process"]] = o
            else:
                req = by_proc.pop(o["process"])["value"]
                assert [(f, k) for f, k, _ in req] == [(f, k) for f, k, _ in o["value"]]
                if o["type"] == ":fail":
                    assert o["value"] == req and o["error"][0] == ":txn-conflict"
                else:
                    assert all(v is not None or f == ":r" for f, _, v in o["value"])
        res = E.check_txn_history(rows, pay)
        assert res["valid?"] is True and res["anomalies"] == [], res
        assert res["txn-count"] == len(inv)


@pytest.mark.parametrize("kw", [dict(latency=2), dict(latency=20, latency_dist="exponential", rate=100),
                                dict(latency=5, nemesis=["partition"], nemesis_interval=2), dict(node_count=3, rate=200, latency=1, key_count=2),
                                dict(latency=10, p_loss=0.02)])
def test_multi_key_node_histories_are_strict_serializable(kw):
    """oracle/mk_nodes.inc (demo/js/multi_key_txn.js): whatever the schedule, a transaction only completes through a root cas against
    the exact map it read, so the list-append analysis finds nothing; several messages per transaction, retries included; a transaction
    that keeps losing the root under contention, or whose message vanished, outlives the client's timeout (:info)."""
    cfg = _cfg(bin="multi-key-txn", **kw)
    r = O.run(cfg, 0, 4)
    for i in range(4):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, cfg.n_nodes, A.WL_TXN_LIST_APPEND) if o["process"] != ":nemesis"]
        inv = [o for o in ops if o["type"] == ":invoke"]
        done = [o for o in ops if o["type"] != ":invoke"]
        assert len(inv) == len(done) > 20
        assert not any(o["type"] == ":fail" for o in done)             # a lost root cas is retried, never reported
        assert sum(o["type"] == ":ok" for o in done) > 20              # (:info = the client gave up on a transaction still retrying, or lost)
        st = r.stats[i]
        assert int(st["servers_send"]) >= 2 * len(inv)                 # at least the root cas and its reply
        res = E.check_txn_history(rows, pay)
        assert res["valid?"] is True and res["anomalies"] == [], res


def test_conflicts_need_concurrency():
    """With one node there is one worker: no cas can lose the race."""
    cfg = _cfg(node_count=1, rate=50, latency=5)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        rows, pay = r.history(i)
        assert not (((rows["packed"] & 3) == A.T_FAIL).any())


def test_appends_are_unique_per_key_and_keys_rotate():
    cfg = _cfg(rate=200, time_limit=10)
    r = O.run(cfg, 0, 1)
    rows, pay = r.history(0)
    seen, keys = set(), set()
    for o in E.decode_history(rows, pay, 5, A.WL_TXN_LIST_APPEND):
        if o["type"] == ":invoke":
            assert 1 <= len(o["value"]) <= cfg.max_txn_length
            for f, k, v in o["value"]:
                keys.add(k)
                if f == ":append":
                    assert (k, v) not in seen and 1 <= v <= cfg.max_writes_per_key
                    seen.add((k, v))
    assert len(keys) > cfg.key_count  # keys were retired and replaced


def test_txn_payload_roundtrip():
    txn = [[":append", 7, 3], [":r", 7, [1, 2, 3]], [":r", 9, None], [":r", 300, [5, 6, 7, 8, 9]], [":append", 32000, 63]]
    assert E.decode_txn(E.encode_txn(txn)) == txn


def _h(*ops):
    """ops: (type, process, txn) in history order."""
    return E.encode_txn_history([{"type": t, "process": p, "value": v} for t, p, v in ops])


def _check(*ops):
    return E.check_txn_history(*_h(*ops))


A_, R_ = ":append", ":r"


def test_checker_accepts_a_serial_history():
    res = _check((":invoke", 0, [[A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[A_, 1, 1], [R_, 1, [1]]]),
                 (":invoke", 1, [[R_, 1, None], [A_, 1, 2]]), (":ok", 1, [[R_, 1, [1]], [A_, 1, 2]]),
                 (":invoke", 0, [[R_, 1, None]]), (":ok", 0, [[R_, 1, [1, 2]]]))
    assert res["valid?"] is True and res["anomalies"] == []


def test_checker_g0_write_cycle():
    res = _check((":invoke", 0, [[A_, 1, 1], [A_, 2, 1]]), (":invoke", 1, [[A_, 1, 2], [A_, 2, 2]]),
                 (":ok", 0, [[A_, 1, 1], [A_, 2, 1]]), (":ok", 1, [[A_, 1, 2], [A_, 2, 2]]),
                 (":invoke", 2, [[R_, 1, None], [R_, 2, None]]), (":ok", 2, [[R_, 1, [1, 2]], [R_, 2, [2, 1]]]))
    assert res["valid?"] is False and "G0" in res["anomalies"]


def test_checker_g1a_aborted_read():
    res = _check((":invoke", 0, [[A_, 1, 1]]), (":fail", 0, [[A_, 1, 1]]),
                 (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1]]]))
    assert res["valid?"] is False and "G1a" in res["anomalies"]


def test_checker_g1b_intermediate_read():
    res = _check((":invoke", 0, [[A_, 1, 1], [A_, 1, 2]]), (":invoke", 1, [[R_, 1, None]]),
                 (":ok", 1, [[R_, 1, [1]]]), (":ok", 0, [[A_, 1, 1], [A_, 1, 2]]))
    assert res["valid?"] is False and "G1b" in res["anomalies"]


def test_checker_g1c_circular_information_flow():
    res = _check((":invoke", 0, [[A_, 1, 1], [R_, 2, None]]), (":invoke", 1, [[A_, 2, 1], [R_, 1, None]]),
                 (":ok", 0, [[A_, 1, 1], [R_, 2, [1]]]), (":ok", 1, [[A_, 2, 1], [R_, 1, [1]]]))
    assert res["valid?"] is False and "G1c" in res["anomalies"]


def test_checker_g_single_read_skew():
    res = _check((":invoke", 0, [[A_, 1, 1], [A_, 2, 1]]), (":invoke", 1, [[R_, 1, None], [R_, 2, None]]),
                 (":ok", 0, [[A_, 1, 1], [A_, 2, 1]]), (":ok", 1, [[R_, 1, [1]], [R_, 2, None]]),
                 (":invoke", 2, [[R_, 2, None]]), (":ok", 2, [[R_, 2, [1]]]))
    assert res["valid?"] is False and res["anomalies"] == ["G-single"]


def test_checker_g2_write_skew():
    res = _check((":invoke", 0, [[R_, 1, None], [A_, 2, 1]]), (":invoke", 1, [[R_, 2, None], [A_, 1, 1]]),
                 (":ok", 0, [[R_, 1, None], [A_, 2, 1]]), (":ok", 1, [[R_, 2, None], [A_, 1, 1]]),
                 (":invoke", 2, [[R_, 1, None], [R_, 2, None]]), (":ok", 2, [[R_, 1, [1]], [R_, 2, [1]]]))
    assert res["valid?"] is False and res["anomalies"] == ["G2"]


def test_checker_internal_inconsistency():
    res = _check((":invoke", 0, [[A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[A_, 1, 1], [R_, 1, None]]))
    assert res["valid?"] is False and "internal" in res["anomalies"]
    res = _check((":invoke", 0, [[R_, 1, None], [A_, 1, 1], [R_, 1, None]]), (":ok", 0, [[R_, 1, None], [A_, 1, 1], [R_, 1, [1]]]))
    assert res["valid?"] is True


def test_checker_incompatible_orders_and_duplicates():
    res = _check((":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]), (":invoke", 1, [[A_, 1, 2]]), (":ok", 1, [[A_, 1, 2]]),
                 (":invoke", 2, [[R_, 1, None]]), (":ok", 2, [[R_, 1, [1, 2]]]),
                 (":invoke", 3, [[R_, 1, None]]), (":ok", 3, [[R_, 1, [2, 1]]]))
    assert res["valid?"] is False and "incompatible-order" in res["anomalies"]
    res = _check((":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]),
                 (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1, 1]]]))
    assert res["valid?"] is False and "duplicate-elements" in res["anomalies"]


def test_checker_stale_read_is_a_realtime_anomaly_only():
    """Serializable but not strict: T2 starts after T1 completed and does not see its append."""
    ops = ((":invoke", 0, [[A_, 1, 1]]), (":ok", 0, [[A_, 1, 1]]),
           (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, None]]),
           (":invoke", 2, [[R_, 1, None]]), (":ok", 2, [[R_, 1, [1]]]))
    res = _check(*ops)
    assert res["valid?"] is False and "realtime" in res["anomalies"] and "G-single" in res["anomalies"]
    # the same reads issued concurrently with the append are fine
    res = _check(ops[0], ops[2], ops[1], ops[3], ops[4], ops[5])
    assert res["valid?"] is True


def test_checker_indeterminate_appends_may_be_observed():
    res = _check((":invoke", 0, [[A_, 1, 1]]), (":info", 0, [[A_, 1, 1]]),
                 (":invoke", 1, [[R_, 1, None]]), (":ok", 1, [[R_, 1, [1]]]))
    assert res["valid?"] is True and res["info-count"] == 1


CYCLES = {"G0", "G1c", "G-single", "G2"}


def _agree(ops):
    import elle_ref
    rows, pay = E.encode_txn_history([o for o in ops if o.get("process") != ":nemesis"])
    got = E.check_txn_history(rows, pay)
    ref = elle_ref.analyse(ops)
    assert got["valid?"] == ref["valid?"], (got, ref)
    assert bool(CYCLES & set(got["anomalies"])) == ("cycle" in ref["anomalies"]), (got, ref)
    for a in ("G1a", "G1b", "internal", "incompatible-order", "duplicate-elements", "dirty-update", "realtime"):
        assert (a in got["anomalies"]) == (a in ref["anomalies"]), (a, got, ref)
    return got


def test_checker_agrees_with_the_python_restatement_on_engine_and_mutated_histories():
    """csrc/txn_check.cpp against tests/elle_ref.py (written independently): oracle histories (valid), then the same
    histories with a read list corrupted in one of five ways — both must see the same classes of anomaly."""
    import copy
    import random
    cfg = _cfg(rate=60, time_limit=8, latency=5, key_count=3)
    r = O.run(cfg, 0, 3)
    rng = random.Random(7)
    n_bad = 0
    for i in range(3):
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, cfg.n_nodes, A.WL_TXN_LIST_APPEND) if o["process"] != ":nemesis"]
        assert _agree(ops)["valid?"] is True
        reads = [(oi, mi) for oi, o in enumerate(ops) if o["type"] == ":ok" for mi, m in enumerate(o["value"]) if m[0] == ":r" and m[2] and len(m[2]) >= 2]
        for trial in range(40):
            oi, mi = rng.choice(reads)
            mut = copy.deepcopy(ops)
            lst = mut[oi]["value"][mi][2]
            how = trial % 5
            if how == 0:
                lst.pop()                       # stale read: one version behind
            elif how == 1:
                lst[0], lst[-1] = lst[-1], lst[0]   # incompatible order
            elif how == 2:
                lst.append(lst[0])              # duplicate element
            elif how == 3:
                lst.append(61)                  # an element nobody appended
            else:
                del lst[:]                      # sees nothing although elements were visible
                mut[oi]["value"][mi][2] = None
            got = _agree(mut)
            n_bad += got["valid?"] is False
    assert n_bad > 60


# ---- the oracle's node, service and its "a database state is a prefix of one append log" shortcut against a transliteration
#      of single_key_txn.clj + service.clj that keeps the real values ----
def _replay_real_values(cfg, inst):
    import collections
    import services_ref as R
    r = O.run(cfg, inst, 1)
    assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
    rows, pay = r.history(0)
    N = cfg.n_nodes
    SVC = 2 * N
    svc = R.Linearizable()
    nodes = [R.SingleKeyTxnNode(SVC) for _ in range(N)]
    out = collections.defaultdict(collections.deque)
    content, n_ok, n_conflict = {}, 0, 0
    for ev in r.events(0):
        msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
        mid, recv, typ = msg >> 8, (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
        src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
        if not recv:
            if N <= src < SVC:
                txn = [[f[1:], k, v] for f, k, v in E.decode_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])] if typ == "txn" else None
                content[mid] = {"type": typ, "msg_id": b, "txn": txn}
                continue
            assert out[src], (src, typ)
            to, body = out[src].popleft()
            assert to == dest and body["type"] == typ and (body.get("msg_id", body.get("in_reply_to", 0)) & 0xFFFF) == b, (src, dest, to, body, typ, b)
            if typ == "txn_ok":    # the completed transaction, reads filled in from the REAL lists
                got = [[f[1:], k, v] for f, k, v in E.decode_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])]
                assert got == body["txn"], (got, body["txn"])
                n_ok += 1
            elif typ == "error":
                assert body["code"] == a
                n_conflict += a == 30
            content[mid] = body
        elif dest < N:
            body = content[mid]
            if body["type"] == "init":
                out[dest].append((src, {"type": "init_ok", "in_reply_to": body["msg_id"]}))
            elif "in_reply_to" in body:
                nxt = nodes[dest].on_reply(body)
                if nxt:
                    out[dest].append(nxt)
            else:
                out[dest].append(nodes[dest].on_txn(src, body))
        elif dest == SVC:
            body = content[mid]
            out[SVC].append((src, dict(svc.handle(src, body, None), in_reply_to=body["msg_id"])))
    assert not any(out.values())
    return n_ok, n_conflict


@pytest.mark.parametrize("kw", [dict(), dict(latency=20, latency_dist="exponential", p_loss=0.05), dict(node_count=3, rate=200, latency=2),
                                dict(nemesis=["partition"], nemesis_interval=2, latency=5), dict(key_count=2, max_txn_length=8, max_writes_per_key=40, rate=150)])
def test_oracle_equals_transliterated_node_and_service_with_real_values(kw):
    base = dict(node_count=5, rate=80, time_limit=8, latency=5, seed=77, journal_capacity=200000)
    base.update(kw)
    cfg = E.test_config("txn-list-append", **base)
    ok = conflict = 0
    for inst in range(3):
        a, b = _replay_real_values(cfg, inst)
        ok += a; conflict += b
    assert ok > 50 and (conflict > 0 or kw.get("p_loss"))
