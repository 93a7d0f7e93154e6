"""kafka workload (workload/kafka.clj over demo/clojure/kafka.clj) — CPU side: the oracle's restatement keeps the promises of the
reference's node (dense logs, one message per offset, committed offsets that only grow), the checker (msim_check_kafka_rows) finds the
anomalies workload/kafka.clj:21-70 describes — the example the reference prints (:42-60) included — and agrees with an independent
Python restatement on real and on corrupted histories.  PARITY UNPINNED beyond that example: the node is babashka-only, the generator and
checker are [upstream] jepsen.tests.kafka."""
import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E

import kafka_check_ref as KR
import oracle_lib as O

KF = A.WL_KAFKA


def _cfg(**kw):
    args = dict(workload="kafka", node_count=3, rate=60.0, time_limit=6.0, latency=5, seed=21)
    args.update(kw)
    return E.test_config(**args)


def _ops(rows, pay, n):
    return E.decode_history(rows, pay, n, KF)


def test_defaults_and_limits():
    cfg = _cfg()
    assert cfg.node_program == A.NODE_KAFKA and cfg.key_count == 4 and cfg.max_writes_per_key == 1024
    with pytest.raises(E.EngineError):
        _cfg(key_count=9)          # {key offset} maps keep insertion order only up to 8 entries
    with pytest.raises(E.EngineError):
        _cfg(concurrency=7)        # one worker per node, or a multiple of the node count (round 6: kafkag_kernel<>)
    assert _cfg(concurrency=6).concurrency == 6


@pytest.mark.parametrize("kw", [dict(), dict(latency=0, rate=120.0), dict(node_count=5, latency=20, latency_dist="exponential"),
                                dict(nemesis=("partition",), nemesis_interval=2.0), dict(p_loss=0.05, latency=10, latency_dist="uniform"),
                                dict(key_count=2, max_writes_per_key=40, rate=150.0, node_count=2)])
def test_oracle_histories_keep_the_logs_promises(kw):
    cfg = _cfg(**kw)
    o = O.run(cfg, 0, 5)
    for i in range(5):
        assert o.meta[i]["flags"] == 0
        rows, pay = o.history(i)
        ops = _ops(rows, pay, cfg.n_nodes)
        # every acknowledged send got its own offset; offsets of a key are dense from 0; a poll returns what was sent
        sent, acked = {}, {}
        for op in ops:
            if op["f"] == ":send" and op["type"] == ":invoke":
                sent.setdefault(op["value"][0][1], set()).add(op["value"][0][2])
            if op["f"] == ":send" and op["type"] == ":ok":
                _, k, (off, msg) = op["value"][0]
                assert off not in acked.setdefault(k, {}), (i, op)
                acked[k][off] = msg
        for op in ops:
            if op["f"] == ":poll" and op["type"] == ":ok" and len(op["value"][0]) > 1:
                for k, pairs in op["value"][0][1].items():
                    assert [o_ for o_, _ in pairs] == list(range(pairs[0][0], pairs[0][0] + len(pairs))) if pairs else True
                    assert len(pairs) <= 32      # one chunk per key and poll (demo/clojure/kafka.clj:19-21,112-139)
                    for off, msg in pairs:
                        assert msg in sent[k] and acked.get(k, {}).get(off, msg) == msg, (i, op)
        res = E.check_kafka_history(rows, pay)
        ref = KR.check(ops)
        assert res["valid?"] is True and res["anomalies"] == [], (i, res)
        assert ref["valid?"] is True and ref["unobserved-count"] == res["unobserved-count"] and ref["acked-count"] == res["acked-count"]
        if not kw.get("p_loss") and not kw.get("nemesis"):
            # the final polls read every key from the beginning until nothing comes: nothing acknowledged stays unobserved
            assert res["unobserved-count"] == 0, (i, res)
        assert {":send", ":poll", ":assign"} <= {op["f"] for op in ops}


def test_a_poll_commits_what_it_saw_and_assign_resumes_there():
    """workload/kafka.clj:207-230: after an assign without :seek-to-beginning? a client resumes at (or below) committed + 1"""
    cfg = _cfg(node_count=2, rate=80.0, time_limit=8.0, latency=2, seed=5)
    o = O.run(cfg, 0, 4)
    resumed = 0
    for i in range(4):
        ops = _ops(*o.history(i), 2)
        high = {}      # key -> the highest offset a completed poll has seen: a lower bound of the committed offset (:141-149)
        top = {}       # key -> the highest offset acknowledged so far: nothing above it can be committed
        held = {}      # process -> keys its client holds an offset of (those win over the committed ones, :215-217)
        at_invoke, fresh = {}, {}
        for op in ops:
            p = op["process"]
            if op["f"] == ":send" and op["type"] == ":ok":
                top[op["value"][0][1]] = max(top.get(op["value"][0][1], -1), op["value"][0][2][0])
            if op["f"] == ":assign" and op["type"] == ":invoke":
                at_invoke[p] = dict(high)
            if op["f"] == ":assign" and op["type"] == ":ok":
                if op.get("seek-to-beginning?"):
                    fresh.pop(p, None)
                else:
                    fresh[p] = (set(op["value"]) - held.get(p, set()), at_invoke[p])
                held[p] = set(op["value"])
            if op["f"] == ":poll" and op["type"] == ":invoke" and p in fresh:
                keys, lo = fresh.pop(p)
                for k in keys:   # (or (offsets k) (committed k) 0): the client resumes AT the committed offset (its message is polled again)
                    assert lo.get(k, 0) <= op["offsets"][k] <= max(top.get(k, 0), 0), (i, op, lo, top)
                    resumed += op["offsets"][k] > 0
            if op["f"] == ":poll" and op["type"] == ":ok" and len(op["value"][0]) > 1:
                for k, pairs in op["value"][0][1].items():
                    if pairs:
                        high[k] = max(high.get(k, -1), pairs[-1][0])
    assert resumed > 0


def _h(*ops):
    return [dict(op, index=i, time=i * 1000) for i, op in enumerate(ops)]


def _send(p, k, msg, off=None, typ=":ok"):
    inv = {"type": ":invoke", "process": p, "f": ":send", "value": [[":send", k, msg]]}
    done = {"type": typ, "process": p, "f": ":send", "value": [[":send", k, [off, msg] if typ == ":ok" else msg]]}
    return [inv, done]


def _poll(p, msgs):
    return [{"type": ":invoke", "process": p, "f": ":poll", "value": [[":poll"]]},
            {"type": ":ok", "process": p, "f": ":poll", "value": [[":poll", msgs]]}]


def _check(ops):
    rows, pay = E.encode_kafka_history(ops)
    res = E.check_kafka_history(rows, pay)
    assert E.decode_history(rows, pay, 1, KF)[-1]["value"] == ops[-1]["value"]     # (the encoding loses nothing)
    ref = KR.check(ops)
    assert ref["anomalies"] == res["anomalies"] and ref["valid?"] == res["valid?"], (res, ref)
    return res


def test_reference_poll_skip_example():
    """workload/kafka.clj:42-60: process 0 polls key "56" (here: "5") and sees offsets 4 and 5 (messages 5 and 6), then offset 8
    (message 9): "The client unexpectedly jumped three offsets ahead, skipping messages 7 and 8" — a poll-skip, :delta 3."""
    ops = _h(*(sum((_send(1, "5", m, m - 1) for m in range(1, 10)), [])),          # offsets 0..8 hold messages 1..9
             *_poll(0, {"5": [[4, 5], [5, 6]]}), *_poll(0, {"5": [[8, 9]]}),
             *_poll(2, {"5": [[o_, o_ + 1] for o_ in range(9)]}))                # (somebody else reads the whole log: nothing is lost)
    res = _check(ops)
    assert res["anomalies"] == ["poll-skip"] and res["valid?"] is False
    # the same two polls across an assign are fine (:64-70), and so is a jump over offsets nobody knows to exist
    a = [{"type": ":invoke", "process": 0, "f": ":assign", "value": ["5"]}, {"type": ":ok", "process": 0, "f": ":assign", "value": ["5"]}]
    ok = _h(*ops[:20], *a, *ops[20:])     # (after the first poll)
    assert _check(ok)["anomalies"] == []
    sparse = _h(*_send(1, "5", 5, 4), *_send(1, "5", 6, 5), *_send(1, "5", 9, 8), *_poll(0, {"5": [[4, 5], [5, 6]]}), *_poll(0, {"5": [[8, 9]]}))   # offsets 6, 7 hold nothing
    assert _check(sparse)["anomalies"] == []   # "Offsets may be sparse" (:5-6)


def test_each_anomaly_by_hand():
    lost = _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_send(0, "1", 3, 2), *_poll(1, {"1": [[0, 1]]}), *_poll(2, {"1": [[2, 3]]}))
    res = _check(lost)
    assert "lost-write" in res["anomalies"] and res["lost-count"] == 1          # offset 1 was acknowledged, 2 was polled, 1 never
    unobs = _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[0, 1]]}))
    res = _check(unobs)
    assert res["anomalies"] == [] and res["unobserved-count"] == 1 and res["valid?"] is True   # no recency requirement (:27-28)
    nm_poll = _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[0, 1], [1, 2]]}), *_poll(1, {"1": [[1, 2]]}))
    assert _check(nm_poll)["anomalies"] == ["nonmonotonic-poll"]                 # "2 then 2" (:33)
    nm_send = _h(*_send(0, "1", 1, 3), *_send(0, "1", 2, 3))
    assert "nonmonotonic-send" in _check(nm_send)["anomalies"]
    int_skip = _h(*(sum((_send(0, "1", m, m - 1) for m in range(1, 6)), [])), *_poll(1, {"1": [[0, 1], [3, 4]]}))
    assert _check(int_skip)["anomalies"] == ["int-poll-skip", "lost-write"]     # (1 and 2 were acknowledged and jumped over)
    int_nm = _h(*_send(0, "1", 1, 0), *_send(0, "1", 2, 1), *_poll(1, {"1": [[1, 2], [0, 1]]}))
    assert _check(int_nm)["anomalies"] == ["int-nonmonotonic-poll"]
    incons = _h(*_send(0, "1", 1, 0), *_poll(1, {"1": [[0, 7]]}))
    assert "inconsistent-offsets" in _check(incons)["anomalies"]
    dup = _h(*_send(0, "1", 1, 0), *_poll(1, {"1": [[0, 1], [1, 1]]}))
    res = _check(dup)
    assert "duplicate" in res["anomalies"] and res["duplicate-count"] == 1
    aborted = _h(*_send(0, "1", 1, typ=":fail"), *_poll(1, {"1": [[0, 1]]}))
    assert _check(aborted)["anomalies"] == ["aborted-read"]
    assert _check(_h(*_send(0, "1", 1, typ=":info"), *_poll(1, {"1": [[0, 1]]})))["anomalies"] == []   # an indeterminate send may have happened


def test_encoder_splits_long_polls_and_refuses_what_the_layout_cannot_hold():
    """A poll run longer than the header's 8-bit count is split into blocks (the format allows several per key); keys, messages and
    offsets outside the layout raise instead of being masked into a different history."""
    ops = _h(*(sum((_send(1, "3", m + 1, m) for m in range(300)), [])), *_poll(0, {"3": [[m, m + 1] for m in range(300)]}))
    res = _check(ops)
    assert res["valid?"] is True and res["anomalies"] == [] and res["lost-count"] == 0 and res["acked-count"] == 300
    rows, pay = E.encode_kafka_history(ops)
    assert E.decode_history(rows, pay, 1, KF)[-1]["value"][0][1]["3"] == [[m, m + 1] for m in range(300)]
    for bad in (_send(0, "9", 1, 0), _send(0, "1", 2047, 0), _send(0, "1", 1, 2047), _poll(0, {"8": [[0, 1]]}), _poll(0, {"1": [[0, 70000]]}),
                [{"type": ":invoke", "process": 0, "f": ":assign", "value": ["12"]}]):
        with pytest.raises(E.EngineError):
            E.encode_kafka_history(_h(*bad))


def test_checker_agrees_with_restatement_on_corrupted_histories():
    cfg = _cfg(node_count=3, rate=80.0, time_limit=5.0)
    o = O.run(cfg, 0, 3)
    rng = np.random.default_rng(9)
    seen = set()
    for i in range(3):
        ops = _ops(*o.history(i), 3)
        for trial in range(20):
            mut = [dict(op) for op in ops]
            for _ in range(1 + trial % 3):
                j = int(rng.integers(len(mut)))
                op = mut[j]
                if op["f"] == ":poll" and op["type"] == ":ok" and len(op["value"][0]) > 1:
                    msgs = {k: [list(p) for p in v] for k, v in op["value"][0][1].items()}
                    ks = [k for k in msgs if msgs[k]]
                    if ks:
                        k = ks[int(rng.integers(len(ks)))]
                        what = int(rng.integers(3))
                        if what == 0:
                            del msgs[k][int(rng.integers(len(msgs[k])))]          # a message missing from the middle: skip
                        elif what == 1:
                            msgs[k][-1][0] += int(rng.integers(1, 4))             # a jump
                        else:
                            msgs[k][-1][1] = int(rng.integers(1, 30))             # another message at that offset
                    op["value"] = [[":poll", msgs]]
                elif op["f"] == ":send" and op["type"] == ":ok":
                    v = op["value"][0]
                    op["value"] = [[":send", v[1], [max(0, v[2][0] - int(rng.integers(0, 3))), v[2][1]]]]
            mut = [dict(op, index=n_) for n_, op in enumerate(mut)]
            rows, pay = E.encode_kafka_history(mut)
            got, ref = E.check_kafka_history(rows, pay), KR.check(mut)
            assert got["anomalies"] == ref["anomalies"] and got["lost-count"] == ref["lost-count"] and got["duplicate-count"] == ref["duplicate-count"], (i, trial, got, ref)
            seen |= set(ref["anomalies"])
    assert {"poll-skip", "inconsistent-offsets"} <= seen or len(seen) >= 3, seen


def test_history_edn_of_kafka_ops():
    cfg = _cfg(node_count=2, rate=40.0, time_limit=3.0)
    o = O.run(cfg, 0, 1)
    rows, pay = o.history(0)
    native = E.history_edn_native(cfg, rows, pay)
    assert native == E.history_edn(_ops(rows, pay, 2))
    assert ':f :send, :value [[:send "' in native and ":f :poll, :value [[:poll {" in native and ":seek-to-beginning? true" in native


def test_net_journal_of_a_kafka_run_is_writable_as_fressian():
    import fressian_reader as FR
    cfg = _cfg(node_count=2, rate=30.0, time_limit=2.0)
    cfg.journal_capacity = 20000
    o = O.run(cfg, 0, 1)
    ev = o.events(0)
    blob = E.journal_fressian(cfg, ev, o.history(0)[1])
    events = FR.read_journal(blob)
    assert len(events) == len(ev)
    types = {str(e["message"]["body"]["type"]) for e in events}
    assert {"send", "poll", "commit_offsets", "cas", "read"} <= types, types


# ---- the oracle's node and lin-kv against an independent transliteration of demo/clojure/kafka.clj + service.clj with real values ----
def _replay_through_model(cfg, inst):
    """Runs the oracle with the journal on, then replays its network schedule (what was sent, what was delivered and when,
    journal.clj:220-239) through tests/kafka_ref.py: every message a node or the service emits must be the one the oracle emitted."""
    import collections
    import kafka_ref
    o = O.run(cfg, inst, 1)
    assert o.meta[0]["flags"] == 0 and o.meta[0]["n_events"] <= cfg.journal_capacity
    pay = o.history(0)[1]
    N = cfg.n_nodes
    SVC = 2 * N
    T = A.MSG_TYPES
    nodes = [kafka_ref.KafkaNode() for _ in range(N)]
    svc = kafka_ref.LinKV()
    out = collections.defaultdict(collections.deque)
    content, versions, n_checked = {}, {}, collections.Counter()

    def block(a):
        return [int(w) for w in pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)]]

    for ev in o.events(0):
        msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
        mid, recv, typ = msg >> 8, (msg >> 7) & 1, T[msg & 0x7F]
        src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
        if not recv:
            if N <= src < SVC:    # a workload client's request: its real body, from the oracle's encoding
                if typ == "send":
                    body = {"type": "send", "key": str(a & 7), "msg": a >> 6}
                elif typ == "poll":
                    body = {"type": "poll", "offsets": {str(w & 7): w >> 8 for w in block(a)}}
                elif typ == "list_committed_offsets":
                    body = {"type": typ, "keys": [str(w & 7) for w in block(a)]}
                elif typ == "commit_offsets":
                    body = {"type": typ, "offsets": {k: v[-1][0] for k, v in E.decode_poll(block(a)).items() if v}}
                else:
                    body = {"type": "init"}
                content[mid] = dict(body, msg_id=b)
                continue
            assert out[src], f"endpoint {src} sent {typ} (message {mid}), the model had nothing to send"
            to, body = out[src].popleft()
            to = {"lin-kv": SVC}.get(to, to)
            assert (body["type"], to) == (typ, dest), (mid, src, body, typ, dest)
            assert (body.get("msg_id", body.get("in_reply_to")) & 0xFFFF) == b, (mid, body, b)
            n_checked[typ] += 1
            if src < N and dest == SVC:
                if typ == "read":
                    assert body["key"] == ("offsets" if a >> 31 else f"log-{a & 7}-{a >> 8}"), (mid, body, hex(a))
                elif a >> 31:
                    assert body["key"] == "offsets"
                else:
                    assert body["key"] == f"log-{a & 7}-{(a >> 3) & 63}" and len(body["from"]) == (a >> 9) & 31 and body["to"] == body["from"] + [a >> 14], (mid, body, hex(a))
            elif src < N:
                if typ == "send_ok":
                    assert body["offset"] == a
                elif typ == "poll_ok":
                    assert list(E.decode_poll(block(a)).items()) == list(body["msgs"].items()), (mid, body, E.decode_poll(block(a)))
                elif typ == "list_committed_offsets_ok":
                    assert {str(w & 7): (w >> 8) & 0x7FFFFF for w in block(a) if w >> 31} == body["offsets"], (mid, body)
                elif typ == "error":
                    assert body["code"] == a
            else:   # the service's replies
                if typ == "error":
                    assert body["code"] == a
                elif typ == "read_ok" and isinstance(body["value"], list):
                    assert len(body["value"]) == a
                elif typ == "read_ok":       # the offsets map travels as its version: one version, one value
                    assert versions.setdefault(a, body["value"]) == body["value"], (mid, a, versions[a], body["value"])
            content[mid] = body
        elif dest < N:
            for to, body in nodes[dest].handle(src, content[mid]):
                out[dest].append((to, body))
        elif dest == SVC:
            req = content[mid]
            out[SVC].append((src, dict(svc.handle(req), in_reply_to=req["msg_id"])))
    assert not any(out.values()), {k: len(v) for k, v in out.items()}
    assert len(set(map(repr, versions.values()))) == len(versions)     # different versions, different maps
    return n_checked


@pytest.mark.parametrize("kw", [dict(), dict(latency=0, rate=100.0), dict(node_count=4, latency=25, latency_dist="exponential", p_loss=0.1),
                                dict(nemesis=("partition",), nemesis_interval=1.5, latency=10), dict(node_count=2, key_count=2, max_writes_per_key=50, rate=200.0, latency=3)])
def test_oracle_node_and_service_equal_transliterated_reference(kw):
    cfg = _cfg(**kw)
    cfg.journal_capacity = 400000
    for inst in range(3):
        n = _replay_through_model(cfg, inst)
        if kw.get("p_loss"):    # a lost lin-kv message leaves its request waiting for ever (no RPC timeout in node.clj:121-129): few operations complete
            assert n["send_ok"] + n["poll_ok"] > 6 and n["cas"] > 3 and n["read"] > 15, n
        else:
            assert n["send_ok"] > 10 and n["poll_ok"] > 5 and n["cas"] > 10 and n["commit_offsets_ok"] > 0, n
