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
        _cfg(concurrency=6)        # one worker per node in this build


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
