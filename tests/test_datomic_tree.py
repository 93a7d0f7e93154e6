"""txn-list-append over the Datomic-style node (demo/ruby/datomic_list_append.rb; oracle/dt_nodes.inc).  PARITY UNPINNED against the
reference: no Ruby here.  What is checked instead:
  * the hash is Ruby's (Zlib.crc32 of the decimal string, mod 128) — against Python's zlib;
  * a replay: the oracle's net journal of whole runs is fed, receive by receive, to tests/datomic_ref.py (the Ruby classes in Python
    with materialised tree nodes, maps and lists, real lin-kv / lww-kv services), which must send exactly the messages the journal
    holds next — destination, type, msg_id / in_reply_to, pointers, error codes, completed transactions;
  * the tree's invariants over those runs (splits at the ninth key, chains in two-wide ranges, lazily loaded paths);
  * every history passes the list-append checker as strict-serializable; message counts follow the protocol's arithmetic."""
import ctypes as C
import zlib

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import datomic_ref as R
import oracle_lib as O


def _cfg(**kw):
    base = dict(workload="txn-list-append", bin="datomic", node_count=3, rate=60, time_limit=6, latency=3, seed=11, journal_capacity=500000)
    base.update(kw)
    return E.test_config(**base)


def _word(ptr):
    if ptr == "empty":
        return 0
    node, n = ptr[1:].split("-")
    return (int(node) << 20) | int(n)


def test_hash_is_rubys_crc32_of_the_decimal_string():
    lib = O.load()
    lib.oracle_dt_hash.argtypes = [C.c_uint32]
    lib.oracle_dt_hash.restype = C.c_uint32
    for k in list(range(0, 1200)) + [32766, 32767]:
        assert lib.oracle_dt_hash(k) == zlib.crc32(str(k).encode()) % 128 == R.tree_hash(k)


def replay(cfg, instance):
    """Feeds the oracle's journal to the reference classes; returns statistics of the run."""
    lib = O.load()
    ora = O.run(cfg, instance, 1)
    assert int(ora.meta[0]["flags"]) == 0
    rows, pay = ora.history(0)
    ev = ora.events(0)
    N = cfg.n_nodes
    CS = max(cfg.concurrency, N)   # client worker slots (several per node with --concurrency k n: worker t talks to node t mod N)
    LIN, LWW = N + CS, N + CS + 1
    names = [f"n{i}" for i in range(N)] + [f"c{i}" for i in range(CS)] + ["lin-kv", "lww-kv"]
    idx = {n: i for i, n in enumerate(names)}
    outq = {i: [] for i in range(len(names))}
    ctr = [0]

    def rand_int(n):
        v = lib.oracle_draw32(cfg.seed, instance, 12, ctr[0])
        ctr[0] += 1
        return (v * n) >> 32

    now = [0]
    nodes = [R.DatomicListAppendNode((lambda i: lambda dest, body: outq[i].append((dest, body)))(i), clock=lambda: now[0]) for i in range(N)]
    lin, lww = R.LinKV(), R.LwwKV(rand_int)
    inflight, stats = {}, {"loads": 0, "load_retries": 0, "writes": 0, "cas_ok": 0, "cas_lost": 0, "max_depth": 0, "splits": 0, "txn_ok": 0, "await_timeouts": 0, "late_cas_ok": 0, "late_cas_lost": 0, "max_waiting": 0, "arrival_order": True}
    arrived = {i: [] for i in range(N)}   # per node: the transactions in the order their requests arrived (client, msg_id)

    def check_a(src, dest, body, a, req_key):
        t = body["type"]
        if src < N and dest == LIN:
            return a == (0 if t == "read" else _word(body["value"] if t == "write" else body["to"]))
        if src < N and dest == LWW:
            return a == _word(body["key"])
        if src == LIN:
            return a == (_word(body["value"]) if t == "read_ok" else body["code"] if t == "error" else 0)
        if src == LWW:
            return a == (body["code"] if t == "error" else _word(req_key))
        if t == "txn_ok":
            got = E.decode_txn(pay[(a & 0xFFFFFF):(a & 0xFFFFFF) + (a >> 24)])
            want = [[":append" if f == "append" else ":r", k, v] for f, k, v in body["txn"]]
            return got == want
        return a == (body["code"] if t == "error" else 0)

    for i in range(len(ev)):
        msg, route, a = int(ev["msg"][i]), int(ev["route"][i]), int(ev["a"][i])
        mid, typ, recv = msg >> 8, A.MSG_TYPES[msg & 0x7F], bool(msg & 0x80)
        src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
        now[0] = int(ev["time_us"][i])
        if not recv:
            if N <= src < LIN:   # a client's request: the journal says what it is
                if typ == "init":
                    body = {"type": "init", "node_id": names[dest], "node_ids": names[:N], "msg_id": b}
                else:
                    txn = [["append" if f == ":append" else "r", k, v] for f, k, v in E.decode_txn(pay[(a & 0xFFFFFF):(a & 0xFFFFFF) + (a >> 24)])]
                    body = {"type": "txn", "txn": txn, "msg_id": b}
                inflight[mid] = {"src": names[src], "dest": names[dest], "body": body}
                continue
            if not outq[src] and src < N:   # nothing arrived, yet the node speaks: a Promise#await of 5 s ago gives up now
                stats["await_timeouts"] += nodes[src].fire_due(now[0])
            assert outq[src], f"event {i}: the journal has {names[src]} send {typ} to {names[dest]}; the reference program sent nothing"
            d, body = outq[src].pop(0)
            assert idx[d] == dest and body["type"] == typ, (i, names[src], d, body, names[dest], typ)
            assert (body.get("msg_id") or body.get("in_reply_to") or 0) & 0xFFFF == b, (i, body, b)
            assert check_a(src, dest, body, a, body.pop("_key", None)), (i, names[src], names[dest], body, hex(a))
            inflight[mid] = {"src": names[src], "dest": d, "body": body, "t": now[0]}
            if src < N and N <= dest < LIN and typ != "init_ok":   # a transaction's answer: the lock was taken in arrival order (datomic_list_append.rb:348, node.rb:147-183)
                if not arrived[src] or arrived[src].pop(0) != (d, body.get("in_reply_to")):
                    stats["arrival_order"] = False
            if src < N and dest == LWW:
                stats["writes" if typ == "write" else "loads"] += 1
            continue
        m = inflight.pop(mid)
        assert idx[m["dest"]] == dest
        if dest < N:
            if m["body"]["type"] == "txn":
                arrived[dest].append((m["src"], m["body"]["msg_id"]))
            nodes[dest].handle(m)
            stats["max_waiting"] = max(stats["max_waiting"], len(nodes[dest].lock_waiters))
        elif dest == LIN or dest == LWW:
            rep = (lin if dest == LIN else lww).handle(m["body"])
            rep = {**rep, "in_reply_to": m["body"]["msg_id"], "_key": m["body"].get("key")}
            outq[dest].append((m["src"], rep))
            if dest == LWW and m["body"]["type"] == "read" and rep["type"] == "error":
                stats["load_retries"] += 1
            if dest == LIN and m["body"]["type"] == "cas":
                stats["cas_ok" if rep["type"] == "cas_ok" else "cas_lost"] += 1
                if now[0] - m["t"] > 5_000_000:   # its sender's Promise#await has given up: the request still names ITS from / to
                    stats["late_cas_ok" if rep["type"] == "cas_ok" else "late_cas_lost"] += 1
        else:
            stats["txn_ok"] += m["body"]["type"] == "txn_ok"
    assert not any(outq.values()), {names[k]: v[:2] for k, v in outq.items() if v}
    assert not inflight or cfg.p_loss_q32 or cfg.latency_mean_ms >= 1000   # (a lost message is sent and never received; with latencies of seconds the test ends over messages under way)
    stats["lost"] = len(inflight)

    # the committed tree, walked in the store: every leaf holds the keys of its range, a full leaf was split
    store = {}
    for r in lww.replicas:
        store.update(r)

    def walk(ptr, depth):
        nd = store[ptr]
        stats["max_depth"] = max(stats["max_depth"], depth)
        lo, hi = nd["range"]
        if nd["type"] == "leaf":
            for k, _ in nd["pairs"]:
                assert lo <= R.tree_hash(k) < hi
            return {k: v for k, v in nd["pairs"]}
        stats["splits"] += 1
        assert len(nd["branches"]) == R.BRANCH_FACTOR and nd["branches"][-1][0] == hi
        out = {}
        for _, child in nd["branches"]:
            out.update(walk(child, depth + 1))
        return out
    final = walk(lin.m["root"], 1)
    # ... and holds, key by key, what the history's acknowledged appends say (the order of a key's list = commit order)
    h = E.decode_history(rows, pay, N, cfg.workload)
    acked = {}
    for op in h:
        if op["type"] == ":ok" and op["process"] != ":nemesis":
            for f, k, v in op["value"]:
                if f == ":append":
                    acked.setdefault(k, set()).add(v)
    for k, vs in acked.items():
        assert vs <= set(final.get(k, [])), (k, vs, final.get(k))
    stats["keys"] = len(final)
    stats["history"] = (rows, pay)
    stats["stats"] = ora.stats[0]
    return stats


@pytest.mark.parametrize("kw,instance", [
    (dict(), 0), (dict(), 1),
    (dict(node_count=1, rate=40), 2),
    (dict(node_count=5, rate=100, time_limit=8, latency=5, nemesis=["partition"], nemesis_interval=3), 3),
    (dict(node_count=2, rate=100, time_limit=10, latency=2, latency_dist="exponential"), 4),
    (dict(node_count=3, rate=150, time_limit=10, latency=0, key_count=16, max_writes_per_key=2), 5),   # many keys: deep trees, chains
    (dict(node_count=5, rate=60, time_limit=25, latency=10, p_loss=0.02), 6),                            # lost messages: Promise#await gives up after 5 s
    (dict(node_count=3, rate=80, time_limit=25, latency=30, latency_dist="exponential", p_loss=0.05), 7),
])
def test_replay_against_the_reference_classes(kw, instance):
    cfg = _cfg(**kw)
    st = replay(cfg, instance)
    rows, pay = st["history"]
    res = E.check_txn_history(rows, pay)
    assert res["valid?"] is True, res
    if cfg.p_loss_q32:
        assert st["lost"] > 0 and st["await_timeouts"] > 0 and res["info-count"] > 0 and res["ok-count"] > 20   # the nodes recover: transactions keep completing
    else:
        assert res["info-count"] == 0 and st["await_timeouts"] == 0
        assert st["txn_ok"] == res["ok-count"] and st["cas_lost"] == res["fail-count"]
    assert st["loads"] > 0 and st["writes"] >= st["cas_ok"] + 1


LATE2 = dict(node_count=2, rate=10, time_limit=60, latency=2500, latency_dist="exponential")
LATE3 = dict(node_count=3, rate=12, time_limit=60, latency=2000, latency_dist="exponential", key_count=3)


@pytest.mark.parametrize("kw,instance,ok,lost", [(LATE2, 6, 1, 0), (LATE2, 11, 1, 0), (LATE3, 0, 2, 0), (LATE3, 6, 1, 1), (LATE3, 20, 1, 1), (LATE3, 26, 1, 0)])
def test_a_cas_served_after_its_sender_gave_up_is_still_its_own(kw, instance, ok, lost):
    """Latencies of seconds (exponential, mean 2 - 2.5 s): a cas reaches lin-kv more than Promise#await's 5 s after it was sent — its sender has
    answered error 0 and holds the NEXT transaction by then.  The request names its own `from` / `to` (datomic_list_append.rb:376-388): the
    reference classes, whose lin-kv compares the `from` in the message and whose store holds the sender's tree, and the oracle must agree on
    every reply and on every list a later transaction reads.  Round 5's oracle took `from` and the transaction from whatever the sender held
    at delivery: on each of these runs it answered the wrong reply type or completed a later transaction with the wrong lists."""
    st = replay(_cfg(**kw), instance)
    assert (st["late_cas_ok"], st["late_cas_lost"]) == (ok, lost), {k: v for k, v in st.items() if k not in ("history", "stats")}
    rows, pay = st["history"]
    res = E.check_txn_history(rows, pay)
    assert res["valid?"] is not False and not res["anomalies"], res   # (with awaits giving up all over a run may acknowledge nothing: :unknown)


@pytest.mark.parametrize("kw,instance,depth", [
    (dict(node_count=1, concurrency=10, rate=100, time_limit=8, latency=2), 0, 3),     # the reference's own invocation for this workload: --node-count 1 --concurrency 10n --rate 100
    (dict(node_count=2, concurrency=20, rate=300, time_limit=5, latency=3, latency_dist="uniform"), 1, 4),
    (dict(node_count=5, concurrency=10, rate=200, time_limit=6, latency=5, nemesis=["partition"], nemesis_interval=2), 2, 1),
    (dict(node_count=3, concurrency=9, rate=150, time_limit=20, latency=10, p_loss=0.03), 3, 2),   # lost messages: awaits give up while others wait for the lock
])
def test_several_workers_per_node_queue_behind_the_lock_in_arrival_order(kw, instance, depth):
    """`--concurrency 10n` (doc/05-datomic/01-single-node.md:257,322): ten workers talk to one node, their transactions reach it within a
    millisecond of each other and wait for its @txn_lock (datomic_list_append.rb:347-372; every message runs in its own thread, node.rb:147-183).
    The reference classes — the lock as a FIFO of coroutines — replayed against the oracle's journal: every message is the oracle's, the queue
    gets at least `depth` deep, and every answer names the oldest transaction the node has not answered yet (arrival order)."""
    st = replay(_cfg(**kw), instance)
    assert st["max_waiting"] >= depth, st["max_waiting"]
    assert st["arrival_order"]
    rows, pay = st["history"]
    res = E.check_txn_history(rows, pay)
    assert res["valid?"] is True, res
    assert res["ok-count"] > (10 if kw.get("p_loss") else 50)


def test_many_keys_grow_branches_and_chains():
    st = replay(_cfg(node_count=3, rate=150, time_limit=10, latency=0, key_count=16, max_writes_per_key=2), 5)
    assert st["keys"] > 128 and st["max_depth"] >= 4 and st["splits"] > 9, {k: v for k, v in st.items() if k not in ("history", "stats")}


def test_message_arithmetic_single_node():
    """One node, nothing lost, nothing contended: every transaction costs a root read; a transaction that appends also costs its new tree
    nodes' writes, one cas and — at the next transaction — the loads of what it wrote (each repeated while lww-kv draws the other replica)."""
    cfg = _cfg(node_count=1, rate=40, time_limit=8, latency=2)
    st = replay(cfg, 7)
    rows, pay = st["history"]
    h = [op for op in E.decode_history(rows, pay, 1, cfg.workload) if op["type"] == ":invoke"]
    n_txn, n_app = len(h), sum(1 for op in h if any(f == ":append" for f, _, _ in op["value"]))
    assert st["cas_lost"] == 0 and st["cas_ok"] == n_app
    s = st["stats"]
    # requests: init + txns from the client; 2 init writes; per txn a root read; loads (+ retries); writes; a cas per appending txn — and as many replies
    assert int(s["clients_send"]) == 2 * (1 + n_txn)
    assert int(s["servers_send"]) == 2 * (2 + n_txn + st["loads"] + st["writes"] - 1 + n_app)


@pytest.mark.parametrize("kw", [dict(latency=2), dict(latency=20, latency_dist="exponential", rate=100, node_count=5),
                                dict(latency=5, nemesis=["partition"], nemesis_interval=2, node_count=5), dict(node_count=3, rate=200, latency=1, key_count=2),
                                dict(latency=10, p_loss=0.02, node_count=5, time_limit=20)])
def test_histories_are_strict_serializable_whatever_the_schedule(kw):
    """A transaction only completes through a root cas against the exact pointer it read (or changes nothing): the list-append analysis finds
    nothing; a lost cas is REPORTED (error 30 => :fail with :txn-conflict, datomic_list_append.rb:385), never retried; a lost message leaves
    the node's lock taken (Promise#await's 5 s are not modelled), so its clients time out (:info) from then on."""
    base = dict(journal_capacity=0)
    base.update(kw)
    cfg = _cfg(**base)
    r = O.run(cfg, 0, 4)
    for i in range(4):
        assert r.meta["flags"][i] == 0
        rows, pay = r.history(i)
        ops = [o for o in E.decode_history(rows, pay, cfg.n_nodes, A.WL_TXN_LIST_APPEND) if o["process"] != ":nemesis"]
        done = [o for o in ops if o["type"] != ":invoke"]
        assert len(done) > 20
        for o in done:
            if o["type"] == ":fail":
                assert o["error"][0] == ":txn-conflict"
        if not cfg.p_loss_q32:
            assert not any(o["type"] == ":info" for o in done)
        res = E.check_txn_history(rows, pay)
        assert res["valid?"] is True and res["anomalies"] == [], res


@pytest.mark.parametrize("case", range(32))
def test_replay_random_options(case):
    """The same replay over random option sets: cluster sizes, rates, latency distributions, loss, key pools from one key to many, transactions of
    one to eight micro-ops, short-lived keys (deep trees), the partition nemesis."""
    import random
    rng = random.Random(0xDA70 + case)
    kw = dict(node_count=rng.choice([1, 2, 3, 5, 7]), rate=rng.choice([20, 50, 100, 200]), time_limit=rng.choice([4, 8, 14]), seed=rng.randrange(1 << 30),
              key_count=rng.choice([1, 3, 10, 16]), max_txn_length=rng.choice([1, 4, 8]), max_writes_per_key=rng.choice([2, 16, 40]))
    lat = rng.choice([0, 1, 5, 20])
    kw.update(latency=lat, latency_dist=rng.choice(["constant", "uniform", "exponential"]) if lat else "constant")
    if rng.random() < 0.3:
        kw["p_loss"] = rng.choice([0.02, 0.1])
    if rng.random() < 0.3 and kw["node_count"] >= 3:
        kw.update(nemesis=["partition"], nemesis_interval=rng.choice([1, 3]))
    cfg = _cfg(**kw)
    st = replay(cfg, rng.randrange(1 << 20))
    rows, pay = st["history"]
    res = E.check_txn_history(rows, pay)
    assert res["valid?"] is True or (res["valid?"] == "unknown" and res["ok-count"] == 0), res   # (nothing acknowledged: elle says :unknown)
