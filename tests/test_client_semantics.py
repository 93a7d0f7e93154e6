"""Maelstrom's synchronous client (src/maelstrom/client.clj:66-172) re-derived from the net journal: a transliteration of
`recv!` / `throw-errors!` / `with-errors` is run over what every client endpoint sent and received (journal.clj:220-239) and
must predict every completion row of the history — when it happens, :ok / :fail / :info, the error, the next process id.
Independent of the oracle's client code (no slots, no rounds): test infrastructure only."""
import collections

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

# resources/errors.edn:2-43: code -> (name, definite?)
ERRORS = {0: ("timeout", False), 1: ("node-not-found", True), 10: ("not-supported", True), 11: ("temporarily-unavailable", True),
          12: ("malformed-request", True), 13: ("crash", False), 14: ("abort", True), 20: ("key-does-not-exist", True),
          21: ("key-already-exists", True), 22: ("precondition-failed", True), 30: ("txn-conflict", True)}
# the idempotent :f sets of the workloads' with-errors (broadcast.clj:200, lin_kv.clj:52, echo.clj:35, txn_list_append.clj:113,
# txn_rw_register.clj:120, unique_ids.clj:51)
IDEMPOTENT = {"broadcast": {":read"}, "lin-kv": {":read"}, "echo": set(), "txn-list-append": set(), "txn-rw-register": set(), "unique-ids": set()}
ERR_NAME = {"net-timeout": ":net-timeout", "temporarily-unavailable": ":temporarily-unavailable", "key-does-not-exist": ":key-does-not-exist",
            "precondition-failed": ":precondition-failed", "txn-conflict": ":txn-conflict"}


def _rpc_outcome(t_send, msg_id, recvs, timeout_us, f, idempotent):
    """recv! (client.clj:81-118) + throw-errors! (:126-139) + with-errors (:155-172) for one request.  `recvs` = the envelopes
    (time, in_reply_to, type, a) this client endpoint took off its queue while it waited for this request, in journal order."""
    deadline = t_send + timeout_us
    for t, irt, typ, a in recvs:
        assert t <= deadline, "recv! returned a message after its deadline"
        if irt != msg_id:
            continue                                  # a reply to a request we gave up on: (recur)
        if typ != "error":
            return t, ":ok", None
        name, definite = ERRORS.get(a, ("unknown", False))
        return t, (":fail" if definite or f in idempotent else ":info"), ERR_NAME.get(name, name)
    return deadline, (":fail" if f in idempotent else ":info"), ":net-timeout"   # Client read timeout


CASES = [
    ("broadcast", dict(node_count=5, rate=40, time_limit=12, latency=30, latency_dist="exponential", p_loss=0.1, nemesis=["partition"], nemesis_interval=3), 5000),
    ("broadcast", dict(bin="broadcast-ack-retry", node_count=5, concurrency=10, rate=40, time_limit=10, latency=20, p_loss=0.05), 5000),
    ("echo", dict(node_count=2, rate=50, time_limit=8, p_loss=0.2), 5000),
    ("lin-kv", dict(bin="raft", node_count=5, rate=40, time_limit=25, latency=10, nemesis=["partition"], nemesis_interval=4), 1000),   # lin_kv.clj:54: max(10 x latency, 1000)
    ("lin-kv", dict(bin="lin-kv-proxy", proxy_service="lin-kv", node_count=3, rate=60, time_limit=10, latency=150, p_loss=0.1), 1500),
    ("txn-list-append", dict(node_count=5, rate=80, time_limit=10, latency=10, p_loss=0.05), 5000),
    ("txn-rw-register", dict(node_count=3, rate=80, time_limit=10, latency=10, p_loss=0.1, nemesis=["partition"], nemesis_interval=3), 5000),
    ("unique-ids", dict(node_count=3, rate=100, time_limit=8, latency=5, p_loss=0.1), 5000),
]


@pytest.mark.parametrize("workload,kw,timeout_ms", CASES)
def test_history_completions_follow_from_the_journal(workload, kw, timeout_ms):
    cfg = E.test_config(workload, seed=17, journal_capacity=400000, **kw)
    N, C = cfg.n_nodes, cfg.concurrency
    CS = max(N, C)
    seen = collections.Counter()
    for inst in range(3):
        r = O.run(cfg, inst, 1)
        assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
        rows, pay = r.history(0)
        ops = E.decode_history(rows, pay, N, E.WORKLOADS[workload])
        # per client endpoint, in journal order: every request with the envelopes received before the next request
        rpcs = collections.defaultdict(list)
        for ev in r.events(0):
            msg, route = int(ev["msg"]), int(ev["route"])
            recv, typ = (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
            src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
            if not recv and N <= src < N + CS:
                if kw.get("bin") == "raft" and dest != (src - N) % N:
                    continue                      # a Raft follower passing the client's request on to the leader, :src unchanged (raft.py:543-546)
                rpcs[src - N].append((int(ev["time_us"]), b, typ, []))
            elif recv and N <= dest < N + CS:
                rpcs[dest - N][-1][3].append((int(ev["time_us"]), b, typ, int(ev["a"])))   # a client only receives while it waits
        # what the history says, per worker: [(invoke time us, f, process), (completion time us, type, error)] ...
        hist = collections.defaultdict(list)
        for op in ops:
            if op["process"] == ":nemesis":
                continue
            slot = op["process"] % C
            err = op.get("error")
            hist[slot].append((op["time"] // 1000, op["type"], op["f"], op["process"], err[0] if isinstance(err, list) else err))
        for slot in range(CS):
            mine, process = collections.deque(hist.get(slot, [])), slot
            for t_send, msg_id, typ, got in rpcs[slot]:
                if typ in ("init", "topology"):   # db/setup! and the broadcast client's setup!: 10 s / default timeout, no history rows
                    t, kind, _ = _rpc_outcome(t_send, msg_id, got, (10000 if typ == "init" else 5000) * 1000, None, set())
                    assert kind == ":ok"
                    continue
                inv = mine.popleft()
                assert inv[:2] == (t_send, ":invoke") and inv[3] == process, (slot, inv, t_send, process)
                t, kind, err = _rpc_outcome(t_send, msg_id, got, timeout_ms * 1000, inv[2], IDEMPOTENT[workload])
                done = mine.popleft()
                assert (done[0], done[1], done[2], done[3], done[4]) == (t, kind, inv[2], process, err), (slot, done, t, kind, err)
                seen[kind] += 1
                seen[err] += 1
                if kind == ":info":
                    process += C                  # the worker's process crashed: a new one takes over [upstream interpreter]
            assert not mine
    assert seen[":ok"] > 30 and seen[":net-timeout"] > 0
    if workload == "lin-kv":
        assert seen[":fail"] > 0
