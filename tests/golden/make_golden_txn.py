#!/usr/bin/env python3
"""Records golden conversations of the reference's OWN transactional node, demo/js/single_key_txn.js (the JavaScript twin
of demo/clojure/single_key_txn.clj, which needs babashka and cannot run here): two real node processes over pipes, the
harness playing the clients and the `lin-kv` service (a dict with the read / cas-with-create_if_not_exists semantics of
service.clj:31-61).  Every message the nodes print is recorded in order, together with what the harness answered.
Committed as tests/golden/txn_transitions.json, replayed against the oracle's node + service transition functions by
tests/test_golden_transitions.py (`/root/reference` does not exist on the GPU box).

One difference between the two demos is visible in the recording and documented in the test: for a missing key the JS
node uses `[]` as the state it read (getKey(k, [])), the Clojure node `nil`; with create_if_not_exists both create the key.

    python tests/golden/make_golden_txn.py            # needs /root/reference and node
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import Node, REF, msg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "txn_transitions.json")


class LinKV:
    """service.clj:31-61 (PersistentKV) behind Linearizable (:141-155): one request at a time."""

    def __init__(self):
        self.m = {}

    def handle(self, body):
        k = body["key"]
        if body["type"] == "read":
            return {"type": "read_ok", "value": self.m[k]} if k in self.m else {"type": "error", "code": 20, "text": "key does not exist"}
        if body["type"] == "cas":
            if k in self.m:
                if body["from"] == self.m[k]:
                    self.m[k] = body["to"]
                    return {"type": "cas_ok"}
                return {"type": "error", "code": 22, "text": "current value %r is not %r" % (self.m[k], body["from"])}
            if body.get("create_if_not_exists"):
                self.m[k] = body["to"]
                return {"type": "cas_ok"}
            return {"type": "error", "code": 20, "text": "key does not exist"}
        raise ValueError(body)


def main():
    ids = ["n0", "n1"]
    nodes = {n: Node(["node", "single_key_txn.js"], os.path.join(REF, "demo/js")) for n in ids}
    kv = LinKV()
    log = []       # {"to": endpoint, "msg": message delivered to it} and {"from": node, "msg": message it printed}
    pending = []   # service requests printed by nodes, not yet answered

    def deliver(m, quiet=0.15):
        log.append({"deliver": m})
        if m["dest"] == "lin-kv":
            reply = kv.handle(m["body"])
            reply["in_reply_to"] = m["body"]["msg_id"]
            out = [msg("lin-kv", m["src"], **reply)]
            log.append({"emitted_by": "lin-kv", "out": out})
            return out
        nodes[m["dest"]].send(m)
        out = nodes[m["dest"]].drain(quiet=quiet)
        log.append({"emitted_by": m["dest"], "out": out})
        return out

    def settle(outs):
        """Delivers service traffic until nothing is pending: node -> service requests and service -> node replies."""
        queue = list(outs)
        done = []
        while queue:
            m = queue.pop(0)
            if m["dest"] == "lin-kv" or m["src"] == "lin-kv":
                queue.extend(deliver(m))
            else:
                done.append(m)  # a reply to a client
        return done

    try:
        for i, n in enumerate(ids):
            nodes[n].send(msg("c%d" % i, n, type="init", msg_id=1, node_id=n, node_ids=ids))
            out = nodes[n].drain(quiet=1.5, total=4.0)
            log.append({"deliver": msg("c%d" % i, n, type="init", msg_id=1, node_id=n, node_ids=ids)})
            log.append({"emitted_by": n, "out": out})
        # 1. the first transaction ever: root missing -> created
        settle(deliver(msg("c2", "n0", type="txn", msg_id=1, txn=[["append", 1, 1], ["r", 1, None]])))
        # 2. reads of existing + missing keys, append to a fresh key, read-your-writes
        settle(deliver(msg("c3", "n1", type="txn", msg_id=1, txn=[["r", 1, None], ["append", 2, 5], ["r", 2, None], ["r", 7, None]])))
        # 3. a conflict: both nodes read the same state, n0's cas wins, n1's loses -> error 30
        o0 = deliver(msg("c2", "n0", type="txn", msg_id=2, txn=[["append", 1, 2]]))
        o1 = deliver(msg("c3", "n1", type="txn", msg_id=2, txn=[["append", 1, 3], ["r", 1, None]]))
        r0 = deliver(o0[0])      # read by n0
        r1 = deliver(o1[0])      # read by n1: same state
        c0 = deliver(r0[0])      # n0 gets the state, emits cas
        c1 = deliver(r1[0])      # n1 gets the state, emits cas
        settle(deliver(c0[0]))   # n0's cas succeeds
        settle(deliver(c1[0]))   # n1's cas: code 22 -> error 30 to the client
        # 4. a read-only transaction writes back the state it read
        settle(deliver(msg("c3", "n1", type="txn", msg_id=3, txn=[["r", 1, None], ["r", 2, None]])))
        # 5. several appends to one key inside a transaction
        settle(deliver(msg("c2", "n0", type="txn", msg_id=3, txn=[["append", 3, 1], ["append", 3, 2], ["r", 3, None], ["append", 1, 4], ["r", 1, None]])))
    finally:
        for n in nodes.values():
            n.close()
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden_txn.py", "source": "demo/js/single_key_txn.js", "node_ids": ids,
                   "final_store": kv.m, "log": log}, f, indent=1, sort_keys=True)
    print(len(log), "log entries;", "final store:", kv.m)


if __name__ == "__main__":
    main()
