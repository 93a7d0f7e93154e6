#!/usr/bin/env python3
"""Records golden vectors from the reference's runnable Raft node, demo/python/raft.py.

One real raft.py process plays n1 of a 5-node cluster; this script plays the network: it feeds init, client
requests, request_vote / append_entries from peers, and (using the msg_ids the node actually emitted) the
votes and acks that make it leader and let it commit.  Every line the node prints is recorded.  Output:
tests/golden/raft_transitions.json, replayed against the oracle by tests/test_golden_transitions.py.

    python tests/golden/make_golden_raft.py       # needs /root/reference and python3; takes ~12 s
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_golden import Node, REF, msg  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "raft_transitions.json")
NODES = ["n0", "n1", "n2"]
# NOTE (reference quirk): raft.py:391-410 builds its append_entries callbacks inside a `for node in other_nodes()`
# loop; Python closures bind `_node`/`_ni`/`_entries` late, so EVERY ack is credited to the last peer.  With 3 nodes
# and acks coming from the last peer (n2) raft.py and the canonical demo/ruby/raft.rb behave identically, which is
# what this recording uses (the docs also run raft with --node-count 3, doc/06-raft/04-committing.md:418).


def main():
    node = Node([sys.executable, "raft.py"], os.path.join(REF, "demo/python"))
    steps = []

    def feed(m, quiet=0.15, total=3.0):
        node.send(m)
        out = node.drain(quiet=quiet, total=total)
        steps.append({"in": m, "out": out})
        return out

    def wait(sec):
        out = node.drain(quiet=sec, total=sec + 0.3)
        steps.append({"wait_ms": int(sec * 1000), "out": out})
        return out

    try:
        feed(msg("c0", "n1", type="init", msg_id=1, node_id="n1", node_ids=NODES), quiet=0.8)
        feed(msg("c10", "n1", type="read", msg_id=1, key=0))                               # no leader known -> error 11
        feed(msg("n2", "n1", type="request_vote", msg_id=5, term=1, candidate_id="n2", last_log_index=1, last_log_term=0))
        feed(msg("n0", "n1", type="request_vote", msg_id=7, term=1, candidate_id="n0", last_log_index=1, last_log_term=0))  # already voted
        e1 = {"term": 1, "op": {"type": "write", "key": 0, "value": 3, "msg_id": 1, "client": "c11"}}
        feed(msg("n2", "n1", type="append_entries", msg_id=6, term=1, leader_id="n2", prev_log_index=1, prev_log_term=0,
                 entries=[e1], leader_commit=1))
        feed(msg("c10", "n1", type="read", msg_id=2, key=0))                               # proxied to the leader, src unchanged
        feed(msg("n2", "n1", type="append_entries", msg_id=8, term=1, leader_id="n2", prev_log_index=2, prev_log_term=1,
                 entries=[], leader_commit=2))                                             # commit -> apply (no reply: not leader)
        feed(msg("n2", "n1", type="append_entries", msg_id=9, term=1, leader_id="n2", prev_log_index=5, prev_log_term=1,
                 entries=[], leader_commit=2))                                             # gap -> success false
        feed(msg("n0", "n1", type="append_entries", msg_id=3, term=0, leader_id="n0", prev_log_index=1, prev_log_term=0,
                 entries=[], leader_commit=0))                                             # stale term -> false
        feed(msg("n0", "n1", type="request_vote", msg_id=2, term=3, candidate_id="n0", last_log_index=1, last_log_term=0))  # newer term, older log
        out = wait(4.6)                                                                    # election timeout (2-4 s): become candidate
        rv = [o for o in out if o["body"]["type"] == "request_vote"]
        assert len(rv) == 2, out
        term = rv[0]["body"]["term"]
        by_dest = {o["dest"]: o["body"]["msg_id"] for o in rv}
        out = feed(msg("n0", "n1", type="request_vote_res", in_reply_to=by_dest["n0"], term=term, vote_granted=True), quiet=0.03, total=0.04)  # majority of 3 -> leader -> heartbeats
        assert len([o for o in out if o["body"]["type"] == "append_entries"]) == 2, out
        out = feed(msg("c12", "n1", type="write", msg_id=1, key=1, value=4), quiet=0.2, total=0.2)  # appended; > 50 ms later: replicated (re-sent until acked)
        ae2 = {o["dest"]: o["body"]["msg_id"] for o in out if o["body"]["type"] == "append_entries" and o["body"]["entries"]}
        assert len(ae2) == 2, out
        feed(msg("n2", "n1", type="append_entries_res", in_reply_to=ae2["n2"], term=term, success=True), quiet=0.03, total=0.04)  # majority -> commit -> apply -> write_ok
        out = feed(msg("c10", "n1", type="cas", msg_id=3, key=1, **{"from": 4, "to": 2}), quiet=0.2, total=0.2)
        ae3 = {o["dest"]: o["body"]["msg_id"] for o in out if o["body"]["type"] == "append_entries" and len(o["body"]["entries"]) >= 1}
        feed(msg("n2", "n1", type="append_entries_res", in_reply_to=ae3["n2"], term=term, success=True), quiet=0.03, total=0.04)  # -> cas_ok
        feed(msg("c10", "n1", type="cas", msg_id=4, key=1, **{"from": 4, "to": 0}), quiet=0.2, total=0.2)            # will fail: value is 2
        out = steps[-1]["out"]
        ae4 = {o["dest"]: o["body"]["msg_id"] for o in out if o["body"]["type"] == "append_entries" and len(o["body"]["entries"]) >= 1}
        feed(msg("n2", "n1", type="append_entries_res", in_reply_to=ae4["n2"], term=term, success=True), quiet=0.03, total=0.04)  # -> error 22
        feed(msg("c10", "n1", type="read", msg_id=5, key=7), quiet=0.2, total=0.2)                                   # key 7 absent
        out = steps[-1]["out"]
        ae5 = {o["dest"]: o["body"]["msg_id"] for o in out if o["body"]["type"] == "append_entries" and len(o["body"]["entries"]) >= 1}
        feed(msg("n2", "n1", type="append_entries_res", in_reply_to=ae5["n2"], term=term, success=True), quiet=0.03, total=0.04)  # -> error 20
    finally:
        node.close()
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden_raft.py", "source": "demo/python/raft.py", "node": "n1",
                   "node_ids": NODES, "steps": steps}, f, indent=1, sort_keys=True)
    for s in steps:
        print(("wait %d" % s["wait_ms"]) if "wait_ms" in s else s["in"]["body"]["type"], "->",
              [(o["dest"], o["body"]["type"]) for o in s["out"]])


if __name__ == "__main__":
    main()
