#!/usr/bin/env python3
"""A Maelstrom node process for the process-faithful CPU stand-in (tools/process_harness_rate.py, bench.py): the fire-and-forget
broadcast node of doc/03-broadcast/01-broadcast.md:525-547 + 02-performance.md:61-67 (dedup, forward to every neighbour but the
sender), speaking the wire protocol of doc/protocol.md — one JSON message per line on stdin / stdout.  Written for this
repository (it is what the engine's MSIM_NODE_BCAST_FF program does, as a process); the reference's own demo/js/gossip.js is used
instead when node.js and the reference tree are at hand."""
import json
import sys


def main():
    node_id, neighbors, seen, next_id = None, [], set(), 0
    out = sys.stdout
    for line in sys.stdin:
        msg = json.loads(line)
        body, src = msg["body"], msg["src"]
        t = body["type"]
        reply = None
        if t == "broadcast":
            m = body["message"]
            if m not in seen:
                seen.add(m)
                for nb in neighbors:
                    if nb != src:
                        out.write(json.dumps({"src": node_id, "dest": nb, "body": {"type": "broadcast", "message": m}}) + "\n")
            if "msg_id" in body:
                reply = {"type": "broadcast_ok"}
        elif t == "read":
            reply = {"type": "read_ok", "messages": sorted(seen)}
        elif t == "init":
            node_id = body["node_id"]
            reply = {"type": "init_ok"}
        elif t == "topology":
            neighbors = body["topology"].get(node_id, [])
            reply = {"type": "topology_ok"}
        if reply is not None:
            next_id += 1
            reply["msg_id"] = next_id
            reply["in_reply_to"] = body["msg_id"]
            out.write(json.dumps({"src": node_id, "dest": src, "body": reply}) + "\n")
        out.flush()


if __name__ == "__main__":
    main()
