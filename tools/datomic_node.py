#!/usr/bin/env python3
"""A Maelstrom node PROCESS for txn-list-append: the Ruby classes of demo/ruby/datomic_list_append.rb (Tree / Leaf / Branch with real maps, lists
and JSON-shaped nodes, @@cache, the node's lock, Promise) as tests/datomic_ref.py writes them out in Python, behind the wire protocol — one JSON
message per line on stdin / stdout (doc/protocol.md).  Test infrastructure: there is no Ruby in this image, so this is the closest thing to the
reference's `--bin demo/ruby/datomic_list_append.rb` that runs here; `tests/test_process_bridge.py` starts it under maelstrom_amd/bridge.py (real
processes, pipes, the bridge's own lin-kv / lww-kv services holding REAL values, its own scheduler) and holds the histories to the oracle's.
Promise#await's 5 s run on the wall clock here (as in the reference); a loss-free virtual-time run never gets there."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import datomic_ref as R  # noqa: E402


def main():
    out = sys.stdout
    me = [None]

    def send(dest, body):
        out.write(json.dumps({"src": me[0], "dest": dest, "body": body}) + "\n")
        out.flush()

    node = R.DatomicListAppendNode(send, clock=lambda: int(time.monotonic() * 1e6))
    for line in sys.stdin:
        line = line.strip()
        if not line:
            continue
        msg = json.loads(line)
        if msg["body"].get("type") == "init":
            me[0] = msg["body"]["node_id"]
        node.fire_due(int(time.monotonic() * 1e6))
        node.handle(msg)


if __name__ == "__main__":
    main()
