#!/usr/bin/env python3
"""A unique-ids node that takes its ids from Maelstrom's `lin-tso` timestamp-oracle service (service.clj:116-132,290-296;
doc/services.md "lin-tso"): `generate` -> {type "ts"} RPC to lin-tso -> `generate_ok` with the timestamp as id.  The reference ships
the service but no demo that uses it; this node is what exercises it under the process bridge (maelstrom_amd/bridge.py)."""
import json
import sys


def main():
    node_id, next_id, waiting = None, 0, {}
    for line in sys.stdin:
        msg = json.loads(line)
        body, src = msg["body"], msg["src"]
        t = body["type"]
        out = []
        if t == "init":
            node_id = body["node_id"]
            out.append((src, {"type": "init_ok", "in_reply_to": body["msg_id"]}))
        elif t == "generate":
            next_id += 1
            waiting[next_id] = (src, body["msg_id"])
            out.append(("lin-tso", {"type": "ts", "msg_id": next_id}))
        elif t == "ts_ok":
            client, mid = waiting.pop(body["in_reply_to"])
            out.append((client, {"type": "generate_ok", "id": body["ts"], "in_reply_to": mid}))
        for dest, b in out:
            sys.stdout.write(json.dumps({"src": node_id, "dest": dest, "body": b}) + "\n")
        sys.stdout.flush()


if __name__ == "__main__":
    main()
