#!/usr/bin/env python3
"""Records golden node-transition vectors from the reference's OWN runnable node programs.

The reference (Clojure/JVM + Ruby/babashka demos) cannot run in the build container, but three of its
demo node programs can: demo/python/echo.py, demo/js/gossip.js and demo/js/crdt_gset.js.
This script spawns each as a real process, speaks the wire protocol of doc/protocol.md over its
stdin/stdout exactly as maelstrom.process does (process.clj:136-166), and records, for a fixed input
script, every message the node prints.  The result is committed as tests/golden/node_transitions.json
and replayed against the CPU oracle's node transition functions by tests/test_golden_transitions.py
(`/root/reference` does not exist on the GPU box, so only the committed fixture travels).

    python tests/golden/make_golden.py            # needs /root/reference, python3, node
"""
import json
import os
import select
import subprocess
import sys
import time

REF = os.environ.get("MAELSTROM_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "node_transitions.json")


class Node:
    def __init__(self, argv, cwd):
        self.p = subprocess.Popen(argv, cwd=cwd, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=0)
        self.buf = b""

    def send(self, msg):
        self.p.stdin.write((json.dumps(msg) + "\n").encode())
        self.p.stdin.flush()

    def drain(self, quiet=0.1, total=3.0):
        """All lines printed until the node has been quiet for `quiet` seconds."""
        out, t_end, t_quiet = [], time.time() + total, time.time() + quiet
        while time.time() < min(t_end, t_quiet):
            r, _, _ = select.select([self.p.stdout], [], [], 0.05)
            if r:
                chunk = os.read(self.p.stdout.fileno(), 65536)
                if not chunk:
                    break
                self.buf += chunk
                t_quiet = time.time() + quiet
                while b"\n" in self.buf:
                    line, self.buf = self.buf.split(b"\n", 1)
                    if line.strip():
                        out.append(json.loads(line))
        return out

    def close(self):
        self.p.kill()
        self.p.wait()


def run_script(argv, cwd, script):
    node = Node(argv, cwd)
    steps = []
    try:
        for step in script:
            if "wait" in step:
                outs = node.drain(quiet=step["wait"], total=step["wait"] + 0.2)
                steps.append({"wait_ms": int(step["wait"] * 1000), "out": outs})
            else:
                node.send(step)
                first = not steps  # interpreter start-up: give the init reply time to appear
                steps.append({"in": step, "out": node.drain(quiet=0.8 if first else 0.1)})
    finally:
        node.close()
    return steps


def msg(src, dest, **body):
    return {"src": src, "dest": dest, "body": body}


def main():
    nodes5 = ["n0", "n1", "n2", "n3", "n4"]
    cases = {}
    # ---- echo: demo/python/echo.py (a13) ----
    cases["echo.py"] = {
        "source": "demo/python/echo.py", "node": "n1", "node_ids": ["n0", "n1", "n2"],
        "steps": run_script([sys.executable, "echo.py"], os.path.join(REF, "demo/python"), [
            msg("c0", "n1", type="init", msg_id=1, node_id="n1", node_ids=["n0", "n1", "n2"]),
            msg("c3", "n1", type="echo", msg_id=1, echo="Please echo 35"),
            msg("c3", "n1", type="echo", msg_id=2, echo="Please echo 101"),
            msg("c4", "n1", type="echo", msg_id=1, echo="Please echo 0"),
        ])}
    # ---- broadcast with ack + retry: demo/js/gossip.js (a14 ii) ----
    topo = {"n0": ["n1"], "n1": ["n0", "n2", "n4"], "n2": ["n1", "n3"], "n3": ["n2"], "n4": ["n1"]}
    cases["gossip.js"] = {
        "source": "demo/js/gossip.js", "node": "n1", "node_ids": nodes5, "neighbors": topo["n1"],
        "steps": run_script(["node", "gossip.js"], os.path.join(REF, "demo/js"), [
            msg("c0", "n1", type="init", msg_id=1, node_id="n1", node_ids=nodes5),
            msg("c5", "n1", type="topology", msg_id=1, topology=topo),
            msg("c10", "n1", type="broadcast", msg_id=1, message=7),          # new, from a client
            msg("n0", "n1", type="broadcast_ok", in_reply_to=0),              # ack of the first RPC (to n0)
            msg("n2", "n1", type="broadcast", msg_id=9, message=7),           # duplicate from a peer
            msg("n2", "n1", type="broadcast", msg_id=10, message=8),          # new, from a peer
            msg("c10", "n1", type="read", msg_id=2),
            {"wait": 0.95},                                                    # 1 s RPC timeout -> first retries only
        ])}
    # ---- g-set CRDT: demo/js/crdt_gset.js (a15) ----
    cases["crdt_gset.js"] = {
        "source": "demo/js/crdt_gset.js", "node": "n1", "node_ids": nodes5,
        "steps": run_script(["node", "crdt_gset.js"], os.path.join(REF, "demo/js"), [
            msg("c0", "n1", type="init", msg_id=1, node_id="n1", node_ids=nodes5),
            msg("c10", "n1", type="add", msg_id=1, element=3),
            msg("c10", "n1", type="add", msg_id=2, element=5),
            msg("c10", "n1", type="read", msg_id=3),
            msg("n2", "n1", type="replicate", value=[9, 3]),
            msg("c10", "n1", type="read", msg_id=4),
            {"wait": 5.3},                                                     # setInterval(5000): replicate to all peers
        ])}
    # ---- PN-counter CRDT: demo/js/crdt_pn_counter.js (workload/pn_counter.clj) ----
    cases["crdt_pn_counter.js"] = {
        "source": "demo/js/crdt_pn_counter.js", "node": "n1", "node_ids": nodes5,
        "steps": run_script(["node", "crdt_pn_counter.js"], os.path.join(REF, "demo/js"), [
            msg("c0", "n1", type="init", msg_id=1, node_id="n1", node_ids=nodes5),
            msg("c10", "n1", type="add", msg_id=1, delta=3),
            msg("c10", "n1", type="add", msg_id=2, delta=-2),
            msg("c10", "n1", type="add", msg_id=3, delta=0),
            msg("c10", "n1", type="read", msg_id=4),
            msg("n2", "n1", type="replicate", value={"plus": {"n2": 4, "n1": 1, "n0": 7}, "minus": {"n0": 2, "n1": 5}}),
            msg("c10", "n1", type="read", msg_id=5),
            msg("n3", "n1", type="replicate", value={"plus": {"n2": 1}, "minus": {"n3": 9}}),
            msg("c10", "n1", type="read", msg_id=6),
            {"wait": 5.3},                                                     # setInterval(5000): replicate to all peers
        ])}
    with open(OUT, "w") as f:
        json.dump({"generated_by": "tests/golden/make_golden.py", "reference": "jepsen-io/maelstrom demo node programs",
                   "cases": cases}, f, indent=1, sort_keys=True)
    for k, c in cases.items():
        print(k, [len(s["out"]) for s in c["steps"]])


if __name__ == "__main__":
    main()
