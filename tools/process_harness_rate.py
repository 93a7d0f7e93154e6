#!/usr/bin/env python3
"""Stand-in for the reference's cost structure ("CPU reference beside it", item 2): real node processes of the
reference's own demo/js/gossip.js, one per node, JSON lines over pipes, routed by a trivial in-memory network as fast as
possible — no JVM, no virtual time, no latency, no journal.  Prints the message rate one n=25 grid cluster sustains on this
machine's cores, i.e. what process-per-node + JSON + pipes cost before Maelstrom itself adds its share.
With node.js and /root/reference at hand (build container) the nodes are the reference's demo/js/gossip.js (acknowledged gossip);
anywhere else they are tools/harness_node.py (python3, the fire-and-forget node).  Usage: tools/process_harness_rate.py [n_nodes] [n_broadcasts]"""
import json
import os
import select
import subprocess
import sys
import time

JS = "/root/reference/demo/js/gossip.js"
PY_NODE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "harness_node.py")


def node_command(prefer_reference=True):
    """(argv, description) of the node process: the reference's own gossip.js if it can run here, else this repository's python node"""
    import shutil
    if prefer_reference and os.path.exists(JS) and shutil.which("node"):
        return ["node", JS], "node demo/js/gossip.js (reference demo, acknowledged gossip)"
    return [sys.executable, PY_NODE], "python3 tools/harness_node.py (fire-and-forget broadcast node, one process per node)"


def grid(n):
    side = 1
    while side * side < n:
        side += 1
    nb = {i: [] for i in range(n)}
    for i in range(side):
        for j in range(side):
            a = i * side + j
            if a >= n:
                continue
            if j + 1 < side and a + 1 < n:
                nb[a].append(a + 1); nb[a + 1].append(a)
            if a + side < n:
                nb[a].append(a + side); nb[a + side].append(a)
    return nb


def measure(n=25, k=2000, prefer_reference=True):
    argv, what = node_command(prefer_reference)
    procs = [subprocess.Popen(argv, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=0) for _ in range(n)]
    fd2node = {p.stdout.fileno(): i for i, p in enumerate(procs)}
    bufs = {i: b"" for i in range(n)}
    names = [f"n{i}" for i in range(n)]
    topo = {names[i]: [names[j] for j in v] for i, v in grid(n).items()}
    sent = 0

    def to(i, msg):
        nonlocal sent
        procs[i].stdin.write((json.dumps(msg) + "\n").encode())
        sent += 1

    def pump(idle):
        """route node -> node messages until nothing has been printed for `idle` seconds"""
        nonlocal sent
        while True:
            r, _, _ = select.select(list(fd2node), [], [], idle)
            if not r:
                return
            for fd in r:
                i = fd2node[fd]
                bufs[i] += os.read(fd, 1 << 16)
                while b"\n" in bufs[i]:
                    line, bufs[i] = bufs[i].split(b"\n", 1)
                    m = json.loads(line)
                    if m["dest"].startswith("n"):
                        to(int(m["dest"][1:]), m)
                    else:
                        sent += 1      # a reply to the client: counted like net/send! counts it

    for i in range(n):
        to(i, {"src": "c0", "dest": names[i], "body": {"type": "init", "msg_id": 1, "node_id": names[i], "node_ids": names}})
    pump(0.3)
    for i in range(n):
        to(i, {"src": "c0", "dest": names[i], "body": {"type": "topology", "msg_id": 2, "topology": topo}})
    pump(0.3)
    sent = 0
    t0 = time.perf_counter()
    for v in range(k):
        to(v % n, {"src": "c0", "dest": names[v % n], "body": {"type": "broadcast", "msg_id": 3 + v, "message": v}})
        if v % 4 == 3:
            pump(0.002)    # a few cascades in flight at a time: the pipes must never fill up in both directions
    pump(0.5)
    dt = time.perf_counter() - t0 - 0.5
    for p in procs:
        p.kill()
    for p in procs:
        p.wait()
    return {"harness": "%s x %d over pipes, grid" % (what, n), "broadcasts": k, "messages": sent, "seconds": round(dt, 2),
            "msgs_per_s": round(sent / dt), "host_cpus": len(os.sched_getaffinity(0))}


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 25
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
    print(json.dumps(measure(n, k, prefer_reference="--python-nodes" not in sys.argv)))


if __name__ == "__main__":
    main()
