"""Generic process bridge: Maelstrom's node execution for ARBITRARY `--bin` programs on the engine's
deterministic scheduler — the CPU path beside the GPU built-ins, with the same round semantics and the same
counter-based RNG, so a run with a real node binary is reproducible from (seed, instance) and, for a program that behaves like a
built-in node, emits the history the engine emits.

What it replaces, with the reference's own behaviour (file:line under /root/reference):
  * process.clj:168-215 start-node! / stop-node!: one OS process per node, stdin / stdout pipes, stderr to a log file;
  * process.clj:154-166 stdin-thread: `recv!` an envelope -> one JSON line {src, dest, body} on the node's stdin;
  * process.clj:136-152 stdout-thread + parse-msg :35-66: every line the node prints is parsed, validated (a map with :dest and a
    :body map) and `send!` with :src = the node's own id;
  * net.clj:189-247 send! / recv! (ids for every send, latency only between servers, loss after the journal, partitions consulted at
    poll time, head-of-line blocking with ms-truncated sleeps) — restated in Python below, as the oracle restates it in C;
  * client.clj:41-172 sync RPC clients (5 s timeout, stale replies skipped, `with-errors` definite / indefinite mapping from
    resources/errors.edn), db.clj:46-69 the init handshake, core.clj:67-80 generator phases, the workloads' request bodies
    (doc/workloads.md) for echo, broadcast, g-set, g-counter, pn-counter, unique-ids, lin-kv and txn-list-append;
  * service.clj:31-132,141-263,290-296 the built-in services lin-kv, seq-kv, lww-kv and lin-tso as endpoints any node may call.

Virtual time and real programs.  A reactive node (everything it ever prints is the reaction to a line it just read) runs in
virtual time: after the due envelopes of a round are written, the bridge reads what the nodes print until all of them are DONE —
on Linux that is read off /proc (a node is done when it has consumed its stdin and every thread of it sleeps, twice in a row with
no CPU time slice in between; NodeProcess.idle), which holds however loaded the machine is; without /proc, until every node has
been quiet for `settle_ms` of real time.  What was printed is the round's output — no wall-clock sleeps, thousands of virtual
seconds per real second.  A
node with timers of its own (periodic replication, election timeouts) needs `clock="real"`: virtual microseconds then follow the
wall clock, spontaneous output is sent at the instant it is seen — the reference's own behaviour (it has no other mode).

Host-side tool: pure Python, no device.  `python -m maelstrom_amd.bridge test -w broadcast --bin ./node.py --node-count 5 ...`
"""
import argparse
import fcntl
import json
import math
import os
import select
import struct
import subprocess
import sys
import termios
import time

MASK64 = (1 << 64) - 1
PHI = 0x9E3779B97F4A7C15
INF = 0xFFFFFFFF
S_GEN, S_GEN2, S_GEN3, S_LATENCY, S_LOSS, S_NEM_STAGGER, S_NEM_SPEC, S_NEM_SHUFFLE, S_NEM_PICK, S_SVC = 1, 2, 3, 4, 5, 7, 8, 9, 10, 12
LOG2_Q24 = [int(round(math.log2(1.0 + i / 256.0) * (1 << 24))) for i in range(257)]   # tools/gen_log2_table.py

# resources/errors.edn: code -> (name, definite?)
ERRORS = {0: ("timeout", False), 1: ("node-not-found", True), 10: ("not-supported", True), 11: ("temporarily-unavailable", True),
          12: ("malformed-request", True), 13: ("crash", False), 14: ("abort", True), 20: ("key-does-not-exist", True),
          21: ("key-already-exists", True), 22: ("precondition-failed", True), 30: ("txn-conflict", True)}


def mix64(z):
    z &= MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & MASK64
    return z ^ (z >> 31)


class Rng:
    """key = mix64(seed + phi (instance + 1)); draw64(stream, ctr) = mix64(key + ((stream << 48) | ctr) phi)"""

    def __init__(self, seed, instance):
        self.key = mix64(seed + PHI * (instance + 1))

    def draw64(self, stream, ctr):
        return mix64(self.key + (((stream << 48) | ctr) * PHI))

    def draw32(self, stream, ctr):
        return self.draw64(stream, ctr) >> 32


def scale32(r, n):
    return (r * n) >> 32


def neg_ln_q16(r):
    """-ln((r + 1) / 2^32) in Q16, integers only (the oracle's and the engine's sampler, net.clj:77)"""
    if r == 0xFFFFFFFF:
        return 0
    v = r + 1
    e = v.bit_length() - 1
    m = (v << (31 - e)) & 0xFFFFFFFF
    idx, f = (m >> 23) & 0xFF, (m >> 7) & 0xFFFF
    l0, l1 = LOG2_Q24[idx], LOG2_Q24[idx + 1]
    lg = (e << 24) + l0 + (((l1 - l0) * f) >> 16)
    return (((32 << 24) - lg) * 2977044472) >> 40


# ---- services (service.clj) ---------------------------------------------------------------------------------------------------
class PersistentKV:
    """service.clj:31-61; state = dict (copied on change: states are values, service.clj:180 compares them)"""

    def __init__(self, m=None):
        self.m = {} if m is None else m

    def handle(self, body):
        k = _key(body.get("key"))
        t = body["type"]
        if t == "read":
            if k in self.m:
                return self, {"type": "read_ok", "value": self.m[k]}
            return self, {"type": "error", "code": 20, "text": "key does not exist"}
        if t == "write":
            return PersistentKV({**self.m, k: body.get("value")}), {"type": "write_ok"}
        if t == "cas":
            if k in self.m:
                if body.get("from") == self.m[k]:
                    return PersistentKV({**self.m, k: body.get("to")}), {"type": "cas_ok"}
                return self, {"type": "error", "code": 22, "text": f"current value {json.dumps(self.m[k])} is not {json.dumps(body.get('from'))}"}
            if body.get("create_if_not_exists"):
                return PersistentKV({**self.m, k: body.get("to")}), {"type": "cas_ok"}
            return self, {"type": "error", "code": 20, "text": "key does not exist"}
        return self, {"type": "error", "code": 10, "text": f"unsupported request type {t}"}

    def same(self, other):
        return self.m == other.m


class LWWKV:
    """service.clj:65-114: values carry the Lamport clock of their write"""

    def __init__(self, clock=0, m=None):
        self.clock, self.m = clock, ({} if m is None else m)

    def handle(self, body):
        k = _key(body.get("key"))
        t = body["type"]
        if t == "read":
            if k in self.m:
                return self, {"type": "read_ok", "value": self.m[k][1]}
            return self, {"type": "error", "code": 20, "text": "key does not exist"}
        if t == "write":
            return LWWKV(self.clock + 1, {**self.m, k: (self.clock, body.get("value"))}), {"type": "write_ok"}
        if t == "cas":
            if k in self.m:
                if body.get("from") == self.m[k][1]:
                    return LWWKV(self.clock + 1, {**self.m, k: (self.clock, body.get("to"))}), {"type": "cas_ok"}
                return self, {"type": "error", "code": 22, "text": f"current value {json.dumps(self.m[k][1])} is not {body.get('from')}"}
            return self, {"type": "error", "code": 20, "text": "key does not exist"}
        return self, {"type": "error", "code": 10, "text": f"unsupported request type {t}"}


class PersistentTSO:
    """service.clj:116-132: {:type "ts"} -> {:type "ts_ok", :ts n}, n = 0, 1, 2, ..."""

    def __init__(self, ts=0):
        self.ts = ts

    def handle(self, body):
        if body["type"] == "ts":
            return PersistentTSO(self.ts + 1), {"type": "ts_ok", "ts": self.ts}
        return self, {"type": "error", "code": 10, "text": f"unsupported request type {body['type']}"}

    def same(self, other):
        return self.ts == other.ts


def _key(k):
    return json.dumps(k, sort_keys=True)   # JSON keys of any shape as dict keys


class Linearizable:
    """service.clj:141-155"""

    def __init__(self, svc):
        self.svc = svc

    def handle(self, src, body, rand_int):
        self.svc, res = self.svc.handle(body)
        return res


class Sequential:
    """service.clj:161-210: a ring of the last `size` states; a request that does not change the state it is tried on may be served
    from any state between the client's last one and the newest (rand-int), anything else runs on the newest and appends a state.
    A state that has left the ring raises in the worker: no reply (service.clj:259-260)."""

    def __init__(self, svc, size=32):
        self.ring, self.size, self.last, self.clients = {0: svc}, size, 0, {}

    def handle(self, src, body, rand_int):
        ci = self.clients.get(src, 0)
        index = ci + rand_int(self.last - ci + 1)
        if index not in self.ring:
            return None
        state = self.ring[index]
        new, res = state.handle(body)
        if new is state or new.same(state):
            self.clients[src] = index
            return res
        new, res = self.ring[self.last].handle(body)
        self.last += 1
        self.ring[self.last] = new
        self.ring.pop(self.last - self.size, None)
        self.clients[src] = self.last
        return res


class Eventual:
    """service.clj:214-243 over LWWKV with 2 replicas.  As written in the reference the `let` binds replicas' twice and the second
    binding starts again from the un-merged vector: the merge is computed and dropped, the replicas never exchange state.  Restated
    as written: three rand-int draws per request, the third picks the replica that serves it."""

    def __init__(self, svc, n=2):
        self.replicas = [svc] * n

    def handle(self, src, body, rand_int):
        n = len(self.replicas)
        rand_int(n); rand_int(n)
        i = rand_int(n)
        self.replicas[i], res = self.replicas[i].handle(body)
        return res


SERVICES = ("lin-kv", "seq-kv", "lww-kv", "lin-tso")   # service.clj:290-296 default-services


def default_services():
    return {"lww-kv": Eventual(LWWKV()), "seq-kv": Sequential(PersistentKV()), "lin-kv": Linearizable(PersistentKV()),
            "lin-tso": Linearizable(PersistentTSO())}


# ---- node processes (process.clj) ---------------------------------------------------------------------------------------------
class NodeProcess:
    def __init__(self, argv, node_id, log_dir=None):
        err = open(os.path.join(log_dir, f"{node_id}.log"), "wb") if log_dir else subprocess.DEVNULL   # process.clj:195-199
        self.p = subprocess.Popen(argv, stdin=subprocess.PIPE, stdout=subprocess.PIPE, stderr=err, bufsize=0)
        self.node_id, self.buf, self.fd = node_id, b"", self.p.stdout.fileno()
        os.set_blocking(self.fd, False)

    def write(self, msg):
        try:
            self.p.stdin.write((json.dumps(msg) + "\n").encode())   # process.clj:160-164
        except (BrokenPipeError, ValueError):
            pass   # the node crashed: it stays silent, like a dead process under the reference

    def lines(self):
        """complete lines printed since the last call"""
        out = []
        try:
            while True:
                chunk = os.read(self.fd, 1 << 16)
                if not chunk:
                    break
                self.buf += chunk
        except BlockingIOError:
            pass
        while b"\n" in self.buf:
            line, self.buf = self.buf.split(b"\n", 1)
            if line.strip():
                out.append(line)
        return out

    def idle(self):
        """(quiet?, activity counter) from /proc: quiet = the node has read all of its stdin and every thread of it (and of its
        children) sleeps; the counter changes whenever one of them ran.  None where /proc does not tell (then only the quiet
        period decides)."""
        try:
            pending = struct.unpack("i", fcntl.ioctl(self.p.stdin.fileno(), termios.FIONREAD, b"\0\0\0\0"))[0]
            quiet, ticks, todo = pending == 0, 0, [self.p.pid]
            while todo:
                pid = todo.pop()
                for tid in os.listdir(f"/proc/{pid}/task"):
                    with open(f"/proc/{pid}/task/{tid}/stat", "rb") as f:
                        st = f.read().rsplit(b")", 1)[1].split()   # fields after "(comm)": state ...
                    if st[0] not in (b"S", b"I", b"Z", b"X"):
                        quiet = False
                    with open(f"/proc/{pid}/task/{tid}/schedstat", "rb") as f:
                        ticks += int(f.read().split()[2])             # timeslices run on a CPU
                    try:
                        with open(f"/proc/{pid}/task/{tid}/children", "rb") as f:
                            todo.extend(int(c) for c in f.read().split())
                    except OSError:
                        pass
            return quiet, ticks
        except (OSError, ValueError, IndexError):
            return None

    def stop(self):   # process.clj:202-215 stop-node!
        try:
            self.p.stdin.close()
        except Exception:
            pass
        try:
            self.p.wait(timeout=1.0)
        except subprocess.TimeoutExpired:
            self.p.kill()
            self.p.wait()


PH_INIT, PH_INIT_WAIT, PH_TOPO, PH_TOPO_WAIT, PH_MAIN_START, PH_MAIN, PH_DRAIN, PH_NEM_FINAL, PH_SLEEP, PH_FINAL, PH_FINAL_WAIT, PH_DONE = range(12)
TOPOLOGIES = {"grid": 0, "line": 1, "total": 2, "tree": 3, "tree2": 3, "tree3": 4, "tree4": 5}
WORKLOADS = ("echo", "broadcast", "g-set", "pn-counter", "g-counter", "unique-ids", "lin-kv", "txn-list-append")
REUSABLE = ("lin-kv", "unique-ids", "txn-list-append")   # lin_kv.clj:74-76, unique_ids.clj:59-61, txn_list_append.clj:94-99
HAS_FINAL = ("broadcast", "g-set", "pn-counter", "g-counter")
IDEMPOTENT = {"broadcast": ("read",), "lin-kv": ("read",)}   # the with-errors sets: broadcast.clj:200, lin_kv.clj:52


def topology(kind, n):
    """workload/broadcast.clj:40-185 -> adjacency lists in ascending order"""
    adj = [set() for _ in range(n)]

    def link(a, b):
        adj[a].add(b); adj[b].add(a)
    t = TOPOLOGIES[kind]
    if t == 0:
        side = 1
        while side * side < n:
            side += 1
        for i in range(side):
            for j in range(side):
                a = i * side + j
                if a >= n:
                    continue
                if j + 1 < side and a + 1 < n:
                    link(a, a + 1)
                if a + side < n:
                    link(a, a + side)
    elif t == 1:
        for i in range(n - 1):
            link(i, i + 1)
    elif t == 2:
        for i in range(n):
            for j in range(i + 1, n):
                link(i, j)
    else:
        b = {3: 2, 4: 3, 5: 4}[t]
        for i in range(1, n):
            link(i, (i - 1) // b)
    return [sorted(s) for s in adj]


class Bridge:
    """One test instance on real node processes.  Options follow core.clj:136-229 (and maelstrom_amd.engine.test_config)."""

    def __init__(self, workload, bin, node_count=5, concurrency=None, rate=5.0, time_limit=60.0, latency=0, latency_dist="constant",
                 topology="grid", nemesis=(), nemesis_interval=10.0, p_loss=0.0, seed=0, instance=0, client_timeout_ms=5000, quiesce_ms=10000,
                 settle_ms=3.0, clock="virtual", log_dir=None, journal=False, key_count=10, max_txn_length=4, max_writes_per_key=16):
        if workload not in WORKLOADS:
            raise ValueError(f"workload {workload!r} is not bridged (one of {WORKLOADS})")
        if latency_dist == "exponential" and latency == 0:
            raise ValueError("exponential latency needs a non-zero mean (net.clj:77 divides by zero)")
        self.wl, self.N = workload, node_count
        self.C = concurrency if concurrency is not None else (2 * node_count if workload == "lin-kv" else node_count)
        if workload == "lin-kv" and self.C % (2 * node_count):
            raise ValueError("lin-kv: concurrency must be a multiple of 2 x node-count")
        self.CS = max(self.C, self.N)
        self.rate_mhz, self.time_limit_us = int(round(rate * 1000)), int(round(time_limit * 1000)) * 1000
        self.lat_ms, self.lat_dist, self.p_loss_q32 = int(latency), latency_dist, min(int(p_loss * 2 ** 32), 2 ** 32 - 1)
        self.topo_kind, self.nemesis = topology, "partition" in set(nemesis)
        self.nem_interval_us, self.quiesce_us = int(round(nemesis_interval * 1000)) * 1000, quiesce_ms * 1000
        self.timeout_ms = client_timeout_ms
        if workload == "lin-kv":
            self.timeout_ms = max(10 * self.lat_ms, 1000)   # lin_kv.clj:54
        # txn-list-append ([upstream] elle list-append generator as DESIGN.md §2.4 restates it): a pool of key-count active keys,
        # exponentially more traffic on the later ones, a key retires after max-writes-per-key appends (core.clj:167-169,191-199)
        self.key_count, self.max_txn_length, self.max_writes = key_count, max_txn_length, max_writes_per_key
        self.t_active, self.t_next_val, self.t_next_key = list(range(16)), [1] * 16, key_count
        self.rng = Rng(seed, instance)
        self.settle_s, self.clock = settle_ms / 1000.0, clock
        self.names = [f"n{i}" for i in range(self.N)] + [f"c{k}" for k in range(self.CS)] + list(SERVICES)
        self.ep = {nm: i for i, nm in enumerate(self.names)}
        self.E = len(self.names)
        self.services = default_services()
        self.svc_ctr = {s: 0 for s in SERVICES}
        argv = bin if isinstance(bin, (list, tuple)) else [bin]
        self.procs = [NodeProcess(list(argv), self.names[i], log_dir) for i in range(self.N)]
        # net (net.clj:79-103)
        self.inbox = [[] for _ in range(self.E)]
        self.committed = [None] * self.E
        self.deliver_at = [0] * self.E
        self.part = [set() for _ in range(self.N)]    # part[dest] = srcs whose packets dest drops (net.clj:109-110)
        self.next_id, self.loss_on = 0, False
        self.stats = {k: 0 for k in ("all_send", "all_recv", "clients_send", "clients_recv", "servers_send", "servers_recv")}
        self.journal = [] if journal else None
        self.out = []
        # clients
        self.cl = [dict(busy=False, kind=None, mark=False, want=0, timeout_at=0, next_msg_id=0, f=None, value=None, process=k, final=False,
                        m_f=None, m_value=None, m_final=False) for k in range(self.CS)]
        # scheduler
        self.T, self.phase, self.cutoff, self.gen_next, self.gen_k, self.next_value = 0, PH_INIT, 0, 0, 0, 0
        self.nem_next, self.nem_j, self.sleep_until, self.rounds = 0, 0, 0, 0
        self.cur_key, self.key_procs, self.key_reg = 0, 0, [0] * self.CS
        self.history, self.errors = [], []
        self.adj = topology_lists = globals()["topology"](topology, self.N)
        self.t0 = time.monotonic()

    # ---- net ----
    def is_client(self, e):
        return self.N <= e < self.N + self.CS

    def send(self, src, dest, body):
        self.out.append((src, dest, body))

    def commit_sends(self):   # net.clj:189-221, canonical order = staging order
        for src, dest, body in self.out:
            mid = self.next_id
            self.next_id += 1
            cl = self.is_client(src) or self.is_client(dest)
            self.stats["all_send"] += 1
            self.stats["clients_send" if cl else "servers_send"] += 1
            if self.journal is not None:
                self.journal.append({"id": len(self.journal), "time": self.T * 1000, "type": ":send",
                                     "message": {"id": mid, "src": self.names[src], "dest": self.names[dest], "body": body}})
            if cl:
                lat = 0
            elif self.lat_dist == "constant":
                lat = self.lat_ms
            elif self.lat_dist == "uniform":
                lat = scale32(self.rng.draw32(S_LATENCY, mid), 2 * self.lat_ms)
            else:
                lat = (self.lat_ms * neg_ln_q16(self.rng.draw32(S_LATENCY, mid))) >> 16
            if self.loss_on and self.p_loss_q32 and self.rng.draw32(S_LOSS, mid) < self.p_loss_q32:
                continue   # net.clj:214: lost after the journal saw it
            self.inbox[dest].append((self.T + lat * 1000, mid, src, body))
        self.out = []

    def poll(self, e):   # net.clj:223-247
        if self.is_client(e) and not self.cl[e - self.N]["busy"]:
            return
        box = self.inbox[e]
        while self.committed[e] is None and box:
            k = min(range(len(box)), key=lambda i: (box[i][0], box[i][1]))
            q = box[k]
            box[k] = box[-1]
            box.pop()
            if e < self.N and q[2] < self.N and q[2] in self.part[e]:
                continue   # partitioned: dropped at poll time, no :recv (net.clj:234)
            self.committed[e] = q
            self.deliver_at[e] = self.T if q[0] <= self.T else self.T + ((q[0] - self.T) // 1000) * 1000

    def recv_log(self, e, q):
        cl = self.is_client(q[2]) or self.is_client(e)
        self.stats["all_recv"] += 1
        self.stats["clients_recv" if cl else "servers_recv"] += 1
        if self.journal is not None:
            self.journal.append({"id": len(self.journal), "time": self.T * 1000, "type": ":recv",
                                 "message": {"id": q[1], "src": self.names[q[2]], "dest": self.names[e], "body": q[3]}})

    # ---- history ----
    def row(self, type_, f, process, value, error=None, final=False):
        op = {"index": len(self.history), "time": self.T * 1000, "type": type_, "f": f, "process": process, "value": value}
        if error is not None:
            op["error"] = error
        if final:
            op["final?"] = True
        self.history.append(op)

    # ---- clients (client.clj) ----
    def request_body(self, c):
        f, v, wl = c["f"], c["value"], self.wl
        if f == ":echo":
            return {"type": "echo", "echo": v}
        if f == ":broadcast":
            return {"type": "broadcast", "message": v}
        if f == ":add":
            return {"type": "add", "element": v} if wl == "g-set" else {"type": "add", "delta": v}
        if f == ":generate":
            return {"type": "generate"}
        if f == ":write":
            return {"type": "write", "key": v[0], "value": v[1]}
        if f == ":cas":
            return {"type": "cas", "key": v[0], "from": v[1][0], "to": v[1][1]}
        if f == ":txn":   # txn_list_append.clj:101-113: micro-ops [f k v] with f "r" / "append"
            return {"type": "txn", "txn": [[m[0][1:], m[1], m[2]] for m in v]}
        return {"type": "read", "key": v[0]} if wl == "lin-kv" else {"type": "read"}

    def client_invoke(self, slot):
        c = self.cl[slot]
        c["mark"], c["busy"] = False, True
        ep = self.N + slot
        if c["kind"] == "init":   # db.clj:46-69
            dest, c["next_msg_id"] = slot, 0
            body = {"type": "init", "node_id": self.names[slot], "node_ids": self.names[: self.N]}
        elif c["kind"] == "topo":   # broadcast.clj:195-197
            dest, c["next_msg_id"] = slot, 0
            body = {"type": "topology", "topology": {self.names[a]: [self.names[b] for b in nb] for a, nb in enumerate(self.adj)}}
        else:
            c["f"], c["value"], c["final"] = c["m_f"], c["m_value"], c["m_final"]
            dest = c["process"] % self.N
            self.row(":invoke", c["f"], c["process"], c["value"], final=c["final"])
            body = self.request_body(c)
        c["next_msg_id"] += 1
        c["want"] = c["next_msg_id"]
        body["msg_id"] = c["want"]
        c["timeout_at"] = self.T + (self.timeout_ms if c["kind"] == "op" else 10000) * 1000
        self.send(ep, dest, body)

    def client_complete(self, slot, type_, value, error=None):
        c = self.cl[slot]
        c["busy"] = False
        if c["kind"] != "op":
            if type_ != ":ok":
                self.errors.append(f"{c['kind']} of {self.names[slot]} failed: {error}")
            return
        self.pend[slot] = (type_, c["f"], c["process"], value, error, c["final"])
        if type_ == ":info":   # crashed process: new process id, fresh client unless Reusable [upstream interpreter]
            c["process"] += self.C
            if self.wl not in REUSABLE:
                c["next_msg_id"] = 0
                self.inbox[self.N + slot] = []

    def client_deliver(self, slot, body):
        c = self.cl[slot]
        if not c["busy"] or body.get("in_reply_to") != c["want"]:
            return   # stale reply (client.clj:105-107)
        t = body.get("type")
        if t == "error":   # client.clj:125-138,163-172
            name, definite = ERRORS.get(body.get("code"), (f"unknown-{body.get('code')}", False))
            idem = c["f"][1:] in IDEMPOTENT.get(self.wl, ())
            self.client_complete(slot, ":fail" if (definite or idem) else ":info", c["value"], [":" + name, body.get("text")])
            return
        f = c["f"]
        if c["kind"] != "op":
            self.client_complete(slot, ":ok", None)
        elif f == ":echo":
            self.client_complete(slot, ":ok", {k: v for k, v in body.items() if k not in ("msg_id", "in_reply_to")})
        elif f == ":read" and self.wl == "broadcast":
            self.client_complete(slot, ":ok", sorted(body.get("messages", [])))
        elif f == ":read" and self.wl == "g-set":
            self.client_complete(slot, ":ok", sorted(body.get("value", [])))
        elif f == ":read" and self.wl == "lin-kv":
            self.client_complete(slot, ":ok", [c["value"][0], body.get("value")])
        elif f == ":read":
            self.client_complete(slot, ":ok", body.get("value"))
        elif f == ":generate":
            self.client_complete(slot, ":ok", body.get("id"))
        elif f == ":txn":   # the completed transaction, reads filled in (txn_list_append.clj:40-52,114-119)
            self.client_complete(slot, ":ok", [[":" + str(m[0]), m[1], m[2]] for m in body.get("txn", [])])
        else:
            self.client_complete(slot, ":ok", c["value"])

    def client_timeout(self, slot):   # client.clj:96-103 + :158-162
        c = self.cl[slot]
        idem = c["f"] is not None and c["f"][1:] in IDEMPOTENT.get(self.wl, ())
        v = None if (c["f"] == ":read" and self.wl != "lin-kv") else c["value"]
        self.client_complete(slot, ":fail" if idem else ":info", v, ":net-timeout")

    # ---- scheduler (core.clj:67-80 through the [upstream] generator interpreter, as DESIGN.md §2.2 fixes it) ----
    def any_busy(self, n):
        return any(c["busy"] for c in self.cl[:n])

    def gen_live(self):
        return self.rate_mhz > 0 and self.gen_next < self.cutoff

    def nem_live(self):
        return self.nemesis and self.nem_next < self.cutoff

    def resolve(self):
        while True:
            ph = self.phase
            if ph == PH_INIT_WAIT and not self.any_busy(self.CS):
                self.phase = PH_TOPO if self.wl == "broadcast" else PH_MAIN_START
            elif ph == PH_TOPO_WAIT and not self.any_busy(self.CS):
                self.phase = PH_MAIN_START
            elif ph == PH_MAIN_START:
                self.cutoff = self.T + self.time_limit_us
                self.gen_next = self.nem_next = self.T
                for c in self.cl:
                    c["next_msg_id"] = 0
                self.loss_on, self.phase = True, PH_MAIN
            elif ph == PH_MAIN and not (self.gen_live() or self.nem_live()) and not (self.rate_mhz == 0 and self.T < self.cutoff):
                self.phase = PH_DRAIN
            elif ph == PH_DRAIN and not self.any_busy(self.C):
                hf = self.wl in HAS_FINAL
                self.phase = PH_NEM_FINAL if (self.nemesis and hf) else PH_SLEEP if hf else PH_DONE
                if self.phase == PH_SLEEP:
                    self.sleep_until = self.T + self.quiesce_us
            elif ph == PH_FINAL_WAIT and not self.any_busy(self.C):
                self.phase = PH_DONE
            else:
                return

    def sched_due(self):
        ph, T = self.phase, self.T
        if ph in (PH_INIT, PH_TOPO, PH_NEM_FINAL, PH_FINAL):
            return T
        if ph == PH_SLEEP:
            return self.sleep_until
        if ph == PH_MAIN:
            d = INF
            if self.nem_live():
                d = max(self.nem_next, T)
            if self.gen_live() and any(not c["busy"] for c in self.cl[: self.C]):
                d = min(d, max(self.gen_next, T))
            if self.rate_mhz == 0 and not self.nem_live():
                d = min(d, self.cutoff)
            return d
        return INF

    def start_partition(self, j, spec):   # [upstream] jepsen.nemesis.combined partition-package, as DESIGN.md §2.4 restates it
        n, rng = self.N, self.rng

        def shuffle():
            perm = list(range(n))
            for i in range(n - 1, 0, -1):
                k = scale32(rng.draw32(S_NEM_SHUFFLE, (j << 16) | i), i + 1)
                perm[i], perm[k] = perm[k], perm[i]
            return perm

        def complete(comp):
            for d in range(n):
                for x in range(n):
                    if comp[d] != comp[x]:
                        self.part[d].add(x)
        comp = [0] * n
        if spec == 0:
            comp[scale32(rng.draw32(S_NEM_PICK, j), n)] = 1
            complete(comp)
        elif spec in (1, 3):
            perm = shuffle()
            for i in range(n // 2 if spec == 1 else (n - 1) // 3):
                comp[perm[i]] = 1
            complete(comp)
        else:
            perm = shuffle()
            m = n // 2 + 1
            for i in range(n):
                c = perm[(i + m // 2) % n]
                vis = {perm[(i + k) % n] for k in range(m)}
                self.part[c] |= {x for x in range(n) if x not in vis}

    def nemesis_rows(self, f, v1, v2):
        self.row(":info", f, ":nemesis", v1)
        self.row(":info", f, ":nemesis", v2)

    def sched_act(self):
        T, ph = self.T, self.phase
        if ph == PH_INIT:
            for c in self.cl[: self.N]:
                c["mark"], c["kind"] = True, "init"
            self.phase = PH_INIT_WAIT
        elif ph == PH_TOPO:
            for c in self.cl[: self.N]:
                c["mark"], c["kind"] = True, "topo"
            self.phase = PH_TOPO_WAIT
        elif ph == PH_MAIN:
            if self.nem_live() and self.nem_next <= T:
                j = self.nem_j
                self.nem_j += 1
                if j % 2 == 0:
                    spec = scale32(self.rng.draw32(S_NEM_SPEC, j), 4)
                    self.start_partition(j, spec)
                    grudge = {self.names[d]: [self.names[s] for s in sorted(self.part[d])] for d in range(self.N) if self.part[d]}
                    self.nemesis_rows(":start-partition", (":one", ":majority", ":majorities-ring", ":minority-third")[spec], [":isolated", grudge])
                else:
                    self.part = [set() for _ in range(self.N)]   # heal! net.clj:112-113
                    self.nemesis_rows(":stop-partition", None, ":network-healed")
                self.nem_next = T + ((self.rng.draw32(S_NEM_STAGGER, j) * (2 * self.nem_interval_us)) >> 32)
            if self.gen_live() and self.gen_next <= T:
                free = [k for k in range(self.C) if not self.cl[k]["busy"]]
                if free:
                    k = self.gen_k
                    self.gen_k += 1
                    h = self.rng.draw64(S_GEN, k)
                    r_hi, r_lo = h >> 32, h & 0xFFFFFFFF
                    slot = free[scale32(r_lo, len(free))]
                    c = self.cl[slot]
                    c["mark"], c["kind"], c["m_final"] = True, "op", False
                    wl = self.wl
                    if wl == "lin-kv":   # [upstream] jepsen.tests.linearizable-register as DESIGN.md §2.4 restates it
                        if self.key_reg[slot] != 1 + c["process"]:
                            if self.key_procs == 20:
                                self.cur_key += 1
                                self.key_procs = 0
                                self.key_reg = [0] * self.CS
                            self.key_reg[slot] = 1 + c["process"]
                            self.key_procs += 1
                        h2 = self.rng.draw64(S_GEN2, k)
                        v1, v2, key = scale32(h2 >> 32, 5), (((h2 >> 20) & 0xFFF) * 5) >> 12, self.cur_key & 0xFF
                        if slot < self.N:
                            c["m_f"], c["m_value"] = ":read", [key, None]
                        elif scale32(h2 & 0xFFFFFFFF, 3) == 0:
                            c["m_f"], c["m_value"] = ":write", [key, v1]
                        else:
                            c["m_f"], c["m_value"] = ":cas", [key, [v1, v2]]
                    elif wl == "txn-list-append":
                        n_mops = 1 + scale32(self.rng.draw64(S_GEN2, k) >> 32, self.max_txn_length)
                        txn = []
                        for j in range(n_mops):
                            h3 = self.rng.draw64(S_GEN3, k * 8 + j)
                            ki = (scale32(h3 >> 32, (1 << self.key_count) - 1) + 1).bit_length() - 1   # P(ki) = 2^ki / (2^kc - 1)
                            key = self.t_active[ki]
                            if h3 & 1:
                                v = self.t_next_val[ki]
                                self.t_next_val[ki] += 1
                                txn.append([":append", key, v])
                                if self.t_next_val[ki] > self.max_writes:   # key used up: a fresh one takes its place in the pool
                                    self.t_active[ki], self.t_next_val[ki] = self.t_next_key, 1
                                    self.t_next_key += 1
                            else:
                                txn.append([":r", key, None])
                        c["m_f"], c["m_value"] = ":txn", txn
                    elif wl == "unique-ids":
                        c["m_f"], c["m_value"] = ":generate", None
                    elif wl == "echo":
                        c["m_f"], c["m_value"] = ":echo", f"Please echo {(r_lo >> 4) & 127}"   # echo.clj:72-75
                    elif wl == "g-counter":   # g_counter.clj:37-41: gen/filter skips negative adds
                        rr, a = r_lo, 0
                        d = ((((rr >> 4) & 0xFFFF) * 10) >> 16) - 5
                        while not (rr & 1) and d < 0 and a < 15:
                            a += 1
                            rr = self.rng.draw64(S_GEN2, k * 16 + a) & 0xFFFFFFFF
                            d = ((((rr >> 4) & 0xFFFF) * 10) >> 16) - 5
                        c["m_f"], c["m_value"] = (":read", None) if (rr & 1 or d < 0) else (":add", d)
                    elif r_lo & 1:
                        c["m_f"], c["m_value"] = ":read", None
                    elif wl == "pn-counter":
                        c["m_f"], c["m_value"] = ":add", ((((r_lo >> 4) & 0xFFFF) * 10) >> 16) - 5   # pn_counter.clj:134-135
                    else:
                        c["m_f"], c["m_value"] = (":broadcast" if wl == "broadcast" else ":add"), self.next_value
                        self.next_value += 1
                    self.gen_next = T + ((r_hi * (2 * (1000000000 // self.rate_mhz))) >> 32)   # gen/stagger (/ rate), core.clj:68
        elif ph == PH_NEM_FINAL:
            self.part = [set() for _ in range(self.N)]
            self.nemesis_rows(":stop-partition", None, ":network-healed")
            self.phase, self.sleep_until = PH_SLEEP, T + self.quiesce_us
        elif ph in (PH_SLEEP, PH_FINAL):
            if ph == PH_SLEEP and T < self.sleep_until:
                return
            for c in self.cl[: self.C]:   # (gen/each-thread {:f :read, :final? true}), broadcast.clj:240, g_set.clj:61
                c["mark"], c["kind"], c["m_f"], c["m_value"] = True, "op", ":read", None
                c["m_final"] = self.wl in ("broadcast", "pn-counter", "g-counter")
            self.phase = PH_FINAL_WAIT

    # ---- node execution (process.clj:136-166) ----
    def collect(self, wrote, must_answer=()):
        """Reads what the nodes print until every one of them has been quiet for settle_s; returns {node: [lines...]} in print order.
        `must_answer`: nodes that owe a reply before quiet counts (the init handshake: db.clj:46-69 waits up to 10 s of real time for
        init_ok — an interpreter needs tens of milliseconds to start)."""
        got = {}
        fds = {p.fd: i for i, p in enumerate(self.procs)}
        owed = set(must_answer)
        hard = time.monotonic() + 10.0
        deadline = time.monotonic() + (self.settle_s if wrote else 0.0)
        probe = self.clock == "virtual" and wrote      # /proc says when the nodes are done, however loaded the machine is
        last = None
        while True:
            now = time.monotonic()
            left = (hard - now) if owed else (deadline - now)
            if probe and not owed:
                left = min(max(left, 0.0), 0.0003) if last is None else 0.0003
            r, _, _ = select.select(list(fds), [], [], max(left, 0.0))
            if not r:
                if probe and not owed and now < hard:
                    cur = [self.procs[i].idle() for i in fds.values()]
                    if any(c is None for c in cur):
                        probe = False          # no /proc: the quiet period alone decides
                        continue
                    if all(c[0] for c in cur) and cur == last:
                        return got             # twice the same: everybody asleep, nobody ran in between, nothing left to read
                    last = cur
                    continue
                return got
            last = None
            for fd in r:
                i = fds[fd]
                lines = self.procs[i].lines()
                if lines:
                    got.setdefault(i, []).extend(lines)
                    owed.discard(i)
                    deadline = time.monotonic() + self.settle_s
                elif self.procs[i].p.poll() is not None:
                    fds.pop(fd, None)   # EOF: the process is gone
                    owed.discard(i)
            if not fds:
                return got

    def ingest(self, node, lines):
        """process.clj:35-66 parse-msg + net.clj:166-176 validation; a bad line is logged and dropped (the reference throws in the
        node's stdout thread)."""
        for line in lines:
            try:
                m = json.loads(line)
                body, dest = m["body"], m["dest"]
                if not isinstance(body, dict) or dest not in self.ep:
                    raise ValueError(f"unknown dest {dest!r}" if isinstance(body, dict) else "body is not a map")
            except Exception as ex:
                self.errors.append(f"{self.names[node]} printed a malformed message ({ex}): {line[:200]!r}")
                continue
            self.send(node, self.ep[dest], body)

    def service_step(self, e):
        q = self.committed[e]
        self.committed[e] = None
        self.recv_log(e, q)
        name = self.names[e]

        def rand_int(n):
            v = scale32(self.rng.draw32(S_SVC, self.svc_ctr[name]), n)
            self.svc_ctr[name] += 1
            return v
        try:
            res = self.services[name].handle(self.names[q[2]], q[3], rand_int)
        except Exception as ex:   # service.clj:259-260: logged, no reply
            self.errors.append(f"error in service worker {name}: {ex}")
            res = None
        if res is not None:
            self.send(e, q[2], {**res, "in_reply_to": q[3].get("msg_id")})

    # ---- the round loop (oracle: run_instance) ----
    def run(self, round_limit=50_000_000):
        N, E = self.N, self.E
        try:
            while True:
                self.resolve()
                if self.phase == PH_DONE:
                    break
                self.rounds += 1
                if self.rounds > round_limit:
                    self.errors.append("round limit")
                    break
                if self.clock == "real":   # spontaneous output of nodes with timers: sent at the instant it is seen
                    for i, lines in self.collect(False).items():
                        self.ingest(i, lines)
                tn = self.sched_due()
                for e in range(E):
                    if self.committed[e] is not None and self.deliver_at[e] < tn:
                        tn = self.deliver_at[e]
                tt = min([c["timeout_at"] for c in self.cl if c["busy"]], default=INF)
                if tn == INF and tt == INF and not self.out:
                    self.errors.append("stuck: nothing will ever happen")
                    break
                timeout_round = tt < tn and not self.out
                T = max(self.T, tt if timeout_round else tn)
                if self.clock == "real":
                    now = int((time.monotonic() - self.t0) * 1e6)
                    if self.out:
                        T = max(self.T, now)   # something was printed: a round now
                        timeout_round = False
                    elif T > now:
                        time.sleep(min((T - now) / 1e6, 0.001))
                        self.rounds -= 1
                        continue
                self.T = T
                self.pend = {}
                if timeout_round:
                    for k, c in enumerate(self.cl):
                        if c["busy"] and c["timeout_at"] <= T:
                            self.client_timeout(k)
                    self.flush_completions()
                    continue
                if self.sched_due() <= T:
                    self.sched_act()
                for k, c in enumerate(self.cl):
                    if c["mark"]:
                        self.client_invoke(k)
                self.commit_sends()
                for e in range(E):
                    self.poll(e)
                # R3: one input per node, then the services (endpoint order)
                wrote, owing = False, []
                for n in range(N):
                    q = self.committed[n]
                    if q is not None and self.deliver_at[n] <= T:
                        self.committed[n] = None
                        self.recv_log(n, q)
                        self.procs[n].write({"src": self.names[q[2]], "dest": self.names[n], "body": q[3]})
                        wrote = True
                        if q[3].get("type") == "init":
                            owing.append(n)
                got = self.collect(wrote, owing) if (wrote or self.clock == "real") else {}
                for n in sorted(got):
                    self.ingest(n, got[n])
                for e in range(N + self.CS, E):
                    if self.committed[e] is not None and self.deliver_at[e] <= T:
                        self.service_step(e)
                self.commit_sends()
                for e in range(E):
                    self.poll(e)
                # R4: the clients' recv! loops, envelope k of every client before envelope k+1 of any
                any_ = True
                while any_:
                    any_ = False
                    for k in range(self.CS):
                        e = N + k
                        q = self.committed[e]
                        if q is not None and self.deliver_at[e] <= T:
                            self.committed[e] = None
                            self.recv_log(e, q)
                            self.client_deliver(k, q[3])
                            self.poll(e)
                            any_ = True
                self.flush_completions()
        finally:
            for p in self.procs:
                p.stop()
        return self.history

    def flush_completions(self):
        for k in sorted(self.pend):
            type_, f, process, value, error, final = self.pend[k]
            self.row(type_, f, process, value, error, final)
        self.pend = {}

    def net_stats(self):
        """the :net :stats map of net/checker.clj:28-41,55-67"""
        ops = sum(1 for op in self.history if op["type"] == ":invoke" and op["process"] != ":nemesis")
        s = self.stats
        m = {k: {"send-count": s[f"{k}_send"], "recv-count": s[f"{k}_recv"], "msg-count": s[f"{k}_send"]} for k in ("all", "clients", "servers")}
        if ops:
            m["all"]["msgs-per-op"] = m["all"]["msg-count"] / ops
            m["servers"]["msgs-per-op"] = m["servers"]["msg-count"] / ops
        return m


def _edn(v):
    if v is None:
        return "nil"
    if isinstance(v, bool):
        return "true" if v else "false"
    if isinstance(v, str):
        return v if v.startswith(":") else json.dumps(v)
    if isinstance(v, dict):
        return "{" + ", ".join(f"{_edn(':' + k if not k.startswith(':') and k.isidentifier() else k)} {_edn(x)}" for k, x in v.items()) + "}"
    if isinstance(v, (list, tuple)):
        return "[" + " ".join(_edn(x) for x in v) + "]"
    return str(v)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m maelstrom_amd.bridge", description="Maelstrom test of an arbitrary node binary on the deterministic scheduler (CPU)")
    ap.add_argument("cmd", choices=["test"])
    ap.add_argument("-w", "--workload", required=True, choices=WORKLOADS)
    ap.add_argument("--bin", required=True, help="node program; its own arguments follow a bare --")
    ap.add_argument("--node-count", type=int, default=5)
    ap.add_argument("--concurrency", type=int)
    ap.add_argument("--rate", type=float, default=5.0)
    ap.add_argument("--time-limit", type=float, default=60.0)
    ap.add_argument("--latency", type=int, default=0)
    ap.add_argument("--latency-dist", default="constant", choices=["constant", "uniform", "exponential"])
    ap.add_argument("--topology", default="grid", choices=sorted(TOPOLOGIES))
    ap.add_argument("--nemesis", action="append", default=[], choices=["partition"])
    ap.add_argument("--nemesis-interval", type=float, default=10.0)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--key-count", type=int, default=10)              # core.clj:167-169
    ap.add_argument("--max-txn-length", type=int, default=4)          # core.clj:191-194
    ap.add_argument("--max-writes-per-key", type=int, default=16)     # core.clj:196-199
    ap.add_argument("--clock", default="virtual", choices=["virtual", "real"])
    ap.add_argument("--settle-ms", type=float, default=3.0)
    ap.add_argument("--log-dir")
    ap.add_argument("--history", help="write history.edn here")
    argv = list(sys.argv[1:] if argv is None else argv)
    bin_args = []
    if "--" in argv:   # everything after -- goes to the node program
        k = argv.index("--")
        argv, bin_args = argv[:k], argv[k + 1:]
    a = ap.parse_args(argv)
    a.bin_args = bin_args
    if a.log_dir:
        os.makedirs(a.log_dir, exist_ok=True)
    b = Bridge(a.workload, [a.bin] + a.bin_args, node_count=a.node_count, concurrency=a.concurrency, rate=a.rate, time_limit=a.time_limit,
               latency=a.latency, latency_dist=a.latency_dist, topology=a.topology, nemesis=a.nemesis, nemesis_interval=a.nemesis_interval,
               seed=a.seed, clock=a.clock, settle_ms=a.settle_ms, log_dir=a.log_dir, key_count=a.key_count, max_txn_length=a.max_txn_length,
               max_writes_per_key=a.max_writes_per_key)
    t0 = time.time()
    hist = b.run()
    if a.history:
        with open(a.history, "w") as f:
            for op in hist:
                f.write("{" + ", ".join(f":{k} {_edn(v)}" for k, v in op.items()) + "}\n")
    oks = sum(1 for op in hist if op["type"] == ":ok")
    inv = sum(1 for op in hist if op["type"] == ":invoke")
    print(json.dumps({"ops": inv, "ok": oks, "virtual_seconds": b.T / 1e6, "wall_seconds": round(time.time() - t0, 2), "rounds": b.rounds,
                      "net": b.net_stats(), "errors": b.errors[:10]}))
    return 0 if not b.errors else 1


if __name__ == "__main__":
    sys.exit(main())
