"""The oracle's CRDT nodes against the reference's own JavaScript: real `node demo/js/crdt_gset.js` / `crdt_pn_counter.js`
processes (from /root/reference — this container only; the test skips elsewhere) are driven over their pipes with the oracle's
complete network schedule of a run (its journal), and everything they print — add_ok, read_ok with the set / the counter
value, the replicate messages with the whole state — must be what the oracle's node sent, in order.  The processes' only
timer (setInterval 5 s, crdt_gset.js:50-58) is taken over by a preload shim and fired when the oracle's node ticks, so the
run is in virtual time.  The engine follows demo/ruby/g_set.rb for WHEN the first tick happens (at start-up; the JS waits 5 s):
the shim fires the callback whenever the oracle ticks, the payloads are the processes' own.  Test infrastructure only."""
import json
import os
import select
import shutil
import signal
import subprocess

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

JS = "/root/reference/demo/js"
needs_reference = pytest.mark.skipif(shutil.which("node") is None or not os.path.exists(JS), reason="needs node.js and the reference tree")

SHIM = """
const cbs = [];
global.setInterval = (f, ms) => { cbs.push(f); return cbs.length; };   // virtual time: the harness says when
global.setTimeout = (f, ms) => 0;   // node.js' 1 s RPC timeout (node.js:85): never in virtual time — the Clojure node the engine follows has none
process.on('SIGUSR1', () => { for (const f of cbs) f(); });
"""


class Proc:
    def __init__(self, script, shim):
        self.p = subprocess.Popen(["node", "-r", shim, os.path.join(JS, script)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, bufsize=0)
        self.buf, self.ping = b"", 0

    def write(self, msg):
        self.p.stdin.write((json.dumps(msg) + "\n").encode())

    def readline(self):
        while b"\n" not in self.buf:
            r, _, _ = select.select([self.p.stdout], [], [], 10.0)
            assert r, "the node process printed nothing for 10 s"
            chunk = os.read(self.p.stdout.fileno(), 65536)
            assert chunk, "the node process exited"
            self.buf += chunk
        line, self.buf = self.buf.split(b"\n", 1)
        return json.loads(line)

    def barrier(self, me):
        """everything written so far has been handled once a read from a made-up client is answered"""
        self.ping += 1
        self.write({"src": "c999", "dest": me, "body": {"type": "read", "msg_id": 1000000 + self.ping}})
        m = self.readline()
        assert m["body"]["type"] == "read_ok" and m["body"]["in_reply_to"] == 1000000 + self.ping

    def tick(self):
        self.p.send_signal(signal.SIGUSR1)

    def close(self):
        self.p.kill()
        self.p.wait()


def _bitmap(words):
    return {w * 32 + b for w, x in enumerate(words) for b in range(32) if (int(x) >> b) & 1}


@needs_reference
@pytest.mark.parametrize("workload,script,kw", [
    ("g-set", "crdt_gset.js", dict(node_count=5, rate=30, time_limit=12, latency=20, latency_dist="exponential", p_loss=0.05)),
    ("g-set", "crdt_gset.js", dict(node_count=3, rate=40, time_limit=8)),
    ("pn-counter", "crdt_pn_counter.js", dict(node_count=5, rate=30, time_limit=12, latency=30, latency_dist="uniform", p_loss=0.1)),
    ("g-counter", "crdt_pn_counter.js", dict(node_count=3, rate=40, time_limit=8, latency=5)),
    # further shapes for the replay only (the recorded digests stay as they are)
    ("g-set", "crdt_gset.js", dict(node_count=7, rate=40, time_limit=14, latency=40, latency_dist="exponential", nemesis=["partition"], nemesis_interval=2)),
    ("pn-counter", "crdt_pn_counter.js", dict(node_count=4, concurrency=12, rate=60, time_limit=12, latency=10, nemesis=["partition"], nemesis_interval=3)),
    ("g-set", "crdt_gset.js", dict(node_count=2, rate=20, time_limit=22, latency=300)),
])
def test_reference_js_crdt_processes_print_what_the_oracle_sends(workload, script, kw, tmp_path):
    shim = tmp_path / "shim.js"
    shim.write_text(SHIM)
    cfg = E.test_config(workload, seed=61, journal_capacity=400000, **kw)
    N = cfg.n_nodes
    name = lambda e: f"n{e}" if e < N else f"c{e}"
    r = O.run(cfg, 0, 1)
    assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
    _, pay = r.history(0)
    procs = [Proc(script, str(shim)) for _ in range(N)]
    try:
        content, owed = {}, [0] * N       # owed[n] = replicate lines process n has printed and the journal has not reached yet
        n_rep = n_reads = 0
        for ev in r.events(0):
            msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
            mid, recv, typ = msg >> 8, (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
            src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
            if recv:
                if dest < N:
                    procs[dest].write(content[mid])
                continue
            if src >= N:                                  # a client's request
                body = {"type": typ, "msg_id": b}
                if typ == "init":
                    body.update(node_id=name(dest), node_ids=[name(i) for i in range(N)])
                elif typ == "add" and workload == "g-set":
                    body["element"] = a
                elif typ == "add":
                    body["delta"] = a - (1 << 32) if a & 0x80000000 else a
                content[mid] = {"src": name(src), "dest": name(dest), "body": body}
                continue
            if typ == "replicate" and owed[src] == 0:     # the oracle's node ticks: so does the process
                procs[src].barrier(name(src))
                procs[src].tick()
                owed[src] = N - 1
            m = procs[src].readline()
            mb = m["body"]
            assert (m["src"], m["dest"], mb["type"]) == (name(src), name(dest), typ), (m, typ, src, dest)
            if typ == "replicate":
                owed[src] -= 1
                n_rep += 1
            else:
                assert mb["in_reply_to"] == b
            if typ == "read_ok":
                n_reads += 1
                if workload == "g-set":                   # the payload bitmap of the reply the oracle's node sent
                    assert set(mb["value"]) == _bitmap(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])
                else:
                    assert mb["value"] == (a - (1 << 32) if a & 0x80000000 else a)
            content[mid] = m
        assert all(o == 0 for o in owed) and n_rep >= 2 * N * (N - 1) and n_reads > 10
    finally:
        for p in procs:
            p.close()


CASES = [("g-set", "crdt_gset.js", dict(node_count=5, rate=30, time_limit=12, latency=20, latency_dist="exponential", p_loss=0.05)),
         ("g-set", "crdt_gset.js", dict(node_count=3, rate=40, time_limit=8)),
         ("pn-counter", "crdt_pn_counter.js", dict(node_count=5, rate=30, time_limit=12, latency=30, latency_dist="uniform", p_loss=0.1)),
         ("g-counter", "crdt_pn_counter.js", dict(node_count=3, rate=40, time_limit=8, latency=5))]


TXN_CASES = [dict(), dict(node_count=3, rate=150, latency=2), dict(latency=15, latency_dist="exponential"),
             dict(key_count=2, max_txn_length=8, max_writes_per_key=40, rate=120)]
TXN_BASE = dict(node_count=5, rate=80, time_limit=8, latency=5, seed=63, journal_capacity=300000)


def run_digest(workload, kw):
    """sha256 over all :send events and the payload area of the run the processes reproduced"""
    import hashlib
    if workload == "txn-list-append":
        kw = dict(TXN_BASE, **kw)
    if workload == "broadcast":
        kw = dict(kw, bin="broadcast-ack-retry", seed=64, journal_capacity=600000)
    cfg = E.test_config(workload, **(kw if "seed" in kw else dict(kw, seed=61, journal_capacity=400000)))
    r = O.run(cfg, 0, 1)
    ev = r.events(0)
    sends = ev[((ev["msg"] >> 7) & 1) == 0]
    h = hashlib.sha256(np.stack([sends["time_us"], sends["msg"], sends["a"], sends["route"]], axis=1).astype(np.uint32).tobytes())
    h.update(r.history(0)[1].tobytes())
    return h.hexdigest()


_GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "js_crdt_replay_digests.json")


def test_runs_still_match_the_recorded_js_replays():
    """needs neither node.js nor the reference tree: the runs the real processes reproduced (tests/golden/make_golden_js_replay.py)
    are still the runs the oracle produces"""
    gold = json.load(open(_GOLD))
    assert len(gold) == len(CASES) + len(TXN_CASES) + len(GOSSIP_CASES)
    for j, kw in enumerate(GOSSIP_CASES):
        g = gold[str(len(CASES) + len(TXN_CASES) + j)]
        assert g["workload"] == "broadcast" and g["options"] == json.loads(json.dumps(kw))
        assert run_digest("broadcast", kw) == g["digest"], f"gossip case {j}: regenerate with tests/golden/make_golden_js_replay.py after checking the replay"
    for j, kw in enumerate(TXN_CASES):
        g = gold[str(len(CASES) + j)]
        assert g["workload"] == "txn-list-append" and g["options"] == json.loads(json.dumps(kw))
        assert run_digest("txn-list-append", kw) == g["digest"], f"txn case {j}: regenerate with tests/golden/make_golden_js_replay.py after checking the replay"
    for i, (workload, _script, kw) in enumerate(CASES):
        assert gold[str(i)]["workload"] == workload and gold[str(i)]["options"] == json.loads(json.dumps(kw))
        assert run_digest(workload, kw) == gold[str(i)]["digest"], f"case {i}: regenerate with tests/golden/make_golden_js_replay.py after checking the replay"


@needs_reference
def test_reference_echo_js_process_prints_what_the_oracle_sends(tmp_path):
    """demo/js/echo.js over a whole run of the echo workload (echo.clj:72-75 payloads), lossy network"""
    shim = tmp_path / "shim.js"
    shim.write_text(SHIM)
    cfg = E.test_config("echo", seed=62, journal_capacity=100000, node_count=2, rate=60, time_limit=8, p_loss=0.1)
    N = cfg.n_nodes
    name = lambda e: f"n{e}" if e < N else f"c{e}"
    r = O.run(cfg, 0, 1)
    assert r.meta["flags"][0] == 0
    procs = [Proc("echo.js", str(shim)) for _ in range(N)]
    try:
        content, n = {}, 0
        for ev in r.events(0):
            msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
            mid, recv, typ = msg >> 8, (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
            src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
            if recv:
                if dest < N:
                    procs[dest].write(content[mid])
            elif src >= N:
                body = {"type": typ, "msg_id": b}
                if typ == "init":
                    body.update(node_id=name(dest), node_ids=[name(i) for i in range(N)])
                else:
                    body["echo"] = f"Please echo {a}"
                content[mid] = {"src": name(src), "dest": name(dest), "body": body}
            else:
                m = procs[src].readline()
                assert (m["src"], m["dest"], m["body"]["type"], m["body"]["in_reply_to"]) == (name(src), name(dest), typ, b)
                if typ == "echo_ok":
                    assert m["body"]["echo"] == f"Please echo {a}"
                    n += 1
        assert n > 15   # every lost message parks its worker for the 5 s timeout
    finally:
        for p in procs:
            p.close()


@needs_reference
@pytest.mark.parametrize("kw", [dict(), dict(node_count=3, rate=150, latency=2), dict(latency=15, latency_dist="exponential"),
                                dict(key_count=2, max_txn_length=8, max_writes_per_key=40, rate=120)])   # = TXN_CASES
def test_reference_single_key_txn_js_processes_print_what_the_oracle_sends(kw, tmp_path):
    """txn-list-append: real `node demo/js/single_key_txn.js` processes (the runnable twin of single_key_txn.clj) against a
    transliteration of service.clj's lin-kv that keeps the REAL database value, driven with the oracle's schedule.  The two
    known differences between the reference's own twins are mapped, not hidden: JS numbers its RPCs from 0, Clojure (and the
    engine) from 1; for a missing root JS cas-es from [], Clojure from nil — both create it."""
    import collections
    import services_ref as R
    shim = tmp_path / "shim.js"
    shim.write_text(SHIM)
    base = dict(node_count=5, rate=80, time_limit=8, latency=5, seed=63, journal_capacity=300000)
    base.update(kw)
    cfg = E.test_config("txn-list-append", **base)
    N = cfg.n_nodes
    SVC = 2 * N
    name = lambda e: f"n{e}" if e < N else ("lin-kv" if e == SVC else f"c{e}")
    r = O.run(cfg, 0, 1)
    assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
    _, pay = r.history(0)
    procs = [Proc("single_key_txn.js", str(shim)) for _ in range(N)]
    svc, svc_out = R.Linearizable(), collections.deque()
    fname = {":r": "r", ":append": "append"}
    try:
        content, n_ok, n_conflict = {}, 0, 0
        for ev in r.events(0):
            msg, a, route = int(ev["msg"]), int(ev["a"]), int(ev["route"])
            mid, recv, typ = msg >> 8, (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
            src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
            if recv:
                if dest < N:
                    procs[dest].write(content[mid])
                elif dest == SVC:                          # the service handles the node's request with the real value
                    req = content[mid]
                    svc_out.append({"src": "lin-kv", "dest": req["src"], "body": dict(svc.handle(req["src"], req["body"], None), in_reply_to=req["body"]["msg_id"])})
                continue
            if N <= src < SVC:                            # a client's request
                body = {"type": typ, "msg_id": b}
                if typ == "init":
                    body.update(node_id=name(dest), node_ids=[name(i) for i in range(N)])
                else:
                    body["txn"] = [[fname[f], k, v] for f, k, v in E.decode_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])]
                content[mid] = {"src": name(src), "dest": name(dest), "body": body}
            elif src == SVC:
                m = svc_out.popleft()
                assert (m["dest"], m["body"]["type"]) == (name(dest), typ), (m, typ, dest)
                if typ == "error":
                    assert m["body"]["code"] == a
                content[mid] = m
            else:
                m = procs[src].readline()
                mb = m["body"]
                assert (m["src"], m["dest"], mb["type"]) == (name(src), name(dest), typ), (m, typ, src, dest)
                if dest == SVC:
                    assert mb["msg_id"] + 1 == b          # JS counts RPCs from 0, the Clojure node from 1
                else:
                    assert mb["in_reply_to"] == b
                    if typ == "txn_ok":
                        want = [[fname[f], k, v] for f, k, v in E.decode_txn(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])]
                        assert mb["txn"] == want, (mb["txn"], want)
                        n_ok += 1
                    elif typ == "error":
                        assert mb["code"] == a == 30
                        n_conflict += 1
                content[mid] = m
        assert not svc_out and n_ok > 100 and n_conflict > 0
    finally:
        for p in procs:
            p.close()


# ---- broadcast with acknowledgements and retries: real demo/js/gossip.js processes in virtual time ----
CLOCK_SHIM = """
const fs = require('fs');
let now = 0, seq = 0, timers = [];
global.setTimeout = (f, ms) => { timers.push({t: now + ms * 1000, s: seq++, f}); return seq; };   // node.js' 1 s RPC timeout, in virtual us
global.setInterval = (f, ms) => 0;
process.on('SIGUSR1', () => {          // the harness moved the clock: run what is due, then say so
  now = parseInt(fs.readFileSync(process.env.MSIM_CLOCK_FILE, 'utf8'));
  for (;;) {
    timers.sort((a, b) => a.t - b.t || a.s - b.s);
    if (!timers.length || timers[0].t > now) break;
    timers.shift().f();
  }
  setImmediate(() => console.log(JSON.stringify({__clock__: now})));   // after the promise callbacks (the retries) have printed
});
"""


class ClockedProc(Proc):
    def __init__(self, script, shim, clock_file):
        self.clock_file, self.now, self.lines = clock_file, 0, []
        env = dict(os.environ, MSIM_CLOCK_FILE=clock_file)
        self.p = subprocess.Popen(["node", "-r", shim, os.path.join(JS, script)], stdin=subprocess.PIPE, stdout=subprocess.PIPE,
                                  stderr=subprocess.DEVNULL, bufsize=0, env=env)
        self.buf, self.ping = b"", 0

    def advance(self, t):
        """virtual time reaches t: timers that are due fire (their output is kept in order), then the marker comes back"""
        if t == self.now:
            return
        with open(self.clock_file, "w") as f:
            f.write(str(t))
        self.now = t
        self.p.send_signal(signal.SIGUSR1)
        while True:
            m = Proc.readline(self)
            if "__clock__" in m:
                assert m["__clock__"] == t
                return
            self.lines.append(m)

    def next_line(self):
        return self.lines.pop(0) if self.lines else Proc.readline(self)


GOSSIP_CASES = [dict(node_count=5, rate=30, time_limit=10, latency=20, latency_dist="exponential", p_loss=0.1),
                dict(node_count=9, rate=40, time_limit=8, latency=5, nemesis=["partition"], nemesis_interval=2),
                dict(node_count=5, rate=20, time_limit=10, latency=50, topology="line", p_loss=0.2)]


@needs_reference
@pytest.mark.parametrize("kw", GOSSIP_CASES)
def test_reference_gossip_js_processes_print_what_the_oracle_sends(kw, tmp_path):
    """The acknowledged, retrying broadcast node (doc/03-broadcast/02-performance.md:406-441 ≡ demo/js/gossip.js) against real
    gossip.js processes whose RPC timeouts run in the oracle's virtual time.  Mapped differences: gossip.js gossips first and
    acknowledges last, the tutorial's final version (which the engine follows) acknowledges first — what a node emits for one
    input is compared as a set; JS numbers RPCs from 0, the engine from 1.  The shapes keep round trips far below the 1 s
    retry timeout: beyond that gossip.js ignores a late acknowledgement of a timed-out attempt, the tutorial's node accepts it."""
    import collections
    shim = tmp_path / "clock_shim.js"
    shim.write_text(CLOCK_SHIM)
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", seed=64, journal_capacity=600000, **kw)
    N = cfg.n_nodes
    name = lambda e: f"n{e}" if e < N else f"c{e}"
    adj = np.zeros((N, 4), dtype=np.uint32)
    assert O.load().oracle_topology(cfg.topology, N, adj.ctypes.data) == 0
    topo = {name(i): [name(j) for j in E.bitmap_to_list(adj[i])] for i in range(N)}
    r = O.run(cfg, 0, 1)
    assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
    _, pay = r.history(0)
    procs = [ClockedProc("gossip.js", str(shim), str(tmp_path / f"clock{i}")) for i in range(N)]
    try:
        content = {}
        want = [collections.defaultdict(list) for _ in range(N)]    # key -> ids of the messages the oracle's node emitted for its current input
        got = [collections.defaultdict(list) for _ in range(N)]     # key -> the lines the process printed for it
        n_retry = n_gossip = n_reads = 0

        def pair(n):
            """a printed line and an emitted message with the same key are the same message, whatever their order"""
            for key in list(want[n]):
                while want[n][key] and got[n].get(key):
                    content[want[n][key].pop(0)] = got[n][key].pop(0)
                if not want[n][key]:
                    del want[n][key]
            for key in [k for k, v in got[n].items() if not v]:
                del got[n][key]

        def settle(n):
            pair(n)
            assert not want[n] and not got[n], (n, dict(want[n]), dict(got[n]))
        for ev in r.events(0):
            t, msg, a, route = int(ev["time_us"]), int(ev["msg"]), int(ev["a"]), int(ev["route"])
            mid, recv, typ = msg >> 8, (msg >> 7) & 1, A.MSG_TYPES[msg & 0x7F]
            src, dest, b = route & 0xFF, (route >> 8) & 0xFF, route >> 16
            if recv:
                if dest < N:
                    settle(dest)
                    procs[dest].advance(t)
                    procs[dest].write(content[mid])
                continue
            if src >= N:                                  # a client's request
                body = {"type": typ, "msg_id": b}
                if typ == "init":
                    body.update(node_id=name(dest), node_ids=[name(i) for i in range(N)])
                elif typ == "topology":
                    body["topology"] = topo
                elif typ == "broadcast":
                    body["message"] = a
                content[mid] = {"src": name(src), "dest": name(dest), "body": body}
                continue
            if t != procs[src].now:                       # nothing was delivered to this node now: its retry timers are due
                settle(src)
                procs[src].advance(t)
                n_retry += 1
            m = procs[src].next_line()
            mb = m["body"]
            assert m["src"] == name(src)
            want[src][(name(dest), typ, a if typ == "broadcast" else None, b)].append(mid)
            got[src][(m["dest"], mb["type"], mb.get("message"), mb["msg_id"] + 1 if "msg_id" in mb else mb["in_reply_to"] + (1 if m["dest"].startswith("n") else 0))].append(m)   # RPC ids: JS from 0, engine from 1
            pair(src)
            if typ == "broadcast":
                n_gossip += 1
            if mb["type"] == "read_ok":                   # only one message per read: same line
                n_reads += 1
                assert typ == "read_ok" and set(mb["messages"]) == _bitmap(pay[a & 0xFFFFFF:(a & 0xFFFFFF) + (a >> 24)])
        for n in range(N):
            settle(n)
        assert n_gossip > 100 and n_reads > 10 and (n_retry > 5 or not (cfg.p_loss_q32 or cfg.nemesis_mask))
    finally:
        for p in procs:
            p.close()
