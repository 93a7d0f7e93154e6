"""Pins the oracle's node state-transition functions against golden vectors
recorded from the reference's own node programs (tests/golden/make_golden.py -> node_transitions.json):
demo/python/echo.py, demo/js/gossip.js (ack + 1 s retry), demo/js/crdt_gset.js.

Compared per input step: the multiset of emitted (dest, type, payload, in_reply_to).  msg_id numbering
differs between the reference's own demo runtimes (js starts at 0, ruby at 1) and is not compared;
RPC-ness (msg_id present or not) is."""
import ctypes as C
import json
import os

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "node_transitions.json")))["cases"]


_SLOTS = {}


def _ep(name, n_nodes):
    """endpoint name -> oracle endpoint index (nodes first, then one client slot per distinct client name)."""
    if name[0] == "n":
        return int(name[1:])
    slots = _SLOTS.setdefault(n_nodes, {})
    if name not in slots:
        slots[name] = len(slots)
        assert slots[name] < n_nodes, "more client names than client slots in this fixture"
    return n_nodes + slots[name]


def _trace(cfg, node, inputs):
    lib = O.load()
    lib.oracle_node_trace.argtypes = [C.POINTER(A.Config), C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p]
    lib.oracle_node_trace.restype = C.c_int
    lib.oracle_msg_type.argtypes = [C.c_char_p]
    lib.oracle_msg_type.restype = C.c_uint32
    inp = np.array(inputs, dtype=np.uint32).reshape(-1, 4)
    out = np.zeros((256, 5), dtype=np.uint32)
    pay = np.zeros(cfg.max_payload_words, dtype=np.uint32)
    fin = np.zeros(cfg.max_values // 32, dtype=np.uint32)
    n = lib.oracle_node_trace(C.byref(cfg), node, inp.ctypes.data, len(inp), out.ctypes.data, 256, pay.ctypes.data, fin.ctypes.data)
    assert n >= 0
    return out[:n], pay, fin


def _t(name):
    lib = O.load()
    lib.oracle_msg_type.argtypes = [C.c_char_p]
    lib.oracle_msg_type.restype = C.c_uint32
    v = lib.oracle_msg_type(name.encode())
    assert v, name
    return v


def _norm_ref(out):
    b = out["body"]
    payload = b.get("echo", b.get("message", None))
    if b["type"] == "read_ok":
        payload = tuple(sorted(b.get("messages", b.get("value"))))
    if b["type"] == "replicate":
        payload = tuple(sorted(b["value"]))
    if isinstance(payload, str):
        payload = int(payload.rsplit(" ", 1)[1])
    return (out["dest"], b["type"], payload, b.get("in_reply_to"), "msg_id" in b)


def test_echo_py(lib):
    case = GOLD["echo.py"]
    n = len(case["node_ids"])
    cfg = E.test_config("echo", node_count=n, rate=5, time_limit=5)
    inputs, names = [], {}
    for st in case["steps"]:
        m = st["in"]; b = m["body"]
        a = int(b["echo"].rsplit(" ", 1)[1]) if b["type"] == "echo" else 0
        inputs.append([_ep(m["src"], n), _t(b["type"]), a, b["msg_id"]]); names[_ep(m["src"], n)] = m["src"]
    out, _, _ = _trace(cfg, 1, inputs)
    for i, st in enumerate(case["steps"]):
        got = sorted((names[int(o[1])], {_t("init_ok"): "init_ok", _t("echo_ok"): "echo_ok"}[int(o[2])],
                      int(o[3]) if int(o[2]) == _t("echo_ok") else None, int(o[4]), False) for o in out if o[0] == i)
        assert got == sorted(_norm_ref(o) for o in st["out"]), (i, got, st["out"])


def test_gossip_js_ack_retry(lib):
    case = GOLD["gossip.js"]
    n = len(case["node_ids"])
    # gossip.js uses the topology it is sent; n1's neighbours are n0, n2, n4 in the golden script = tree2 rooted at n0? no:
    # use a config whose adjacency for n1 equals the golden topology: tree3 on 5 nodes gives n0-{n1,n2,n3}, n1-{n0,n4}; so
    # replay against node n1 of the golden by checking targets through the oracle's own topology for a node with the
    # same neighbour set: tree2 (n1: parent n0, children n3, n4).  Neighbour NAMES are mapped positionally.
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=n, topology="tree2", rate=5, time_limit=5)
    adj = np.zeros((n, 4), dtype=np.uint32)
    assert O.load().oracle_topology(A.TOPO_TREE2, n, adj.ctypes.data) == 0
    mine = E.bitmap_to_list(adj[1])
    assert len(mine) == len(case["neighbors"]) == 3
    ren = dict(zip(case["neighbors"], [f"n{x}" for x in mine]))  # golden neighbour name -> oracle neighbour name
    ren.update({"n1": "n1"})

    def node_name(x):
        return ren.get(x, x)
    inputs, names = [], {}
    for st in case["steps"]:
        if "in" not in st:
            inputs.append([0, 0, st["wait_ms"] * 1000 + 60000, 0]); continue
        m = st["in"]; b = m["body"]; src = node_name(m["src"]) if m["src"][0] == "n" else m["src"]
        a = b.get("message", 0)
        if b["type"] == "broadcast_ok":
            a = 7  # the engine carries the acked value with the ack (the reference finds it via its callback table)
        inputs.append([_ep(src, n), _t(b["type"]), a, b.get("msg_id", b.get("in_reply_to", 0)) if b["type"] != "broadcast_ok" else 1])
        names[_ep(src, n)] = m["src"]
    out, pay, fin = _trace(cfg, 1, inputs)
    inv = {v: k for k, v in ren.items()}
    tn = {_t(k): k for k in ("init_ok", "topology_ok", "broadcast", "broadcast_ok", "read_ok")}
    for i, st in enumerate(case["steps"]):
        got = []
        for o in (o for o in out if o[0] == i):
            typ = tn[int(o[2])]
            dest = int(o[1])
            dname = inv.get(f"n{dest}", f"n{dest}") if dest < n else names[dest]
            if typ == "broadcast":
                got.append((dname, typ, int(o[3]), None, True))
            elif typ == "read_ok":
                words = int(o[3]) >> 24; off = int(o[3]) & 0xFFFFFF
                got.append((dname, typ, tuple(E.bitmap_to_list(pay[off:off + words])), int(o[4]), False))
            else:
                got.append((dname, typ, None, int(o[4]), False))
        assert sorted(got, key=repr) == sorted((_norm_ref(o) for o in st["out"]), key=repr), (i, got, st["out"])


def test_crdt_gset_js(lib):
    case = GOLD["crdt_gset.js"]
    n = len(case["node_ids"])
    cfg = E.test_config("g-set", node_count=n, rate=5, time_limit=5)
    # the engine's replicate message carries a snapshot reference; replay the peer's replicate as adds of its elements
    inputs, names, step_of = [], {}, []
    for si, st in enumerate(case["steps"]):
        if "in" not in st:
            inputs.append([0, 0, st["wait_ms"] * 1000, 0]); step_of.append(si); continue
        m = st["in"]; b = m["body"]
        if b["type"] == "replicate":
            for x in b["value"]:
                inputs.append([_ep("c63", n), _t("add"), x, 0]); step_of.append(-1)  # set-union of the payload
            continue
        inputs.append([_ep(m["src"], n), _t(b["type"]), b.get("element", 0), b["msg_id"]]); step_of.append(si)
        names[_ep(m["src"], n)] = m["src"]
    out, pay, fin = _trace(cfg, 1, inputs)
    tn = {_t(k): k for k in ("init_ok", "add_ok", "read_ok", "replicate")}
    ticks = 0
    for si, st in enumerate(case["steps"]):
        got = []
        for o in out:
            if step_of[int(o[0])] != si:
                continue
            typ, dest = tn[int(o[2])], int(o[1])
            if typ == "replicate":
                got.append((f"n{dest}", typ, tuple(E.bitmap_to_list(fin)), None, False))
            elif typ == "read_ok":
                words = int(o[3]) >> 24; off = int(o[3]) & 0xFFFFFF
                got.append((names[dest], typ, tuple(E.bitmap_to_list(pay[off:off + words])), int(o[4]), False))
            else:
                got.append((names[dest], typ, None, int(o[4]), False))
        ref = sorted((_norm_ref(o) for o in st["out"]), key=repr)
        # (node.rb's `every` also fires once at start-up, node.rb:129-138, where crdt_gset.js's setInterval
        #  does not; the trace hook only runs timers inside the wait step, so the 5 s tick is what is compared)
        assert sorted(got, key=repr) == ref, (si, got, st["out"])
