"""The net journal in the reference's on-disk format (net/journal.clj:55-141): the library's Fressian writer
(msim_journal_fressian_rows, host code, no device) read back by an independently written reader, folded like
maelstrom.net.checker (net/checker.clj:28-41) and compared with the engine's own statistics; plus hand-assembled byte vectors for
the encodings the format description fixes."""
import numpy as np

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import fressian_reader as F
import oracle_lib as O


def _journal(cfg, inst=0):
    ora = O.run(cfg, inst, 1)
    assert ora.meta["flags"][0] == 0 and ora.meta["n_events"][0] <= cfg.journal_capacity
    return ora, ora.events(0), ora.history(0)[1]


def _endpoint_is_client(name):
    return name.startswith("c")


def test_round_trip_broadcast_journal():
    cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=5, latency=10, seed=3, journal_capacity=60000)
    ora, events, payload = _journal(cfg)
    data = E.journal_fressian(cfg, events, payload)
    evs = F.read_journal(data)
    assert len(evs) == len(events) > 1000
    assert [e["id"] for e in evs] == list(range(len(evs)))                       # dense ids (journal.clj:225-239)
    assert [e["time"] for e in evs] == [int(t) * 1000 for t in events["time_us"]]
    assert all(e["type"] in ("send", "recv") and isinstance(e["type"], F.Keyword) for e in evs)
    # net/checker.clj:28-41: send / recv / distinct-message counts for all, client-involving and server-only messages
    stats = {k: {"send-count": 0, "recv-count": 0, "ids": set()} for k in ("all", "clients", "servers")}
    for e in evs:
        m = e["message"]
        cl = _endpoint_is_client(m["src"]) or _endpoint_is_client(m["dest"])
        for k in ("all", "clients" if cl else "servers"):
            stats[k]["send-count" if e["type"] == "send" else "recv-count"] += 1
            stats[k]["ids"].add(m["id"])
    want = E.journal_stats(events, cfg.n_nodes)
    for k in stats:
        assert stats[k]["send-count"] == want[k]["send-count"] and stats[k]["recv-count"] == want[k]["recv-count"]
        assert len(stats[k]["ids"]) == want[k]["msg-count"]
    st = ora.stats[0]
    assert stats["all"]["send-count"] == int(st["all_send"]) and stats["servers"]["recv-count"] == int(st["servers_recv"])
    # bodies: wire fields of doc/protocol.md / doc/workloads.md, keyword keys
    kinds = {}
    for e in evs:
        b = e["message"]["body"]
        assert all(isinstance(k, F.Keyword) for k in b)
        kinds.setdefault(b["type"], b)
    assert set(kinds) >= {"init", "init_ok", "topology", "topology_ok", "broadcast", "broadcast_ok", "read", "read_ok"}
    assert kinds["init"]["node_ids"] == ["n0", "n1", "n2", "n3", "n4"] and kinds["init"]["node_id"].startswith("n") and kinds["init"]["msg_id"] == 1
    topo = kinds["topology"]["topology"]
    assert sorted(topo) == ["n0", "n1", "n2", "n3", "n4"] and topo["n0"] == ["n1", "n3"] and topo["n4"] == ["n1", "n3"]   # 5-node grid, side 3
    assert "in_reply_to" in kinds["broadcast_ok"] and "msg_id" not in kinds["init_ok"]
    # every read_ok lists exactly the elements of the history's :ok read with that reply
    decoded = E.decode_history(*ora.history(0), cfg.n_nodes, cfg.workload)
    reads = [op["value"] for op in decoded if op["type"] == ":ok" and op["f"] == ":read"]
    got = [e["message"]["body"]["messages"] for e in evs if e["type"] == "recv" and e["message"]["body"]["type"] == "read_ok"]
    assert got == reads
    # gossip carries the value; a client's broadcast carries a msg_id, server gossip does not (fire-and-forget)
    gossip = [e["message"] for e in evs if e["message"]["body"]["type"] == "broadcast" and e["type"] == "send"]
    assert any("msg_id" in m["body"] for m in gossip) and any("msg_id" not in m["body"] for m in gossip)
    assert all(("msg_id" in m["body"]) == m["src"].startswith("c") for m in gossip)


def test_round_trip_other_workloads():
    for wl, kw in (("echo", dict(node_count=3, rate=10, time_limit=5)), ("g-set", dict(node_count=4, rate=10, time_limit=6)),
                   ("pn-counter", dict(node_count=3, rate=10, time_limit=6)), ("unique-ids", dict(node_count=3, rate=50, time_limit=3)),
                   ("lin-kv", dict(bin="raft", node_count=3, concurrency=6, rate=10, time_limit=5))):
        cfg = E.test_config(wl, seed=5, journal_capacity=200000, **kw)
        ora, events, payload = _journal(cfg)
        evs = F.read_journal(E.journal_fressian(cfg, events, payload))
        assert len(evs) == len(events) > 20
        want = E.journal_stats(events, cfg.n_nodes)
        assert sum(e["type"] == "send" for e in evs) == want["all"]["send-count"]
        types = {e["message"]["body"]["type"] for e in evs}
        if wl == "echo":
            b = next(e["message"]["body"] for e in evs if e["message"]["body"]["type"] == "echo_ok")
            assert b["echo"].startswith("Please echo ") and "in_reply_to" in b
        if wl == "g-set":
            assert {"add", "add_ok", "replicate", "read_ok"} <= types
            assert all(isinstance(e["message"]["body"]["value"], list) for e in evs if e["message"]["body"]["type"] == "read_ok")
        if wl == "unique-ids":
            b = next(e["message"]["body"] for e in evs if e["message"]["body"]["type"] == "generate_ok")
            assert len(b["id"]) == 3 and b["id"][2].startswith("n")
        if wl == "lin-kv":
            assert {"request_vote", "append_entries", "append_entries_res"} <= types


def test_known_byte_encodings():
    """One event assembled by hand from the format description: struct definitions, caches, ints, strings, closed list."""
    cfg = E.test_config("broadcast", node_count=2, rate=1, time_limit=1, journal_capacity=16)
    ev = np.zeros(2, dtype=E.EVENT_DT)
    #   send of message id 5, type broadcast_ok (8), from n1 to c0 (endpoint 2), in_reply_to 7, at t = 3 us
    ev[0] = (3, (5 << 8) | 8, 0, 1 | (2 << 8) | (7 << 16))
    ev[1] = (20000, (5 << 8) | 0x80 | 8, 0, 1 | (2 << 8) | (7 << 16))   # its :recv at 20 ms
    data = E.journal_fressian(cfg, ev, np.zeros(0, np.uint32))
    want = bytes([0xEF, 0xDC]) + b"ev" + bytes([0x04,                   # STRUCTTYPE "ev" 4
                  0x00,                                                  # :id 0
                  0x5B, 0xB8,                                            # :time 3000 ns: 0x50 + (3000 >> 8), 3000 & 0xFF
                  0xCD, 0xCA, 0xF7, 0xCD, 0xDE]) + b"send" + bytes([     # cache :send (slot 0) = key(nil, cache "send" (slot 1))
                  0xEF, 0xDD]) + b"msg" + bytes([0x04,                   # STRUCTTYPE "msg" 4
                  0x05,                                                  # message id
                  0xCD, 0xDC]) + b"n1" + bytes([0xCD, 0xDC]) + b"c0" + bytes([   # src, dest cached (slots 2, 3)
                  0xC0, 0xED,                                            # map, closed list
                  0xCD, 0xCA, 0xF7, 0xCD, 0xDE]) + b"type" + bytes([     # key :type (slots 4, 5)
                  0xCD, 0xE3, 0x0C]) + b"broadcast_ok" + bytes([         # cached value (slot 6), 12 chars > 7: STRING + length
                  0xCD, 0xCA, 0xF7, 0xCD, 0xE3, 0x0B]) + b"in_reply_to" + bytes([0x07,   # key (slots 7, 8), value 7
                  0xFD])
    second = bytes([0xA0, 0x01,                                          # struct cache 0 = "ev", id 1
                    0x73, 0x31, 0x2D, 0x00,                              # 20_000_000 ns = 0x1312D00: 0x72 + (v >> 24), three low bytes
                    0xCD, 0xCA, 0xF7, 0xCD, 0xDE]) + b"recv" + bytes([   # :recv (slots 9, 10)
                    0xA1, 0x05, 0x82, 0x83,                              # "msg", id, cached n1, c0
                    0xC0, 0xED, 0x84, 0x86, 0x87, 0x07, 0xFD])           # :type broadcast_ok :in_reply_to 7 from the cache
    assert data == want + second, (data.hex(), (want + second).hex())
    evs = F.read_journal(data)
    assert evs[1]["time"] == 20_000_000 and evs[1]["message"]["body"] == {"type": "broadcast_ok", "in_reply_to": 7}


def test_wide_ints_and_long_caches_round_trip():
    cfg = E.test_config("g-set", node_count=40, rate=100, time_limit=12, latency=20, seed=9, journal_capacity=400000)   # > 32 cached strings
    ora, events, payload = _journal(cfg)
    evs = F.read_journal(E.journal_fressian(cfg, events, payload))
    assert len(evs) == len(events) and evs[-1]["time"] == int(events["time_us"][-1]) * 1000 > 2 ** 33   # 5-byte packed ints
    assert {e["message"]["src"] for e in evs} >= {f"n{i}" for i in range(40)}


def test_transaction_and_key_value_bodies_are_real_and_engine_abstractions_say_so():
    """txn / txn_ok carry the micro-ops of doc/workloads.md, lin-kv reads / writes / cas their key / value / from / to; what the
    engine never materialises (Raft's internal RPCs, the transactional nodes' storage traffic, replicate snapshots) is marked
    `:elided true`; the service endpoints carry the names of service.clj:290-296."""
    cfg = E.test_config("txn-list-append", bin="multi-key-txn", node_count=3, rate=40, time_limit=4, latency=2, seed=8, journal_capacity=200000)
    ora, events, payload = _journal(cfg)
    evs = F.read_journal(E.journal_fressian(cfg, events, payload))
    names = {e["message"]["src"] for e in evs} | {e["message"]["dest"] for e in evs}
    assert {"lin-kv", "lww-kv", "n0", "c0"} <= names
    decoded = [o for o in E.decode_history(*ora.history(0), cfg.n_nodes, cfg.workload)]
    want_ok = [[[f[1:], k, v] for f, k, v in o["value"]] for o in decoded if o["type"] == ":ok"]
    got_ok = [e["message"]["body"]["txn"] for e in evs if e["type"] == "recv" and e["message"]["body"]["type"] == "txn_ok"]
    assert got_ok == want_ok and len(got_ok) > 20
    want_inv = [[[f[1:], k, v] for f, k, v in o["value"]] for o in decoded if o["type"] == ":invoke"]
    got_inv = [e["message"]["body"]["txn"] for e in evs if e["type"] == "send" and e["message"]["body"]["type"] == "txn"]
    assert got_inv == want_inv
    storage = [e["message"]["body"] for e in evs if e["message"]["dest"] in ("lin-kv", "lww-kv")]
    assert storage and all(b.get("elided") is True for b in storage)

    cfg = E.test_config("lin-kv", bin="lin-kv-proxy", proxy_service="seq-kv", node_count=3, concurrency=6, rate=40, time_limit=4, seed=8, journal_capacity=200000)
    ora, events, payload = _journal(cfg)
    evs = F.read_journal(E.journal_fressian(cfg, events, payload))
    assert "seq-kv" in {e["message"]["dest"] for e in evs}
    bodies = [e["message"]["body"] for e in evs if e["type"] == "send" and e["message"]["src"].startswith("c")]
    ops = [o for o in E.decode_history(*ora.history(0), cfg.n_nodes, cfg.workload) if o["type"] == ":invoke"]
    assert len(bodies) == len(ops) + cfg.n_nodes
    for b, o in zip([b for b in bodies if b["type"] != "init"], ops):
        k, v = o["value"]
        assert b["key"] == k and b["type"] == o["f"][1:]
        if o["f"] == ":write":
            assert b["value"] == v
        if o["f"] == ":cas":
            assert [b["from"], b["to"]] == v

    cfg = E.test_config("lin-kv", bin="raft", node_count=3, concurrency=6, rate=10, time_limit=5, seed=5, journal_capacity=200000)
    ora, events, payload = _journal(cfg)
    evs = F.read_journal(E.journal_fressian(cfg, events, payload))
    assert all(e["message"]["body"].get("elided") is True for e in evs if e["message"]["body"]["type"] in ("request_vote", "append_entries", "append_entries_res"))
