"""Pins the CPU oracle against every known-answer the reference's docs hold for this path:
exact message counts printed by `maelstrom.net.checker` in the tutorial transcripts, and checker verdicts."""
import math

import numpy as np
import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O
import setfull_ref as R


def _ops(rows, f=None):
    typ = rows["packed"] & 3
    ff = (rows["packed"] >> 2) & 31
    m = typ == A.T_INVOKE
    if f is not None:
        m &= ff == f
    return int(m.sum())


def test_kat1_echo_message_count(lib):
    """doc/02-echo/index.md:367-397: total msgs = 2*ops + 2*N (12 ops, 1 node => 26)."""
    for n in (1, 3):
        cfg = E.test_config("echo", node_count=n, rate=5, time_limit=10, seed=3)
        r = O.run(cfg, 0, 4)
        for i in range(4):
            rows, _ = r.history(i)
            assert r.meta[i]["flags"] == 0
            assert int(r.stats[i]["all_send"]) == 2 * _ops(rows) + 2 * n
            assert int(r.stats[i]["all_recv"]) == int(r.stats[i]["all_send"])
            assert int(r.stats[i]["servers_send"]) == 0


# doc/03-broadcast/02-performance.md: server msgs per broadcast = 2E - N + 1 (dedup + skip-sender)
@pytest.mark.parametrize("topology,n,per_bcast,cite", [
    ("grid", 25, 56, ":87-92 (56 280 = 56 x 1005)"),
    ("line", 25, 24, ":110-115 (24 120)"),
    ("total", 25, 576, ":234-237 (587 520 = 576 x 1020)"),
    ("tree4", 25, 24, ":249-254 (24 744)"),
    ("grid", 5, 6, ":71-76 (5 916 = 6 x 986)"),
])
def test_kat2_broadcast_server_msgs_per_broadcast(lib, topology, n, per_bcast, cite):
    cfg = E.test_config("broadcast", node_count=n, rate=100, time_limit=20, topology=topology, seed=11)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        rows, _ = r.history(i)
        assert r.meta[i]["flags"] == 0
        nb = _ops(rows, A.F_BROADCAST)
        assert 800 < nb < 1200
        assert int(r.stats[i]["servers_send"]) == per_bcast * nb, cite
        assert int(r.stats[i]["servers_recv"]) == per_bcast * nb


def test_kat3_broadcast_without_skip_sender(lib):
    """02-performance.md:22-28,43: 5-node grid, echo-to-sender variant: 10 server msgs per broadcast (9 980 = 10 x 998)."""
    cfg = E.test_config("broadcast", bin="broadcast-ff-echoback", node_count=5, rate=100, time_limit=20, seed=5)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        rows, _ = r.history(i)
        assert int(r.stats[i]["servers_send"]) == 10 * _ops(rows, A.F_BROADCAST)


def test_kat4_g_set_replication_rounds(lib):
    """doc/04-crdts/01-g-set.md:200-216: --time-limit 10, 5 nodes: servers 80 = 4 rounds x 5 x 4."""
    cfg = E.test_config("g-set", node_count=5, rate=5, time_limit=10, seed=2)
    r = O.run(cfg, 0, 8)
    for i in range(8):
        assert r.meta[i]["flags"] == 0
        assert int(r.stats[i]["servers_send"]) == 80


def test_kat5_client_message_count(lib):
    """01-broadcast.md:552-558: clients 130 = 2*(55 ops + 5 init + 5 topology); 01-g-set.md:200-206: 104 = 2*(47 + 5)."""
    cfg = E.test_config("broadcast", node_count=5, rate=10, time_limit=5, seed=9)
    r = O.run(cfg, 0, 8)
    for i in range(8):
        rows, _ = r.history(i)
        assert int(r.stats[i]["clients_send"]) == 2 * (_ops(rows) + 5 + 5)
        assert int(r.stats[i]["clients_recv"]) == int(r.stats[i]["clients_send"])
    cfg = E.test_config("g-set", node_count=5, rate=5, time_limit=10, seed=9)
    r = O.run(cfg, 0, 8)
    for i in range(8):
        rows, _ = r.history(i)
        assert int(r.stats[i]["clients_send"]) == 2 * (_ops(rows) + 5)


def test_kat7_set_full_result_shape_and_verdict(lib):
    """01-broadcast.md:564-577: healthy 5-node broadcast => :valid? true, nothing lost/stale at latency 0."""
    cfg = E.test_config("broadcast", node_count=5, rate=10, time_limit=5, seed=1)
    r = O.run(cfg, 0, 4)
    for i in range(4):
        rows, pay = r.history(i)
        res = R.set_full(E.decode_history(rows, pay, 5))
        assert set(res) >= {"valid?", "attempt-count", "stable-count", "lost-count", "lost", "never-read-count", "never-read",
                            "stale-count", "stale", "stable-latencies", "duplicated-count", "duplicated"}
        assert res["valid?"] is True and res["lost-count"] == 0 and res["stale-count"] == 0
        assert res["stable-count"] == res["attempt-count"] == _ops(rows, A.F_BROADCAST)
        assert res["stable-latencies"] == {0: 0, 0.5: 0, 0.95: 0, 0.99: 0, 1: 0}


def test_kat8_latency_scaling(lib):
    """02-performance.md:185-194: grid n=25 at 100 ms constant latency: worst stable latency ~ 8 hops x 100 ms;
    :140-158: line at 10 ms: <= 24 x 10 ms."""
    cfg = E.test_config("broadcast", node_count=25, rate=20, time_limit=10, latency=100, seed=4)
    r = O.run(cfg, 0, 2)
    for i in range(2):
        rows, pay = r.history(i)
        res = R.set_full(E.decode_history(rows, pay, 25))
        assert res["valid?"] is True
        assert 400 <= res["stable-latencies"][1] <= 800
    cfg = E.test_config("broadcast", node_count=25, rate=20, time_limit=10, latency=10, topology="line", seed=4)
    rows, pay = O.run(cfg, 0, 1).history(0)
    res = R.set_full(E.decode_history(rows, pay, 25))
    assert 100 <= res["stable-latencies"][1] <= 240


def test_partition_with_retry_loses_nothing(lib):
    """02-performance.md:519-541: tree4 + partitions + ack/retry => valid, send-count > recv-count (drops)."""
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=10, time_limit=20, topology="tree4",
                        nemesis=["partition"], nemesis_interval=5, seed=13)
    r = O.run(cfg, 0, 8)
    dropped = 0
    for i in range(8):
        rows, pay = r.history(i)
        assert r.meta[i]["flags"] == 0
        res = R.set_full(E.decode_history(rows, pay, 5))
        assert res["valid?"] is True and res["lost-count"] == 0
        dropped += int(r.stats[i]["servers_send"]) - int(r.stats[i]["servers_recv"])
    assert dropped > 0


def test_history_shape_pairs_and_times(lib):
    """Every invoke has exactly one completion by the same process; :time is non-decreasing; final reads are flagged."""
    cfg = E.test_config("broadcast", node_count=5, rate=20, time_limit=5, latency=20, seed=6)
    rows, pay = O.run(cfg, 0, 1).history(0)
    h = E.decode_history(rows, pay, 5)
    open_ops = {}
    last_t = 0
    for op in h:
        assert op["time"] >= last_t
        last_t = op["time"]
        if op["type"] == ":invoke":
            assert op["process"] not in open_ops
            open_ops[op["process"]] = op
        else:
            inv = open_ops.pop(op["process"])
            assert inv["f"] == op["f"]
    assert not open_ops
    finals = [op for op in h if op.get("final?")]
    assert len(finals) == 2 * 5 and all(op["f"] == ":read" for op in finals)
    assert E.history_edn(h).count("\n") == len(h)


def test_exponential_sampler_and_rng_are_pinned(lib):
    """Golden values of the integer-only samplers (any change here changes every history)."""
    lib_o = O.load()
    assert [lib_o.oracle_neg_ln_q16(r) for r in (0, 1, 0x7FFFFFFF, 0x80000000, 0xFFFFFFFE, 0xFFFFFFFF)] == \
        [1453634, 1408208, 45426, 45426, 0, 0]
    # -ln(u) in Q16 tracks the real function to < 2e-5 relative
    for r in (12345, 0x12345678, 0xC0000000, 0xFFFF0000):
        want = -math.log((r + 1) / 2**32)
        assert abs(lib_o.oracle_neg_ln_q16(r) / 65536 - want) < 3e-5 * max(1.0, want)
    assert lib_o.oracle_draw32(42, 0, 1, 0) == lib_o.oracle_draw32(42, 0, 1, 0)
    assert len({lib_o.oracle_draw32(42, i, 4, 7) for i in range(64)}) == 64


def test_topologies_match_reference_builders(lib):
    """broadcast.clj:40-185: grid 5x5 interior degree 4, corners 2; E: grid 40, line 24, total 300, trees 24."""
    lib_o = O.load()
    for topo, edges in ((A.TOPO_GRID, 40), (A.TOPO_LINE, 24), (A.TOPO_TOTAL, 300), (A.TOPO_TREE2, 24), (A.TOPO_TREE3, 24), (A.TOPO_TREE4, 24)):
        adj = np.zeros((25, 4), dtype=np.uint32)
        assert lib_o.oracle_topology(topo, 25, adj.ctypes.data) == 0
        deg = [len(E.bitmap_to_list(adj[i])) for i in range(25)]
        assert sum(deg) == 2 * edges
        for i in range(25):
            for j in E.bitmap_to_list(adj[i]):
                assert i in E.bitmap_to_list(adj[j]) and i != j
    adj = np.zeros((25, 4), dtype=np.uint32)
    lib_o.oracle_topology(A.TOPO_GRID, 25, adj.ctypes.data)
    assert E.bitmap_to_list(adj[0]) == [1, 5] and E.bitmap_to_list(adj[12]) == [7, 11, 13, 17]
    adj = np.zeros((5, 4), dtype=np.uint32)
    lib_o.oracle_topology(A.TOPO_GRID, 5, adj.ctypes.data)  # side 3: n0 n1 n2 / n3 n4 (doc: 5 edges... E=5)
    assert sum(len(E.bitmap_to_list(adj[i])) for i in range(5)) == 2 * 5
    lib_o.oracle_topology(A.TOPO_TREE4, 25, (adj25 := np.zeros((25, 4), dtype=np.uint32)).ctypes.data)
    assert E.bitmap_to_list(adj25[0]) == [1, 2, 3, 4] and E.bitmap_to_list(adj25[1]) == [0, 5, 6, 7, 8]


def test_journal_reproduces_net_checker_stats(lib):
    """maelstrom.net.checker folds the journal into {:all :clients :servers} x {send,recv,msg}-count
    (net/checker.clj:28-41): the journal the oracle emits must fold to exactly the engine's counters, every
    :recv must follow its :send, and a lost or partition-dropped message has a :send but no :recv
    (02-performance.md:519-529: 1277 sent / 734 received)."""
    cfg = E.test_config("broadcast", bin="broadcast-ack-retry", node_count=5, rate=10, time_limit=20, topology="tree4",
                        nemesis=["partition"], nemesis_interval=5, latency=10, seed=13, journal_capacity=100000)
    r = O.run(cfg, 0, 4)
    for i in range(4):
        ev = r.events(i)
        assert r.meta[i]["flags"] == 0 and len(ev) == r.meta[i]["n_events"]
        st = E.journal_stats(ev, 5)
        for k in ("all", "clients", "servers"):
            assert st[k]["send-count"] == int(r.stats[i][f"{k}_send"]) and st[k]["recv-count"] == int(r.stats[i][f"{k}_recv"])
            assert st[k]["msg-count"] == st[k]["send-count"]          # every id is journalled at send (net.clj:208)
        assert st["servers"]["send-count"] > st["servers"]["recv-count"]  # partitions drop at recv time (net.clj:234)
        seen_send = set()
        last_t = 0
        for e in E.decode_journal(ev, 5):
            assert e["time"] >= last_t; last_t = e["time"]
            mid = e["message"]["id"]
            if e["type"] == ":send":
                assert mid == len(seen_send)                          # ids count every send!, from 0 (net.clj:103,197)
                seen_send.add(mid)
            else:
                assert mid in seen_send
