"""The simulated network (src/maelstrom/net.clj:178-247) re-derived from the net journal: from every `send!` the journal
records, the counter-based RNG and the nemesis rows of the history, a transliteration of latency-for / send! / recv! decides
which envelope each receiver takes off its queue, when, and whether it survives — and must reproduce every :recv event of
every node and service, in order.  Covers: latency only between servers (util.clj:7-16), the three distributions, loss
decided at send but journalled first, the (deadline, id) priority queue, take-the-head-even-if-not-due with head-of-line
blocking, the ms-truncated sleep, partitions consulted at take time (a dropped head leaves no :recv).
Independent of the oracle's queues and round machinery: test infrastructure only."""
import collections

import pytest

from maelstrom_amd import _abi as A
from maelstrom_amd import engine as E
import oracle_lib as O

S_LATENCY, S_LOSS = 4, 5   # RNG streams


def _latency_ms(lib, cfg, inst, mid):
    r = lib.oracle_draw32(cfg.seed, inst, S_LATENCY, mid)
    if cfg.latency_dist == A.LAT_CONSTANT:
        return cfg.latency_mean_ms                                   # ConstantDistribution, net.clj:65-68
    if cfg.latency_dist == A.LAT_UNIFORM:
        return (r * 2 * cfg.latency_mean_ms) >> 32                   # integer-distribution 0 (2 mean), :70-73
    return (cfg.latency_mean_ms * lib.oracle_neg_ln_q16(r)) >> 16    # exponential, :75-77 (the engine's integer sampler)


CASES = [
    ("broadcast", dict(node_count=9, rate=60, time_limit=8, latency=40, latency_dist="exponential", p_loss=0.05)),
    ("broadcast", dict(node_count=5, rate=40, time_limit=10, latency=25, latency_dist="uniform", nemesis=["partition"], nemesis_interval=2)),
    ("broadcast", dict(bin="broadcast-ack-retry", node_count=5, rate=30, time_limit=10, latency=30, p_loss=0.1, nemesis=["partition"], nemesis_interval=2)),
    ("broadcast", dict(node_count=25, rate=100, time_limit=5, topology="total")),
    ("broadcast", dict(node_count=25, rate=100, time_limit=20)),                          # BASELINE configs[1], the headline shape
    ("broadcast", dict(node_count=25, rate=100, time_limit=10, latency=100, latency_dist="exponential")),   # its latency sweep
    ("txn-rw-register", dict(node_count=3, rate=60, time_limit=8, latency=15, latency_dist="exponential", p_loss=0.05, nemesis=["partition"], nemesis_interval=2)),
    ("txn-list-append", dict(node_count=3, rate=60, time_limit=8, latency=10, latency_dist="uniform", p_loss=0.02)),
    ("lin-kv", dict(bin="lin-kv-proxy", proxy_service="seq-kv", node_count=3, rate=60, time_limit=8, latency=20, latency_dist="exponential")),
]


@pytest.mark.parametrize("workload,kw", CASES)
def test_every_delivery_follows_from_sends_rng_and_partitions(workload, kw):
    lib = O.load()
    cfg = E.test_config(workload, seed=29, journal_capacity=600000, **kw)
    N = cfg.n_nodes
    CS = max(N, cfg.concurrency)
    is_client = lambda e: N <= e < N + CS
    delivered = dropped = lost = waited = 0
    for inst in range(2):
        r = O.run(cfg, inst, 1)
        assert r.meta["flags"][0] == 0 and r.meta["n_events"][0] <= cfg.journal_capacity
        ops = E.decode_history(*r.history(0), N, E.WORKLOADS[workload])
        # partitions[dest] = the sources whose packets dest drops, as a timeline (net.clj:105-113; grudge in the 2nd start row)
        part_at = [(0, {})]
        for op in ops:
            if op["process"] == ":nemesis" and op["f"] == ":start-partition" and isinstance(op["value"], list):
                part_at.append((op["time"] // 1000, {int(d[1:]): {int(s[1:]) for s in srcs} for d, srcs in op["value"][1].items()}))
            elif op["process"] == ":nemesis" and op["f"] == ":stop-partition" and op["value"] == ":network-healed":
                part_at.append((op["time"] // 1000, {}))

        def partitioned(t, dest, src):
            cur = {}
            for t0, g in part_at:
                if t0 <= t:
                    cur = g
            return src in cur.get(dest, ())
        # events with their poll slot: a round is [client sends][node recvs][node sends][client recvs]; receivers poll after
        # the client sends and after the node sends
        events, rnd, phase, last_t, last_ep, loss_on = [], 0, 0, -1, -1, False
        for ev in r.events(0):
            msg, route, t = int(ev["msg"]), int(ev["route"]), int(ev["time_us"])
            recv, src, dest = (msg >> 7) & 1, route & 0xFF, (route >> 8) & 0xFF
            kind = (3 if is_client(dest) else 1) if recv else (0 if is_client(src) else 2)
            ep = dest if recv else src
            # a new round: time moved, the phase went backwards, or — inside a phase — the endpoint order did (deliveries run
            # in node order, one per node; sends in sender order)
            if t != last_t or kind < phase or (kind == phase and (ep <= last_ep if kind == 1 else ep < last_ep)):
                rnd += 1
            phase, last_t, last_ep = kind, t, ep
            events.append((t, rnd * 2 + (0 if kind == 0 else 1), msg >> 8, recv, A.MSG_TYPES[msg & 0x7F], src, dest))
        queue = collections.defaultdict(list)          # dest -> [(deadline, id, src, arrival slot, arrival time)]
        free_slot = collections.defaultdict(lambda: (0, 0))   # dest -> (slot, time) of its last delivery
        slot_time = {}
        for t, slot, mid, recv, typ, src, dest in events:
            slot_time[slot] = t
            if typ not in ("init", "init_ok", "topology", "topology_ok"):
                loss_on = True                         # the main phase has begun: every later message may be lost (net.clj:214)
            if not recv:
                if is_client(dest):
                    continue
                lat = 0 if is_client(src) else _latency_ms(lib, cfg, inst, mid)
                if loss_on and cfg.p_loss_q32 and lib.oracle_draw32(cfg.seed, inst, S_LOSS, mid) < cfg.p_loss_q32:
                    lost += 1
                    continue                           # journalled, then lost
                queue[dest].append((t + lat * 1000, mid, src, slot, t))
                continue
            if is_client(dest):
                continue
            # a node / service delivery: which envelope would recv! have handed over?
            while True:
                q = queue[dest]
                assert q, f"{dest} received {mid} at {t} but its queue is empty"
                fs, ft = free_slot[dest]
                first = min(e[3] for e in q)
                take_slot = max(fs, first)
                take_t = slot_time[take_slot]
                head = min(e for e in q if e[3] <= take_slot)    # the priority queue's head at that moment: (deadline, id)
                q.remove(head)
                if src_blocked := partitioned(take_t, dest, head[2]):
                    dropped += 1                       # dropped at take time: no :recv, recv! polls again (net.clj:232-234)
                    free_slot[dest] = (take_slot, take_t)
                    continue
                due = take_t if head[0] <= take_t else take_t + ((head[0] - take_t) // 1000) * 1000   # (Thread/sleep (long dt))
                assert (head[1], due) == (mid, t), (dest, "expected", head, "due", due, "journal", mid, t, "taken at", take_t)
                waited += due > take_t
                delivered += 1
                free_slot[dest] = (slot, t)
                break
    assert delivered > 300
    if cfg.p_loss_q32:
        assert lost > 5
    if cfg.nemesis_mask:
        assert dropped > 5
    if cfg.latency_mean_ms:
        assert waited > 50
