"""How many rounds would a WINDOWED round loop need?  (developer tool; reads an oracle run's net journal)

The round loop of DESIGN.md §2.2 handles one event time per round.  A window round handles, per node, the FIRST pending
delivery whose time lies before H = min(next scheduler / client-request round, T + (lookahead ms) * 1000 - 999, min over the
nodes of their SECOND event) — the condition under which nothing handled in the round can depend on anything else
handled in it, and every send of the round precedes every send of the next.  This script replays the deliveries of an
oracle run under that rule and prints oracle rounds vs window rounds.

    python tools/window_rounds_estimate.py [latency_ms] [dist] [K]
"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))

from maelstrom_amd import _abi as A   # noqa: E402
from maelstrom_amd import engine as E   # noqa: E402
import oracle_lib as O   # noqa: E402

S_LATENCY = 4


def latency_ms(lib, cfg, inst, mid):
    r = lib.oracle_draw32(cfg.seed, inst, S_LATENCY, mid)
    if cfg.latency_dist == A.LAT_CONSTANT:
        return cfg.latency_mean_ms
    if cfg.latency_dist == A.LAT_UNIFORM:
        return (r * 2 * cfg.latency_mean_ms) >> 32
    return (cfg.latency_mean_ms * lib.oracle_neg_ln_q16(r)) >> 16


def main():
    lat = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dist = sys.argv[2] if len(sys.argv) > 2 else "exponential"
    K = int(sys.argv[3]) if len(sys.argv) > 3 else 24
    lib = O.load()
    cfg = E.test_config("broadcast", node_count=25, rate=100, time_limit=20, latency=lat, latency_dist=dist, seed=7, journal_capacity=400000)
    N = cfg.n_nodes
    INST = int(os.environ.get("INST", "0")); r = O.run(cfg, INST, 1)
    assert r.meta["flags"][0] == 0
    ev = r.events(0)
    per_node = [[] for _ in range(N)]   # delivery times per node (server envelopes and client requests)
    general = set()                     # times of rounds that need the scheduler / client machinery
    sends_at = {}
    lat_at = {}                       # (t, node) -> server sends of that delivery
    n_sends = 0
    first_id_at = {}
    for e in ev:
        msg, route, t = int(e["msg"]), int(e["route"]), int(e["time_us"])
        recv, src, dest = (msg >> 7) & 1, route & 0xFF, (route >> 8) & 0xFF
        mid = msg >> 8
        if recv:
            if dest < N:
                per_node[dest].append(t)
                if src >= N:
                    general.add(t)
        else:
            if src >= N:
                general.add(t)
            elif dest < N:
                sends_at[(t, src)] = sends_at.get((t, src), 0) + 1
                lat_at.setdefault((t, src), []).append(latency_ms(lib, cfg, INST, mid))
                first_id_at.setdefault((t, src), mid)
            n_sends += 1
    general = sorted(general)
    import bisect
    ptr = [0] * N
    rounds = handled_tot = gen_rounds = 0
    hist = {}
    over_k = 0
    binds = {}
    next_id_guess = 0
    while True:
        t1 = [per_node[n][ptr[n]] if ptr[n] < len(per_node[n]) else None for n in range(N)]
        live = [t for t in t1 if t is not None]
        if not live:
            break
        Tm = min(live)
        gi = bisect.bisect_left(general, Tm)
        G = general[gi] if gi < len(general) else 1 << 62
        if G == Tm:
            H = Tm + 1
            gen_rounds += 1
        else:
            if cfg.latency_dist == A.LAT_CONSTANT:
                m = lat
            else:
                # the ids the round's sends will take start at the id of the first send at or after Tm
                cand = [first_id_at[k] for k in first_id_at if k[0] >= Tm]
                nid = min(cand) if cand else 0
                m = min(latency_ms(lib, cfg, INST, nid + j) for j in range(K))
            H0 = min(G, Tm + max(m * 1000 - 999, 1))
            B = 1 << 62
            for n in range(N):
                if t1[n] is None:
                    continue
                t2 = per_node[n][ptr[n] + 1] if ptr[n] + 1 < len(per_node[n]) else 1 << 62
                B = min(B, max(t2, t1[n] + 1))
            H = min(H0, B)
            bind = 'B' if B <= H0 else ('G' if G <= Tm + max(m * 1000 - 999, 1) else 'L')
            binds[bind] = binds.get(bind, 0) + 1
            if os.environ.get('NOB'): H = H0
            if os.environ.get('ACT'):
                Hc = min(G, B) if not os.environ.get('NOB') else G
                cmin = 1 << 62
                for n in range(N):
                    if t1[n] is not None and t1[n] < Hc:
                        for l in lat_at.get((t1[n], n), []):
                            cmin = min(cmin, t1[n] + max(l * 1000 - 999, 1))
                H = min(Hc, max(cmin, Tm + 1))
        h = 0
        s = 0
        for n in range(N):
            if t1[n] is not None and t1[n] < H:
                s += sends_at.get((t1[n], n), 0)
                ptr[n] += 1
                h += 1
        if s > K:
            over_k += 1
        rounds += 1
        handled_tot += h
        hist[h] = hist.get(h, 0) + 1
    print(f"latency {lat} ms {dist}: oracle rounds {int(r.meta['n_rounds'][0])}, deliveries {handled_tot}, window rounds {rounds} "
          f"(of them at scheduler / request times {gen_rounds}), deliveries per window round {handled_tot / rounds:.2f}, rounds with more than K={K} sends {over_k}")
    print("  binding constraint:", binds)
    print("  handled-per-round histogram:", dict(sorted(hist.items())))


if __name__ == "__main__":
    main()
