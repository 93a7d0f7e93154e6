#!/usr/bin/env python3
"""Developer tool (GPU box): the headline shape (broadcast n=25 grid, 20 s) over a list of (latency ms, distribution, rate) points —
simulation ms per batch, rounds per cluster, ns per wave-round — to see what a round costs as queues deepen.
    python tools/latency_probe.py 0:constant:100 100:constant:100 100:constant:50 100:exponential:100 ...   [env N=4096]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402


def main():
    n = int(os.environ.get("N", "4096"))
    for spec in sys.argv[1:]:
        lat, dist, rate = spec.split(":")
        cfg = E.test_config("broadcast", node_count=25, rate=float(rate), time_limit=20, latency=int(lat), latency_dist=dist, seed=99)
        with E.Engine(cfg) as eng:
            eng.run(0, n)
            eng.run(n, n)
            sim_ms, _ = eng.kernel_ms()
            eng.fetch()
            rounds = [eng.meta(i).n_rounds for i in range(n)]
            pair_max = [max(rounds[i], rounds[min(i + 1, n - 1)]) for i in range(0, n, 2)]
            msgs = sum(int(eng.net_stats_raw(i).all_send) for i in range(0, n, 16)) * 16
            flagged = sum(1 for i in range(n) if eng.meta(i).flags)
        print(json.dumps({"latency_ms": int(lat), "dist": dist, "rate": float(rate), "instances": n, "sim_ms": round(sim_ms, 3), "rounds_mean": sum(rounds) / n, "rounds_max": max(rounds),
                          "wave_rounds_mean": sum(pair_max) / len(pair_max), "us_per_wave_round_of_slowest_wave": round(sim_ms * 1000 / max(pair_max), 4), "msgs": msgs, "flagged": flagged,
                          "inbox": cfg.inbox_capacity, "spill": cfg.spill_capacity}), flush=True)


if __name__ == "__main__":
    main()
