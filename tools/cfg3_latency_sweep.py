#!/usr/bin/env python3
"""the wide g-set kernel's lazy replicate merge keeps ONE pending tick per node; when latencies are comparable to the 5 s
replicate period ticks interleave and a delivery of another tick flushes first.  Times BASELINE cfg3's shape (n = 100, rate 100, 20 s) at
exponential latencies from 100 ms to 5 s and prints ms per batch and ns per simulated message."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
for lat in (100, 500, 1000, 2500, 5000):
    cfg = E.test_config("g-set", node_count=100, rate=100, time_limit=20, latency=lat, latency_dist="exponential", seed=99)
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.run(n, n)
        ms = eng.kernel_ms()[0]
        eng.fetch()
        msgs = sum(int(eng.net_stats_raw(i).all_send) for i in range(n))
        flagged = sum(1 for i in range(n) if eng.meta(i).flags)
    print(json.dumps({"latency_ms": lat, "instances": n, "sim_ms": round(ms, 1), "msgs": msgs, "ns_per_msg": round(ms * 1e6 / msgs, 3), "flagged": flagged}), flush=True)
