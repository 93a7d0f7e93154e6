#!/usr/bin/env python3
"""Developer tool (GPU box): the wide kernel with part of every node queue in LDS (inbox_capacity k, the spill area the rest) — sim ms per batch."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maelstrom_amd import engine as E  # noqa: E402

CASES = {
    "cfg3": (dict(workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 16384),
    "bcast100exp": (dict(workload="broadcast", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 2048),
    "bcast100": (dict(workload="broadcast", node_count=100, rate=100, time_limit=20), 2048),
}
for name in sys.argv[1].split(","):
    kw, n = CASES[name]
    base = E.test_config(seed=99, **kw)
    depth = base.inbox_capacity + base.spill_capacity
    for k in [int(x) for x in sys.argv[2].split(",")]:
        cfg = E.test_config(seed=99, inbox_capacity=k, spill_capacity=depth - k, **kw) if k else base
        with E.Engine(cfg) as eng:
            eng.run(0, n)
            eng.run(n, n)
            sim_ms = eng.kernel_ms()[0]
            eng.fetch()
            flagged = sum(1 for i in range(0, n, 64) if eng.meta(i).flags)
        print(json.dumps({"case": name, "lds_envelopes_per_node": k, "depth": depth, "sim_ms": round(sim_ms, 2), "flagged_sampled": flagged}), flush=True)
