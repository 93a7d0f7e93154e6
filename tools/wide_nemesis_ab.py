#!/usr/bin/env python3
"""Developer tool (GPU box): the wide kernel's programs under the partition nemesis, sim ms per batch (A/B of builds via MSIM_LIB)."""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from maelstrom_amd import engine as E  # noqa: E402

CASES = {
    "bcast": dict(workload="broadcast", node_count=64, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5),
    "ack": dict(workload="broadcast", bin="broadcast-ack-retry", node_count=64, rate=50, time_limit=10, latency=10, nemesis=["partition"], nemesis_interval=5),
    "gset": dict(workload="g-set", node_count=64, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5),
    "pn": dict(workload="pn-counter", node_count=64, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5),
}
n = int(os.environ.get("N", "2048"))
for name in sys.argv[1].split(","):
    cfg = E.test_config(seed=99, **CASES[name])
    with E.Engine(cfg) as eng:
        eng.run(0, n)
        eng.run(n, n)
        sim_ms = eng.kernel_ms()[0]
        eng.fetch()
        flagged = sum(1 for i in range(0, n, 64) if eng.meta(i).flags)
    print(json.dumps({"case": name, "lib": os.path.basename(os.environ.get("MSIM_LIB", "libmaelsim.so")), "instances": n, "sim_ms": round(sim_ms, 2), "flagged_sampled": flagged}), flush=True)
