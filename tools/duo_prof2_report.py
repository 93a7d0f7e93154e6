#!/usr/bin/env python3
"""Developer tool (GPU box): the headline shape with the DUO_PROF2 build (tools/variant_lib.sh p2 duo.hip -DDUO_PROF2): cycles of a
wavefront by section of the gossip round.  Env: N, LAT, DIST, MSIM_LIB."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_p2.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload="broadcast", bin="broadcast-ff", node_count=25, rate=100, time_limit=20, latency=int(os.environ.get("LAT", "0")), latency_dist=os.environ.get("DIST", "constant"), seed=2026)
if kw["latency"] == 0:
    kw["inbox_capacity"] = 6
n = int(os.environ.get("N", "4096"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    a = np.array([[eng.meta(i).n_events, *eng.meta(i).reserved, eng.meta(i + 1).n_events, *eng.meta(i + 1).reserved, eng.meta(i).n_rounds] for i in range(0, n, 2)], dtype=np.float64)
a[:, :8] *= 64
if os.environ.get("GENERAL"):   # the DUO_PROF3 build
    names = ["gossip rounds", "GENERAL: R1 scheduler", "GENERAL: R2 invoke + poll", "GENERAL: R3 dedup, read copies", "GENERAL: ids + arrivals", "GENERAL: poll", "GENERAL: rows", "GENERAL: scheduler's view"]
else:
    names = ["R0 time / round kind", "R3 dedup (seen set)", "ids (prefix sum)", "arrivals (pull, latency draw, push)", "poll + general rounds + loop"]
tot = a[:, :8].sum(axis=1).mean()
print(f"latency {kw['latency']} ms {kw['latency_dist']}, {n} instances: sim kernel {sim_ms:.3f} ms, cycles per wavefront {tot:.3e}, cluster rounds {a[:, 8].mean():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:38s} {a[:, i].mean():12.3e} cycles  {100 * a[:, i].mean() / tot:5.1f} %")
