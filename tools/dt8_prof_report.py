#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE configs[4] over the Datomic-style node with the D8_PROF build (tools/variant_lib.sh d8prof dt8.hip -DD8_PROF): cycles a wavefront
spends in each section of the round.  Env: N (instances)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_d8prof.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload="txn-list-append", bin="datomic", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
n = int(os.environ.get("N", "32768"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    v = np.array([[eng.meta(i).n_events, *eng.meta(i).reserved, eng.meta(i + 1).n_events, *eng.meta(i + 1).reserved, eng.meta(i + 2).n_events] for i in range(0, n - 7, 8)], dtype=np.float64)
    rounds = np.array([eng.meta(i).n_rounds for i in range(n)], dtype=np.float64)
names = ["top + R0 time", "R1 scheduler", "R2 invoke", "R3 nodes + services", "completed txns -> payload", "commit + poll", "R4 clients", "rows"]
if os.environ.get("FINE"):   # the -DD8_PROF2 build
    names = ["top .. R2 invoke", "R3 handlers (nodes, lin-kv, lww-kv)", "R3 apply_txn", "R3 answer + unlock", "completed txns -> payload", "commit: arrivals", "commit: poll", "R4 clients + rows"]
cyc = v[:, :8] * 64
wr = v[:, 8]
tot = cyc.sum(axis=1)
print(f"{n} instances: sim kernel {sim_ms:.3f} ms; wave-rounds {wr.mean():.0f} (cluster rounds {rounds.mean():.0f}); cycles per wavefront {tot.mean():.3e} = {tot.mean() / wr.mean():.0f} per wave-round")
for i, nm in enumerate(names):
    print(f"  {nm:28s} {cyc[:, i].mean() / wr.mean():8.0f} cycles/wave-round  {100 * cyc[:, i].mean() / tot.mean():5.1f} %")
