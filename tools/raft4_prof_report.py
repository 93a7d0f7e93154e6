#!/usr/bin/env python3
"""Developer tool (GPU box): runs the Raft lin-kv batch (BASELINE configs[3]) with the R4_PROF build (tools/variant_lib.sh r4prof raft4.hip -DR4_PROF) and
prints the cycles a wavefront spends in each section of the round.  Env: N (instances), PART=1 (partitions + 10 ms latency)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_r4prof.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload="lin-kv", bin="raft", node_count=5, rate=30, time_limit=60, seed=99)
if os.environ.get("PART"):
    kw.update(latency=10, nemesis=["partition"], nemesis_interval=10)
n = int(os.environ.get("N", "8192"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    v = np.array([[x for g in range(4) for x in (eng.meta(i + g).n_events, *eng.meta(i + g).reserved)] for i in range(0, n - 3, 4)], dtype=np.float64)
    rounds = np.array([eng.meta(i).n_rounds for i in range(n)], dtype=np.float64)
names = ["top + R0 time", "R1 scheduler", "R2 invoke", "R3 message", "R3 action", "R3 next-action time", "commit + poll", "R4 clients", "rows"]
cyc = v[:, :9] * 64
wr = v[:, 9]
tot = cyc.sum(axis=1)
print(f"{kw}\n{n} instances: sim kernel {sim_ms:.3f} ms; wave-rounds {wr.mean():.0f} (cluster rounds {rounds.mean():.0f}); cycles per wavefront {tot.mean():.3e} = {tot.mean() / wr.mean():.0f} per wave-round")
for i, nm in enumerate(names):
    print(f"  {nm:22s} {cyc[:, i].mean() / wr.mean():8.0f} cycles/wave-round  {100 * cyc[:, i].mean() / tot.mean():5.1f} %")
