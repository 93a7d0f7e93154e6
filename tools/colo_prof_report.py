#!/usr/bin/env python3
"""Developer tool (GPU box): acknowledged gossip n=25 + partitions with the COLO_PROF build (tools/variant_lib.sh cprof k_general_c.hip -DCOLO_PROF): cycles of a
wavefront of sim_kernel_colo<> by kind of round.  Env: N (instances), MSIM_LIB."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_cprof.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload="broadcast", bin="broadcast-ack-retry", node_count=25, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=10, seed=99)
n = int(os.environ.get("N", "2048"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    a = []
    for i in range(0, n, max(1, n // 512)):
        st, m = eng.net_stats_raw(i), eng.meta(i)
        a.append([st.all_send, st.all_recv, st.clients_send, st.clients_recv, st.servers_send, st.servers_recv, m.n_rounds])
a = np.array(a, dtype=np.float64)
names = ["gossip round: time", "gossip round: wake-ups and deliveries", "gossip round: COMMIT", "gossip round: polls", "every other round"]
tot = a[:, :5].sum(axis=1)
slow = int(np.argmax(tot))
print(f"acknowledged gossip n=25 + partitions, {n} instances: sim kernel {sim_ms:.2f} ms; rounds mean {a[:, 6].mean():.0f} max {a[:, 6].max():.0f}, of them gossip rounds mean {a[:, 5].mean():.0f}; cycles per wavefront mean {tot.mean():.3e} max {tot.max():.3e}")
for label, row in (("mean", a.mean(axis=0)), ("slowest wavefront", a[slow])):
    print(f" {label}: gossip rounds {row[5]:.0f} of {row[6]:.0f}, {row[:4].sum() / max(row[5], 1):.0f} cycles each; other rounds {row[4] / max(row[6] - row[5], 1):.0f} cycles each")
    for i, nm in enumerate(names):
        print(f"  {nm:40s} {row[i]:12.3e} cycles  {100 * row[i] / row[:5].sum():5.1f} %")
