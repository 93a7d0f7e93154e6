#!/usr/bin/env python3
"""Developer tool (GPU box): kafka n=5 + partitions with the K8_PROF build (tools/variant_lib.sh k8prof kafka8.hip -DK8_PROF): cycles of a wavefront of
kafka8_kernel<> by section of the round (the first cluster of every wavefront carries the counters).  Env: N (instances), MSIM_LIB."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_k8prof.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload="kafka", node_count=5, rate=100, time_limit=20, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
n = int(os.environ.get("N", "16384"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    a = []
    for i in range(0, n, 64):   # (instance 0 of every eighth wavefront)
        st, m = eng.net_stats_raw(i), eng.meta(i)
        a.append([st.all_send, st.all_recv, st.clients_send, st.clients_recv, st.servers_send, st.servers_recv, m.reserved[0] * 64, m.reserved[1] * 64, m.n_events])
a = np.array(a, dtype=np.float64)
names = ["phase checks + R0 (time) + timeouts", "R1 scheduler / generator", "R2 invocations", "R3 nodes (+ reply payloads)", "R3 service", "COMMIT + polls", "R4 clients", "rows"]
tot = a[:, :8].sum(axis=1)
print(f"kafka n=5 + partitions, {n} instances, eight clusters per wavefront: sim kernel {sim_ms:.2f} ms; wave-rounds mean {a[:, 8].mean():.0f} max {a[:, 8].max():.0f}; cycles per wavefront mean {tot.mean():.3e} = {tot.mean() / a[:, 8].mean():.0f} per wave-round")
for i, nm in enumerate(names):
    print(f"  {nm:40s} {a[:, i].mean() / a[:, 8].mean():8.0f} cycles per wave-round  {100 * a[:, i].mean() / tot.mean():5.1f} %")
