#!/usr/bin/env python3
"""Developer tool (GPU box): BASELINE cfg3 (or WL=broadcast) with the WIDE_PROF build (tools/variant_lib.sh wprof k_wide_gset.hip -DWIDE_PROF): cycles of a wavefront of
sim_kernel_wide<> by section of the round.  Env: N (instances), WL, NODES, LAT, DIST, PLOSS, MSIM_LIB, MSIM_DEV_FLAGS."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_wprof.so"))
sys.path.insert(0, ROOT)
from maelstrom_amd import engine as E  # noqa: E402

kw = dict(workload=os.environ.get("WL", "g-set"), node_count=int(os.environ.get("NODES", "100")), rate=100, time_limit=20,
          latency=int(os.environ.get("LAT", "100")), latency_dist=os.environ.get("DIST", "exponential"), p_loss=float(os.environ.get("PLOSS", "0")), seed=99)
n = int(os.environ.get("N", "16384"))
cfg = E.test_config(**kw)
with E.Engine(cfg) as eng:
    eng.run(0, n)
    eng.run(n, n)
    sim_ms = eng.kernel_ms()[0]
    eng.fetch()
    a = []
    for i in range(0, n, max(1, n // 512)):
        st, m = eng.net_stats_raw(i), eng.meta(i)
        a.append([st.all_send, st.all_recv, st.clients_send, st.clients_recv, st.servers_send, st.servers_recv, m.reserved[0] * 64, m.reserved[1] * 64, m.reserved[2] * 64, m.n_events * 64, m.n_rounds, m.n_payload_words, m.flags & 0xFFFF, m.flags >> 16])
a = np.array(a, dtype=np.float64)
names = ["phase checks + R0 (time) + the windows' bounds", "quiet-round test + windows: who is eligible", "quiet rounds + windows: deliveries noted", "quiet rounds + windows: polls", "general: scheduler .. arrivals", "general: sort passes", "general: polls", "rows", "lone-operation path + its test", "general: R4 (clients)"]
NS = len(names)
tot = a[:, :NS].sum(axis=1).mean()
print(f"{kw['workload']} n={kw['node_count']} latency {kw['latency']} ms {kw['latency_dist']}, {n} instances: sim kernel {sim_ms:.2f} ms, cycles per wavefront {tot:.3e}, rounds {a[:, NS].mean():.0f} ({tot / a[:, NS].mean():.0f} cycles per round)")
print(f"  turns of the round loop {a[:, NS + 1].mean():.0f}; quiet windows entered {a[:, NS + 2].mean():.0f}, turns of their loops {a[:, NS + 3].mean():.0f}")
for i, nm in enumerate(names):
    print(f"  {nm:34s} {a[:, i].mean():12.3e} cycles  {100 * a[:, i].mean() / tot:5.1f} %")
