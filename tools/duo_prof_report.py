#!/usr/bin/env python3
"""Developer tool (GPU box): runs the headline batch with the DUO_PROF build (tools/variant_lib.sh prof duo.hip -DDUO_PROF) and prints, per wavefront,
the number of wave-rounds, how many of them were GENERAL rounds, and the cycles spent in each kind."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ.setdefault("MSIM_LIB", os.path.join(ROOT, "maelstrom_amd", "libmaelsim_prof.so"))
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
    m = np.array([[eng.meta(i).n_events, eng.meta(i).reserved[0], eng.meta(i).reserved[1], eng.meta(i).reserved[2], eng.meta(i).n_rounds] for i in range(0, n, 2)], dtype=np.float64)
ngen, nwave, cgen, ctot, rounds = m.T
cgen *= 64; ctot *= 64
print(f"latency {kw['latency']} ms {kw['latency_dist']}, {n} instances: sim kernel {sim_ms:.3f} ms")
print(f"per wavefront: wave-rounds {nwave.mean():.0f} (cluster rounds {rounds.mean():.0f}), GENERAL {ngen.mean():.0f} ({100 * ngen.mean() / nwave.mean():.1f} %)")
print(f"cycles per wavefront {ctot.mean():.3e} (max {ctot.max():.3e}); in GENERAL rounds {cgen.mean():.3e} ({100 * cgen.mean() / ctot.mean():.1f} %)")
print(f"cycles per GENERAL round {cgen.mean() / ngen.mean():.0f}, per gossip round {(ctot.mean() - cgen.mean()) / (nwave.mean() - ngen.mean()):.0f}")
