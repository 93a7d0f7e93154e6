#!/usr/bin/env python3
"""Where does dtg4_kernel (four clusters of the Datomic-style txn-list-append node with several workers per node per wavefront, csrc/dtg4.hip)
overtake dtg_kernel<> (one)?  Both layouts at several batch sizes and shapes (MSIM_DEV_FLAGS bit 9 = one cluster per wavefront, bit 10 = the
packed layout whatever the launch)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

SHAPES = {
    "n=1 c=10 rate100 30s lat0 (doc/05-datomic/01-single-node.md:257's invocation)": dict(node_count=1, concurrency=10, rate=100, time_limit=30, latency=0),
    "n=1 c=10 rate100 30s lat5": dict(node_count=1, concurrency=10, rate=100, time_limit=30, latency=5),
    "n=2 c=12 rate100 30s lat5 + partitions": dict(node_count=2, concurrency=12, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10),
}
for name, kw in SHAPES.items():
    for n in (2048, 4096, 8192, 16384):
        out = {"shape": name, "clusters": n}
        for lay, flags in (("one", 0x200), ("four", 0x400)):
            cfg = E.test_config("txn-list-append", bin="datomic", seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n)
                eng.run(n, n)
                out[lay] = round(eng.kernel_ms()[0], 2)
        print(json.dumps(out), flush=True)
