#!/usr/bin/env python3
"""Where does dt8_kernel (eight Datomic-style clusters per wavefront) overtake dt_kernel<> (one)?  Both layouts at several batch sizes and node
counts (MSIM_DEV_FLAGS bit 9 = one cluster per wavefront)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

for nodes in (2, 5):
    for n in (1024, 4096, 8192, 16384, 32768):
        out = {"nodes": nodes, "clusters": n}
        for name, flags in (("one", 0x200), ("eight", 0)):
            cfg = E.test_config("txn-list-append", bin="datomic", node_count=nodes, rate=100, time_limit=30, latency=5, nemesis=["partition"] if nodes >= 3 else (), nemesis_interval=10, seed=99)
            with E.Engine(cfg) as eng:
                if flags:
                    eng.set_dev_flags(flags)
                eng.run(0, n)
                eng.run(n, n)
                out[name] = round(eng.kernel_ms()[0], 1)
        print(json.dumps(out), flush=True)
