#!/usr/bin/env python3
"""where does kafka8_kernel (eight clusters per wavefront) overtake kafka_kernel<> (one)?  Times both layouts at 4096 / 8192 /
12288 / 16384 clusters for 1, 3, 5 and 7 nodes (MSIM_DEV_FLAGS bit 9 = one cluster per wavefront, bit 10 = the packed layout or fail)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

for nodes in (1, 3, 5, 7):
    for n in (4096, 8192, 12288, 16384):
        out = {"nodes": nodes, "clusters": n}
        for name, flags in (("one", 0x200), ("eight", 0x400)):
            cfg = E.test_config("kafka", node_count=nodes, rate=100, time_limit=20, latency=5, nemesis=["partition"] if nodes >= 3 else (), nemesis_interval=10, seed=99)
            try:
                with E.Engine(cfg) as eng:
                    eng.set_dev_flags(flags)
                    eng.run(0, n)
                    eng.run(n, n)
                    out[name] = round(eng.kernel_ms()[0], 2)
            except E.EngineError as ex:
                out[name] = str(ex)[:80]
        print(json.dumps(out), flush=True)
