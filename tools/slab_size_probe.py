#!/usr/bin/env python3
"""Does the size of the per-instance slabs matter to the latency-bound transactional kernels (TLB reach: every cluster's hot words lie in its own slabs)?
cfg5 over the three list-append nodes with the default payload capacity (worst case: every read returns a full list) and with tighter ones."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from maelstrom_amd import engine as E  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
for b in (None, "multi-key-txn", "datomic"):
    for cap in (0, 48000, 24000):
        kw = dict(workload="txn-list-append", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
        if b:
            kw["bin"] = b
        if cap:
            kw["max_payload_words"] = cap
        cfg = E.test_config(**kw)
        with E.Engine(cfg) as eng:
            eng.run(0, n)
            eng.run(n, n)
            ms = eng.kernel_ms()[0]
            eng.fetch()
            fl = sum(1 for i in range(0, n, 61) if eng.meta(i).flags)
        print(json.dumps({"bin": b or "single-key-txn", "clusters": n, "max_payload_words": cfg.max_payload_words, "sim_ms": round(ms, 1), "flagged_sample": fl}), flush=True)
