#!/bin/bash
# round 3, call af: hat_kernel<> with the long-list passes done by the whole wavefront too: parity, sim ms over batch sizes for both layouts
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3af; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "txn_rw_register" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 900 python - > $O/sweep.txt 2>&1 <<'P'
import sys, time
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
shapes = {"n=2 rate100 30s + partitions": dict(workload="txn-rw-register", node_count=2, rate=100, time_limit=30, nemesis=["partition"], nemesis_interval=10),
          "n=5 rate100 30s lat5 + partitions": dict(workload="txn-rw-register", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10),
          "n=3 rate100 30s lat5 (healthy)": dict(workload="txn-rw-register", node_count=3, rate=100, time_limit=30, latency=5)}
for name, kw in shapes.items():
    for n in (2048, 4096, 8192, 16384):
        row = []
        for flags in (0x400, 0x200):
            cfg = E.test_config(seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n); eng.run(n, n)
                row.append(eng.kernel_ms()[0])
        print(f"{name:40s} {n:6d} clusters: hat8 {row[0]:8.2f} ms   hat_kernel {row[1]:8.2f} ms", flush=True)
P
cat $O/sweep.txt
