#!/bin/bash
# round 3, call aq: what small broadcast clusters cost in the two-per-wavefront kernel (is an eight-per-wavefront broadcast kernel worth building?)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3aq; mkdir -p $O
timeout 600 python - > $O/sweep.txt 2>&1 <<'P'
import sys
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
for name, kw in {"broadcast n=5 rate100 20s lat0": dict(workload="broadcast", node_count=5, rate=100, time_limit=20),
                 "broadcast n=5 rate100 20s lat10": dict(workload="broadcast", node_count=5, rate=100, time_limit=20, latency=10),
                 "broadcast n=5 rate100 20s lat10 + partitions (colo)": dict(workload="broadcast", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5),
                 "broadcast ack-retry n=5 rate100 20s lat10 + partitions (colo)": dict(workload="broadcast", bin="broadcast-ack-retry", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5)}.items():
    for n in (16384,):
        cfg = E.test_config(seed=99, **kw)
        with E.Engine(cfg) as eng:
            eng.run(0, n); eng.run(n, n)
            ms = eng.kernel_ms()[0]
            eng.fetch(); msgs = sum(int(eng.net_stats_raw(i).all_send) for i in range(64)) / 64
        print(f"{name:64s} {n:6d} clusters: {ms:8.2f} ms   ({msgs:.0f} msgs per cluster)", flush=True)
P
cat $O/sweep.txt
