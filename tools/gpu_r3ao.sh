#!/bin/bash
# round 3, call ao: crdt8 with states of up to 64 words: the g-set shape of call an again, parity
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ao; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "g_set_parity or crdt" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -1 $O/tests.log
timeout 900 python - > $O/sweep.txt 2>&1 <<'P'
import sys
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
shapes = {"g-set n=5 rate100 20s lat10 + partitions": dict(workload="g-set", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=10)}
for name, kw in shapes.items():
    for n in (4096, 16384, 65536):
        row = []
        for flags in (0x400, 0x200):
            cfg = E.test_config(seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n); eng.run(n, n); eng.check()
                row.append(eng.kernel_ms())
        print(f"{name:48s} {n:6d} clusters: crdt8 {row[0][0]:8.2f} ms (check {row[0][1]:.2f})   one cluster per wavefront {row[1][0]:8.2f} ms", flush=True)
P
cat $O/sweep.txt
