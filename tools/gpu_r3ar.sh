#!/bin/bash
# round 3, call ar: bcast8_kernel (eight broadcast clusters per wavefront, all four programs): parity, fuzz, the shapes of call aq against duo / the colocated kernel
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ar; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "broadcast" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
MSIM_FUZZ_CASES=400 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 800 -k "test_random_options" > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 900 python - > $O/sweep.txt 2>&1 <<'P'
import sys
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
shapes = {"broadcast n=5 rate100 20s lat0": dict(workload="broadcast", node_count=5, rate=100, time_limit=20),
          "broadcast n=5 rate100 20s lat10": dict(workload="broadcast", node_count=5, rate=100, time_limit=20, latency=10),
          "broadcast n=5 rate100 20s lat10 + partitions": dict(workload="broadcast", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5),
          "broadcast ack-retry n=5 rate100 20s lat10 + partitions": dict(workload="broadcast", bin="broadcast-ack-retry", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=5)}
for name, kw in shapes.items():
    for n in (4096, 16384):
        row = []
        for flags in (0x8400, 0x0 if "partitions" not in name else 0x200):
            cfg = E.test_config(seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n); eng.run(n, n)
                row.append(eng.kernel_ms()[0])
        print(f"{name:56s} {n:6d} clusters: bcast8 {row[0]:8.2f} ms   {'duo' if 'partitions' not in name else 'one cluster per wavefront'} {row[1]:8.2f} ms", flush=True)
P
cat $O/sweep.txt
