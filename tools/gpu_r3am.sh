#!/bin/bash
# round 3, call am: uid8_kernel (eight echo / unique-ids clusters per wavefront): parity, the unique-ids bench shape against the colocated kernel, batch sizes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3am; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "echo or unique_ids or client_only" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
MSIM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 800 -k "test_random_options" > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 900 python - > $O/sweep.txt 2>&1 <<'P'
import sys
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
shapes = {"unique-ids n=3 rate1000 10s lat5 + partitions": dict(workload="unique-ids", node_count=3, rate=1000, time_limit=10, latency=5, nemesis=["partition"], nemesis_interval=10),
          "echo n=5 rate500 10s": dict(workload="echo", node_count=5, rate=500, time_limit=10)}
for name, kw in shapes.items():
    for n in (4096, 8192, 16384, 65536):
        row = []
        for flags in (0x400, 0x200):
            cfg = E.test_config(seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n); eng.run(n, n)
                row.append(eng.kernel_ms()[0])
        print(f"{name:48s} {n:6d} clusters: uid8 {row[0]:8.2f} ms   one cluster per wavefront {row[1]:8.2f} ms", flush=True)
P
cat $O/sweep.txt
timeout 600 python tools/bench_configs.py "unique-ids n=3 rate1000 10s lat5 + partitions" > $O/uniq.jsonl 2> $O/uniq.err; cut -c1-400 $O/uniq.jsonl
