#!/bin/bash
# round 3, call an: crdt8_kernel (eight g-set / pn-counter / g-counter clusters per wavefront): parity, fuzz, batch sizes against the colocated kernel, the bench configs
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3an; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "g_set_parity or pn_counter or g_counter or crdt or echo or unique_ids or client_only" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
MSIM_FUZZ_CASES=300 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 800 -k "test_random_options" > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 900 python - > $O/sweep.txt 2>&1 <<'P'
import sys
sys.path.insert(0, ".")
from maelstrom_amd import engine as E
shapes = {"pn-counter n=5 rate100 20s lat100 exponential": dict(workload="pn-counter", node_count=5, rate=100, time_limit=20, latency=100, latency_dist="exponential"),
          "g-set n=5 rate100 20s lat10 + partitions": dict(workload="g-set", node_count=5, rate=100, time_limit=20, latency=10, nemesis=["partition"], nemesis_interval=10)}
for name, kw in shapes.items():
    for n in (2048, 4096, 16384, 65536):
        row = []
        for flags in (0x400, 0x200):
            cfg = E.test_config(seed=99, **kw)
            with E.Engine(cfg) as eng:
                eng.set_dev_flags(flags)
                eng.run(0, n); eng.run(n, n)
                row.append(eng.kernel_ms()[0])
        print(f"{name:48s} {n:6d} clusters: crdt8 {row[0]:8.2f} ms   one cluster per wavefront {row[1]:8.2f} ms", flush=True)
P
cat $O/sweep.txt
timeout 600 python tools/bench_configs.py "cfg1 echo n=3" "pn-counter n=5 rate100 20s lat100 exponential" "g-counter n=5 rate100 20s lat10" > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-400 $O/cfg.jsonl
