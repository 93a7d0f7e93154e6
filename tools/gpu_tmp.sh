#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wide; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_checker_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -k "wide or widest or misuse" --timeout 120 > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert|differ" $O/pytest.log | head -30
timeout 200 python - > $O/timing.txt 2>&1 <<'PY'
from maelstrom_amd import engine as E
import numpy as np
def run(name, n, **kw):
    cfg = E.test_config(**kw)
    with E.Engine(cfg) as eng:
        eng.run(0, n); eng.run(n, n)
        sim = eng.kernel_ms()[0]
        eng.check(); chk = eng.kernel_ms()[1]
        eng.fetch()
        msgs = sum(eng.net_stats_raw(i).all_send for i in range(n))
        res = eng.check_results()
        flags = sum(1 for i in range(n) if eng.meta(i).flags)
    print(name, n, "sim_ms", round(sim, 2), "check_ms", round(chk, 2), "msgs", msgs, "msgs/s", f"{msgs / ((sim + chk) / 1e3):.3e}", "valid", int((res["valid"] == 1).sum()), "flagged", flags)
run("broadcast n=100 grid lat0", 2048, workload="broadcast", node_count=100, rate=100, time_limit=20, latency=0, seed=1)
run("broadcast n=100 grid lat100exp", 2048, workload="broadcast", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", seed=1)
run("g-set n=100 cfg3", 4096, workload="g-set", node_count=100, rate=100, time_limit=20, latency=100, latency_dist="exponential", seed=99)
PY
cat $O/timing.txt | tail -5
