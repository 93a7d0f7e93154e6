#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2z}; mkdir -p $O
timeout 600 python -m pytest tests/test_checker_gpu.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 300 python tools/bench_configs.py "unique-ids n=3 rate1000 10s lat5 + partitions" "pn-counter n=5 rate100 20s lat100 exponential" "g-counter n=5 rate100 20s lat10" > $O/uid.log 2>&1; cat $O/uid.log
