#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2t}; mkdir -p $O
timeout 300 python -m pytest tests/test_txn_check_gpu.py -m gpu -q -x --timeout 120 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
MSIM_DEV_FLAGS=4096 timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > $O/cfg5.log 2>&1; cat $O/cfg5.log
