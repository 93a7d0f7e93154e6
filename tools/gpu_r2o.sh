#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2o}; mkdir -p $O
timeout 180 python -m pytest tests/test_lin_check_gpu.py -m gpu -q --timeout 120 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
MSIM_DEV_FLAGS=4096 timeout 600 python tools/bench_configs.py "cfg4 lin-kv raft + partitions lat10" "cfg4 lin-kv raft n=5 c=10 rate30 60s" > $O/cfg4.log 2>&1; cat $O/cfg4.log
