#!/bin/bash
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/tmp
timeout 300 python -m pytest tests/test_txn_check_gpu.py -m gpu -q -x --timeout 120 2>&1 | tail -3
MSIM_DEV_FLAGS=4096 timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" 2>&1 | tail -2
