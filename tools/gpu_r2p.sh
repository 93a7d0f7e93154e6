#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2p}; mkdir -p $O
MSIM_DEV_FLAGS=4096 timeout 600 python tools/bench_configs.py "cfg4 lin-kv raft + partitions lat10" "cfg4 lin-kv raft n=5 c=10 rate30 60s" > $O/cfg4p.log 2>&1; cat $O/cfg4p.log
