#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2x}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 600 python tools/duo_debug.py n25-lat0 n25-lat10 n25-exp100 n9-echoback n12-spill raft raft-part raft-exp-loss raft-n3c6 > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -30
timeout 300 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat0" "cfg2 broadcast n=25 grid lat10" "cfg2 broadcast n=25 grid lat100 exponential" "cfg4 lin-kv raft n=5 c=10 rate30 60s" "cfg4 lin-kv raft + partitions lat10" > $O/cfg.log 2>&1; cat $O/cfg.log
timeout 300 python bench.py --cpu-sample 0 --no-gather --no-fetch 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
