#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2r}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 600 python tools/duo_debug.py n25-exp100 n25-uni50 n5-exp20 n25-total-exp n5-exp200-tiny n9-echoback-uni n25-lat10 n25-lat0 > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -30
timeout 300 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat100 exponential" "cfg2 broadcast n=25 grid lat0" > $O/cfg2.log 2>&1; cat $O/cfg2.log
