#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2g}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 900 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -40; grep "exp\|uni" $O/debug.log
timeout 900 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat0" "cfg2 broadcast n=25 grid lat10" "cfg2 broadcast n=25 grid lat100" "cfg2 broadcast n=25 grid lat100 exponential" "cfg2 broadcast n=25 total lat100" > $O/configs.jsonl 2> $O/configs.err
python3 -c "
import json
for l in open('$O/configs.jsonl'):
    d=json.loads(l); print(d['config'], 'sim_ms %.1f'%d['sim_ms'], 'msgs/s %.3g'%d['msgs_per_s'], 'valid', d['valid'], 'flagged', d['flagged'])"
