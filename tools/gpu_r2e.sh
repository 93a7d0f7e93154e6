#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2e}; mkdir -p $O
MSIM_DEV_FLAGS=256 timeout 600 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
for t in prof noswpf; do
  L=$PWD/maelstrom_amd/libmaelsim_$t.so
  MSIM_LIB=$L MSIM_DEV_FLAGS=256 timeout 300 python tools/duo_debug.py n25-lat0 n25-lat10 n12-spill > $O/debug_$t.log 2>&1
  MSIM_LIB=$L timeout 200 python tools/duo_prof_report.py > $O/prof_${t}_lat0.txt 2>&1
  MSIM_LIB=$L LAT=10 timeout 200 python tools/duo_prof_report.py > $O/prof_${t}_lat10.txt 2>&1
done
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-gather > $O/bench_duo.json 2> $O/bench_duo.err
grep -c identical $O/debug.log; grep -v identical $O/debug.log | head -20; cat $O/debug_*.log | grep -v "n events\|^$" | head; cat $O/prof_*; python3 -c "
import json; d=json.load(open('$O/bench_duo.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
