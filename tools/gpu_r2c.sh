#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2c}; mkdir -p $O
MSIM_DEV_FLAGS=256 timeout 600 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_checker_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 200 python tools/duo_prof_report.py > $O/prof_lat0.txt 2>&1
LAT=10 timeout 200 python tools/duo_prof_report.py > $O/prof_lat10.txt 2>&1
LAT=100 timeout 200 python tools/duo_prof_report.py > $O/prof_lat100.txt 2>&1
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-gather > $O/bench_duo.json 2> $O/bench_duo.err
grep -c identical $O/debug.log; grep -v identical $O/debug.log | head -20; tail -3 $O/pytest.log; cat $O/prof_lat0.txt $O/prof_lat10.txt $O/prof_lat100.txt; python3 -c "
import json; d=json.load(open('$O/bench_duo.json')); print(d['value'], d['ms_per_step'], d['kernel_ms'])"
