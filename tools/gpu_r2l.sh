#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2l}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 900 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -40
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
N=4096 LAT=100 DIST=exponential timeout 300 python tools/duo_prof_report.py > $O/prof_exp100.txt 2>&1
N=4096 LAT=50 DIST=uniform timeout 300 python tools/duo_prof_report.py > $O/prof_uni50.txt 2>&1
N=4096 LAT=0 timeout 300 python tools/duo_prof_report.py > $O/prof_lat0.txt 2>&1
N=4096 LAT=10 timeout 300 python tools/duo_prof_report.py > $O/prof_lat10.txt 2>&1
N=4096 LAT=100 timeout 300 python tools/duo_prof_report.py > $O/prof_lat100.txt 2>&1
cat $O/prof_*.txt
