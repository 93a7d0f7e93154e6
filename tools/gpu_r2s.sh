#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2s}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 600 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -30
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat10" "cfg2 broadcast n=25 grid lat100" "cfg2 broadcast n=25 grid lat100 exponential" > $O/cfg2.log 2>&1; cat $O/cfg2.log
