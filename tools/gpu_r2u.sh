#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2u}; mkdir -p $O
MSIM_DEV_FLAGS=256 timeout 600 python tools/duo_debug.py txn txn-lat5 txn-part txn-exp-loss txn-n3 txn-n7 txn-n1 txn-len6 > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep identical $O/debug.log; grep -v "identical" $O/debug.log | head -30
timeout 600 python -m pytest tests/test_bench_shapes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_txn_check_gpu.py -m gpu -q -x --timeout 300 -k "txn or fuzz" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 300 python tools/txn8_prof_report.py > $O/prof.txt 2>&1; cat $O/prof.txt
timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > $O/cfg5.log 2>&1; cat $O/cfg5.log
