#!/bin/bash
# round 3, call x: txn_check_lds_kernel after Kahn's steps were spread over the lanes: tests, cfg5 check time by threads per history, phase split
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3x; mkdir -p $O
timeout 600 python -m pytest tests/test_txn_check_gpu.py tests/test_elle_reference_vectors.py tests/test_txn_list_append.py -q -x --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
C5="cfg5 txn-list-append n=5 rate100 30s lat5 + partitions"
for wg in 512 256 128; do echo "-- LDS kernel, $wg threads per history"; MSIM_TXN_WG=$wg MSIM_DEV_FLAGS=0x3000 timeout 300 python tools/bench_configs.py "$C5" 2> $O/cfg5_wg$wg.err | tee -a $O/cfg5_wg.jsonl | cut -c150-330; grep "txn-check" $O/cfg5_wg$wg.err | tail -2; done
echo "-- HBM tables"; MSIM_DEV_FLAGS=0x1000 timeout 300 python tools/bench_configs.py "$C5" 2> $O/cfg5_hbm.err | cut -c150-330; grep "txn-check" $O/cfg5_hbm.err | tail -2
echo "-- phases, LDS kernel"; MSIM_DEV_FLAGS=0x2000 timeout 400 python tools/txn_check_prof_report.py 2>&1 | tail -10 | tee $O/tc_lds.txt
