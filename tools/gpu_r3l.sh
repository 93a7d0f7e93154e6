#!/bin/bash
# round 3, call l: where a wavefront of the wide kernel spends its cycles (WIDE_PROF build): cfg3 with the sets in LDS / in HBM,
# wide broadcast at latency 0 and 100 ms exponential; new tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3l; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_txn_check_gpu.py -m gpu -q -x -k "wide_g_set or device_pass" --timeout 600 > $O/new_tests.log 2>&1; tail -2 $O/new_tests.log
timeout 300 python tools/wide_prof_report.py > $O/wprof_cfg3_setl.txt 2>&1; cat $O/wprof_cfg3_setl.txt
MSIM_DEV_FLAGS=0x4000 timeout 300 python tools/wide_prof_report.py > $O/wprof_cfg3_hbm.txt 2>&1; cat $O/wprof_cfg3_hbm.txt
WL=broadcast LAT=0 DIST=constant N=2048 timeout 300 python tools/wide_prof_report.py > $O/wprof_bcast_lat0.txt 2>&1; cat $O/wprof_bcast_lat0.txt
WL=broadcast N=2048 timeout 300 python tools/wide_prof_report.py > $O/wprof_bcast_exp.txt 2>&1; cat $O/wprof_bcast_exp.txt
