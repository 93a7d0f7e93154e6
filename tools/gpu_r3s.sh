#!/bin/bash
# round 3, call s: mk8_kernel (eight clusters of the canonical txn-list-append node per wavefront): parity on the device, cfg5-mk timing with / without
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3s; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_txn_check_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "multi_key or txn" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
CM="cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions"
MSIM_DEV_FLAGS=0x1000 timeout 600 python tools/bench_configs.py "$CM" 2>$O/err.log | tee $O/cfg5mk.jsonl | cut -c1-330; grep "mk8" $O/err.log | tail -1
echo "-- one cluster per wavefront (MSIM_DEV_FLAGS=0x200)"; MSIM_DEV_FLAGS=0x200 timeout 600 python tools/bench_configs.py "$CM" 2>>$O/err.log | tee $O/cfg5mk_single.jsonl | cut -c1-330
