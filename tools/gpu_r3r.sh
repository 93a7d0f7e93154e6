#!/bin/bash
# round 3, call r: mk_kernel with two transaction slots per node in LDS (9.6 KiB per cluster instead of 21.6): parity, cfg5-mk timing
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3r; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_txn_check_gpu.py -m gpu -q -x -k "multi_key or txn" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python tools/bench_configs.py "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" 2>$O/err.log | tee $O/cfg5.jsonl | cut -c1-330
