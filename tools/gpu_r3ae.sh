#!/bin/bash
# round 3, call ae: hat8_kernel as it ships (8-lane groups, batches of >= 8192 clusters): parity, the bench shapes, the rw check
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ae; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py -k "txn_rw_register" -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python -m pytest tests/test_rw_check_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x --timeout 500 > $O/tests2.log 2>&1; tail -2 $O/tests2.log
timeout 600 python tools/bench_configs.py "txn-rw-register hat n=2 rate100 30s + partitions" "txn-rw-register hat n=5 rate100 30s lat5 + partitions" > $O/hat.jsonl 2> $O/hat.err
cut -c1-420 $O/hat.jsonl; tail -2 $O/hat.err
