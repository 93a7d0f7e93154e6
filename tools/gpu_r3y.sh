#!/bin/bash
# round 3, call y: txn8 with queues of 8 + batched spill scans; the LDS txn-check kernel as the default: cfg5 both nodes; tests
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3y; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_txn_check_gpu.py tests/test_edge_cases_gpu.py tests/test_elle_reference_vectors.py -m gpu -q -x -k "txn or list_append or elle" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" 2>$O/err.log | tee $O/cfg5.jsonl | cut -c1-330
