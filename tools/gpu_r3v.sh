#!/bin/bash
# round 3, call v: mk8 after the batched handler loads: parity, timing, cycle split
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3v; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_txn_check_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "multi_key or txn" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python tools/bench_configs.py "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" 2>$O/err.log | tee $O/cfg5mk.jsonl | cut -c1-330
timeout 600 python tools/mk8_prof_report.py > $O/mk8_prof.txt 2>&1; cat $O/mk8_prof.txt
