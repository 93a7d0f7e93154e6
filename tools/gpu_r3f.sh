#!/bin/bash
# round 3, call f: check_kernel with adjacent pairs settled by a shuffle, single guard per read; checker tests + headline profile
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3f; mkdir -p $O
timeout 600 python -m pytest tests/test_set_full_batch_gpu.py tests/test_checker_gpu.py tests/test_edge_cases_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 500 bash tools/profile_headline.sh r3f/headline full > $O/headline_prof.log 2>&1; grep "check_kernel" $O/headline_prof.log | tail -4
