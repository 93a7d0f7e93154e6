#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2y}; mkdir -p $O
timeout 600 python -m pytest tests/test_edge_cases_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
