#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wide; mkdir -p $O
timeout 300 python -m pytest tests/test_parity_gpu.py tests/test_checker_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -k "wide or widest or misuse" --timeout 120 > $O/pytest.log 2>&1; grep -E "passed|failed|Error|assert|differ|^FAILED" $O/pytest.log | head -30
