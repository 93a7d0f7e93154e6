#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2o}; mkdir -p $O
timeout 180 python -m pytest tests/test_lin_check_gpu.py -m gpu -q --timeout 120 > $O/pytest.log 2>&1; tail -15 $O/pytest.log
timeout 400 python -m pytest tests/test_checker_gpu.py tests/test_parity_gpu.py -m gpu -q --timeout 120 -x > $O/pytest2.log 2>&1; tail -5 $O/pytest2.log
