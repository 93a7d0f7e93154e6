#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-full_tests}; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
