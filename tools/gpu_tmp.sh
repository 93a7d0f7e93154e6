#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/wide; mkdir -p $O
MSIM_FUZZ_CASES=400 timeout 420 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -k "wide" --timeout 60 -x > $O/fuzz.log 2>&1; grep -E "passed|failed|Error|assert|differ|^FAILED" $O/fuzz.log | head -20
