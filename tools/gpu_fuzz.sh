#!/bin/bash
# a wide differential sweep on the GPU box: tools/gpu_fuzz.sh [cases [instances]]
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/fuzz
MSIM_FUZZ_CASES=${1:-300} MSIM_FUZZ_INSTANCES=${2:-9} timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 300 -n 8 ${3:+-k $3} > gpurun_out/fuzz/pytest.log 2>&1; tail -8 gpurun_out/fuzz/pytest.log
