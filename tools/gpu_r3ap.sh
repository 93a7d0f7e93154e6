#!/bin/bash
# round 3, call ap: the differential sweeps at 17 instances per option set (two full wavefronts and a partial one in the eight-per-wavefront layouts)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ap; mkdir -p $O
MSIM_FUZZ_CASES=400 MSIM_FUZZ_INSTANCES=17 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 1200 -k "test_random_options or rw_register or kafka" > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
