#!/bin/bash
# round 3, call ay: rw_check_kernel with its tables sized by the configuration (one launch for 16384 histories instead of four): tests, the demo shape
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ay; mkdir -p $O
timeout 200 python -m pytest tests/test_rw_check_gpu.py -m gpu -q -x > $O/tests.log 2>&1; tail -1 $O/tests.log
MSIM_DEV_FLAGS=0x1000 timeout 200 python tools/bench_configs.py "txn-rw-register hat n=2 rate100 30s + partitions" "txn-rw-register hat n=5 rate100 30s lat5 + partitions" > $O/hat.jsonl 2> $O/hat.err; cut -c1-420 $O/hat.jsonl; grep "rw-check" $O/hat.err | tail -2
