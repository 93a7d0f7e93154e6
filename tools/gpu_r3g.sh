#!/bin/bash
# round 3, call g: rw-register analysis on the device: tests, timings of the two txn-rw-register configs (device vs host check)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3g; mkdir -p $O
timeout 600 python -m pytest tests/test_rw_check_gpu.py tests/test_set_full_batch_gpu.py -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
MSIM_DEV_FLAGS=4096 timeout 400 python tools/bench_configs.py "txn-rw-register hat n=2 rate100 30s + partitions" "txn-rw-register hat n=5 rate100 30s lat5 + partitions" > $O/rw.jsonl 2> $O/rw.err; cut -c1-420 $O/rw.jsonl; grep "rw-check" $O/rw.err | tail -4
MSIM_DEV_FLAGS=2048 timeout 400 python tools/bench_configs.py "txn-rw-register hat n=2 rate100 30s + partitions" "txn-rw-register hat n=5 rate100 30s lat5 + partitions" > $O/rw_host.jsonl 2> $O/rw_host.err; cut -c1-420 $O/rw_host.jsonl
