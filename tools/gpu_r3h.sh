#!/bin/bash
# round 3, call h: kafka kernel on the device; whole GPU suite
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3h; mkdir -p $O
timeout 300 python -m pytest tests/test_kafka_gpu.py -m gpu -q -x --timeout 300 > $O/kafka.log 2>&1; tail -3 $O/kafka.log
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
