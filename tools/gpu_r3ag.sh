#!/bin/bash
# round 3, call ag: kafka_kernel with the committed-offset list bisected: parity, the bench config
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ag; mkdir -p $O
timeout 600 python -m pytest tests/test_kafka_gpu.py tests/test_kafka_check_gpu.py -m gpu -q -x --timeout 500 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 600 python tools/bench_configs.py "kafka n=5 rate100 20s lat5 + partitions" > $O/kafka.jsonl 2> $O/kafka.err; cut -c1-400 $O/kafka.jsonl; tail -2 $O/kafka.err
