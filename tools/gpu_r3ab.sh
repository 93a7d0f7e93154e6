#!/bin/bash
# round 3, call ab: the kafka checker's device pass (csrc/kafka_check_dev.hip): its tests, the kafka tests, the bench config (device vs host checker)
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ab; mkdir -p $O
timeout 600 python -m pytest tests/test_kafka_check_gpu.py tests/test_kafka_gpu.py -m gpu -q -x --timeout 500 > $O/tests.log 2>&1; tail -3 $O/tests.log
MSIM_DEV_FLAGS=0x1000 timeout 600 python tools/bench_configs.py "kafka n=5 rate100 20s lat5 + partitions" > $O/kafka_dev.jsonl 2> $O/kafka_dev.err; cut -c1-400 $O/kafka_dev.jsonl; grep "kafka-check" $O/kafka_dev.err | tail -4
MSIM_DEV_FLAGS=0x800 timeout 600 python tools/bench_configs.py "kafka n=5 rate100 20s lat5 + partitions" > $O/kafka_host.jsonl 2> $O/kafka_host.err; cut -c1-400 $O/kafka_host.jsonl
