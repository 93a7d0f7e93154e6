#!/bin/bash
# round 3, call aj: the kafka checker's tables sized by what the histories name; the unique-ids table in LDS: tests, the two bench configs
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3aj; mkdir -p $O
timeout 600 python -m pytest tests/test_kafka_check_gpu.py tests/test_kafka_gpu.py tests/test_checker_gpu.py -m gpu -q -x --timeout 500 > $O/tests.log 2>&1; tail -2 $O/tests.log
MSIM_DEV_FLAGS=0x1000 timeout 600 python tools/bench_configs.py "kafka n=5 rate100 20s lat5 + partitions" "unique-ids n=3 rate1000 10s lat5 + partitions" > $O/cfg.jsonl 2> $O/cfg.err; cut -c1-400 $O/cfg.jsonl; grep "kafka-check" $O/cfg.err | tail -2
MSIM_DEV_FLAGS=0x2000 timeout 600 python tools/bench_configs.py "unique-ids n=3 rate1000 10s lat5 + partitions" > $O/uniq_hbm.jsonl 2> $O/uniq_hbm.err; cut -c1-400 $O/uniq_hbm.jsonl
