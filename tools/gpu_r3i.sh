#!/bin/bash
# round 3, call i: kafka: engine check test, timing; bench after all changes
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3i; mkdir -p $O
timeout 300 python -m pytest tests/test_kafka_gpu.py -m gpu -q -x --timeout 300 > $O/kafka.log 2>&1; tail -3 $O/kafka.log
timeout 300 python tools/bench_configs.py "kafka n=5 rate100 20s lat5 + partitions" > $O/kafka.jsonl 2> $O/kafka.err; cut -c1-420 $O/kafka.jsonl; tail -3 $O/kafka.err
timeout 300 python bench.py --steps 10 --warmup 2 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r3i/bench.json")); print({k:d[k] for k in ("value","ms_per_step","histories_per_sec","kernel_ms")}); print(d["roofline"]["frac"], d.get("cpu_baseline"))
PY
