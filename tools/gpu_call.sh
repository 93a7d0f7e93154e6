#!/bin/bash
# One gpurun call = one experiment directory under gpurun_out/: `gpurun -- tools/gpu_call.sh <tag> <step> [<step> ...]`.
# Steps: tests (pytest -m gpu), bench (default bench.py line), configs (tools/bench_configs.py, all or the names in CONFIGS),
#        probe (tools/latency_probe.py over PROBE), prof-headline / prof-config (tools/profile_*.sh), or any shell command in quotes.
cd "$GRAFT_REPO_ROOT"
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for step in "$@"; do
  case "$step" in
    tests) timeout 1700 python -m pytest tests -m gpu -q -x --timeout 300 > $O/pytest.log 2>&1; tail -3 $O/pytest.log ;;
    bench) timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 1500 $O/bench.json ;;
    configs) timeout 1500 python tools/bench_configs.py ${CONFIGS:+"$CONFIGS"} > $O/configs.jsonl 2> $O/configs.err; cat $O/configs.jsonl | cut -c1-330 ;;
    probe) timeout 900 python tools/latency_probe.py $PROBE > $O/probe.jsonl 2> $O/probe.err; cat $O/probe.jsonl ;;
    *) echo "== $step"; timeout 1500 bash -c "$step" ;;
  esac
done
