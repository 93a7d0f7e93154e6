#!/bin/bash
# round 3, call b: GPU suite after the LDS txn check, the single-sweep set-full checker, lin-tso; timings + counters of both checkers
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=gpurun_out/r3b; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
MSIM_DEV_FLAGS=4096 timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" > $O/cfg5.jsonl 2> $O/cfg5.err; cut -c1-420 $O/cfg5.jsonl; grep "txn-check" $O/cfg5.err | tail -8
MSIM_DEV_FLAGS=12288 timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > $O/cfg5_hbm.jsonl 2> $O/cfg5_hbm.err; cut -c1-420 $O/cfg5_hbm.jsonl; grep "txn-check" $O/cfg5_hbm.err | tail -4
timeout 200 python bench.py --steps 10 --warmup 2 --cpu-sample 0 --no-gather --no-fetch > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; python -c "import json;d=json.load(open('$O/bench.json'));print(d['kernel_ms'], d['roofline']['secondary']['valu_issue_frac'] if 'secondary' in d['roofline'] else None)"
timeout 400 bash tools/profile_config.sh r3b/cfg5 "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > $O/cfg5_prof.log 2>&1; grep -v "compact" $O/cfg5_prof.log | tail -40
timeout 400 bash tools/profile_headline.sh r3b/headline full > $O/headline_prof.log 2>&1; grep "check_kernel" $O/headline_prof.log | tail -20
