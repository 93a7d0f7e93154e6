#!/bin/bash
# round 3, call a: the whole GPU suite (mk_kernel's first run on hardware), cfg5 over both nodes, the profiles round 2 lacked
# (random-latency duo, wide cfg3, wide broadcast), the headline batch sweep
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/bench_configs.py "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > $O/cfg5.jsonl 2> $O/cfg5.err; cut -c1-400 $O/cfg5.jsonl
for b in 4096 8192 16384; do timeout 200 python bench.py --instances $b --steps 5 --warmup 2 --cpu-sample 0 --no-gather --no-fetch > $O/bench_$b.json 2> $O/bench_$b.err; cut -c1-300 $O/bench_$b.json; done
timeout 400 bash tools/profile_config.sh r3a/duo_exp100 "cfg2 broadcast n=25 grid lat100 exponential" > $O/duo_exp100.log 2>&1; tail -30 $O/duo_exp100.log
timeout 400 bash tools/profile_config.sh r3a/wide_cfg3 "cfg3 g-set n=100 lat100 exponential" > $O/wide_cfg3.log 2>&1; tail -30 $O/wide_cfg3.log
timeout 500 bash tools/profile_config.sh r3a/wide_bcast100_exp "broadcast n=100 grid lat100 exponential" > $O/wide_bcast_exp.log 2>&1; tail -30 $O/wide_bcast_exp.log
timeout 400 bash tools/profile_config.sh r3a/cfg5_mk "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" > $O/cfg5_mk.log 2>&1; tail -30 $O/cfg5_mk.log
