#!/bin/bash
# round 3, call q: the committed profiles of the round's final kernels: headline (kernel trace + SQ / FETCH / WRITE passes), cfg3, cfg5; full GPU suite; bench
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > gpurun_out/r3q_suite.log 2>&1; tail -2 gpurun_out/r3q_suite.log
bash tools/profile_headline.sh r3q_head full > gpurun_out/r3q_head.log 2>&1; tail -12 gpurun_out/r3q_head.log
bash tools/profile_config.sh r3q_cfg3 "cfg3 g-set n=100 lat100 exponential" > gpurun_out/r3q_cfg3.log 2>&1; tail -12 gpurun_out/r3q_cfg3.log
cd $R; timeout 300 python bench.py --steps 20 --warmup 3 > gpurun_out/r3q_bench.json 2> gpurun_out/r3q_bench.err; cut -c1-400 gpurun_out/r3q_bench.json
for n in 8192 16384; do timeout 300 python bench.py --steps 5 --warmup 2 --instances $n --cpu-sample 0 --no-gather --no-fetch >> gpurun_out/r3q_batch_sweep.jsonl 2>> gpurun_out/r3q_bench.err; done; cut -c1-260 gpurun_out/r3q_batch_sweep.jsonl
timeout 600 python tools/bench_configs.py > gpurun_out/r3q_other_configs.jsonl 2> gpurun_out/r3q_other.err; cut -c1-300 gpurun_out/r3q_other_configs.jsonl
