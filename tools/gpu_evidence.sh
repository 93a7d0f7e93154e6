#!/bin/bash
# round-2 evidence run: headline profile (trace + PMC passes), cfg4 / cfg5 profiles, every other config, the default bench line
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT
timeout 600 bash tools/profile_headline.sh r02_headline full > gpurun_out/r02_headline.log 2>&1
timeout 600 bash tools/profile_config.sh r02_cfg4 "cfg4 lin-kv raft n=5 c=10 rate30 60s" > gpurun_out/r02_cfg4.log 2>&1
timeout 600 bash tools/profile_config.sh r02_cfg5 "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > gpurun_out/r02_cfg5.log 2>&1
cd $R
timeout 900 python tools/bench_configs.py > gpurun_out/r02_other_configs.jsonl 2> gpurun_out/r02_other_configs.err
timeout 600 python bench.py > gpurun_out/r02_bench.json 2> gpurun_out/r02_bench.err
tail -3 gpurun_out/r02_headline.log; cut -c1-330 gpurun_out/r02_other_configs.jsonl; cat gpurun_out/r02_bench.json | cut -c1-1500
