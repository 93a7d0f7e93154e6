#!/bin/bash
# A round's evidence in one gpurun call: `gpurun -- tools/gpu_evidence.sh r04z` -> gpurun_out/<tag>_*: the headline profile (kernel trace +
# PMC passes), the profiles of cfg3 / cfg4 / cfg5, every other config, the full GPU suite and the default bench line.  Copy what is to be
# judged into profiles/.
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT; TAG=${1:-evidence}
timeout 600 bash tools/profile_headline.sh ${TAG}_headline full > gpurun_out/${TAG}_headline.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg3 "cfg3 g-set n=100 lat100 exponential" > gpurun_out/${TAG}_cfg3.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg4 "cfg4 lin-kv raft n=5 c=10 rate30 60s" > gpurun_out/${TAG}_cfg4.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg5 "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > gpurun_out/${TAG}_cfg5.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg5dt "cfg5-datomic txn-list-append datomic n=5 rate100 30s lat5 + partitions" > gpurun_out/${TAG}_cfg5dt.log 2>&1
cd $R
timeout 900 python tools/bench_configs.py > gpurun_out/${TAG}_other_configs.jsonl 2> gpurun_out/${TAG}_other_configs.err
timeout 1700 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${TAG}_gpu_suite.txt 2>&1
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err
tail -3 gpurun_out/${TAG}_headline.log; cut -c1-330 gpurun_out/${TAG}_other_configs.jsonl; tail -3 gpurun_out/${TAG}_gpu_suite.txt; cut -c1-1500 gpurun_out/${TAG}_bench_default.json
