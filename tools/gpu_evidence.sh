#!/bin/bash
# A round's evidence in one gpurun call: `gpurun -- tools/gpu_evidence.sh r04z` -> gpurun_out/<tag>_*: the headline profile (kernel trace +
# PMC passes), the profiles of cfg3 / cfg4 / cfg5, every other config, the full GPU suite and the default bench line.  Copy what is to be
# judged into profiles/.
cd "$GRAFT_REPO_ROOT"
R=$GRAFT_REPO_ROOT; TAG=${1:-evidence}
timeout 600 bash tools/profile_headline.sh ${TAG}_headline full > gpurun_out/${TAG}_headline.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg3 "cfg3 g-set n=100 lat100 exponential" > gpurun_out/${TAG}_cfg3.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg3loss "cfg3 g-set n=100 lat100 exponential p_loss 0.05" > gpurun_out/${TAG}_cfg3loss.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg4 "cfg4 lin-kv raft n=5 c=10 rate30 60s" > gpurun_out/${TAG}_cfg4.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg5 "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" > gpurun_out/${TAG}_cfg5.log 2>&1
timeout 600 bash tools/profile_config.sh ${TAG}_cfg5dt "cfg5-datomic txn-list-append datomic n=5 rate100 30s lat5 + partitions" > gpurun_out/${TAG}_cfg5dt.log 2>&1
cd $R
timeout 900 python tools/bench_configs.py > gpurun_out/${TAG}_other_configs.jsonl 2> gpurun_out/${TAG}_other_configs.err
timeout 1700 python -m pytest tests -m gpu -q --timeout 300 > gpurun_out/${TAG}_gpu_suite.txt 2>&1
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench.err
# the driver's own command under rocprofv3's kernel trace (no counters: launches are not serialised, durations are what bench.py's HIP events see)
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_driver_cmd/trace -o t -- python $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > $R/gpurun_out/${TAG}_driver_cmd_bench.json 2> $R/gpurun_out/${TAG}_driver_cmd.log;
  python $R/tools/rocpd_summary.py $(find $R/gpurun_out/${TAG}_driver_cmd -name "*_results.db" | sort) > $R/gpurun_out/${TAG}_driver_cmd_summary.txt 2>&1; rm -rf $R/gpurun_out/${TAG}_driver_cmd )
# the N > 1 path of bench.py on this one-GPU box: eight ranks on device 0, gloo instead of RCCL (a dry run of the code path, not a scaling number)
MSIM_BENCH_ONE_DEVICE=1 MSIM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --config cfg4 --steps 2 --warmup 1 > gpurun_out/${TAG}_dryrun_gpus8_cfg4.json 2> gpurun_out/${TAG}_dryrun_gpus8_cfg4.err; echo "dry run --gpus 8 cfg4 rc=$?"
MSIM_BENCH_ONE_DEVICE=1 MSIM_BENCH_BACKEND=gloo timeout 900 python bench.py --gpus 8 --steps 3 --warmup 1 --instances 1024 > gpurun_out/${TAG}_dryrun_gpus8_cfg2.json 2> gpurun_out/${TAG}_dryrun_gpus8_cfg2.err; echo "dry run --gpus 8 cfg2 rc=$?"
tail -3 gpurun_out/${TAG}_headline.log; cut -c1-330 gpurun_out/${TAG}_other_configs.jsonl; tail -3 gpurun_out/${TAG}_gpu_suite.txt; cut -c1-1500 gpurun_out/${TAG}_bench_default.json; cut -c1-600 gpurun_out/${TAG}_dryrun_gpus8_cfg4.json; head -8 gpurun_out/${TAG}_driver_cmd_summary.txt
