#!/bin/bash
# round 3, call p: instruction-cache counters of the wide kernel (cfg3) and of the headline kernel
cd "$GRAFT_REPO_ROOT"; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r3p; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "icache\|IFETCH\|SQ_INST_LEVEL\|SQC_" | head -40 > $O/counters_avail.txt; head -30 $O/counters_avail.txt
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $O/ic_cfg3 -o c -- python $R/tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" > $O/ic_cfg3.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU -d $O/sq_cfg3 -o s -- python $R/tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" > $O/sq_cfg3.log 2>&1
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $O/ic_head -o c -- python $R/bench.py --steps 3 --warmup 2 --cpu-sample 0 --no-gather --no-fetch > $O/ic_head.log 2>&1
python $R/tools/rocpd_summary.py $(find $O -name "*_results.db" | sort) > $O/summary.txt 2>&1; grep -v "compact\|__amd\|check_kernel" $O/summary.txt | tail -40
tail -3 $O/ic_cfg3.log
