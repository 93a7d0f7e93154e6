#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2k}; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/bench_configs.py cfg2_exp"
sed -i 's/^CONFIGS = {/CONFIGS = {\n    "cfg2_exp": (dict(workload="broadcast", node_count=25, rate=100, time_limit=20, latency=100, latency_dist="exponential"), 4096),/' $R/tools/bench_configs.py
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $R/$O/pmc_sq -o s -- $CMD > $R/$O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d $R/$O/pmc_cyc -o c -- $CMD > $R/$O/pmc_cyc.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O -name "*_results.db" | sort) 2>&1 | grep "sim_kernel_duo" > $O/summary.txt
cat $O/summary.txt
