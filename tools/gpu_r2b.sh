#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2b; mkdir -p $O
timeout 200 python tools/duo_prof_report.py > $O/prof_lat0.txt 2>&1
LAT=10 timeout 200 python tools/duo_prof_report.py > $O/prof_lat10.txt 2>&1
LAT=100 timeout 200 python tools/duo_prof_report.py > $O/prof_lat100.txt 2>&1
N=8192 timeout 200 python tools/duo_prof_report.py > $O/prof_lat0_8192.txt 2>&1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 3 --warmup 2 --cpu-sample 0 --no-gather"
timeout 300 rocprofv3 --kernel-trace --stats -d $R/$O/trace -o t -- $CMD > $R/$O/trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_SMEM -d $R/$O/pmc_sq -o s -- $CMD > $R/$O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d $R/$O/pmc_cyc -o c -- $CMD > $R/$O/pmc_cyc.log 2>&1
cd $R
python tools/rocpd_summary.py $(find $O -name "*_results.db" | sort) 2>&1 | grep -v "at::native\|rocclr" > $O/summary.txt
cat $O/prof_lat0.txt $O/prof_lat10.txt $O/prof_lat100.txt $O/prof_lat0_8192.txt; grep "sim_kernel_duo\|check_kernel" $O/summary.txt | head -40
