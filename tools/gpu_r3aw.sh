#!/bin/bash
# round 3, call aw: rocprofv3 on the unique-ids bench shape (uid8_kernel + unique_check_lds_kernel): kernel trace + the two SQ counter passes
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r3aw; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
CFG="unique-ids n=3 rate1000 10s lat5 + partitions"
timeout 60 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/trace.log" 2>&1
timeout 60 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES -d "$OUT/pmc_sq" -o s -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_sq.log" 2>&1
timeout 60 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY -d "$OUT/pmc_cyc" -o c -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_cyc.log" 2>&1
DBS=$(find "$OUT" -name "*_results.db" | sort)
python $ROOT/tools/rocpd_summary.py $DBS > "$OUT/summary.txt" 2>&1
python $ROOT/tools/rocpd_summary.py --counters "$OUT/counters.json" $DBS > /dev/null 2>&1
grep -v "compact\|__amd" "$OUT/summary.txt" | tail -30 | cut -c1-200
