#!/bin/bash
# round 3, call ax: rocprofv3 kernel trace + SQ instruction / cycle counters on three more shapes of the last stretch (crdt8, hat8 + rw check, kafka + its checker)
ROOT=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
i=0
for CFG in "pn-counter n=5 rate100 20s lat100 exponential" "txn-rw-register hat n=2 rate100 30s + partitions" "kafka n=5 rate100 20s lat5 + partitions"; do
  i=$((i+1)); OUT=$ROOT/gpurun_out/r3ax/c$i; mkdir -p $OUT
  timeout 50 rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/trace.log" 2>&1
  timeout 50 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY -d "$OUT/pmc_sq" -o s -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_sq.log" 2>&1
  DBS=$(find "$OUT" -name "*_results.db" | sort)
  { echo "### $CFG"; python $ROOT/tools/rocpd_summary.py $DBS 2>&1 | grep -v "compact\|__amd"; } > "$OUT/summary.txt"
  find "$OUT" -name "*_results.db" -delete
done
cat $ROOT/gpurun_out/r3ax/c*/summary.txt | grep -v "^==" | cut -c1-170 | grep "ms\|kernel\|###" | head -40
