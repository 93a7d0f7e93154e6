#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace + separate PMC passes (SQ instruction counts, SQ cycles, FETCH_SIZE,
# WRITE_SIZE) for one tools/bench_configs.py config.
# Usage: tools/profile_config.sh <tag> "<config name>"   -> gpurun_out/<tag>/summary.txt, counters.json
set -u
TAG=$1; CFG=$2
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_BRANCH SQ_WAVES -d "$OUT/pmc_sq" -o s -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_sq.log" 2>&1
rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_ANY -d "$OUT/pmc_cyc" -o c -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_cyc.log" 2>&1
rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w -- python $ROOT/tools/bench_configs.py "$CFG" > "$OUT/pmc_write.log" 2>&1
DBS=$(find "$OUT" -name "*_results.db" | sort)
python $ROOT/tools/rocpd_summary.py $DBS > "$OUT/summary.txt" 2>&1
python $ROOT/tools/rocpd_summary.py --counters "$OUT/counters.json" $DBS
# the raw rocprofv3 databases are tens of MiB per pass and gpurun merges at most 64 MiB back: the summaries are what is kept (KEEP_RAW=1 keeps everything)
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf "$OUT/trace" "$OUT"/pmc_*
grep -v "compact\|__amd" "$OUT/summary.txt" | tail -60
