#!/bin/bash
# Runs on the GPU box (gpurun): rocprofv3 kernel trace + separate PMC passes for the headline bench, summarised by
# tools/rocpd_summary.py.  Usage: tools/profile_headline.sh <tag> [full]   -> gpurun_out/<tag>/{trace,pmc_*}, summary.txt, counters.json
set -u
TAG=${1:-prof}
ROOT=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$ROOT/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
CMD="python $ROOT/bench.py --steps 3 --warmup 3 --cpu-sample 0 --no-gather --no-fetch"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- $CMD > "$OUT/trace.log" 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_WAVES SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d "$OUT/pmc_sq" -o s -- $CMD > "$OUT/pmc_sq.log" 2>&1
if [ "${2:-}" = "full" ]; then
  rocprofv3 --pmc FETCH_SIZE -d "$OUT/pmc_fetch" -o f -- $CMD > "$OUT/pmc_fetch.log" 2>&1
  rocprofv3 --pmc WRITE_SIZE -d "$OUT/pmc_write" -o w -- $CMD > "$OUT/pmc_write.log" 2>&1
  rocprofv3 --pmc SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_SALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS -d "$OUT/pmc_cyc" -o c -- $CMD > "$OUT/pmc_cyc.log" 2>&1
fi
DBS=$(find "$OUT" -name "*_results.db" | sort)
python $ROOT/tools/rocpd_summary.py $DBS > "$OUT/summary.txt" 2>&1
python $ROOT/tools/rocpd_summary.py --counters "$OUT/counters.json" $DBS
if [ "${2:-}" = "full" ]; then python $ROOT/tools/rocpd_summary.py --traffic "$OUT/traffic.json" $DBS; fi
cp "$OUT"/trace/*kernel_stats.csv "$OUT/kernel_stats.csv" 2>/dev/null || find "$OUT/trace" -name "*stats*" | head
# the raw rocprofv3 databases are tens of MiB per pass and gpurun merges at most 64 MiB back: the summaries are what is kept (KEEP_RAW=1 keeps everything)
[ "${KEEP_RAW:-0}" = 1 ] || rm -rf "$OUT/trace" "$OUT"/pmc_*
tail -40 "$OUT/summary.txt"
