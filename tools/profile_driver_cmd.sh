set -u
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out/r05e_driver_cmd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o t -- python $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 > "$OUT/bench.json" 2> "$OUT/trace.log"
DBS=$(find "$OUT" -name "*_results.db" | sort)
echo "dbs: $DBS" | head -3
python $ROOT/tools/rocpd_summary.py $DBS > "$OUT/summary.txt" 2>&1
rm -rf "$OUT/trace"
head -12 "$OUT/summary.txt"; cut -c1-200 "$OUT/bench.json"
