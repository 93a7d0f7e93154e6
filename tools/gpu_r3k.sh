#!/bin/bash
# round 3, call k: full GPU suite after the duo trims / multi-wavefront txn check / SETL; A/B timings
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3k; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/gpu_suite.log 2>&1; tail -4 $O/gpu_suite.log
timeout 300 python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err; python - <<'PY'
import json; d=json.load(open("gpurun_out/r3k/bench.json")); print({k:d[k] for k in ("value","ms_per_step","histories_per_sec","kernel_ms")}); print(d["roofline"]["frac"])
PY
for c in "cfg3 g-set n=100 lat100 exponential" "broadcast n=100 grid lat0" "broadcast n=100 grid lat100 exponential"; do
  timeout 300 python tools/bench_configs.py "$c" >> $O/wide_setl.jsonl 2>> $O/wide.err
  MSIM_DEV_FLAGS=16384 timeout 300 python tools/bench_configs.py "$c" >> $O/wide_hbm.jsonl 2>> $O/wide.err
done
echo "-- SETL"; cut -c1-330 $O/wide_setl.jsonl; echo "-- HBM sets"; cut -c1-330 $O/wide_hbm.jsonl; tail -3 $O/wide.err
C5="cfg5 txn-list-append n=5 rate100 30s lat5 + partitions"
for wg in 512 256 128 64; do echo "-- txn check, $wg threads per history"; MSIM_TXN_WG=$wg MSIM_DEV_FLAGS=4096 timeout 300 python tools/bench_configs.py "$C5" 2> $O/cfg5_wg$wg.err | tee -a $O/cfg5_wg.jsonl | cut -c1-330; grep "txn-check" $O/cfg5_wg$wg.err | tail -2; done
echo "-- txn check, HBM tables"; MSIM_DEV_FLAGS=12288 timeout 300 python tools/bench_configs.py "$C5" 2> $O/cfg5_hbm.err | tee $O/cfg5_hbm.jsonl | cut -c1-330; grep "txn-check" $O/cfg5_hbm.err | tail -2
timeout 300 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat100 exponential" "cfg2 broadcast n=25 grid lat10" | tee $O/cfg2_lat.jsonl | cut -c1-330
