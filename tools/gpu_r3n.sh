#!/bin/bash
# round 3, call n: cfg3 after the register diet / lossy lone operations / fences (timing, non-temporal queue traffic A/B, cycle split); wide broadcast sanity
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3n; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "wide or g_set or gset or cfg3 or counter" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
C3="cfg3 g-set n=100 lat100 exponential"
timeout 300 python tools/bench_configs.py "$C3" "$C3 p_loss 0.05" "$C3 p_loss 0.5" 2>$O/cfg3.err | tee $O/cfg3.jsonl | cut -c1-330
echo "-- non-temporal queue traffic"; MSIM_LIB=maelstrom_amd/libmaelsim_wnt.so timeout 300 python tools/bench_configs.py "$C3" "$C3 p_loss 0.05" 2>>$O/cfg3.err | tee $O/cfg3_nt.jsonl | cut -c1-330
echo "-- again, default"; timeout 300 python tools/bench_configs.py "$C3" 2>>$O/cfg3.err | tee -a $O/cfg3.jsonl | cut -c1-330
echo "-- lone operations off"; MSIM_DEV_FLAGS=2 timeout 300 python tools/bench_configs.py "$C3" 2>>$O/cfg3.err | tee $O/cfg3_nolone.jsonl | cut -c1-330
timeout 300 python tools/wide_prof_report.py > $O/wprof_cfg3.txt 2>&1; cat $O/wprof_cfg3.txt
timeout 300 python tools/bench_configs.py "broadcast n=100 grid lat0" "broadcast n=100 grid lat100 exponential" "pn-counter n=5 rate100 20s lat100 exponential" 2>>$O/cfg3.err | tee $O/wide_other.jsonl | cut -c1-330
