#!/bin/bash
# round 3, call m: cfg3 with the lone-operation path (timing with / without, cycle split), wide parity again
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3m; mkdir -p $O
timeout 600 python -m pytest tests/test_parity_gpu.py tests/test_fuzz_gpu.py tests/test_bench_shapes_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x -k "wide or g_set or gset or cfg3" --timeout 600 > $O/tests.log 2>&1; tail -2 $O/tests.log
timeout 300 python tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" "cfg3 g-set n=100 lat100 exponential p_loss 0.05" "cfg3 g-set n=100 lat100 exponential p_loss 0.5" 2>$O/cfg3.err | tee $O/cfg3.jsonl | cut -c1-330
MSIM_DEV_FLAGS=2 timeout 300 python tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" 2>>$O/cfg3.err | tee $O/cfg3_nolone.jsonl | cut -c1-330
MSIM_DEV_FLAGS=0x4000 timeout 300 python tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" 2>>$O/cfg3.err | tee $O/cfg3_hbm.jsonl | cut -c1-330
timeout 300 python tools/wide_prof_report.py > $O/wprof_cfg3_setl.txt 2>&1; cat $O/wprof_cfg3_setl.txt
