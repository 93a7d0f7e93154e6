#!/bin/bash
# round 3, call ai: duo RND with the messages' latency draws taken 32 at a time: parity, the latency sweep of cfg2
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ai; mkdir -p $O
timeout 900 python -m pytest tests/test_parity_gpu.py tests/test_bench_shapes_gpu.py tests/test_edge_cases_gpu.py -m gpu -q -x --timeout 800 > $O/tests.log 2>&1; tail -2 $O/tests.log
MSIM_FUZZ_CASES=120 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 800 -k "test_random_options" > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
timeout 600 python tools/bench_configs.py "cfg2 broadcast n=25 grid lat0" "cfg2 broadcast n=25 grid lat10" "cfg2 broadcast n=25 grid lat100" "cfg2 broadcast n=25 grid lat100 exponential" "cfg2 broadcast n=25 total lat100" > $O/cfg2.jsonl 2> $O/cfg2.err
cut -c1-330 $O/cfg2.jsonl; tail -2 $O/cfg2.err
