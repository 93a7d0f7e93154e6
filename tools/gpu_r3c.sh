#!/bin/bash
# round 3, call c2: wide kernel with run + suffix queues held to two wavefronts per SIMD: whole GPU suite, wide timings
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3c2; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -3
timeout 400 python tools/bench_configs.py "cfg3 g-set n=100 lat100 exponential" "cfg3 g-set n=100 lat100 exponential p_loss 0.05" "cfg3 g-set n=100 lat100 exponential p_loss 0.5" "broadcast n=100 grid lat100 exponential" "broadcast n=100 grid lat0" > $O/wide.jsonl 2> $O/wide.err; cut -c1-330 $O/wide.jsonl
