#!/bin/bash
# round 3, call at: bench.py's multi-rank path on the one-GPU box (two ranks on one device, gloo): the headline and the cfg4 leg with the history gather
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3at; mkdir -p $O
MSIM_BENCH_BACKEND=gloo MSIM_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 > $O/bench2.json 2> $O/bench2.err; cut -c1-400 $O/bench2.json; tail -2 $O/bench2.err
MSIM_BENCH_BACKEND=gloo MSIM_BENCH_ONE_DEVICE=1 timeout 600 python bench.py --gpus 2 --steps 2 --warmup 1 --config cfg4 > $O/bench2_cfg4.json 2> $O/bench2_cfg4.err; cut -c1-600 $O/bench2_cfg4.json; tail -2 $O/bench2_cfg4.err
