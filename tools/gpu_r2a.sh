#!/bin/bash
# round-2 GPU call A: bring-up of the two-clusters-per-wavefront kernel
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r2a; mkdir -p $O
export MSIM_DEV_FLAGS=256
timeout 600 python tools/duo_debug.py > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-gather > $O/bench_duo.json 2> $O/bench_duo.err; echo "rc=$?" >> $O/bench_duo.err
MSIM_DEV_FLAGS=768 timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-gather > $O/bench_old.json 2> $O/bench_old.err; echo "rc=$?" >> $O/bench_old.err
timeout 300 python bench.py --steps 5 --warmup 2 --cpu-sample 0 --no-gather --instances 8192 > $O/bench_duo_8192.json 2> $O/bench_duo_8192.err
timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --no-gather --instances 16384 > $O/bench_duo_16384.json 2> $O/bench_duo_16384.err
timeout 1500 python -m pytest tests -m gpu -q -x --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -5 $O/debug.log; cat $O/bench_duo.json | head -c 600; echo; cat $O/bench_old.json | head -c 300; echo; tail -5 $O/pytest.log
