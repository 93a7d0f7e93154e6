#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-bench_dryrun}; mkdir -p $O
timeout 600 python bench.py --cpu-sample 2 > $O/bench1.json 2> $O/bench1.err; echo "rc=$?"; cut -c1-400 $O/bench1.json; python -c "
import json; d=json.load(open('$O/bench1.json')); print(d['value'], d['value_incl_fetch'], d['history_gather'], d.get('history_gather_error'), d.get('incl_fetch_error'))"
MSIM_BENCH_BACKEND=gloo MSIM_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --instances 1024 > $O/bench2.json 2> $O/bench2.err; echo "rc=$?"; python -c "
import json; d=json.load(open('$O/bench2.json')); print(d['n_gpus'], d['value'], d['value_incl_fetch'], d['history_gather'], d.get('history_gather_error'), d.get('incl_fetch_error'))"; tail -3 $O/bench2.err
