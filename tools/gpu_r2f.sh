#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2f}; mkdir -p $O
timeout 900 python -m pytest tests/test_gather_gpu.py tests/test_abi_c_smoke_gpu.py tests/test_checker_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q --timeout 600 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
MSIM_BENCH_ONE_DEVICE=1 MSIM_BENCH_BACKEND=gloo timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench2.json 2> $O/bench2.err; echo "rc=$?" >> $O/bench2.err
tail -15 $O/pytest.log; tail -3 $O/bench.err; python3 - <<PY
import json
for f in ("$O/bench.json", "$O/bench2.json"):
    try:
        d = json.load(open(f))
        print(f, {k: d[k] for k in ("value", "ms_per_step", "kernel_ms", "value_incl_fetch", "history_gather", "checker_parity") if k in d})
        print("  incl_fetch", d.get("incl_fetch"))
        print("  cpu", d.get("cpu_baseline"))
    except Exception as e:
        print(f, "unreadable", e)
PY
tail -5 $O/bench2.err
