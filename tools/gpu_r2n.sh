#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2n}; mkdir -p $O
MSIM_DEV_FLAGS=1280 timeout 600 python tools/duo_debug.py raft raft-lat10 raft-exp-loss raft-part raft-n3c6 raft-n1 raft-n4c8 > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
cat $O/debug.log | head -60
timeout 900 python -m pytest tests/test_bench_shapes_gpu.py tests/test_parity_gpu.py tests/test_fuzz_gpu.py -m gpu -q -x --timeout 600 -k "raft or fuzz" > $O/pytest.log 2>&1; tail -5 $O/pytest.log
timeout 300 python tools/raft4_prof_report.py > $O/prof.txt 2>&1; PART=1 timeout 300 python tools/raft4_prof_report.py >> $O/prof.txt 2>&1; cat $O/prof.txt
timeout 600 python tools/bench_configs.py "cfg4 lin-kv raft n=5 c=10 rate30 60s" "cfg4 lin-kv raft + partitions lat10" > $O/cfg4.log 2>&1; cat $O/cfg4.log
