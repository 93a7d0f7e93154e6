#!/bin/bash
# round 3, call as: the round's last code: full GPU suite, the differential sweeps (300 option sets x 9 instances), smoke(), the default bench line, every bench config
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3as; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > $O/suite.log 2>&1; grep -n "passed\|failed\|error" $O/suite.log | tail -3
MSIM_FUZZ_CASES=300 MSIM_FUZZ_INSTANCES=9 timeout 1500 python -m pytest tests/test_fuzz_gpu.py -m gpu -q -x --timeout 1200 > $O/fuzz.log 2>&1; tail -1 $O/fuzz.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json; tail -1 $O/bench.err
timeout 900 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other.err; python - <<'P'
import json
for l in open("gpurun_out/r3as/other_configs.jsonl"):
    d = json.loads(l); print(f"{d['config']:72s} {d['instances']:6d}  sim {d['sim_ms']:9.2f} ms  check {d['check_ms']:8.2f} ms  valid {d['valid']}  host {d['check_host_rechecks']}")
P
