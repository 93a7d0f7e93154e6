#!/bin/bash
# round 3, call ah: after the kafka checker's device pass, hat8 and the kafka lookup: full GPU suite, smoke(), the default bench line, every bench config
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ah; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q -x --timeout 900 > $O/suite.log 2>&1; grep -n "passed\|failed\|error" $O/suite.log | tail -3
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 400 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-330 $O/bench.json; tail -2 $O/bench.err
timeout 900 python tools/bench_configs.py > $O/other_configs.jsonl 2> $O/other.err; cut -c1-290 $O/other_configs.jsonl
