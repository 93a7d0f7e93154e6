#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final; mkdir -p $O
timeout 420 python -m pytest tests -m gpu -q -x --timeout 200 > $O/pytest.log 2>&1; tail -3 $O/pytest.log
timeout 100 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 240 python bench.py > $O/bench.json 2> $O/bench.err; python -c "import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['value_incl_fetch'], d['incl_fetch']['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'])"
