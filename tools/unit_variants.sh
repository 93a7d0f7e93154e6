#!/bin/bash
# Developer tool (GPU box): A/B builds of ONE translation unit (tools/variant_lib.sh), each held to its parity tests first (pytest -k <expr> with
# MSIM_LIB=<variant>) and then timed on a tools/bench_configs.py configuration.
# usage: tools/unit_variants.sh <out.jsonl> <unit.hip> "<bench_configs name>" "<pytest -k expression>" [tag:"flags" ...]
OUT=$1; UNIT=$2; CFG=$3; KEXPR=$4; shift 4
: > $OUT
for spec in "product:" "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  if [ "$tag" = product ]; then L=""; else tools/variant_lib.sh v_$tag $UNIT $flags > /dev/null 2>&1 || { echo "build of $tag failed"; continue; }; L=_v_$tag; fi
  par=$(MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$L.so timeout 900 python3 -m pytest tests/test_parity_gpu.py tests/test_bench_shapes_gpu.py -m gpu -q -x -p no:cacheprovider -k "$KEXPR" 2>&1 | grep -E "passed|failed" | tail -1)
  MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$L.so python3 tools/bench_configs.py "$CFG" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(json.dumps({'variant': '$tag', 'flags': '''$flags''', 'parity': '''$par''', 'instances': d['instances'], 'sim_ms': round(d['sim_ms'], 1), 'check_ms': round(d['check_ms'], 1), 'valid': d['valid'], 'flagged': d['flagged']}))" >> $OUT
done
cat $OUT
