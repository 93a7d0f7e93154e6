cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f_svc4
: > gpurun_out/r06f_svc4/variants.jsonl
for v in "" ${VARIANTS}; do
  export MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$v.so
  par=$(python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k proxy 2>&1 | grep -E "passed|failed" | tail -1)
  for cfgname in "lin-kv proxy n=5 c=10 rate30 60s lat5"; do
    python3 tools/bench_configs.py "$cfgname" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.readline())
print(json.dumps({'variant': '$v' or 'product', 'parity': '''$par''', 'instances': d['instances'], 'sim_ms': round(d['sim_ms'], 1), 'check_ms': round(d['check_ms'], 1), 'valid': d['valid'], 'flagged': d['flagged']}))" >> gpurun_out/r06f_svc4/variants.jsonl
  done
done
cat gpurun_out/r06f_svc4/variants.jsonl
