cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r06f_guard
for mode in 1 2; do
  echo "== MSIM_GUARD=$mode: the whole GPU suite"
  MSIM_GUARD=$mode timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|msim guard|fault" | tail -4
  echo "== MSIM_GUARD=$mode: the proxy / lin-tso bench shapes (svc4_kernel) and the option sweep of the key-value programs at 200 cases"
  MSIM_GUARD=$mode python tools/bench_configs.py "lin-kv proxy n=5 c=10 rate30 60s lat5" "unique-ids over lin-tso n=3 rate1000 10s lat5 + partitions" 2>&1 | cut -c1-200
  MSIM_GUARD=$mode MSIM_FUZZ_KIND=proxy MSIM_FUZZ_CASES=200 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 600 -n 8 -k random_kv -p no:cacheprovider 2>&1 | grep -E "passed|failed|fault" | tail -2
done
