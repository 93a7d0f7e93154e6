cd $GRAFT_REPO_ROOT
for v in "" _v_w2 _v_w3; do
  export MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$v.so
  par=$(python3 -m pytest tests/test_parity_gpu.py -m gpu -q -x -p no:cacheprovider -k datomic_many 2>&1 | grep -E "passed|failed" | tail -1)
  echo "variant ${v:-product(w4)}: $par"
  python3 tools/bench_configs.py "txn-list-append datomic n=1 c=10 rate100 30s lat5" 2>/dev/null | cut -c1-330
done
MSIM_DEV_FLAGS=0x200 python3 tools/bench_configs.py "txn-list-append datomic n=1 c=10 rate100 30s lat5" 2>/dev/null | cut -c1-330
MSIM_FUZZ_KIND=dt MSIM_FUZZ_CASES=200 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 600 -n 8 -k "random_kv" 2>&1 | tail -2
