#!/bin/bash
# Developer tool (GPU box): the whole GPU suite and the bench shapes / option sweeps of the round's last packed kernels (svc4.hip, txng4.hip, dtg4.hip) under
# fenced device slabs (csrc/guard.cpp), both directions.
cd $GRAFT_REPO_ROOT
for mode in 1 2; do
  echo "== MSIM_GUARD=$mode: the whole GPU suite"
  MSIM_GUARD=$mode timeout 1500 python -m pytest tests -m gpu -q --timeout 600 -p no:cacheprovider 2>&1 | grep -E "passed|failed|msim guard|fault" | tail -4
  echo "== MSIM_GUARD=$mode: the proxy / lin-tso / several-workers txn bench shapes (svc4_kernel, txng4_kernel, dtg4_kernel) and the option sweeps of the key-value programs at 200 cases"
  MSIM_GUARD=$mode python tools/bench_configs.py "lin-kv proxy n=5 c=10 rate30 60s lat5" "unique-ids over lin-tso n=3 rate1000 10s lat5 + partitions" "txn-list-append n=1 c=10 rate100 30s lat5 (single-root node)" "txn-list-append n=5 c=10 rate100 30s lat5 + partitions (single-root node)" "txn-list-append datomic n=1 c=10 rate100 30s lat0" 2>&1 | cut -c1-260
  for kind in proxy txn dt; do MSIM_GUARD=$mode MSIM_FUZZ_KIND=$kind MSIM_FUZZ_CASES=200 timeout 900 python -m pytest tests/test_fuzz_gpu.py -m gpu -q --timeout 600 -n 8 -k random_kv -p no:cacheprovider 2>&1 | grep -E "passed|failed|fault" | tail -2; done
done
