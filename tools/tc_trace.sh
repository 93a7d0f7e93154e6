#!/bin/bash
# Developer tool (GPU box): the list-append device pass of cfg5 under rocprofv3's kernel trace, product build against variants of txn_check_dev.hip
# (tools/variant_lib.sh <tag> txn_check_dev.hip <flags>), and the check's wall time by threads per history.
# usage: tools/tc_trace.sh [variant-tag ...]        ("" = the product build is always run first)
R=${GRAFT_REPO_ROOT:-/root/repo}
CFG="cfg5 txn-list-append n=5 rate100 30s lat5 + partitions"
cd /tmp; export TMPDIR=/tmp
for v in "" "$@"; do
  L=$R/maelstrom_amd/libmaelsim${v:+_$v}.so
  rm -rf /tmp/tcprof_$v
  MSIM_LIB=$L timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/tcprof_$v -o t -- python3 $R/tools/bench_configs.py "$CFG" > /tmp/tc_$v.log 2>&1 < /dev/null
  echo "== variant [$v]: $(grep -o '"check_ms": [0-9.]*' /tmp/tc_$v.log)"
  timeout 120 python3 $R/tools/rocpd_summary.py $(find /tmp/tcprof_$v -name "*_results.db" | sort) < /dev/null 2>&1 | grep -i "txn_check\|kernel\b" | head -6 | cut -c1-220
done
cd $R
for wg in 1024 512 256; do echo "== threads per history $wg: $(MSIM_TXN_WG=$wg timeout 300 python3 tools/bench_configs.py "$CFG" 2>/dev/null < /dev/null | grep -o '"check_ms": [0-9.]*')"; done
