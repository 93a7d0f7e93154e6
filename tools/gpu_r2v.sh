#!/bin/bash
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${1:-r2v}; mkdir -p $O
MSIM_DEV_FLAGS=256 timeout 600 python tools/duo_debug.py txn txn-lat5 txn-part txn-exp-loss txn-n3 txn-n7 txn-n1 txn-len6 > $O/debug.log 2>&1; echo "debug rc=$?" >> $O/debug.log
grep -c identical $O/debug.log; grep -v "identical" $O/debug.log | head -30
for L in "" _t8w3 _t8w4; do echo "lib$L"; MSIM_LIB=$PWD/maelstrom_amd/libmaelsim$L.so timeout 300 python tools/bench_configs.py "cfg5 txn-list-append n=5 rate100 30s lat5 + partitions" 2>&1 | tail -1; done
