#!/bin/bash
# round 3, call ak: mk8_kernel's occupancy: LDS queue slots (RQ) / the transaction slot in LDS or HBM — cfg5 over the canonical node, sim ms per 32768 clusters
cd "$GRAFT_REPO_ROOT"; O=gpurun_out/r3ak; mkdir -p $O
for v in "" _rq6 _rq4 _rq8sl0 _rq4sl0; do
  L=maelstrom_amd/libmaelsim$v.so
  MSIM_LIB=$PWD/$L MSIM_DEV_FLAGS=0x1000 timeout 600 python tools/bench_configs.py "cfg5-mk txn-list-append multi-key n=5 rate100 30s lat5 + partitions" > $O/mk$v.jsonl 2> $O/mk$v.err
  echo "variant '$v': $(grep -o '"sim_ms": [0-9.]*' $O/mk$v.jsonl) $(grep -o 'mk8.*LDS per wavefront' $O/mk$v.err | tail -1)"
done
MSIM_LIB=$PWD/maelstrom_amd/libmaelsim_rq4sl0.so timeout 600 python -m pytest tests/test_parity_gpu.py -k "multi_key" -m gpu -q -x > $O/parity_rq4sl0.log 2>&1; tail -1 $O/parity_rq4sl0.log
