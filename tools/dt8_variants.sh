#!/bin/bash
# Developer tool (GPU box): A/B builds of csrc/dt8.hip (tools/variant_lib.sh) — register budget x LDS queue slots — timed on cfg5 over the
# Datomic-style node at 16384 / 32768 clusters, each variant bit-compared with the oracle first (tools/emu_compare.py on the device).
# usage: tools/dt8_variants.sh <out.jsonl> [tag:flags ...]     e.g.  w4q3:"-DD8_WAVES_PER_EU=4 -DD8_RQ=3u -DD8_CQ=0u"
OUT=$1; shift
: > $OUT
for spec in "product:" "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  if [ "$tag" = product ]; then L=""; else tools/variant_lib.sh d8_$tag dt8.hip $flags > /dev/null 2>&1 || { echo "build of $tag failed"; continue; }; L=_d8_$tag; fi
  ok=$(MSIM_LIB=maelstrom_amd/libmaelsim$L.so python3 tools/emu_compare.py "{'workload':'txn-list-append','bin':'datomic','node_count':5,'rate':100,'time_limit':10,'latency':5,'nemesis':['partition'],'nemesis_interval':3,'n':24,'flags':0x400}" "{'workload':'txn-list-append','bin':'datomic','node_count':3,'rate':150,'time_limit':6,'latency':0,'key_count':16,'max_writes_per_key':2,'n':9,'flags':0x400}" 2>&1 | grep -c ": OK")
  for n in 16384 32768; do
    MSIM_LIB=maelstrom_amd/libmaelsim$L.so python3 - $n "$tag" "$flags" $ok >> $OUT <<'PY'
import sys, json
sys.path.insert(0, '.')
from maelstrom_amd import engine as E
n = int(sys.argv[1])
cfg = E.test_config("txn-list-append", bin="datomic", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
with E.Engine(cfg) as eng:
    eng.run(0, n); eng.run(n, n)
    ms = eng.kernel_ms()[0]
    eng.fetch()
    fl = sum(1 for i in range(0, n, 97) if eng.meta(i).flags)
print(json.dumps({"variant": sys.argv[2], "flags": sys.argv[3], "parity_cases_ok": int(sys.argv[4]), "clusters": n, "sim_ms": round(ms, 1), "flagged_sample": fl}), flush=True)
PY
  done
done
cat $OUT
