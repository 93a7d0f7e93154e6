for L in "" _d8w3 _d8w4 _d8a _d8b _d8c _d8d; do
  for n in 16384 32768; do
    MSIM_LIB=maelstrom_amd/libmaelsim$L.so python3 - $n "$L" <<'PY'
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
print(json.dumps({"lib": sys.argv[2] or "product", "clusters": n, "sim_ms": round(ms, 1), "flagged_sample": fl}), flush=True)
PY
  done
done
