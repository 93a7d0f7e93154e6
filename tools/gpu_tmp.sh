#!/bin/bash
cd "$GRAFT_REPO_ROOT"
timeout 100 python - <<'PY' 2>&1 | tail -8
import sys
sys.path.insert(0, "tests")
from maelstrom_amd import engine as E
import oracle_lib as O
for n, kw in [(36, dict(latency=10, topology="line")), (64, dict(latency=20, latency_dist="uniform", topology="tree4"))]:
    cfg = E.test_config("broadcast", node_count=n, seed=79, rate=100, time_limit=8, **kw)
    ora = O.run(cfg, 0, 3)
    with E.Engine(cfg) as eng:
        eng.run(0, 3); eng.fetch()
        for i in range(3):
            m = eng.meta(i)
            print(n, i, "gpu", m.n_rows, m.n_payload_words, hex(m.flags), m.n_rounds, "oracle", int(ora.meta["n_rows"][i]), int(ora.meta["n_rounds"][i]))
PY
