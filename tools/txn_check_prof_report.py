#!/usr/bin/env python3
"""Developer tool (GPU box): cycles per phase of the list-append device pass on BASELINE configs[4], from a build of
csrc/txn_check_dev.hip with -DTC_PROF linked as maelstrom_amd/libmaelsim_tcprof.so (see the hipcc lines in DESIGN.md §4.6b's history)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MSIM_LIB", os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "maelstrom_amd", "libmaelsim_tcprof.so"))   # tools/variant_lib.sh tcprof txn_check_dev.hip -DTC_PROF
NODE = os.environ.get("TC_NODE", "")    # "" (the single-root node) | multi-key-txn | datomic: which txn-list-append node program wrote the histories
import numpy as np
from maelstrom_amd import engine as E
cfg = E.test_config(workload="txn-list-append", **({"bin": NODE} if NODE else {}), node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
n = 32768
with E.Engine(cfg) as eng:
    eng.run(0, n); eng.check(); eng.run(n, n); eng.check()
    res = eng.check_results()
    print("check ms", eng.kernel_ms()[1])
t = np.concatenate([res["stable_latency_ms"].astype(np.float64), res["never_read_count"][:, None].astype(np.float64), res["duplicated_count"][:, None].astype(np.float64)], axis=1) * 64
names = ["A rows -> transactions (pairing)", "B ranges + clears", "C writer table", "D read checks", "realtime suffix min", "E edges (2 passes) + prefix", "F acyclic (potential sweeps / Kahn)"]
sw = res["stale_count"]
print("sweeps until nothing was raised (1000+ = Kahn decided after that many):", dict(zip(*[x.tolist() for x in np.unique(sw, return_counts=True)])))
tot = t.sum(axis=1).mean()
for i, nm in enumerate(names): print(f"  {nm:36s} {t[:, i].mean():12.3e} cycles {100*t[:, i].mean()/tot:5.1f} %")
print("total", tot)
t0 = res["error_count"].astype(np.int64); t1 = res["stable_count"].astype(np.int64)
life = (t1 - t0) & 0xFFFFFFFF; span = int(((t1 - t0.min()) & 0xFFFFFFFF).max())
print(f"workgroup lifetime (100 MHz counter): mean {life.mean()/100:.1f} us, max {life.max()/100:.1f} us; first start to last end {span/1e5:.2f} ms; "
      f"workgroups alive on average {life.sum()/max(span,1):.0f}; wavefront 0's cycles per lifetime -> {tot/ (life.mean()*10):.2f} GHz")
