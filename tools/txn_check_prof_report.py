#!/usr/bin/env python3
"""Developer tool (GPU box): cycles per phase of the list-append device pass on BASELINE configs[4], from a build of
csrc/txn_check_dev.hip with -DTC_PROF linked as maelstrom_amd/libmaelsim_tcprof.so (see the hipcc lines in DESIGN.md §4.6b's history)."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["MSIM_LIB"] = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "maelstrom_amd", "libmaelsim_tcprof.so")
import numpy as np
from maelstrom_amd import engine as E
cfg = E.test_config(workload="txn-list-append", node_count=5, rate=100, time_limit=30, latency=5, nemesis=["partition"], nemesis_interval=10, seed=99)
n = 32768
with E.Engine(cfg) as eng:
    eng.run(0, n); eng.check(); eng.run(n, n); eng.check()
    res = eng.check_results()
    print("check ms", eng.kernel_ms()[1])
t = np.concatenate([res["stable_latency_ms"].astype(np.float64), res["never_read_count"][:, None].astype(np.float64), res["duplicated_count"][:, None].astype(np.float64)], axis=1) * 64
names = ["A rows -> transactions (pairing)", "B ranges + clears", "C writer table", "D read checks", "realtime suffix min", "E edges (2 passes) + prefix", "F Kahn"]
tot = t.sum(axis=1).mean()
for i, nm in enumerate(names): print(f"  {nm:36s} {t[:, i].mean():12.3e} cycles {100*t[:, i].mean()/tot:5.1f} %")
print("total", tot)
